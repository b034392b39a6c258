"""GPU parity tests (-m gpu) for the mapper's seed + chain stages against oracle/mm2_oracle.c (parity unpinned:
the oracle restates minimap2 2.17, which the reference does not vendor).  Bit-exact on every integer output."""
import numpy as np
import pytest

from map_cases import small_world

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def world(libmpn, oracle_built):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads = small_world()
    gidx = mapper.Index(gen)
    oidx = mb.Index(gen)
    yield gen, reads, gidx, oidx
    gidx.close()
    oidx.close()


def test_sketch_matches_oracle(world):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, _, _ = world
    seqs = [r['seq'] for r in reads] + [gen[0][1][:50000], np.zeros(0, dtype=np.uint8)]
    got = mapper.sketch_batch(seqs)
    for i, s in enumerate(seqs):
        want = mb.sketch(s, 10, 15, i)
        assert np.array_equal(got[i], want), (i, len(s), len(got[i]), len(want))
    # other (w,k)
    got = mapper.sketch_batch(seqs[:6], k=19, w=5)
    for i in range(6):
        assert np.array_equal(got[i], mb.sketch(seqs[i], 5, 19, i))


def test_sketch_adversarial_matches_oracle(libmpn, oracle_built):
    """The position-parallel sketch kernel and the automaton kernel must tile a sequence seamlessly: ties (low-complexity
    sequence), ambiguous bases on and around chunk boundaries (256 positions), lengths around the chunk and window sizes,
    even k (symmetric k-mers), hashes wider than 32 bits, windows wider than the fast kernel takes."""
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b'ACGT', dtype=np.uint8)
    rnd = lambda n: alpha[rng.integers(0, 4, size=n)]  # noqa: E731
    seqs = [np.full(1500, ord('A'), dtype=np.uint8), np.frombuffer(b'AT' * 700, dtype=np.uint8).copy(),
            np.frombuffer(b'ACG' * 500, dtype=np.uint8).copy(), np.tile(rnd(37), 60)]
    for L in (0, 1, 14, 15, 16, 24, 25, 26, 255, 256, 257, 280, 281, 282, 511, 512, 513, 537, 538, 800, 5000):
        seqs.append(rnd(L))
    for pos in (0, 10, 230, 255, 256, 257, 270, 290, 500, 512, 767, 768, 1023, 1024, 1999):
        s = rnd(2000)
        s[pos] = ord('N')
        seqs.append(s)
    s = rnd(4000)
    s[1000:1300] = ord('N')
    s[2047:2049] = ord('n')
    seqs.append(s)
    low = rnd(3000)
    low[500:1500] = np.frombuffer(b'CA' * 500, dtype=np.uint8)
    seqs.append(low)
    seqs.append(np.frombuffer(bytes(rnd(1200)).lower(), dtype=np.uint8).copy())
    for k, w in ((15, 10), (15, 1), (15, 2), (15, 32), (15, 33), (15, 50), (14, 10), (16, 10), (17, 10), (19, 5), (28, 19), (11, 3), (4, 4)):
        got = mapper.sketch_batch(seqs, k=k, w=w)
        for i, sq in enumerate(seqs):
            want = mb.sketch(sq, w, k, i)
            assert np.array_equal(got[i], want), (k, w, i, len(sq), len(got[i]), len(want))


def test_sketch_more_irregular_chunks_than_staging(libmpn, oracle_built):
    """The automaton kernel stages its reports for the chunks the host lists + 2048 of those the fast kernel appends; a sequence
    with an ambiguous base every few hundred positions has more: the rest take the automaton's own second pass."""
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    rng = np.random.default_rng(78)
    alpha = np.frombuffer(b'ACGT', dtype=np.uint8)
    s = alpha[rng.integers(0, 4, size=1_300_000)]
    s[rng.integers(0, len(s), size=len(s) // 300)] = ord('N')
    t = alpha[rng.integers(0, 4, size=40_000)]
    got = mapper.sketch_batch([t, s, t[:3000]])
    for i, sq in enumerate([t, s, t[:3000]]):
        want = mb.sketch(sq, 10, 15, i)
        assert np.array_equal(got[i], want), (i, len(got[i]), len(want))


def test_index_matches_oracle(world):
    import ctypes as ct
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    keys, key_off, pos = gidx.export()
    # oracle index via its lookup function on every key
    L = mb.lib()
    L.mmo_idx_get.argtypes = [ct.c_void_p, ct.c_uint64, ct.POINTER(ct.POINTER(ct.c_uint64))]
    L.mmo_idx_get.restype = ct.c_int64
    assert len(keys) > 1000 and np.all(np.diff(keys.astype(np.int64)) > 0)
    rng = np.random.default_rng(0)
    for ki in rng.integers(0, len(keys), size=300):
        p = ct.POINTER(ct.c_uint64)()
        n = L.mmo_idx_get(oidx.h, int(keys[ki]), ct.byref(p))
        assert n == key_off[ki + 1] - key_off[ki]
        assert [p[j] for j in range(n)] == [int(x) for x in pos[key_off[ki]:key_off[ki + 1]]]
    for f in (2e-4, 1e-2, 0.5):
        assert gidx.mid_occ(f) == oidx.mid_occ(f)


@pytest.mark.parametrize('mid_occ', [0, 3])
def test_seed_chain_matches_oracle(world, mid_occ):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    gopt = mapper.default_opt(mid_occ=mid_occ)
    oopt = mb.default_opt(mid_occ=mid_occ)
    occ = mid_occ if mid_occ > 0 else oidx.mid_occ()
    got = mapper.seed_chain_batch(gidx, gopt, [r['seq'] for r in reads])
    n_chains = 0
    for r, g in zip(reads, got):
        mv = mb.sketch(r['seq'], 10, 15, 0)
        a, rep = mb.collect_anchors(oidx, occ, mv, len(r['seq']))
        u, b = mb.chain(oopt, a)
        assert g['n_anchor'] == len(a), r['name']
        assert g['rep_len'] == rep, r['name']
        assert np.array_equal(g['u'], u), r['name']
        assert np.array_equal(g['b'], b), r['name']
        n_chains += len(u)
    assert n_chains > len(reads) // 2


@pytest.mark.parametrize('min_cnt,max_gap', [(1, 5000), (2, 5000), (5, 5000), (3, 300), (3, 40000), (2, 100)])
def test_seed_chain_other_segment_rules(world, min_cnt, max_gap):
    """the stray-hit filter derives its threshold and bin width from -n and -g: the chains must not depend on either"""
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    gopt = mapper.default_opt(min_cnt=min_cnt, max_gap=max_gap)
    oopt = mb.default_opt(min_cnt=min_cnt, max_gap=max_gap)
    got = mapper.seed_chain_batch(gidx, gopt, [r['seq'] for r in reads])
    n_chains = 0
    for r, g in zip(reads, got):
        mv = mb.sketch(r['seq'], 10, 15, 0)
        a, rep = mb.collect_anchors(oidx, oidx.mid_occ(), mv, len(r['seq']))
        u, b = mb.chain(oopt, a)
        assert g['n_anchor'] == len(a) and g['rep_len'] == rep, r['name']
        assert np.array_equal(g['u'], u) and np.array_equal(g['b'], b), (r['name'], min_cnt, max_gap)
        n_chains += len(u)
    assert n_chains > 10


def test_seed_chain_with_saturated_filter(libmpn, oracle_built):
    """150 near-identical copies of one sequence and no occurrence cut-off: a read minimizer hits every copy, a read's
    hundreds of thousands of hits overfill the filter's counters (it then keeps everything) and every bin is a true locus."""
    from megapath_nano_amd import mapper, synth
    from oracle import mm2_bindings as mb
    rng = np.random.default_rng(5)
    base = synth.ALPHA[rng.integers(0, 4, size=30000)]
    gen = []
    for c in range(150):
        s = base.copy()
        pos = rng.integers(0, len(s), size=300)
        s[pos] = synth.ALPHA[rng.integers(0, 4, size=300)]
        gen.append(('copy%d' % c, s))
    reads = []
    for k in range(6):
        st = int(rng.integers(0, 14000))
        q = base[st:st + 15000].copy()
        pos = rng.integers(0, len(q), size=150)
        q[pos] = synth.ALPHA[rng.integers(0, 4, size=150)]
        reads.append(q if k % 2 == 0 else synth.COMP[q[::-1]])
    gidx, oidx = mapper.Index(gen), mb.Index(gen)
    try:
        gopt, oopt = mapper.default_opt(mid_occ=100000), mb.default_opt(mid_occ=100000)
        got = mapper.seed_chain_batch(gidx, gopt, reads)
        tot = 0
        for q, g in zip(reads, got):
            mv = mb.sketch(q, 10, 15, 0)
            a, rep = mb.collect_anchors(oidx, 100000, mv, len(q))
            u, b = mb.chain(oopt, a)
            assert g['n_anchor'] == len(a) and g['rep_len'] == rep
            assert np.array_equal(g['u'], u) and np.array_equal(g['b'], b)
            tot += len(a)
        assert tot > 6 * 150000   # far more hits per read than the filter has counters
    finally:
        gidx.close()
        oidx.close()


def test_seed_chain_through_a_tandem_repeat(libmpn, oracle_built):
    """A 37-bp unit repeated 220 times inside one target, no occurrence cut-off: a read minimizer of the repeat hits every copy
    on the SAME target, hundreds of off-diagonal anchors lie within max_dist_x of each other, and the predecessor scan of an
    anchor runs far beyond the 128 anchors the chain DP keeps in LDS (its tiles from global memory, marks in both places)."""
    from megapath_nano_amd import mapper, synth
    from oracle import mm2_bindings as mb
    rng = np.random.default_rng(11)
    unit = synth.ALPHA[rng.integers(0, 4, size=37)]
    left, right = synth.ALPHA[rng.integers(0, 4, size=6000)], synth.ALPHA[rng.integers(0, 4, size=6000)]
    rep = np.tile(unit, 220)
    pos = rng.integers(0, len(rep), size=120)          # a few substitutions so that the copies are not all identical
    rep[pos] = synth.ALPHA[rng.integers(0, 4, size=120)]
    gen = [('flank_only', np.concatenate([left, right])), ('with_repeat', np.concatenate([left, rep, right])),
           ('other', synth.ALPHA[rng.integers(0, 4, size=9000)])]
    target = gen[1][1]
    reads = []
    for k in range(8):
        st = int(rng.integers(3000, 5500))
        q = target[st:st + int(rng.integers(6000, 11000))].copy()
        p = rng.integers(0, len(q), size=len(q) // 25)
        q[p] = synth.ALPHA[rng.integers(0, 4, size=len(p))]
        reads.append(q if k % 2 == 0 else synth.COMP[q[::-1]])
    gidx, oidx = mapper.Index(gen), mb.Index(gen)
    try:
        gopt, oopt = mapper.default_opt(mid_occ=100000), mb.default_opt(mid_occ=100000)
        got = mapper.seed_chain_batch(gidx, gopt, reads)
        tot = 0
        for q, g in zip(reads, got):
            mv = mb.sketch(q, 10, 15, 0)
            a, rl = mb.collect_anchors(oidx, 100000, mv, len(q))
            u, b = mb.chain(oopt, a)
            assert g['n_anchor'] == len(a) and g['rep_len'] == rl
            assert np.array_equal(g['u'], u) and np.array_equal(g['b'], b)
            tot += len(a)
        assert tot > 8 * 20000   # the repeat multiplies the hits
    finally:
        gidx.close()
        oidx.close()
