"""GPU tests (-m gpu) at a size the oracle cannot cover in full: several thousand reads through the whole step
(map -> reassign -> counts).  Checked through size-independent properties + oracle parity on a random sample."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def big(libmpn, oracle_built):
    from megapath_nano_amd import mapper, synth
    gen = synth.make_genomes(77, 20, 400000, strain_pairs=2)
    w = np.zeros(20)
    w[[0, 1, 2, 3, 4, 5, 6, 7, 18, 19]] = [8, 4, 2, 2, 1, 1, 1, 1, 0.3, 0.2]
    reads = synth.make_reads(78, gen, 3000, mean_len=6000, weights=w, random_frac=0.02)
    idx = mapper.Index(gen)
    packed = mapper.PackedReads([r['name'] for r in reads], [r['seq'] for r in reads])
    yield gen, reads, idx, packed
    idx.close()


def test_sample_parity_and_truth(big):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, idx, packed = big
    opt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    paf, cols = mapper.map_batch_ex(idx, opt, packed, want_paf=True, want_cols=True)
    by = {}
    for line in paf.splitlines():
        by.setdefault(line.split('\t', 1)[0], []).append(line)
    # (1) oracle parity on a random sample
    oidx = mb.Index(gen)
    oopt = mb.default_opt(best_n=50, pri_ratio=1.0)
    rng = np.random.default_rng(1)
    for i in rng.choice(len(reads), size=40, replace=False):
        r = reads[i]
        want = mb.map_read(oidx, oopt, r['name'], r['seq'])[2]
        assert '\n'.join(by.get(r['name'], [])) + ('\n' if r['name'] in by else '') == want, r['name']
    oidx.close()
    # (2) the primary hit of a simulated read overlaps its origin (strain copies count as the same origin)
    name_to_idx = {g[0]: i for i, g in enumerate(gen)}
    twin = {18: 0, 19: 1, 0: 18, 1: 19}
    ok = tot = 0
    for r in reads:
        if r['genome'] < 0 or len(r['seq']) < 1000:
            continue
        tot += 1
        prim = [l.split('\t') for l in by.get(r['name'], []) if '\ttp:A:P' in l]
        if not prim:
            continue
        f = prim[0]
        gi = name_to_idx[f[5]]
        if (gi == r['genome'] or twin.get(gi) == r['genome']) and int(f[7]) < r['end'] and int(f[8]) > r['start']:
            ok += 1
    assert ok >= 0.97 * tot, (ok, tot)
    # (3) columns agree with the text
    assert len(cols['rid']) == paf.count('\n')
    # (4) idempotence
    paf2, _ = mapper.map_batch_ex(idx, opt, packed, want_paf=True, want_cols=False)
    assert paf2 == paf


def test_step_counts_and_shard_invariance(big):
    """reads-per-name counters: every mapped read is counted once; two shards + summing all-reduce == one run."""
    import threading
    from megapath_nano_amd import mapper
    from megapath_nano_amd.pipeline import Taxonomy, align_and_assign
    from test_reassign_gpu import ThreadAllreduce
    gen, reads, idx, packed = big
    opt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    tax = Taxonomy(np.arange(20), 20, np.arange(20), 20)
    full = align_and_assign(idx, opt, packed, tax, rng=random.Random(5))
    _, cols = mapper.map_batch_ex(idx, opt, packed, want_paf=False, want_cols=True)
    assert int(full['read_count'].sum()) == len(set(cols['read_idx'].tolist()))
    assert full['n_relations'] > 0, 'the strain pair should trigger reassignment'
    # counts do not depend on the tiebreaker stream except through exact score ties
    half = len(reads) // 2
    parts = [mapper.PackedReads([r['name'] for r in reads[:half]], [r['seq'] for r in reads[:half]]),
             mapper.PackedReads([r['name'] for r in reads[half:]], [r['seq'] for r in reads[half:]])]
    ar = ThreadAllreduce(2)
    outs, errs = [None, None], []

    def work(k):
        try:
            outs[k] = align_and_assign(idx, opt, parts[k], tax, allreduce=ar, rng=random.Random(5 + k))
        except Exception as e:  # pragma: no cover
            errs.append(e)
            ar.barrier.abort()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert np.array_equal(outs[0]['read_count'], outs[1]['read_count'])
    assert int(outs[0]['read_count'].sum()) == int(full['read_count'].sum())
    assert np.abs(outs[0]['read_count'] - full['read_count']).sum() <= 4  # only exact-score ties may move
    assert np.array_equal(outs[0]['aligned_bp'], outs[1]['aligned_bp'])


def test_very_long_reads_match_oracle(big):
    """Reads far beyond the typical 8 kb: a 390 kb read covering almost a whole target (forward), a 250 kb one from the
    reverse strand, and a 120 kb chimera of two targets -- thousands of DP windows and tens of thousands of anchors per read,
    alone in their own sub-batches; PAF identical to the oracle's."""
    from megapath_nano_amd import mapper, synth
    from oracle import mm2_bindings as mb
    gen, _, idx, _ = big
    rng = np.random.default_rng(123)
    g3 = np.frombuffer(bytes(gen[3][1]), dtype=np.uint8)
    g5 = np.frombuffer(bytes(gen[5][1]), dtype=np.uint8)
    g7 = np.frombuffer(bytes(gen[7][1]), dtype=np.uint8)
    long_reads = [
        dict(name='long_fwd_390k', seq=synth.ont_errors(rng, g3[5000:395000].copy())),
        dict(name='long_rev_250k', seq=synth.ont_errors(rng, synth.COMP[g5[100000:350000][::-1]])),
        dict(name='chimera_120k', seq=synth.ont_errors(rng, np.concatenate([g7[10000:70000], g3[200000:260000]]))),
    ]
    opt, oopt = mapper.default_opt(best_n=50, pri_ratio=1.0), mb.default_opt(best_n=50, pri_ratio=1.0)
    got = mapper.map_batch(idx, opt, [r['name'] for r in long_reads], [r['seq'] for r in long_reads])
    oidx = mb.Index(gen)
    want = ''.join(mb.map_read(oidx, oopt, r['name'], r['seq'])[2] for r in long_reads)
    oidx.close()
    assert got == want
    prim = {l.split('\t')[0]: l.split('\t') for l in got.splitlines() if '\ttp:A:P' in l}
    assert int(prim['long_fwd_390k'][3]) - int(prim['long_fwd_390k'][2]) > 0.95 * len(long_reads[0]['seq'])
    assert prim['long_rev_250k'][4] == '-'
