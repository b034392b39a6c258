"""GPU parity tests (-m gpu) of the extension-DP kernels (single-wave LDS, workgroup, strip, band) against
the oracle's mmo_extd2 on seeded pairs: global / approximate-max / extension-only / right-aligned modes, band clipping,
z-drop, ambiguous bases, small to large windows.  Bit-exact on scores, end points and CIGAR."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

APPROX, RIGHT, EXTZ, REV = 0x02, 0x08, 0x40, 0x80


def mutate(rng, s, rate):
    out = []
    for c in s:
        r = rng.random()
        if r < rate * 0.4:
            out.append(int((c + rng.integers(1, 4)) % 4))
        elif r < rate * 0.7:
            out.append(int(c))
            out.extend(rng.integers(0, 4, size=int(rng.integers(1, 4))).tolist())
        elif r < rate:
            continue
        else:
            out.append(int(c))
    return np.array(out if out else [0], dtype=np.uint8)


def make_pairs(seed, sizes, tail=False, ambig=False, big_indel=False):
    rng = np.random.default_rng(seed)
    qs, ts = [], []
    for L in sizes:
        t = rng.integers(0, 4, size=L).astype(np.uint8)
        q = mutate(rng, t, 0.12)
        if big_indel and L > 400:
            cut = int(rng.integers(100, L - 200))
            q = np.concatenate([q[:cut], rng.integers(0, 4, size=int(rng.integers(30, 120))).astype(np.uint8), q[cut:]])
        if tail:  # unrelated sequence after the homologous part: the extension must stop (z-drop)
            q = np.concatenate([q, rng.integers(0, 4, size=int(rng.integers(300, 900))).astype(np.uint8)])
            t = np.concatenate([t, rng.integers(0, 4, size=int(rng.integers(300, 900))).astype(np.uint8)])
        if ambig:
            q[rng.integers(0, len(q), size=max(1, len(q) // 50))] = 4
            t[rng.integers(0, len(t), size=max(1, len(t) // 60))] = 4
        qs.append(q)
        ts.append(t)
    return qs, ts


def check(opt, qs, ts, w, zdrop, end_bonus, flag, kernels):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    want = [mb.extd2(q, t, w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag) for q, t in zip(qs, ts)]
    for k in kernels:
        got = mapper.ext_dp_batch(opt, qs, ts, w, zdrop, end_bonus, flag, force_kernel=k)
        for i, (g, e) in enumerate(zip(got, want)):
            keys = ['zdropped', 'n_cigar', 'cigar', 'score'] if flag & APPROX else \
                ['max', 'zdropped', 'max_q', 'max_t', 'mqe', 'mqe_t', 'score', 'reach_end', 'n_cigar', 'cigar']
            for key in keys:
                if key == 'score' and e['zdropped']:
                    continue
                if key in ('mqe', 'score') and g[key] < -10**8 and e[key] < -10**8:
                    continue  # both 'never set' (the two sides use different -inf constants)
                assert g[key] == e[key], (k, i, len(qs[i]), len(ts[i]), key, g[key] if key != 'cigar' else g[key][:6],
                                          e[key] if key != 'cigar' else e[key][:6])


@pytest.fixture(scope='module')
def opt(libmpn, oracle_built):
    from megapath_nano_amd import mapper
    return mapper.default_opt()


def test_gap_fill_windows_all_kernels(opt):
    qs, ts = make_pairs(1, [30, 64, 65, 128, 200, 230, 256, 257, 300, 400, 511])
    check(opt, qs, ts, 751, 400, -1, APPROX, [1, 3, 4, 5, 0])
    qs, ts = make_pairs(2, [220, 260, 310], ambig=True)
    check(opt, qs, ts, 751, 400, -1, APPROX, [1, 3, 4, 5])
    qs, ts = make_pairs(3, [450, 500], big_indel=True)
    check(opt, qs, ts, 751, 400, -1, APPROX, [1, 3, 4, 5])
    qs, ts = make_pairs(11, [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 512, 513, 600, 700])
    check(opt, qs, ts, 751, 400, -1, APPROX, [1, 4, 5])
    check(opt, qs, ts, 751, 400, -1, APPROX | RIGHT, [1, 4, 5])


def test_strip_heights_up_to_16(opt):
    """Every strip height S = ceil(tlen / 64) = 1..16 of the systolic kernel (kernel 4), incl. the unaligned 32-bit
    direction stores of the odd heights and query windows longer / shorter than the target."""
    rng = np.random.default_rng(21)
    qs, ts = [], []
    for S in range(1, 17):
        tlen = int(rng.integers(64 * (S - 1) + 1, 64 * S + 1))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = mutate(rng, t, 0.12)
        if S % 3 == 0:
            q = q[: max(1, len(q) * 2 // 3)]
        if S % 4 == 1:
            q = np.concatenate([q, rng.integers(0, 4, size=40).astype(np.uint8)])
        qs.append(q)
        ts.append(t)
    ts.append(rng.integers(0, 4, size=1024).astype(np.uint8))
    qs.append(mutate(rng, ts[-1], 0.1))
    check(opt, qs, ts, 2000, 400, -1, APPROX, [4, 1])


def test_exact_global_mode(opt):
    qs, ts = make_pairs(4, [50, 180, 260, 420, 700])
    check(opt, qs, ts, 751, 400, -1, 0, [1, 3, 5])
    qs, ts = make_pairs(5, [300, 600], big_indel=True)  # z-drop inside a global fill
    check(opt, qs, ts, 751, 50, -1, 0, [1, 3, 5])


def test_extension_modes_with_zdrop(opt):
    qs, ts = make_pairs(6, [40, 150, 400, 900], tail=True)
    check(opt, qs, ts, 751, 400, -1, EXTZ, [1, 3, 5, 0])
    check(opt, qs, ts, 751, 400, -1, EXTZ | RIGHT | REV, [1, 3, 5, 0])
    check(opt, qs, ts, 751, 400, 50, EXTZ, [1, 3, 5])  # end bonus: reach_end path
    qs, ts = make_pairs(7, [300, 500])
    check(opt, qs, ts, 751, 400, 10, EXTZ, [1, 3, 5])


def test_exact_strip_windows(opt):
    """The exact variants of the systolic strip kernel (kernel 4 where eligible: the band never clips, target up to 1024 rows):
    end extensions to the right and to the left (right-aligned gaps, reversed CIGAR), exact global fills; z-drops in the
    middle of a window, the end bonus, ambiguous bases, every lane-group class, ties between cells of an anti-diagonal
    (low-complexity sequence)."""
    rng = np.random.default_rng(41)
    sizes = [1, 2, 3, 7, 16, 17, 33, 47, 64, 90, 128, 150, 200, 255, 256, 257, 300, 420, 511, 512, 600, 800, 1000]
    qs, ts = make_pairs(41, sizes)
    ts = [np.concatenate([t, rng.integers(0, 4, size=len(t) // 2 + 3).astype(np.uint8)])[:1024] for t in ts]   # target window ~1.5 x query
    for flag in (EXTZ, EXTZ | RIGHT | REV, 0, RIGHT):
        check(opt, qs, ts, 3000, 400, -1, flag, [4, 0])
        check(opt, qs, ts, 3000, 60, 20, flag, [4])
    qs, ts = make_pairs(42, [60, 130, 260, 500], tail=True)          # unrelated tails: the extension z-drops
    qs, ts = [q[:1000] for q in qs], [t[:1024] for t in ts]
    for flag in (EXTZ, EXTZ | RIGHT | REV):
        check(opt, qs, ts, 3000, 100, -1, flag, [4, 1])
        check(opt, qs, ts, 3000, 30, 5, flag, [4])
    qs, ts = make_pairs(43, [80, 240, 480], ambig=True)
    check(opt, qs, ts, 3000, 400, 10, EXTZ, [4])
    check(opt, qs, ts, 3000, 400, 10, EXTZ | RIGHT | REV, [4])
    # low complexity: many equal scores on an anti-diagonal (the tie order of the maximum decides where the alignment ends)
    qs = [np.array(([0, 1] * 200)[:n], dtype=np.uint8) for n in (50, 150, 333)] + [np.zeros(120, dtype=np.uint8)]
    ts = [np.array(([0, 1] * 300)[:n * 3 // 2], dtype=np.uint8) for n in (50, 150, 333)] + [np.zeros(200, dtype=np.uint8)]
    for flag in (EXTZ, EXTZ | RIGHT | REV, 0):
        check(opt, qs, ts, 3000, 400, -1, flag, [4, 1])
        check(opt, qs, ts, 3000, 20, 3, flag, [4])


def test_band_clipping_and_large_windows(opt):
    qs, ts = make_pairs(8, [600, 1500])
    check(opt, qs, ts, 100, 400, -1, 0, [1, 3, 5])       # narrow band: cells outside the previous band
    check(opt, qs, ts, 20, 400, -1, APPROX, [1, 3, 5])
    qs, ts = make_pairs(9, [3000, 5200], tail=True)
    check(opt, qs, ts, 751, 400, -1, EXTZ, [1, 3, 5, 0])
    qs, ts = make_pairs(10, [14000])                  # state arrays in the global scratch
    check(opt, qs, ts, 751, 400, -1, EXTZ, [1, 3, 5])


def test_band_kernel_slot_variants(opt):
    """Every slot count of the band kernel (128/256/512/1024), with the band sliding far enough to wrap the slots several
    times, in approximate, exact and extension modes; the single-wave LDS kernel (1) must agree with the oracle too."""
    qs, ts = make_pairs(12, [900, 2500, 4100])
    for w in (10, 63, 126, 127, 128, 254, 255, 256, 400, 510, 511, 600, 1022, 1023):
        check(opt, qs, ts, w, 400, -1, APPROX, [5])
        check(opt, qs, ts, w, 400, -1, 0, [5])
    qs, ts = make_pairs(13, [700, 2000, 3500], tail=True)
    for w in (100, 200, 500, 751, 1000):
        check(opt, qs, ts, w, 400, -1, EXTZ, [5])
        check(opt, qs, ts, w, 200, 30, EXTZ | RIGHT | REV, [5])
    qs, ts = make_pairs(14, [1800, 2600], ambig=True, big_indel=True)
    check(opt, qs, ts, 300, 100, -1, 0, [1, 5])
    check(opt, qs, ts, 300, 100, -1, APPROX | RIGHT, [1, 5])


def test_lopsided_windows_dispatch(opt):
    """Windows far from square: long query x short target and the reverse, around the strip kernel's class limits
    (query 1024 / 2048 / 4096 bases per lane-group class, target 256 / 512 / 1024 rows): whatever kernel the dispatcher
    picks (0) must agree with the oracle, like the LDS fallback (1)."""
    rng = np.random.default_rng(31)
    qs, ts = [], []
    for qlen, tlen in ((5000, 300), (300, 5000), (1024, 256), (1025, 256), (2048, 512), (2049, 300), (4096, 1024), (4097, 1000),
                       (40, 1024), (1500, 17), (17, 1500), (3, 900)):
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = rng.integers(0, 4, size=qlen).astype(np.uint8)
        n = min(qlen, tlen)
        q[:n] = np.where(rng.random(n) < 0.88, t[:n], q[:n])   # a homologous prefix, the rest unrelated
        qs.append(q)
        ts.append(t)
    check(opt, qs, ts, 6000, 400, -1, APPROX, [0, 4, 1, 6])
    check(opt, qs, ts, 6000, 400, -1, 0, [0, 5])


def test_tiled_banded_strips(opt):
    """Kernel 6: the gap fills beyond the strip kernel's reach -- targets longer than 1024 rows (several tiles, the boundary between
    them through HBM), bands that clip (freshly opened gaps at a row's first and last in-band cell), both together; windows whose
    corner the band does not reach fall back to the other kernels.  Scores and CIGARs against the oracle."""
    qs, ts = make_pairs(12, [900, 2500, 4100])
    for w in (63, 127, 255, 400, 600, 751, 1023, 2000, 6000):
        check(opt, qs, ts, w, 400, -1, APPROX, [6])
    qs, ts = make_pairs(15, [1000, 1024, 1025, 1100, 2047, 2048, 2049, 3000, 5000])
    check(opt, qs, ts, 751, 400, -1, APPROX, [6, 0])
    check(opt, qs, ts, 9000, 400, -1, APPROX, [6])          # no clipping, only tiles
    qs, ts = make_pairs(14, [1800, 2600], ambig=True, big_indel=True)
    check(opt, qs, ts, 300, 100, -1, APPROX, [6])
    qs, ts = make_pairs(16, [30, 200, 700, 790])              # one tile, clipping or not
    check(opt, qs, ts, 20, 400, -1, APPROX, [6])
    check(opt, qs, ts, 751, 400, -1, APPROX, [6])
    # low complexity: ties everywhere
    qs = [np.array(([0, 1] * 1200)[:n], dtype=np.uint8) for n in (1300, 2200)] + [np.zeros(1500, dtype=np.uint8)]
    ts = [np.array(([0, 1] * 1300)[:n + 37], dtype=np.uint8) for n in (1300, 2200)] + [np.zeros(1530, dtype=np.uint8)]
    check(opt, qs, ts, 500, 400, -1, APPROX, [6])


def test_tiled_exact_extensions(opt):
    """Kernel 6 on end extensions and exact fills beyond the exact strip variants' reach (targets > 1024 rows, clipping bands): the
    maximum of every anti-diagonal with ksw2's tie order over IN-BAND cells, z-drop in the middle, the band leaving the matrix
    (st > en), the best cell of the last query column, the end bonus, right-aligned gaps on reversed sequences; low complexity."""
    qs, ts = make_pairs(13, [700, 2000, 3500], tail=True)
    for w in (100, 200, 500, 751, 1000):
        check(opt, qs, ts, w, 400, -1, EXTZ, [6])
        check(opt, qs, ts, w, 200, 30, EXTZ | RIGHT | REV, [6])
    qs, ts = make_pairs(9, [3000, 5200], tail=True)
    check(opt, qs, ts, 751, 400, -1, EXTZ, [6, 0])
    check(opt, qs, ts, 751, 400, -1, EXTZ | RIGHT | REV, [6, 0])
    qs, ts = make_pairs(17, [1100, 1500, 2600, 4800])           # no tail: the extension runs to the end (reach_end / mqe)
    check(opt, qs, ts, 751, 400, 10, EXTZ, [6])
    check(opt, qs, ts, 751, 400, -1, EXTZ | RIGHT | REV, [6])
    check(opt, qs, ts, 751, 400, -1, 0, [6])                    # exact global fill
    check(opt, qs, ts, 300, 100, -1, 0, [6])
    qs, ts = make_pairs(18, [1200, 2500], ambig=True, big_indel=True)
    check(opt, qs, ts, 751, 400, -1, EXTZ, [6])
    check(opt, qs, ts, 751, 100, 5, EXTZ | RIGHT | REV, [6])
    # lopsided: the band leaves the matrix long before the query ends
    rng = np.random.default_rng(3)
    qs, ts = [], []
    for qlen, tlen in ((5000, 1300), (1300, 5000), (4000, 1100), (600, 1500)):
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = rng.integers(0, 4, size=qlen).astype(np.uint8)
        n = min(qlen, tlen)
        q[:n] = np.where(rng.random(n) < 0.9, t[:n], q[:n])
        qs.append(q)
        ts.append(t)
    check(opt, qs, ts, 751, 400, -1, EXTZ, [6])
    check(opt, qs, ts, 200, 400, -1, EXTZ | RIGHT | REV, [6])
    # low complexity: ties on every anti-diagonal
    qs = [np.array(([0, 1] * 1200)[:n], dtype=np.uint8) for n in (1300, 2200)] + [np.zeros(1500, dtype=np.uint8)]
    ts = [np.array(([0, 1] * 1300)[:n + 37], dtype=np.uint8) for n in (1300, 2200)] + [np.zeros(1530, dtype=np.uint8)]
    for flag in (EXTZ, EXTZ | RIGHT | REV, 0):
        check(opt, qs, ts, 500, 400, -1, flag, [6])
        check(opt, qs, ts, 500, 20, 3, flag, [6])
