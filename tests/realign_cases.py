"""Seeded windows for the amplicon realigner (f4): a reference window, candidate haplotypes (prefix + consensus + suffix, as
/root/reference/bin/realignment/realign_illumina_reads.py:593 builds them) and Illumina-like reads drawn from them."""
import numpy as np

BASES = 'ACGT'


def _rand_seq(rng, n):
    return ''.join(BASES[i] for i in rng.integers(0, 4, n))


def _mutate_center(rng, center, kind):
    c = list(center)
    if kind == 'snv':
        for _ in range(int(rng.integers(1, 4))):
            i = int(rng.integers(0, len(c)))
            c[i] = BASES[(BASES.index(c[i]) + int(rng.integers(1, 4))) % 4]
    elif kind == 'ins':
        i = int(rng.integers(1, len(c)))
        c[i:i] = list(_rand_seq(rng, int(rng.integers(1, 12))))
    elif kind == 'del':
        i = int(rng.integers(1, max(2, len(c) - 12)))
        del c[i:i + int(rng.integers(1, 10))]
    elif kind == 'mix':
        i = int(rng.integers(1, len(c) // 2))
        del c[i:i + int(rng.integers(1, 5))]
        j = int(rng.integers(len(c) // 2, len(c)))
        c[j:j] = list(_rand_seq(rng, int(rng.integers(1, 6))))
        k = int(rng.integers(0, len(c)))
        c[k] = BASES[(BASES.index(c[k]) + 1) % 4]
    return ''.join(c)


def make_window(seed, n_reads=40, n_haps=3, prefix=150, center=60, suffix=150, read_len=100, include_ref=True,
                uncovered_hap=False, with_n=True, repeat=False):
    rng = np.random.default_rng(seed)
    pre, cen, suf = _rand_seq(rng, prefix), _rand_seq(rng, center), _rand_seq(rng, suffix)
    if repeat:  # a tandem repeat across the window: k-mers of a read hit several haplotype positions
        unit = _rand_seq(rng, 7)
        cen = (unit * (center // 7 + 1))[:center]
    reference = pre + cen + suf
    kinds = ['snv', 'ins', 'del', 'mix']
    haps = []
    if include_ref:
        haps.append(reference)
    while len(haps) < n_haps:
        haps.append(pre + _mutate_center(rng, cen, kinds[int(rng.integers(0, 4))]) + suf)
    if uncovered_hap:  # a haplotype whose novel middle no read supports
        haps.append(pre + _rand_seq(rng, center + 40) + suf)
    src_haps = haps[:n_haps]
    seqs, positions, cigars = [], [], []
    ref_start = int(rng.integers(1000, 100000))
    for r in range(n_reads):
        h = src_haps[int(rng.integers(0, len(src_haps)))]
        mode = int(rng.integers(0, 10))
        L = read_len if mode < 8 else int(rng.integers(20, read_len + 1))
        L = min(L, len(h))
        s = int(rng.integers(0, len(h) - L + 1))
        q = list(h[s:s + L])
        n_mm = [0, 0, 0, 1, 1, 2, 2, 3, 5, 0][mode]
        for _ in range(n_mm):
            i = int(rng.integers(0, L))
            q[i] = BASES[(BASES.index(q[i]) + int(rng.integers(1, 4))) % 4] if q[i] in BASES else 'A'
        if mode == 9 and L > 40:   # an indel inside the read: only the SSW path can place it
            i = int(rng.integers(10, L - 10))
            if rng.integers(0, 2):
                q[i:i] = list(_rand_seq(rng, int(rng.integers(1, 4))))
            else:
                del q[i:i + int(rng.integers(1, 4))]
        if with_n and rng.integers(0, 8) == 0:
            for _ in range(int(rng.integers(1, 4))):
                q[int(rng.integers(0, len(q)))] = 'N'
        if rng.integers(0, 25) == 0:
            q = list(_rand_seq(rng, L))  # unrelated read
        seqs.append(''.join(q))
        positions.append(ref_start + s)
        cigars.append('%dM' % len(q))
    return dict(seqs=seqs, positions=positions, cigars=cigars, reference=reference, haplotypes=haps, ref_start=ref_start,
                ref_prefix=prefix, ref_suffix=suffix)


def make_cases():
    cases = []
    for seed in range(6):
        cases.append(make_window(100 + seed))
    cases.append(make_window(200, n_reads=120, n_haps=6, prefix=300, center=120, suffix=300, read_len=250))
    cases.append(make_window(201, n_reads=60, n_haps=4, prefix=100, center=40, suffix=100, read_len=150, include_ref=False))
    cases.append(make_window(202, n_reads=50, n_haps=3, uncovered_hap=True))
    cases.append(make_window(203, n_reads=50, n_haps=4, repeat=True, center=90))
    cases.append(make_window(204, n_reads=30, n_haps=2, prefix=40, center=50, suffix=40, read_len=60))
    cases.append(make_window(205, n_reads=1, n_haps=1))
    cases.append(make_window(206, n_reads=80, n_haps=8, prefix=200, center=80, suffix=200, read_len=125, with_n=False))
    cases.append(make_window(207, n_reads=40, n_haps=24, prefix=36, center=82, suffix=111, read_len=89, uncovered_hap=True))  # > 16: std::sort's quicksort phase
    cases.append(make_window(208, n_reads=35, n_haps=18, prefix=60, center=30, suffix=60, read_len=50, include_ref=False, repeat=True))
    return cases
