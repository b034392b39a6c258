"""GPU tests (-m gpu) of the f3 kernels (include/mpn_abundance.h): the stable coordinate sort behind the BAM writer and the
interval union behind the abundance statistic, against numpy and against brute force; the noise-BED branch of the statistic
(/root/reference/bin/megapath_nano.py:516-541)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def test_sort_order_is_numpy_stable_lexsort(libmpn):
    from megapath_nano_amd.abundance import device_sort_order
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 300, 5000, 300000):
        tid = rng.integers(0, 40, size=n).astype(np.int64)
        tid[rng.random(n) < 0.05] = np.int64(1) << 40          # unplaced records sort last
        pos = rng.integers(0, 1 << 28, size=n).astype(np.int64) if n else np.zeros(0, dtype=np.int64)
        pos[rng.random(n) < 0.3] = 12345                       # many ties: stability matters
        rev = rng.integers(0, 2, size=n).astype(np.int64)
        got = device_sort_order(tid, pos, rev)
        want = np.lexsort((rev, pos, tid))
        assert np.array_equal(got, want), n
    # keys that differ only in the top bits, and a constant key
    hi = np.array([5, 1 << 40, 0, 5, 1 << 40], dtype=np.int64)
    assert list(device_sort_order(hi, np.zeros(5, dtype=np.int64), np.zeros(5, dtype=np.int64))) == [2, 0, 3, 1, 4]
    assert list(device_sort_order(np.zeros(4, dtype=np.int64), np.zeros(4, dtype=np.int64), np.zeros(4, dtype=np.int64))) == [0, 1, 2, 3]


def test_cover_by_group_matches_numpy_and_brute_force(libmpn):
    from megapath_nano_amd.abundance import device_cover_by_group, host_cover_by_group
    rng = np.random.default_rng(4)
    # small: brute force over positions
    for trial in range(20):
        n = int(rng.integers(1, 200))
        g = rng.integers(0, 4, size=n).astype(np.int32)
        sq = rng.integers(0, 3, size=n).astype(np.int32)
        st = rng.integers(0, 300, size=n).astype(np.int64)
        en = st + rng.integers(0, 60, size=n)
        got = device_cover_by_group(g, sq, st, en, 4)
        want = np.zeros(4, dtype=np.int64)
        for gi in range(4):
            cov = set()
            for k in np.flatnonzero(g == gi):
                cov |= {(int(sq[k]), p) for p in range(int(st[k]), int(en[k]))}
            want[gi] = len(cov)
        assert np.array_equal(got, want), trial
        assert np.array_equal(host_cover_by_group(g, sq, st, en, 4), want)
    # large: few huge groups (chunks of the sweep run inside one (group, sequence)) and many tiny ones
    n = 600000
    g = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 2000, size=n)).astype(np.int32)
    sq = np.where(g == 0, 0, rng.integers(0, 5, size=n)).astype(np.int32)
    st = rng.integers(0, 50_000_000, size=n).astype(np.int64)
    en = st + rng.integers(0, 20000, size=n)
    assert np.array_equal(device_cover_by_group(g, sq, st, en, 2000), host_cover_by_group(g, sq, st, en, 2000))
    # book-ended intervals merge, empty intervals add nothing
    assert list(device_cover_by_group([0, 0, 0, 1], [7, 7, 7, 7], [0, 10, 30, 5], [10, 20, 30, 6], 2)) == [20, 1]


def test_align_stat_with_noise_bed_device_equals_host_and_brute_force(libmpn):
    from megapath_nano_amd.abundance import align_stat_by_assembly_id
    rng = np.random.default_rng(9)
    rows = []
    for r in range(400):
        for _ in range(int(rng.integers(1, 4))):
            a = str(rng.choice(['A1', 'A2', 'A3']))
            s0 = int(rng.integers(0, 6000))
            rows.append((f'r{r}', 5000, a, a + str(rng.choice(['_c1', '_c2'])), s0, s0 + int(rng.integers(1, 3000)), 100, 5,
                         int(rng.integers(100, 900)), float(rng.random())))
    al = pd.DataFrame(rows, columns=['read_id', 'read_length', 'assembly_id', 'sequence_id', 'sequence_from', 'sequence_to', 'match',
                                     'edit_dist', 'alignment_score', 'alignment_score_tiebreaker'])
    lens = pd.DataFrame({'assembly_id': ['A1', 'A2', 'A3'], 'assembly_length': [9000, 9000, 9000]})
    noise = pd.DataFrame({'sequence_id': ['A1_c1', 'A1_c1', 'A2_c2', 'ZZ'], 'start': [100, 150, 0, 0], 'end': [900, 2500, 4000, 10],
                          'assembly_id': ['A1', 'A1', 'A2', 'ZZ']})
    dev = align_stat_by_assembly_id(al, lens, noise_bed=noise, device=True).set_index('assembly_id')
    host = align_stat_by_assembly_id(al, lens, noise_bed=noise, device=False).set_index('assembly_id')
    pd.testing.assert_frame_equal(dev, host)
    # brute force: best row per (read, assembly), positions covered minus noise positions
    best = {}
    for row in rows:
        key = (row[0], row[2])
        if key not in best or (row[8], row[9]) > (best[key][8], best[key][9]):
            best[key] = row
    noise_pos = set()
    for s_, a_, b_ in zip(noise['sequence_id'], noise['start'], noise['end']):
        noise_pos |= {(s_, p) for p in range(a_, b_)}
    for a in ('A1', 'A2', 'A3'):
        cov = set()
        for (rd, asm), v in best.items():
            if asm == a:
                cov |= {(v[3], p) for p in range(v[4], v[5])}
        assert dev.loc[a, 'covered_bp'] == len(cov - noise_pos)
    assert dev.loc['A1', 'noise_span_bp'] == 800 + 2350 and dev.loc['A2', 'noise_span_bp'] == 4000 and dev.loc['A3', 'noise_span_bp'] == 0
    assert abs(dev.loc['A1', 'adjusted_covered_percent'] - dev.loc['A1', 'covered_bp'] / (9000 - 3150)) < 1e-12


def test_sorted_bam_with_device_sort_equals_host_sort(libmpn, tmp_path):
    """bam.sam_to_sorted_bam with the GPU sort hook writes the same file as with the host sort."""
    from megapath_nano_amd import bam
    from megapath_nano_amd.abundance import device_sort_order
    rng = np.random.default_rng(5)
    lines = ['@SQ\tSN:t1\tLN:3000000\n', '@SQ\tSN:t2\tLN:2000000\n']
    for i in range(4000):
        ref, pos = ('t1', int(rng.integers(1, 2000))) if i % 3 else ('t2', int(rng.integers(1, 1900000)))
        lines.append(f'r{i}\t{16 * (i % 2)}\t{ref}\t{pos}\t60\t10M\t*\t0\t0\tACGTACGTAC\t*\tNM:i:{i % 50}\n')
    lines.append('u1\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t*\n')
    sam = tmp_path / 'x.sam'
    sam.write_text(''.join(lines))
    bam.sam_to_sorted_bam(str(sam), str(tmp_path / 'h.bam'), exclude_flags=0)
    bam.sam_to_sorted_bam(str(sam), str(tmp_path / 'd.bam'), exclude_flags=0, sort_keys=device_sort_order)
    assert open(tmp_path / 'h.bam', 'rb').read() == open(tmp_path / 'd.bam', 'rb').read()
    assert open(tmp_path / 'h.bam.bai', 'rb').read() == open(tmp_path / 'd.bam.bai', 'rb').read()
