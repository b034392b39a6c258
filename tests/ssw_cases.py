"""Seeded SSW test-case generator shared by the oracle pin test, the golden-vector script
and the GPU parity tests.  Pure data generation - no reference code involved.

Case shapes follow SURVEY.md section 8c: random DNA pairs, L in 20..2000, score>=255 (word path),
N bases, maskLen<15, flags {0,1,2,8,15}, both caller matrices (pyssw.py:70-76 leaves the
N row/col at 0; ssw_cpp.cpp:41-46 sets it to -mismatch).
"""
import numpy as np


def build_matrix(match=4, mismatch=6, n_zero=True):
    m = np.zeros((5, 5), dtype=np.int8)
    for i in range(4):
        for j in range(4):
            m[i, j] = match if i == j else -mismatch
    if not n_zero:
        m[4, :] = -mismatch
        m[:, 4] = -mismatch
    return m.reshape(-1)


def mutate(rng, seq, sub=0.04, ins=0.03, dele=0.03, long_indel=0.0):
    out = []
    i = 0
    n = len(seq)
    while i < n:
        r = rng.random()
        if r < long_indel:
            # adjacent insertion followed by deletion: the path the striped passes treat specially
            k = int(rng.integers(4, 12))
            out.extend(rng.integers(0, 4, size=k).tolist())
            i += int(rng.integers(4, 12))
            continue
        r = rng.random()
        if r < sub:
            out.append(int((seq[i] + rng.integers(1, 4)) % 4))
            i += 1
        elif r < sub + ins:
            out.extend(rng.integers(0, 4, size=int(rng.integers(1, 4))).tolist())
        elif r < sub + ins + dele:
            i += int(rng.integers(1, 4))
        else:
            out.append(int(seq[i]))
            i += 1
    if not out:
        out = [0]
    return np.array(out, dtype=np.int8)


def make_case(rng, idx):
    kind = idx % 8
    ref_len = int(rng.choice([30, 64, 100, 257, 400, 600, 1000, 1500]))
    if kind == 7:
        ref_len = int(rng.integers(1800, 2400))
    ref = rng.integers(0, 4, size=ref_len).astype(np.int8)
    # read = mutated window of ref, sometimes with random flanks
    lo = int(rng.integers(0, max(1, ref_len // 2)))
    hi = int(rng.integers(lo + min(16, ref_len - lo), ref_len + 1))
    window = ref[lo:hi]
    if kind == 0:
        read = window.copy()
    elif kind == 1:
        read = mutate(rng, window, 0.02, 0.01, 0.01)
    elif kind == 2:
        read = mutate(rng, window, 0.08, 0.05, 0.05)
    elif kind == 3:
        read = mutate(rng, window, 0.03, 0.02, 0.02, long_indel=0.01)
    elif kind == 4:
        read = np.concatenate([rng.integers(0, 4, size=int(rng.integers(1, 30))).astype(np.int8),
                               mutate(rng, window, 0.05, 0.03, 0.03),
                               rng.integers(0, 4, size=int(rng.integers(1, 30))).astype(np.int8)])
    elif kind == 5:
        read = rng.integers(0, 4, size=int(rng.integers(15, 200))).astype(np.int8)  # unrelated
    elif kind == 6:
        read = mutate(rng, window, 0.05, 0.02, 0.02)
        npos = rng.integers(0, len(read), size=max(1, len(read) // 25))
        read[npos] = 4
        rpos = rng.integers(0, ref_len, size=max(1, ref_len // 40))
        ref[rpos] = 4
    else:
        read = mutate(rng, window, 0.04, 0.03, 0.03, long_indel=0.004)
    read = read[:2048]
    qlen = len(read)
    flag = int(rng.choice([0, 1, 2, 8, 15, 2, 2, 15]))
    filters = int(rng.choice([0, 0, 0, 50, 300]))
    filterd = int(rng.choice([0, 100, 5000]))
    mask = 15 if qlen <= 30 else qlen
    r = rng.random()
    if r < 0.15:
        mask = int(rng.integers(0, 15))
    elif r < 0.4:
        mask = max(15, qlen // 2)
    score_size = int(rng.choice([2, 2, 2, 2, 1, 0]))
    n_zero = bool(rng.random() < 0.6)
    # (2, 30, 3, 1): mismatch so costly that insertion-then-deletion beats it - the one regime where the
    # striped passes' 'E is opened from H before the lazy-F lift' rule changes results
    scoring = [(4, 6, 8, 2), (4, 6, 8, 2), (2, 2, 3, 1), (1, 3, 5, 2), (2, 4, 4, 2), (2, 30, 3, 1),
               (3, 40, 4, 2)][int(rng.integers(0, 7))]
    return dict(read=read, ref=ref, flag=flag, filters=filters, filterd=filterd, mask=mask,
                score_size=score_size, mat=build_matrix(scoring[0], scoring[1], n_zero),
                gap_open=scoring[2], gap_extend=scoring[3])


def make_cases(seed, n):
    rng = np.random.default_rng(seed)
    return [make_case(rng, i) for i in range(n)]
