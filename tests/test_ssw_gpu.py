"""GPU parity tests (run with -m gpu on an MI355X): the HIP SSW path, called through the C-ABI,
against the CPU oracle and the committed golden vectors from the reference's ssw.c.  Bit-exact."""
import numpy as np
import pytest

from ssw_cases import make_cases
from test_ssw_oracle import load_golden, golden_kwargs, expect_tuple

pytestmark = pytest.mark.gpu

STATUS_NULL, STATUS_UNDEF, STATUS_DOMAIN = 1, 2, 3


def as_tuple(r):
    if r['status'] == STATUS_NULL:
        return None
    if r['status'] == STATUS_UNDEF:
        return 'undefined'
    if r['status'] == STATUS_DOMAIN:
        return 'unsupported'
    assert r['status'] == 0, r
    return (r['score1'], r['score2'], r['ref_begin1'], r['ref_end1'], r['read_begin1'], r['read_end1'], r['ref_end2'],
            r['cigar'])


def run_batch_grouped(cases):
    """The batch entry point takes one scoring/flag set per call: group cases by their scalar params."""
    from megapath_nano_amd.ssw_batch import ssw_align_batch
    groups = {}
    for idx, c in enumerate(cases):
        key = (bytes(c['mat']), c['gap_open'], c['gap_extend'], c['flag'], c['filters'], c['filterd'], c['score_size'])
        groups.setdefault(key, []).append(idx)
    out = [None] * len(cases)
    for key, idxs in groups.items():
        c0 = cases[idxs[0]]
        res = ssw_align_batch([cases[i]['read'] for i in idxs], [cases[i]['ref'] for i in idxs], c0['mat'], 5,
                              c0['score_size'], c0['gap_open'], c0['gap_extend'], c0['flag'], c0['filters'],
                              c0['filterd'], [cases[i]['mask'] for i in idxs])
        for i, r in zip(idxs, res):
            out[i] = as_tuple(r)
    return out


def test_batch_matches_golden_reference_vectors(libmpn):
    cases = load_golden()
    kws = [golden_kwargs(c) for c in cases]
    got = run_batch_grouped(kws)
    for c, g in zip(cases, got):
        assert g == expect_tuple(c['expect'])


def test_batch_matches_oracle_seeded(libmpn, oracle_built):
    from oracle.ssw_bindings import oracle_align
    cases = make_cases(777, 1200)
    got = run_batch_grouped(cases)
    for c, g in zip(cases, got):
        want = oracle_align(read=c['read'], ref=c['ref'], mat=c['mat'], gap_open=c['gap_open'],
                            gap_extend=c['gap_extend'], flag=c['flag'], filters=c['filters'], filterd=c['filterd'],
                            mask=c['mask'], score_size=c['score_size'])
        assert g == want, (len(c['read']), len(c['ref']), c['flag'], c['mask'], c['score_size'])


def test_reference_abi_single_calls(libmpn, oracle_built):
    """ssw_init / ssw_align / align_destroy with the pyssw.py prototypes, one pair per call."""
    from oracle import ssw_bindings as sb
    lib = sb.bind_ssw_abi(libmpn)
    for c in make_cases(31337, 40):
        kw = dict(read=c['read'], ref=c['ref'], mat=c['mat'], gap_open=c['gap_open'], gap_extend=c['gap_extend'],
                  flag=c['flag'], filters=c['filters'], filterd=c['filterd'], mask=c['mask'], score_size=c['score_size'])
        assert sb.ssw_abi_align(lib, **kw) == sb.oracle_align(**kw)


def test_pyssw_mirror_matches_reference_wrapper_golden(libmpn):
    """tests/golden/pyssw_golden.json: outputs of the reference's pyssw.SSW.align (score, cigar string with
    `=`/soft clips, ref_begin) on the reference's compiled ssw.c."""
    import json
    import os
    from megapath_nano_amd.pyssw import SSW
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pyssw_golden.json')) as f:
        sets = json.load(f)['sets']
    for st in sets:
        s = SSW()
        s.set_reference_sequence(st['reference'])
        want = [(q['score'], q['cigar'], q['ref_begin']) for q in st['queries']]
        assert [s.align(q['query']) for q in st['queries'][:3]] == want[:3]
        assert s.align_batch([q['query'] for q in st['queries']]) == want


def test_edge_cases(libmpn, oracle_built):
    from megapath_nano_amd.ssw_batch import ssw_align_batch
    from oracle.ssw_bindings import oracle_align
    from ssw_cases import build_matrix
    mat = build_matrix()
    rng = np.random.default_rng(9)
    ref = rng.integers(0, 4, size=500).astype(np.int8)
    reads = [ref[:1].copy(), ref[100:116].copy(), np.full(40, 4, dtype=np.int8), ref[:2048 // 4].copy(),
             np.concatenate([ref, ref, ref, ref, ref[:48]])[:2048], np.zeros(2049, dtype=np.int8), np.zeros(0, dtype=np.int8),
             np.zeros(64 * 2048 + 1, dtype=np.int8)]
    res = ssw_align_batch(reads, [ref] * len(reads), mat, 5, 2, 8, 2, 15, 0, 0, [15] * len(reads))
    assert res[-1]['status'] == 4  # longer than 64 strips of 2048 rows
    assert res[-2]['status'] == STATUS_UNDEF  # empty read: reference indexes pvHStore[-1]
    for r, q in zip(res[:-2], reads[:-2]):
        want = oracle_align(read=q, ref=ref, mat=mat, gap_open=8, gap_extend=2, flag=15, filters=0, filterd=0, mask=15,
                            score_size=2)
        assert as_tuple(r) == want
    # refused domain
    res = ssw_align_batch(reads[:1], [ref], mat, 5, 2, 2, 2, 0, 0, 0, [15])
    assert res[0]['status'] == STATUS_DOMAIN


def test_reads_longer_than_2048_match_oracle_and_reference(libmpn, oracle_built):
    """ssw.c has no read-length limit (ssw_align :762-852); the strip kernel cuts the read into 2048-row strips.  Pairs with
    reads of 2049 .. 9000 bases (boundaries of the strips and of the 16/8-row SSE padding), both score passes, all flag sets,
    against the oracle and, where the box has it, the reference's own compiled ssw.c (oracle/_ref/libssw.so)."""
    import os
    from oracle.ssw_bindings import oracle_align
    from ssw_cases import build_matrix, mutate
    rng = np.random.default_rng(4242)
    cases = []
    for qlen, reflen, kind in ((2049, 2600, 1), (2064, 3000, 1), (4096, 4500, 2), (4097, 4200, 1), (4100, 900, 3), (6000, 7000, 1),
                               (6150, 6100, 0), (9000, 2500, 4), (2500, 9000, 1), (3000, 3300, 5)):
        ref = rng.integers(0, 4, size=reflen).astype(np.int8)
        if kind == 0:
            read = np.resize(ref, qlen).copy()                       # exact (tandem continuation past the end)
        elif kind == 3:
            read = np.concatenate([rng.integers(0, 4, size=(qlen - reflen) // 2).astype(np.int8), mutate(rng, ref, 0.02, 0.01, 0.01),
                                   rng.integers(0, 4, size=qlen).astype(np.int8)])[:qlen]  # the reference inside a longer read
        elif kind == 4:
            read = np.concatenate([rng.integers(0, 4, size=5000).astype(np.int8), mutate(rng, ref[200:2300], 0.03, 0.02, 0.02),
                                   rng.integers(0, 4, size=qlen).astype(np.int8)])[:qlen]  # hit deep inside the third strip
        elif kind == 5:
            read = rng.integers(0, 4, size=qlen).astype(np.int8)                          # unrelated
        else:
            src = np.resize(ref, max(qlen + 200, reflen))
            read = mutate(rng, src[:qlen + 150], 0.01 * kind, 0.01, 0.01)[:qlen]
            if kind == 2:
                read[rng.integers(0, qlen, size=40)] = 4
        for flag, score_size, mask in ((2, 2, qlen), (15, 2, qlen // 2), (0, 1, 15), (8, 2, 10), (1, 0, qlen)):
            cases.append(dict(read=read, ref=ref, flag=flag, filters=0, filterd=0, mask=mask, score_size=score_size,
                              mat=build_matrix(2, 3, True), gap_open=4, gap_extend=1))
    got = run_batch_grouped(cases)
    ref_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'libssw.so')
    n_checked_ref = 0
    for c, g in zip(cases, got):
        kw = dict(read=c['read'], ref=c['ref'], mat=c['mat'], gap_open=c['gap_open'], gap_extend=c['gap_extend'], flag=c['flag'],
                  filters=c['filters'], filterd=c['filterd'], mask=c['mask'], score_size=c['score_size'])
        want = oracle_align(**kw)
        assert g == want, (len(c['read']), len(c['ref']), c['flag'], c['score_size'])
        if os.path.exists(ref_lib):
            from oracle.ssw_bindings import ref_align
            assert ref_align(**kw) == want
            n_checked_ref += 1
    assert any(g is not None and g != 'undefined' and g[0] > 2000 for g in got)
