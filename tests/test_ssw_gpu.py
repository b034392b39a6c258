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
             np.concatenate([ref, ref, ref, ref, ref[:48]])[:2048], np.zeros(0, dtype=np.int8),
             np.zeros(2049, dtype=np.int8)]
    res = ssw_align_batch(reads, [ref] * len(reads), mat, 5, 2, 8, 2, 15, 0, 0, [15] * len(reads))
    assert res[-1]['status'] == 4  # too long for this round
    assert res[-2]['status'] == STATUS_UNDEF  # empty read: reference indexes pvHStore[-1]
    for r, q in zip(res[:-2], reads[:-2]):
        want = oracle_align(read=q, ref=ref, mat=mat, gap_open=8, gap_extend=2, flag=15, filters=0, filterd=0, mask=15,
                            score_size=2)
        assert as_tuple(r) == want
    # refused domain
    res = ssw_align_batch(reads[:1], [ref], mat, 5, 2, 2, 2, 0, 0, 0, [15])
    assert res[0]['status'] == STATUS_DOMAIN
