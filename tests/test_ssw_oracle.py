"""CPU tests: the SSW oracle restatement against (a) golden vectors produced by the reference's own
ssw.c and (b), when oracle/_ref/libssw.so is present, the compiled reference live on seeded cases."""
import json
import os

import numpy as np
import pytest

from ssw_cases import make_cases

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = {c: i for i, c in enumerate('ACGTN')}


def load_golden():
    with open(os.path.join(HERE, 'golden', 'ssw_golden.json')) as f:
        return json.load(f)['cases']


def golden_kwargs(c):
    return dict(read=np.array([CODE[x] for x in c['read']], dtype=np.int8),
                ref=np.array([CODE[x] for x in c['ref']], dtype=np.int8), mat=np.array(c['mat'], dtype=np.int8),
                gap_open=c['gap_open'], gap_extend=c['gap_extend'], flag=c['flag'], filters=c['filters'],
                filterd=c['filterd'], mask=c['mask'], score_size=c['score_size'])


def expect_tuple(e):
    if e is None:
        return None
    return (e['score1'], e['score2'], e['ref_begin1'], e['ref_end1'], e['read_begin1'], e['read_end1'], e['ref_end2'],
            e['cigar'])


def test_oracle_matches_golden(oracle_built):
    from oracle.ssw_bindings import oracle_align
    cases = load_golden()
    assert len(cases) >= 100
    for c in cases:
        assert oracle_align(**golden_kwargs(c)) == expect_tuple(c['expect'])


def test_oracle_matches_compiled_reference(oracle_built):
    from oracle import ssw_bindings as sb
    if not sb.have_ref():
        pytest.skip('oracle/_ref/libssw.so not built (reference tree absent)')
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference chats on stderr for maskLen < 15
    try:
        for c in make_cases(4242, 600):
            kw = dict(read=c['read'], ref=c['ref'], mat=c['mat'], gap_open=c['gap_open'], gap_extend=c['gap_extend'],
                      flag=c['flag'], filters=c['filters'], filterd=c['filterd'], mask=c['mask'],
                      score_size=c['score_size'])
            assert sb.oracle_align(**kw) == sb.ref_align(**kw)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)


def test_oracle_domain_guards(oracle_built):
    from oracle.ssw_bindings import oracle_align
    c = make_cases(1, 1)[0]
    kw = dict(read=c['read'], ref=c['ref'], mat=c['mat'], gap_open=2, gap_extend=2, flag=0, filters=0, filterd=0,
              mask=15, score_size=2)
    assert oracle_align(**kw) == 'unsupported'
    kw.update(gap_open=8, read=np.zeros(0, dtype=np.int8))
    assert oracle_align(**kw) == 'undefined'
