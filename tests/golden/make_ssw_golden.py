"""Generates tests/golden/ssw_golden.json: inputs + outputs of the REFERENCE's own SSW
(/root/reference/bin/realignment/realign/ssw.c compiled in place to oracle/_ref/libssw.so by
oracle/Makefile, driven with the pyssw.py:30-48 prototypes).  Run in the build container only:

    make -C oracle && python tests/golden/make_ssw_golden.py

The JSON is data (inputs and expected outputs); no reference source text is stored.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ssw_cases import make_cases  # noqa: E402
from oracle.ssw_bindings import ref_align  # noqa: E402

ALPHA = 'ACGTN'


def main():
    out = []
    for c in make_cases(20240901, 160):
        if len(c['read']) > 700 or len(c['ref']) > 1100:
            continue
        res = ref_align(read=c['read'], ref=c['ref'], mat=c['mat'], gap_open=c['gap_open'],
                        gap_extend=c['gap_extend'], flag=c['flag'], filters=c['filters'], filterd=c['filterd'],
                        mask=c['mask'], score_size=c['score_size'])
        out.append(dict(read=''.join(ALPHA[x] for x in c['read']), ref=''.join(ALPHA[x] for x in c['ref']),
                        mat=[int(x) for x in c['mat']], gap_open=c['gap_open'], gap_extend=c['gap_extend'],
                        flag=c['flag'], filters=c['filters'], filterd=c['filterd'], mask=c['mask'],
                        score_size=c['score_size'],
                        expect=None if res is None else dict(score1=res[0], score2=res[1], ref_begin1=res[2],
                                                             ref_end1=res[3], read_begin1=res[4], read_end1=res[5],
                                                             ref_end2=res[6], cigar=res[7])))
    with open(os.path.join(HERE, 'ssw_golden.json'), 'w') as f:
        json.dump(dict(source='reference ssw.c via oracle/_ref/libssw.so', cases=out), f, separators=(',', ':'))
    print('wrote', len(out), 'cases')


if __name__ == '__main__':
    main()
