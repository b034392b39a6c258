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


def pyssw_golden():
    """(score, cigar string, ref_begin) from the reference's own Python wrapper
    (/root/reference/bin/realignment/pyssw.py SSW.align) driving the compiled reference ssw.c."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location('ref_pyssw', '/root/reference/bin/realignment/pyssw.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    so = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle', '_ref', 'libssw.so')
    rng = np.random.default_rng(77)
    out = []
    for t in range(6):
        ref = ''.join('ACGTN'[x] for x in rng.choice(5, size=int(rng.integers(120, 500)), p=[.245, .245, .245, .245, .02]))
        s = mod.SSW(lib_path=so)
        s.set_reference_sequence(ref)
        qs = []
        for k in range(8):
            lo = int(rng.integers(0, len(ref) - 40))
            hi = int(rng.integers(lo + 20, min(len(ref), lo + 200)))
            q = list(ref[lo:hi])
            for _ in range(int(rng.integers(0, 6))):
                p = int(rng.integers(0, len(q)))
                r = rng.random()
                if r < 0.4:
                    q[p] = 'ACGT'[int(rng.integers(0, 4))]
                elif r < 0.7:
                    q.insert(p, 'acgtn'[int(rng.integers(0, 5))])
                else:
                    del q[p]
            if k == 7:
                q = list('TTAGGC') + q + list('x')  # soft clips + an unknown letter
            q = ''.join(q)
            score, cigar, beg = s.align(q)
            qs.append(dict(query=q, score=int(score), cigar=cigar, ref_begin=int(beg)))
        out.append(dict(reference=ref, queries=qs))
    with open(os.path.join(HERE, 'pyssw_golden.json'), 'w') as f:
        json.dump(dict(source='reference pyssw.SSW.align via oracle/_ref/libssw.so', sets=out), f, separators=(',', ':'))
    print('wrote pyssw golden', sum(len(x['queries']) for x in out))


if __name__ == '__main__':
    pyssw_golden()
