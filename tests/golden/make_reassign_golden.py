"""Generates tests/golden/reassign_golden.json by IMPORTING the reference's
/root/reference/bin/lib/reassignment.py (Reassign, :66-108) in this container (pandas 2.3.3) and running
it on the seeded tables of tests/reassign_cases.py.  Only inputs and outputs are stored.

    python tests/golden/make_reassign_golden.py
"""
import hashlib
import json
import os
import pickle
import sys
import tempfile

import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference/bin/lib')
import reassignment  # noqa: E402  (the reference module)
from reassign_cases import SPECIES, boundary_case, cases  # noqa: E402


def run(case):
    df = pd.DataFrame(case['table'])
    with tempfile.TemporaryDirectory() as tmp:
        db = os.path.join(tmp, 'db')
        os.makedirs(db)
        with open(os.path.join(db, 'sequence_name'), 'w') as f:
            for sid, desc in SPECIES:
                f.write(f'{sid}\t{desc}\n')
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            try:
                out = reassignment.Reassign(df, db, threads=2, level=case['level'], **case['params'])
            except TypeError as e:
                return dict(exception='TypeError', message=str(e))
            explains = None
            if os.path.exists('i_explains_j_dict.pickle'):
                with open('i_explains_j_dict.pickle', 'rb') as f:
                    explains = {k: sorted(v) for k, v in pickle.load(f).items()}
            wrote_csv = os.path.exists('alignlist_reassigned.csv')
        finally:
            os.chdir(cwd)
    out = out.sort_index()
    best = out.sort_values(['read_id', 'alignment_score', 'alignment_score_tiebreaker']).drop_duplicates(
        subset=['read_id'], keep='last')  # megapath_nano.py:1287
    counts = best.groupby(['name']).count()['read_id']  # megapath_nano.py:3666 (order of ties not pinned)
    aligned_bp = best.assign(aligned_bp=lambda x: x['sequence_to'] - x['sequence_from']).groupby(
        ['species_tax_id'])['aligned_bp'].sum()  # megapath_nano.py:1289
    return dict(explains=explains, wrote_csv=wrote_csv, columns=list(out.columns),
                index=[int(i) for i in out.index], name=list(out['name']), sequence_id=list(out['sequence_id']),
                alignment_score=[int(x) for x in out['alignment_score']],
                is_in_explain_other=[bool(x) for x in out['is_in_explain_other']] if 'is_in_explain_other' in out else None,
                read_count_by_name={k: int(v) for k, v in counts.items()},
                aligned_bp_by_species={str(int(k)): int(v) for k, v in aligned_bp.items()})


def main():
    out = []
    for case in cases() + [boundary_case()]:
        res = run(case)
        # the input table is regenerated from its seed by tests/reassign_cases.py; its digest pins it
        digest = hashlib.sha1(json.dumps(case['table'], sort_keys=True).encode()).hexdigest()
        out.append(dict(name=case['name'], level=case['level'], params=case['params'], table_sha1=digest, expect=res))
        print(case['name'], 'rows', len(case['table']['read_id']), '->',
              res.get('exception') or f"{len(res['index'])} rows, explains={res['explains']}")
    with open(os.path.join(HERE, 'reassign_golden.json'), 'w') as f:
        json.dump(dict(source='reference reassignment.Reassign imported in the build container (pandas %s)' % pd.__version__,
                       sequence_name=SPECIES, cases=out), f, separators=(',', ':'))


if __name__ == '__main__':
    main()
