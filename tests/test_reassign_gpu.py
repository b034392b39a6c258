"""GPU parity tests (-m gpu): the HIP Reassign path through the C-ABI + Python mirror against the golden
outputs of the reference's reassignment.Reassign and, at larger sizes, against the CPU oracle."""
import os

import numpy as np
import pandas as pd
import pytest

from reassign_cases import SPECIES, community
from test_reassign_oracle import check_against_golden, load_golden

pytestmark = pytest.mark.gpu


def write_db(tmp_path):
    db = tmp_path / 'db'
    db.mkdir()
    with open(db / 'sequence_name', 'w') as f:
        for sid, desc in SPECIES:
            f.write(f'{sid}\t{desc}\n')
    return str(db)


def run_mirror(table, db, level, params, tmp_path, **kw):
    from megapath_nano_amd.reassignment import Reassign
    df = pd.DataFrame(table)
    stats = {}
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        out = Reassign(df, db, threads=2, level=level, stats=stats, **params, **kw)
    finally:
        os.chdir(cwd)
    out = out.sort_index()
    # map merged-row index back to input rows: merge is inner on sequence_id with a 1:1 name table here
    known = {sid for sid, _ in SPECIES}
    src_of = [i for i, s in enumerate(table['sequence_id']) if s in known]
    rows = [dict(index=int(ix), src=src_of[int(ix)], name=nm, sequence_id=sid,
                 is_in_explain_other=(io if 'is_in_explain_other' in out else None))
            for ix, nm, sid, io in zip(out.index, out['name'], out['sequence_id'],
                                       out['is_in_explain_other'] if 'is_in_explain_other' in out else [None] * len(out))]
    return dict(explains=stats.get('explains'), rows=rows), stats


@pytest.mark.parametrize('idx', range(7))
def test_mirror_matches_reference_golden(libmpn, tmp_path, idx):
    c, table = load_golden()[idx]
    db = write_db(tmp_path)
    holder = {}

    def res_fn():
        res, stats = run_mirror(table, db, c['level'], c['params'], tmp_path)
        holder['stats'] = stats
        return res

    check_against_golden(c, table, res_fn,
                         lambda rows: holder['stats']['read_count_by_name'],
                         lambda rows: holder['stats']['aligned_bp_by_species'])
    if not c['expect'].get('exception'):
        assert os.path.exists(tmp_path / 'alignlist_reassigned.csv') == c['expect']['wrote_csv']
        assert os.path.exists(tmp_path / 'i_explains_j_dict.pickle') == (c['expect']['explains'] is not None)


@pytest.mark.parametrize('order', ['alphabetical', 'frequency'])
def test_mirror_matches_oracle_large(libmpn, oracle_built, tmp_path, order):
    from oracle import reassign_oracle as ro
    table = community(99, 20000, [50, 30, 2, 1, 40, 30, 1, 25, 5, 1, 1, 0],
                      {0: [(2, 0.4), (3, 0.3), (1, 0.5), (4, 0.1)], 1: [(0, 0.5), (2, 0.3)], 2: [(0, 0.9), (4, 0.5)],
                       3: [(0, 0.8), (4, 0.6)], 4: [(2, 0.2), (3, 0.2), (10, 0.1)], 5: [(6, 0.5)],
                       6: [(5, 0.9), (7, 0.5)], 7: [(6, 0.1)], 8: [(0, 0.05)], 9: [(7, 0.9)],
                       10: [(4, 0.9), (0, 0.5), (7, 0.5)]}, extra_rows=0.2, unknown_seq=True)
    db = write_db(tmp_path)
    want = ro.reassign_oracle(table, SPECIES, level='strain', explainer_order=order)
    got, stats = run_mirror(table, db, 'strain', {}, tmp_path, explainer_order=order, side_files=False)
    assert got['explains'] == want['explains']
    assert [(x['index'], x['name'], x['sequence_id'], bool(x['is_in_explain_other'])) for x in got['rows']] == \
           [(x['index'], x['name'], x['sequence_id'], bool(x['is_in_explain_other'])) for x in want['rows']]
    assert stats['read_count_by_name'] == ro.read_count_by_name(table, want['rows'])
    assert stats['aligned_bp_by_species'] == ro.aligned_bp_by_species(table, want['rows'])
    # size-independent property: every read is counted exactly once
    assert sum(stats['read_count_by_name'].values()) == len({r for r, s in zip(table['read_id'], table['sequence_id'])
                                                             if s != 'NZ_NOT_IN_DB.1'})


class ThreadAllreduce:
    """Summing all-reduce between threads of one process: stands in for RCCL so that the two-shard
    decomposition can be checked on the single GPU of the test box."""

    def __init__(self, n):
        import threading
        self.barrier = threading.Barrier(n)
        self.lock = threading.Lock()
        self.acc = None

    def __call__(self, arr):
        with self.lock:
            self.acc = arr.copy() if self.acc is None else self.acc + arr
        self.barrier.wait()
        arr[:] = self.acc
        if self.barrier.wait() == 0:
            self.acc = None
        self.barrier.wait()


def test_sharded_reads_with_allreduce_equal_global_run(libmpn, tmp_path):
    """Reads split over two 'ranks' + count all-reduce == one global run (the multi-GPU decomposition)."""
    import threading
    from megapath_nano_amd.reassignment import Reassign
    table = community(5, 3000, [50, 0, 2, 1, 30, 0, 0, 0, 0, 0, 0, 0],
                      {0: [(2, 0.5), (3, 0.3)], 2: [(0, 0.9)], 3: [(0, 0.9)]})
    db = write_db(tmp_path)
    df = pd.DataFrame(table)
    g = {}
    full = Reassign(df, db, stats=g, side_files=False)
    reads = sorted(set(table['read_id']))
    half = set(reads[:len(reads) // 3])
    parts = [df[df['read_id'].isin(half)], df[~df['read_id'].isin(half)]]
    universe = sorted(set(table['species_tax_id']))
    ar = ThreadAllreduce(2)
    outs, stats, errs = [None, None], [{}, {}], []

    def work(k):
        try:
            outs[k] = Reassign(parts[k], db, stats=stats[k], side_files=False, allreduce=ar, species_universe=universe)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            ar.barrier.abort()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert stats[0]['read_count_by_name'] == stats[1]['read_count_by_name'] == g['read_count_by_name']
    assert stats[0]['aligned_bp_by_species'] == g['aligned_bp_by_species']
    both = pd.concat(outs)
    key = ['read_id', 'sequence_from', 'alignment_score']
    a = both.sort_values(key)[key + ['name', 'sequence_id']].reset_index(drop=True)
    b = full.sort_values(key)[key + ['name', 'sequence_id']].reset_index(drop=True)
    assert a.equals(b)
