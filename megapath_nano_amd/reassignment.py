"""Host-side mirror of /root/reference/bin/lib/reassignment.py on top of libmpn.so.

`Reassign(align_list, db_folder, error_rate=0.05, ratio=0.05, threads=24, AS_threshold=0, level='species')`
has the reference's signature (:66), returns the same DataFrame (same columns incl. `name` and
`is_in_explain_other`, same index labels, rows in ascending alignment_score) and leaves the same side files in
the current directory (`i_explains_j_dict.pickle` :103-104, `alignlist_reassigned.csv` :107).  Strings are
factorised here; the dedupe, the counters, the per-read relabel and the per-name / per-species reductions run in
HIP kernels (csrc/reassign_kernels.hip) through include/mpn_reassign.h.  `threads` is accepted and ignored.

Extra keyword arguments (not in the reference):
  explainer_order  'alphabetical' (default: what the reference does on pandas >= 1.5) or
                   'frequency' (its behaviour on the pandas <= 1.1 it was written for)
  allreduce        callable(np.ndarray[int64]) -> None summing the array in place over all ranks; given when the
                   reads are sharded over GPUs (one process per GPU).  Called twice: counters, final counts.
                   Name codes are taken from db/sequence_name (identical on every rank); pass
                   species_universe (all species_tax_id of the target set) so species codes agree as well.
  side_files       write the two CWD files (default True, as the reference)
  stats            optional dict that receives read_count_by_name / aligned_bp_by_species / explains
"""
import ctypes as ct
import pickle

import numpy as np
import pandas as pd

from . import _ffi

_bound = False


def _bind():
    global _bound
    lib = _ffi.lib()
    if not _bound:
        P = ct.c_void_p
        lib.mpn_reassign_create.argtypes = [ct.c_int64, ct.c_int32, ct.c_int32, ct.c_int32, P, P, P, P, P, P,
                                            ct.POINTER(ct.c_void_p)]
        lib.mpn_reassign_create.restype = ct.c_int
        lib.mpn_reassign_counts.argtypes = [P, P, P, P]
        lib.mpn_reassign_counts.restype = ct.c_int
        lib.mpn_reassign_apply.argtypes = [P, P, P, P, ct.c_double, ct.c_double, ct.c_double, P, P, P, P, P, P]
        lib.mpn_reassign_apply.restype = ct.c_int
        lib.mpn_reassign_destroy.argtypes = [P]
        lib.mpn_reassign_destroy.restype = None
        _bound = True
    return lib


def species_name(desc, level):
    """reassignment.py:69-70"""
    if level != 'species':
        return desc
    if ' sp. ' not in desc:
        return ' '.join(desc.split(' ', 2)[0:2])
    return ' '.join(desc.split(' ', 3)[0:3])


class ReassignPlan:
    """Integer-coded rows resident in HBM; wraps mpn_reassign_create/counts/apply/destroy."""

    def __init__(self, read_code, name_code, score, tiebreak, aligned_bp, species_code, n_names, n_species):
        lib = _bind()
        self.lib = lib
        n = len(read_code)
        # group rows by read, keeping their relative order (stable)
        self.perm = np.argsort(read_code, kind='stable')
        rc = np.asarray(read_code)[self.perm]
        self.n_reads = int(rc.max()) + 1 if n else 0
        self.read_ptr = np.zeros(self.n_reads + 1, dtype=np.int64)
        if n:
            np.cumsum(np.bincount(rc, minlength=self.n_reads), out=self.read_ptr[1:])
        self.n_rows, self.n_names, self.n_species = n, int(n_names), int(n_species)
        arrs = [np.ascontiguousarray(np.asarray(a)[self.perm], dtype=dt) for a, dt in
                ((name_code, np.int32), (score, np.int32), (tiebreak, np.float64), (aligned_bp, np.int64),
                 (species_code, np.int32))]
        self._plan = ct.c_void_p()
        _ffi.check(lib.mpn_reassign_create(n, self.n_reads, self.n_names, self.n_species, self.read_ptr.ctypes.data,
                                           *[a.ctypes.data for a in arrs], ct.byref(self._plan)),
                   'mpn_reassign_create')

    def counts(self):
        all_count = np.zeros(self.n_names, dtype=np.int64)
        u_count = np.zeros(self.n_names, dtype=np.int64)
        n_multi = np.zeros(1, dtype=np.int64)
        _ffi.check(self.lib.mpn_reassign_counts(self._plan, all_count.ctypes.data, u_count.ctypes.data,
                                                n_multi.ctypes.data), 'mpn_reassign_counts')
        return all_count, u_count, n_multi

    def apply(self, all_count, u_count, name_rank, error_rate, ratio, as_threshold):
        keep = np.zeros(max(self.n_rows, 1), dtype=np.uint8)
        new_name = np.zeros(max(self.n_rows, 1), dtype=np.int32)
        explainer = np.zeros(self.n_names, dtype=np.uint8)
        read_count = np.zeros(self.n_names, dtype=np.int64)
        bp = np.zeros(self.n_species, dtype=np.int64)
        nrel = np.zeros(1, dtype=np.int64)
        all_count = np.ascontiguousarray(all_count, dtype=np.int64)
        u_count = np.ascontiguousarray(u_count, dtype=np.int64)
        name_rank = np.ascontiguousarray(name_rank, dtype=np.int32)
        _ffi.check(self.lib.mpn_reassign_apply(self._plan, all_count.ctypes.data, u_count.ctypes.data,
                                               name_rank.ctypes.data, float(error_rate), float(ratio),
                                               float(as_threshold), keep.ctypes.data, new_name.ctypes.data,
                                               explainer.ctypes.data, read_count.ctypes.data, bp.ctypes.data,
                                               nrel.ctypes.data), 'mpn_reassign_apply')
        # back to the caller's row order
        inv_keep = np.zeros(self.n_rows, dtype=bool)
        inv_name = np.zeros(self.n_rows, dtype=np.int32)
        inv_keep[self.perm] = keep[:self.n_rows].astype(bool)
        inv_name[self.perm] = new_name[:self.n_rows]
        return inv_keep, inv_name, explainer.astype(bool), read_count, bp, int(nrel[0])

    def close(self):
        if self._plan:
            self.lib.mpn_reassign_destroy(self._plan)
            self._plan = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def explains_dict(names, all_count, u_count, explainer, error_rate, ratio):
    """The relation as the reference pickles it ({i: {j, ...}}), from the counters (reassignment.py:27-36)."""
    present = np.flatnonzero(all_count > 0)
    out = {}
    for i in np.flatnonzero(explainer):
        thr = error_rate * float(u_count[i])
        js = [names[j] for j in present if j != i and float(u_count[j]) < thr]
        if js:
            out[names[i]] = set(js)
    return out


def Reassign(align_list, db_folder, error_rate=0.05, ratio=0.05, threads=24, AS_threshold=0, level='species',
             explainer_order='alphabetical', allreduce=None, species_universe=None, side_files=True, stats=None):
    taxon_df = pd.read_csv(f'{db_folder}/sequence_name', sep='\t', header=None, names=['sequence_id', 'name'])
    if level == 'species':
        uniq = {d: species_name(d, level) for d in taxon_df['name'].unique()}
        taxon_df['name'] = taxon_df['name'].map(uniq)
    merged = align_list.merge(right=taxon_df, on=['sequence_id'], how='inner')                       # :71
    read_code, _ = pd.factorize(merged['read_id'], sort=False)
    # name codes come from the database, not from the observed rows, so that every rank of a sharded run
    # agrees on them; sorted => code order is alphabetical order
    names = sorted(taxon_df['name'].unique())
    name_code = pd.Categorical(merged['name'], categories=names).codes.astype(np.int32)
    if species_universe is None:
        sp_code, sp_ids = pd.factorize(merged['species_tax_id'], sort=True)
    else:
        sp_ids = np.array(sorted(set(int(x) for x in species_universe)))
        sp_code = np.searchsorted(sp_ids, merged['species_tax_id'].to_numpy())
        if len(sp_code) and (sp_code.max() >= len(sp_ids) or
                             np.any(sp_ids[sp_code] != merged['species_tax_id'].to_numpy())):
            raise KeyError('species_tax_id outside species_universe')
    score = merged['alignment_score'].to_numpy()
    if len(score) and (score.max() > 2**31 - 1 or score.min() < -2**31):
        raise OverflowError('alignment_score does not fit int32')
    plan = ReassignPlan(read_code, name_code, score, merged['alignment_score_tiebreaker'].to_numpy(dtype=np.float64),
                        (merged['sequence_to'] - merged['sequence_from']).to_numpy(dtype=np.int64), sp_code,
                        max(len(names), 1), max(len(sp_ids), 1))
    try:
        all_count, u_count, n_multi = plan.counts()
        if allreduce is not None:
            allreduce(all_count), allreduce(u_count), allreduce(n_multi)
        if int(n_multi[0]) == 0:
            # reassignment.py:91: functools.reduce over an empty result list
            raise TypeError('reduce() of empty iterable with no initial value')
        if explainer_order == 'alphabetical':
            rank = np.arange(len(names), dtype=np.int32)
        elif explainer_order == 'frequency':
            rank = np.empty(len(names), dtype=np.int32)
            rank[np.argsort(-all_count, kind='stable')] = np.arange(len(names), dtype=np.int32)
        else:
            raise ValueError('explainer_order must be alphabetical or frequency')
        keep, new_name, explainer, read_count, bp, nrel = plan.apply(all_count, u_count, rank, error_rate, ratio,
                                                                     AS_threshold)
    finally:
        plan.close()
    if allreduce is not None:
        allreduce(read_count), allreduce(bp)
    if stats is not None:
        stats['read_count_by_name'] = {names[i]: int(c) for i, c in enumerate(read_count) if c > 0}
        stats['aligned_bp_by_species'] = {int(sp_ids[i]): int(c) for i, c in enumerate(bp) if c != 0}
        stats['n_relations'] = nrel
    # rows in ascending score like :73 (ties: input order; the reference's own tie order is unspecified)
    order = np.argsort(score, kind='stable')
    order = order[keep[order]]
    out = merged.iloc[order].copy()
    if nrel == 0:
        if stats is not None:
            stats['explains'] = None
        return out                                                                                   # :100-101
    explains = explains_dict(names, all_count, u_count, explainer, error_rate, ratio)
    if stats is not None:
        stats['explains'] = {k: sorted(v) for k, v in explains.items()}
    if side_files:
        with open('i_explains_j_dict.pickle', 'wb') as f:                                            # :103-104
            pickle.dump(explains, f)
    oc, nc = name_code[order], new_name[order]
    name_arr = np.array(names, dtype=object)
    out['is_in_explain_other'] = explainer[oc]                                                      # :57
    changed = nc != oc
    if changed.any():
        new_names = name_arr[nc[changed]]
        out.loc[out.index[changed], 'name'] = new_names
        out.loc[out.index[changed], 'sequence_id'] = [f'{n}_reassigned' for n in new_names]         # :56
    if side_files:
        out.to_csv('alignlist_reassigned.csv')                                                       # :107
    return out
