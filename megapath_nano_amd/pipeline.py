"""Integer-coded fast path of the reference's step 3 (`step_placement_to_species`,
/root/reference/bin/megapath_nano.py:1253-1310): Align (:1262) -> Reassign (:1281) -> best alignment per read
(:1287) -> aligned bp per species (:1289) -> reads per name (:3664-3667), without going through DataFrames.
The DataFrame-level mirrors (aligner.Align, reassignment.Reassign) call the same C-ABI entry points; this module is
what bench.py times and what a multi-GPU run executes per rank.
"""
import random

import numpy as np

from . import mapper
from .reassignment import ReassignPlan


def random_block(rnd, n):
    """The next n values of `rnd.random()` (a random.Random, or the random module) as a float64 array, leaving `rnd` where
    n calls would have left it.  Python's generator and numpy's RandomState are the same MT19937 with the same 53-bit
    double construction, so the state is lent to numpy for the block instead of making n Python-level calls."""
    ver, st, gauss = rnd.getstate()
    rs = np.random.RandomState()
    rs.set_state(('MT19937', np.array(st[:-1], dtype=np.uint32), int(st[-1])))
    out = rs.random_sample(int(n))
    _, key, pos = rs.get_state()[:3]
    rnd.setstate((ver, tuple(int(x) for x in key) + (int(pos),), gauss))
    return out


def sharded_tiebreak(rnd, n_rows, shard, allreduce):
    """Tiebreakers for this rank's rows such that the concatenation over ranks equals ONE `random.random()` stream drawn
    in global row order (the reference draws them row by row over the whole table, aligner.py:334-335).  Every rank
    seeds `rnd` identically; ranks hold contiguous read ranges in rank order; the row counts are exchanged as one small
    all-reduce, the draws of the ranks before this one are skipped and those of the ranks after it consumed, so that
    the generator is in the same state on every rank afterwards.  shard = (rank, world)."""
    rank, world = shard
    if world <= 1 or allreduce is None:
        return random_block(rnd, n_rows)
    counts = np.zeros(world, dtype=np.int64)
    counts[rank] = n_rows
    allreduce(counts)
    before, after = int(counts[:rank].sum()), int(counts[rank + 1:].sum())
    random_block(rnd, before)
    out = random_block(rnd, n_rows)
    random_block(rnd, after)
    return out


class Taxonomy:
    """Per target sequence: dense species-name code (reassignment.py:69-71) and dense species_tax_id code."""

    def __init__(self, name_code, n_names, species_code, n_species):
        self.name_code = np.asarray(name_code, dtype=np.int32)
        self.species_code = np.asarray(species_code, dtype=np.int32)
        self.n_names, self.n_species = int(n_names), int(n_species)


def align_and_assign(idx, opt, packed, tax, error_rate=0.05, ratio=0.05, as_threshold=0.0, min_alignment_score=0,
                     allreduce=None, rng=None, reassign=True, shard=(0, 1), use_device=True):
    """One step of the hot path for one batch of reads.  Returns dict(read_count, aligned_bp, n_rows, n_relations).
    With shard=(rank, world) and an all-reduce, `rng` must be seeded identically on every rank (see sharded_tiebreak)."""
    if isinstance(idx, (list, tuple)):
        # a target set held as several index parts (minimap2 -I): every part is mapped, the hits are merged per read like
        # minimap2 --split-prefix merges them (mapper.Hits); column `rid` indexes the concatenated target list
        hits = mapper.Hits(packed, want_text=False)
        try:
            hits.add_parts(list(idx), opt, use_device=use_device)      # all parts are resident: one call, one pipeline
            _, _, c = hits.finish(opt, want_paf=False, want_cols=True)
        finally:
            hits.close()
    else:
        _, c = mapper.map_batch_ex(idx, opt, packed, want_paf=False, want_cols=True, use_device=use_device)
    keep = c['as_'] >= min_alignment_score                                   # aligner.py:311-312
    read_idx = c['read_idx'][keep]
    rid = c['rid'][keep]
    score = c['as_'][keep]
    aligned_bp = (c['re'][keep] - c['rs'][keep]).astype(np.int64)
    n_rows = len(read_idx)
    tiebreak = sharded_tiebreak(rng if rng is not None else random, n_rows, shard, allreduce)  # aligner.py:334-335
    read_count = np.zeros(tax.n_names, dtype=np.int64)
    bp = np.zeros(tax.n_species, dtype=np.int64)
    nrel = 0
    if n_rows:
        plan = ReassignPlan(read_idx, tax.name_code[rid], score, tiebreak, aligned_bp, tax.species_code[rid], tax.n_names,
                            tax.n_species)
        try:
            all_count, u_count, n_multi = plan.counts()
            if allreduce is not None:
                allreduce(all_count), allreduce(u_count), allreduce(n_multi)
            rank = np.arange(tax.n_names, dtype=np.int32)
            if not reassign:  # --no-reassignment: an empty relation leaves every name untouched
                error_rate = -1.0
            _, _, _, read_count, bp, nrel = plan.apply(all_count, u_count, rank, error_rate, ratio, as_threshold)
        finally:
            plan.close()
    elif allreduce is not None:
        z = np.zeros(tax.n_names, dtype=np.int64)
        allreduce(z), allreduce(z.copy()), allreduce(np.zeros(1, dtype=np.int64))
    if allreduce is not None:
        allreduce(read_count), allreduce(bp)                                 # the taxon-count all-reduce
    return dict(read_count=read_count, aligned_bp=bp, n_rows=n_rows, n_relations=nrel)
