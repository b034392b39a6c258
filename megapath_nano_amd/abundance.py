"""Per-assembly abundance statistic over an alignment table.

Reference: `align_list_to_align_stat_by_assembly_id` (/root/reference/bin/megapath_nano.py:485-541) with its helpers
`summary_stat_1` (:440-449), `align_list_to_bed` (:313-330: bedtools sort + merge of (assembly, sequence) intervals),
`bed_to_covered_bp_by_assembly_id` (:333-347) and `summary_stat_2` (:451-482), including the noise-BED branch
(`covered_bed.subtract(noise_bed)` and `noise_span_bp`, :516-541; the noise filters that produce such a BED are off by default,
megapath_nano.py:4985-4996).

Steps: per (read, assembly) keep the alignment with the largest (alignment_score, tiebreaker); sum per assembly; covered bp
= length of the union of the kept alignments' target intervals per (assembly, sequence), where overlapping AND
book-ended intervals merge (`bedtools merge` default distance 0), minus what a noise BED covers; then the derived columns of
summary_stat_2, with the reference's inf -> nan -> 0 clean-up and the rounded `adjusted_total_aligned_bp`.

The interval union runs on the GPU (include/mpn_abundance.h: mpn_cover_by_group -- a radix sort and a segmented sweep;
`device=True`, the default when libmpn.so can reach a GPU); the numpy form below (`device=False`) is the host statement of
the same sums and is what the CPU tests pin against plain loops.  `device_sort_order` is the same sort behind
bam.sam_to_sorted_bam (`samtools sort`).
"""
import ctypes as ct

import numpy as np
import pandas

from . import _ffi

_bound = False


def _lib():
    global _bound
    lib = _ffi.lib()
    if not _bound:
        P = ct.c_void_p
        lib.mpn_sort_order.argtypes = [ct.c_int64, P, P, P]
        lib.mpn_sort_order.restype = ct.c_int
        lib.mpn_cover_by_group.argtypes = [ct.c_int64, P, P, P, P, ct.c_int32, P]
        lib.mpn_cover_by_group.restype = ct.c_int
        _bound = True
    return lib


def device_sort_order(tid_key, pos, rev):
    """Order of BAM records by (reference, position, strand), ties in input order: the GPU form of `samtools sort`
    (bam.sam_to_sorted_bam's sort_keys hook).  tid_key < 2^41, 0 <= pos < 2^62, rev in {0, 1}."""
    tid_key = np.ascontiguousarray(tid_key, dtype=np.uint64)
    lo = np.ascontiguousarray(np.asarray(pos, dtype=np.uint64) << np.uint64(1) | np.asarray(rev, dtype=np.uint64))
    order = np.empty(len(tid_key), dtype=np.int64)
    _ffi.check(_lib().mpn_sort_order(len(tid_key), tid_key.ctypes.data, lo.ctypes.data, order.ctypes.data), 'mpn_sort_order')
    return order


def device_cover_by_group(group, seq, start, end, n_groups):
    """-> int64[n_groups]: union length of [start, end) per (group, seq), summed per group (mpn_cover_by_group)."""
    group = np.ascontiguousarray(group, dtype=np.int32)
    seq = np.ascontiguousarray(seq, dtype=np.int32)
    start = np.ascontiguousarray(start, dtype=np.int64)
    end = np.ascontiguousarray(end, dtype=np.int64)
    out = np.zeros(max(int(n_groups), 1), dtype=np.int64)
    _ffi.check(_lib().mpn_cover_by_group(len(group), group.ctypes.data, seq.ctypes.data, start.ctypes.data, end.ctypes.data, int(n_groups),
                                         out.ctypes.data), 'mpn_cover_by_group')
    return out[:n_groups]


def _codes(col):
    u, inv = np.unique(col.to_numpy(dtype=object).astype(str), return_inverse=True)
    return u, inv


def best_per_read_and_assembly(align_list):
    _, rc = _codes(align_list['read_id'])
    _, ac = _codes(align_list['assembly_id'])
    order = np.lexsort((align_list['alignment_score_tiebreaker'].to_numpy(), align_list['alignment_score'].to_numpy(), ac, rc))
    last = np.ones(len(order), dtype=bool)
    last[:-1] = (rc[order][1:] != rc[order][:-1]) | (ac[order][1:] != ac[order][:-1])
    return align_list.iloc[order[last]]


def host_cover_by_group(ac, sc, start, end, n_groups):
    """numpy statement of mpn_cover_by_group (one global running maximum serves all groups)."""
    if len(ac) == 0:
        return np.zeros(n_groups, dtype=np.int64)
    order = np.lexsort((end, start, sc, ac))
    ac, sc, start, end = ac[order], sc[order], start[order], end[order]
    new_group = np.ones(len(order), dtype=bool)
    new_group[1:] = (ac[1:] != ac[:-1]) | (sc[1:] != sc[:-1])
    gid = np.cumsum(new_group) - 1
    span = int(end.max() - min(start.min(), 0)) + 2
    run_end = np.maximum.accumulate(end + gid * span) - gid * span
    prev_end = np.empty_like(run_end)
    prev_end[0] = 0
    prev_end[1:] = run_end[:-1]
    opens = new_group | (start > prev_end)                      # book-ended intervals (start == previous end) merge
    add = np.where(opens, end - start, np.maximum(end - np.maximum(prev_end, start), 0))
    return np.bincount(ac, weights=add.astype(np.float64), minlength=n_groups).astype(np.int64)


def covered_bp_by_assembly(rows, noise_bed=None, device=None):
    r"""Union length of [sequence_from, sequence_to) per (assembly_id, sequence_id), summed per assembly.
    noise_bed: DataFrame(sequence_id, start, end[, assembly_id]) -- what it covers on a sequence is subtracted
    (`covered_bed.subtract(noise_bed)`, megapath_nano.py:516-518): |A \ N| = |A u N| - |N| per sequence.
    device: None / True = the HIP path (libmpn.so and a GPU are REQUIRED: like every product path of this package it raises
    MpnError without them, there is no silent fallback); False = the numpy restatement the tests compare it with."""
    if rows.shape[0] == 0:
        return {}
    if device is None:
        device = True
    cover = device_cover_by_group if device else host_cover_by_group
    asm, ac = _codes(rows['assembly_id'])
    seqs, sc = _codes(rows['sequence_id'])
    start, end = rows['sequence_from'].to_numpy(dtype=np.int64), rows['sequence_to'].to_numpy(dtype=np.int64)
    ac, sc = ac.astype(np.int32), sc.astype(np.int32)
    if noise_bed is None or noise_bed.shape[0] == 0:
        return dict(zip(asm, cover(ac, sc, start, end, len(asm))))
    # noise intervals on the sequences that carry alignments, once under every assembly that has alignments on that sequence
    # (bedtools subtract matches on the chromosome column alone, which is the sequence_id)
    where = pandas.Index(seqs).get_indexer(noise_bed['sequence_id'].astype(str))
    hit = where >= 0
    pairs = pandas.DataFrame({'ac': ac, 'sc': sc}).drop_duplicates()
    nz = pandas.DataFrame({'sc': where[hit].astype(np.int32), 'start': noise_bed['start'].to_numpy(dtype=np.int64)[hit],
                           'end': noise_bed['end'].to_numpy(dtype=np.int64)[hit]}).merge(pairs, on='sc', how='inner')
    n_ac, n_sc = nz['ac'].to_numpy(dtype=np.int32), nz['sc'].to_numpy(dtype=np.int32)
    n_start, n_end = nz['start'].to_numpy(dtype=np.int64), nz['end'].to_numpy(dtype=np.int64)
    both = cover(np.concatenate([ac, n_ac]), np.concatenate([sc, n_sc]), np.concatenate([start, n_start]), np.concatenate([end, n_end]), len(asm))
    noise = cover(n_ac, n_sc, n_start, n_end, len(asm))
    return dict(zip(asm, both - noise))


def align_stat_by_assembly_id(align_list, assembly_length, assembly_tax=None, noise_bed=None, device=None):
    """align_list: the Align() table.  assembly_length: DataFrame(assembly_id, assembly_length); assembly_tax (optional):
    DataFrame(assembly_id, tax_id, species_tax_id, genus_tax_id, genus_height); noise_bed (optional): DataFrame(sequence_id,
    start, end, assembly_id) as the reference's noise BEDs carry them (name column = assembly_id).  -> DataFrame, one row per
    assembly."""
    best = best_per_read_and_assembly(align_list)
    asm, ac = _codes(best['assembly_id'])
    n = len(asm)

    def total(values):
        return np.bincount(ac, weights=np.asarray(values, dtype=np.float64), minlength=n)

    aligned = (best['sequence_to'] - best['sequence_from']).to_numpy()
    out = pandas.DataFrame({
        'assembly_id': asm,
        'total_number_of_read': np.bincount(ac, minlength=n).astype(np.int64),
        'total_read_bp': total(best['read_length']).astype(np.int64),
        'total_aligned_bp': total(aligned).astype(np.int64),
        'match': total(best['match']).astype(np.int64),
        'edit_dist': total(best['edit_dist']).astype(np.int64),
        'alignment_score': total(best['alignment_score']).astype(np.int64),
        'alignment_score_tiebreaker': total(best['alignment_score_tiebreaker']),
    })
    length = dict(zip(assembly_length['assembly_id'], assembly_length['assembly_length']))
    out['assembly_length'] = np.array([int(length.get(a, 0)) for a in asm], dtype=np.int64)
    for col in ('tax_id', 'species_tax_id', 'genus_tax_id', 'genus_height'):
        lut = dict(zip(assembly_tax['assembly_id'], assembly_tax[col])) if assembly_tax is not None and col in assembly_tax else {}
        out[col] = np.array([int(lut.get(a, 0)) for a in asm], dtype=np.int64)
    cov = covered_bp_by_assembly(best, noise_bed=noise_bed, device=device)
    out['covered_bp'] = np.array([int(cov.get(a, 0)) for a in asm], dtype=np.int64)
    out['noise_span_bp'] = 0
    if noise_bed is not None and noise_bed.shape[0]:
        # bed_to_covered_bp_by_assembly_id(noise_bed): the plain sum of the noise intervals' lengths per assembly (:523-533)
        span = (noise_bed['end'] - noise_bed['start']).groupby(noise_bed['assembly_id'].astype(str)).sum()
        out['noise_span_bp'] = np.array([int(span.get(a, 0)) for a in asm], dtype=np.int64)
    L = out['assembly_length'].to_numpy(dtype=np.float64)
    noise = out['noise_span_bp'].to_numpy(dtype=np.float64)
    tab = out['total_aligned_bp'].to_numpy(dtype=np.float64)

    def clean(x):
        x = np.asarray(x, dtype=np.float64)
        return np.where(np.isfinite(x), x, 0.0)

    with np.errstate(divide='ignore', invalid='ignore'):
        out['average_read_length'] = clean(out['total_read_bp'] / out['total_number_of_read'])
        out['average_depth'] = clean(tab / L)
        out['covered_percent'] = clean(out['covered_bp'] / L)
        out['noise_span_percent'] = clean(noise / L)
        acp = clean(out['covered_bp'] / (L - noise))
        out['adjusted_covered_percent'] = acp
        out['average_identity'] = clean(out['match'] / tab)
        out['average_edit_dist'] = clean(out['edit_dist'] / tab)
        out['average_alignment_score'] = clean(out['alignment_score'] / tab)
        # the reference cleans inf/nan AFTER this block of columns and again after the next one
        aad = acp * tab / (L - noise)
        out['adjusted_average_depth'] = aad
        out['adjusted_total_aligned_bp'] = np.round(clean(aad * L), 0).astype(np.int64)
        out['adjusted_average_depth'] = clean(aad)
    return out
