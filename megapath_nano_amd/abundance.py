"""Per-assembly abundance statistic over an alignment table (host side, numpy).

Reference: `align_list_to_align_stat_by_assembly_id` (/root/reference/bin/megapath_nano.py:485-541) with its helpers
`summary_stat_1` (:440-449), `align_list_to_bed` (:313-330: bedtools sort + merge of (assembly, sequence) intervals),
`bed_to_covered_bp_by_assembly_id` (:333-347) and `summary_stat_2` (:451-482).  Only the default configuration is covered
(no noise BED: every noise filter is off by default, megapath_nano.py:4985-4996).

Steps: per (read, assembly) keep the alignment with the largest (alignment_score, tiebreaker); sum per assembly; covered bp
= length of the union of the kept alignments' target intervals per (assembly, sequence), where overlapping AND
book-ended intervals merge (`bedtools merge` default distance 0); then the derived columns of summary_stat_2, with the
reference's inf -> nan -> 0 clean-up and the rounded `adjusted_total_aligned_bp`.
"""
import numpy as np
import pandas


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


def covered_bp_by_assembly(rows):
    """Union length of [sequence_from, sequence_to) per (assembly_id, sequence_id), summed per assembly."""
    if rows.shape[0] == 0:
        return {}
    asm, ac = _codes(rows['assembly_id'])
    _, sc = _codes(rows['sequence_id'])
    start, end = rows['sequence_from'].to_numpy(dtype=np.int64), rows['sequence_to'].to_numpy(dtype=np.int64)
    order = np.lexsort((end, start, sc, ac))
    ac, sc, start, end = ac[order], sc[order], start[order], end[order]
    new_group = np.ones(len(order), dtype=bool)
    new_group[1:] = (ac[1:] != ac[:-1]) | (sc[1:] != sc[:-1])
    gid = np.cumsum(new_group) - 1
    # running maximum of `end` inside every group: offset the groups so that one global maximum.accumulate serves all
    span = int(end.max() - min(start.min(), 0)) + 2
    run_end = np.maximum.accumulate(end + gid * span) - gid * span
    prev_end = np.empty_like(run_end)
    prev_end[0] = 0
    prev_end[1:] = run_end[:-1]
    opens = new_group | (start > prev_end)                      # book-ended intervals (start == previous end) merge
    add = np.where(opens, end - start, np.maximum(end - np.maximum(prev_end, start), 0))
    per_asm = np.bincount(ac, weights=add.astype(np.float64), minlength=len(asm)).astype(np.int64)
    return dict(zip(asm, per_asm))


def align_stat_by_assembly_id(align_list, assembly_length, assembly_tax=None):
    """align_list: the Align() table.  assembly_length: DataFrame(assembly_id, assembly_length); assembly_tax (optional):
    DataFrame(assembly_id, tax_id, species_tax_id, genus_tax_id, genus_height).  -> DataFrame, one row per assembly."""
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
    cov = covered_bp_by_assembly(best)
    out['covered_bp'] = np.array([int(cov.get(a, 0)) for a in asm], dtype=np.int64)
    out['noise_span_bp'] = 0
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
