"""Human / decoy classification of reads from the alignments of the reference's step 2
(/root/reference/bin/megapath_nano.py:1135-1200 `step_human_and_decoy_filter`; thresholds :5071-5074), on integer codes.

Rules (from the reference): a read's best alignment within a set of assemblies is the one with the largest
(alignment_score, alignment_score_tiebreaker).  A read is HUMAN if its best alignment to a human assembly has
alignment_score >= human_min_alignment_score or alignment_score * 100 / read_length >= human_min_alignment_score_percent.
Human reads are removed with all their alignments; the same test with the decoy thresholds on the best alignment to a
decoy assembly makes a DECOY read.  What is left is MICROBE: its best alignment over all remaining rows is reported,
and every input read that is neither human nor decoy (aligned or not) is a microbe read.

The DataFrame in / DataFrame out interface of the reference is kept; inside, reads are coded as integers, the "best row
per read" is one lexsort, and set membership is a boolean gather -- no joins.  (The per-read best hit is also what
reassign_apply_kernel computes on the GPU for the species-placement stage; here the table is a few rows per read and
stays on the host.)
"""
import numpy as np
import pandas


def _last_of_groups(codes, order):
    """order: row permutation sorted by (code, score, tiebreak); -> rows that close each run of equal codes"""
    sorted_codes = codes[order]
    last = np.ones(len(order), dtype=bool)
    last[:-1] = sorted_codes[1:] != sorted_codes[:-1]
    return order[last]


def _best_rows(read_code, score, tiebreak, mask):
    """Per read (ascending code), the row with the largest (score, tiebreak) among the rows selected by mask."""
    rows = np.flatnonzero(mask)
    if rows.size == 0:
        return rows
    order = rows[np.lexsort((tiebreak[rows], score[rows], read_code[rows]))]
    return _last_of_groups(read_code, order)


def _passes(score, read_length, min_score, min_percent):
    return (score >= min_score) | (score * 100 / read_length >= min_percent)


def human_and_decoy_classify(align_list, human_assembly_list, decoy_assembly_list, read_id_list,
                             human_min_alignment_score=1000, human_min_alignment_score_percent=100,
                             decoy_min_alignment_score=1000, decoy_min_alignment_score_percent=100, key='assembly_id'):
    """-> dict(human_best_align_list, human_read_id_list, decoy_best_align_list, decoy_read_id_list,
               microbe_best_align_list, microbe_read_id_list): the O.* tables the reference fills at :1144-1200."""
    for name, lst in (('human', human_assembly_list), ('decoy', decoy_assembly_list)):
        if pandas.Index(lst['assembly_id']).has_duplicates:
            raise pandas.errors.MergeError(f'{name} assembly list is not unique: not a many-to-one merge')
    ids, read_code = np.unique(align_list['read_id'].to_numpy(dtype=object).astype(str), return_inverse=True)
    score = align_list['alignment_score'].to_numpy()
    tiebreak = align_list['alignment_score_tiebreaker'].to_numpy()
    read_length = align_list['read_length'].to_numpy()
    target = align_list[key].to_numpy(dtype=object)
    is_human = np.isin(target, human_assembly_list['assembly_id'].to_numpy(dtype=object))
    is_decoy = np.isin(target, decoy_assembly_list['assembly_id'].to_numpy(dtype=object))

    def classify(candidates, min_score, min_percent):
        best = _best_rows(read_code, score, tiebreak, candidates)
        best = best[_passes(score[best], read_length[best], min_score, min_percent)]
        caught = np.zeros(len(ids), dtype=bool)
        caught[read_code[best]] = True
        return best, caught

    human_rows, human_read = classify(is_human, human_min_alignment_score, human_min_alignment_score_percent)
    remaining = ~human_read[read_code]
    decoy_rows, decoy_read = classify(is_decoy & remaining, decoy_min_alignment_score, decoy_min_alignment_score_percent)
    microbe_rows = _best_rows(read_code, score, tiebreak, remaining & ~decoy_read[read_code])

    def id_table(rows):
        return align_list.iloc[rows][['read_id', 'read_length']].drop_duplicates()

    reads = read_id_list.drop_duplicates()
    gone = set(ids[human_read | decoy_read])
    keep = ~reads['read_id'].astype(str).isin(gone).to_numpy()
    return dict(human_best_align_list=align_list.iloc[human_rows], human_read_id_list=id_table(human_rows),
                decoy_best_align_list=align_list.iloc[decoy_rows], decoy_read_id_list=id_table(decoy_rows),
                microbe_best_align_list=align_list.iloc[microbe_rows], microbe_read_id_list=reads[keep][['read_id', 'read_length']])


def classify_codes(read_idx, kind, score, tiebreak, read_length, n_reads, human_min_alignment_score=1000,
                   human_min_alignment_score_percent=100, decoy_min_alignment_score=1000, decoy_min_alignment_score_percent=100):
    """The same rules on integer columns only (what bench.py --config c2 times after the mapping call): read_idx[row] in
    [0, n_reads), kind[row] = 0 human target / 1 decoy target / anything else other, read_length[read].
    -> int8[n_reads]: 0 microbe (aligned or not), 1 human, 2 decoy."""
    read_idx = np.asarray(read_idx)
    score, tiebreak, kind = np.asarray(score), np.asarray(tiebreak), np.asarray(kind)
    row_len = np.asarray(read_length)[read_idx]
    out = np.zeros(n_reads, dtype=np.int8)

    def caught(mask, min_score, min_percent):
        best = _best_rows(read_idx, score, tiebreak, mask)
        best = best[_passes(score[best], row_len[best], min_score, min_percent)]
        return read_idx[best]

    out[caught(kind == 0, human_min_alignment_score, human_min_alignment_score_percent)] = 1
    remaining = out[read_idx] == 0
    out[caught((kind == 1) & remaining, decoy_min_alignment_score, decoy_min_alignment_score_percent)] = 2
    return out
