"""The consumers of the placement stage's alignment table (host side, integer-coded inside):

  best_align_list      best alignment per read                      /root/reference/bin/megapath_nano.py:1287
  aligned_species_list aligned bp per species_tax_id                :1289-1290
  read_count_by_name   the `.read_count_by_name` report (both branches of --reassignment)   :3657-3667

With --reassignment the table already carries a `name` column (Reassign()); without it the names come from
db/sequence_name, reduced to species level like reassignment.py:69-70 does, and an alignment whose sequence has no
entry there is counted under its sequence_id (:3663).
"""
import numpy as np
import pandas

from .reassignment import species_name


def best_align_list(align_list):
    """Per read the row with the largest (alignment_score, alignment_score_tiebreaker); rows in ascending read_id."""
    if align_list.shape[0] == 0:
        return align_list.copy()
    ids, code = np.unique(align_list['read_id'].to_numpy(dtype=object).astype(str), return_inverse=True)
    order = np.lexsort((align_list['alignment_score_tiebreaker'].to_numpy(), align_list['alignment_score'].to_numpy(), code))
    last = np.ones(len(order), dtype=bool)
    last[:-1] = code[order][1:] != code[order][:-1]
    return align_list.iloc[order[last]].copy()


def aligned_species_list(best, min_aligned_bp=0):
    """-> (table species_tax_id, aligned_bp over the best alignments; the rows with aligned_bp >= min_aligned_bp)"""
    sp, code = np.unique(best['species_tax_id'].to_numpy(), return_inverse=True)
    bp = np.bincount(code, weights=(best['sequence_to'] - best['sequence_from']).to_numpy(dtype=np.float64), minlength=len(sp)).astype(np.int64)
    table = pandas.DataFrame({'species_tax_id': sp, 'aligned_bp': bp})
    return table, table[table['aligned_bp'] >= min_aligned_bp]


def read_count_by_name(best, db_folder=None, reassignment=False, resolution='species'):
    """-> pandas.Series (index `name`, values = reads) in descending order of the count (ties: by name; the reference's own
    tie order is that of an unstable sort)."""
    if reassignment:
        names = best['name'].to_numpy(dtype=object)
    else:
        table = pandas.read_csv(f'{db_folder}/sequence_name', sep='\t', header=None, names=['sequence_id', 'name'])
        by_seq = {}
        for sid, desc in zip(table['sequence_id'], table['name']):
            by_seq.setdefault(sid, []).append(species_name(desc, resolution) if isinstance(desc, str) else desc)
        names = []
        for sid in best['sequence_id']:
            hit = by_seq.get(sid)
            if not hit:
                names.append(sid)                     # no name on record: the sequence id stands in (:3663)
            else:
                names.extend(n if isinstance(n, str) else sid for n in hit)   # a left join: one row per name entry
        names = np.array(names, dtype=object)
    uniq, cnt = np.unique(names.astype(str), return_counts=True) if len(names) else (np.array([], dtype=str), np.array([], dtype=np.int64))
    order = np.lexsort((uniq, -cnt))
    return pandas.Series(cnt[order].astype(np.int64), index=pandas.Index(uniq[order], name='name'), name='read_id')


def write_read_count_by_name(series, path):
    series.to_csv(path_or_buf=path, sep='\t')
