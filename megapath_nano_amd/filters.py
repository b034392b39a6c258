"""Host-side mirror of the human/decoy classification that consumes Align() output in the reference's step 2
(/root/reference/bin/megapath_nano.py step_human_and_decoy_filter, :1135-1200; thresholds :5071-5074).

Pure pandas post-processing over the align_list DataFrame (a few thousand rows per batch); kept on the host exactly as
in the reference.  The alignment itself (the expensive part of step 2) is megapath_nano_amd.aligner.Align."""
import pandas


def _best_per_read(df):
    # megapath_nano.py:1142 / :1176 / :1195: sort (read_id, alignment_score, tiebreaker), keep the last row per read
    return df.sort_values(['read_id', 'alignment_score', 'alignment_score_tiebreaker']).drop_duplicates(subset=['read_id'], keep='last')


def human_and_decoy_classify(align_list, human_assembly_list, decoy_assembly_list, read_id_list,
                             human_min_alignment_score=1000, human_min_alignment_score_percent=100,
                             decoy_min_alignment_score=1000, decoy_min_alignment_score_percent=100):
    """-> dict(human_best_align_list, human_read_id_list, decoy_best_align_list, decoy_read_id_list,
               microbe_best_align_list, microbe_read_id_list), the O.* members the reference fills at :1144-1200."""
    human_best = _best_per_read(align_list.merge(right=human_assembly_list.set_index('assembly_id'), how='inner',
                                                 left_on='assembly_id', right_index=True, suffixes=['', '_y'], validate='m:1'))
    human_best = human_best.query('alignment_score >= @human_min_alignment_score or '
                                  'alignment_score * 100 / read_length >= @human_min_alignment_score_percent')    # :1144
    human_ids = human_best[['read_id', 'read_length']].sort_values(['read_id', 'read_length']).drop_duplicates()   # :1152
    remaining = align_list.merge(right=human_ids.set_index('read_id').rename(columns={'read_length': 'filtered'}),
                                 how='left', left_on='read_id', right_index=True, suffixes=['', '_y'],
                                 validate='m:1').fillna(0).query('filtered == 0').drop(['filtered'], axis=1)       # :1155-1162
    decoy_best = _best_per_read(remaining.merge(right=decoy_assembly_list.set_index('assembly_id'), how='inner',
                                                left_on='assembly_id', right_index=True, suffixes=['', '_y'], validate='m:1'))
    decoy_best = decoy_best.query('alignment_score >= @decoy_min_alignment_score or '
                                  'alignment_score * 100 / read_length >= @decoy_min_alignment_score_percent')     # :1178
    decoy_ids = decoy_best[['read_id', 'read_length']].sort_values(['read_id', 'read_length']).drop_duplicates()    # :1183
    microbe_best = remaining.merge(right=decoy_ids.set_index('read_id').rename(columns={'read_length': 'filtered'}),
                                   how='left', left_on='read_id', right_index=True, suffixes=['', '_y'], validate='m:1')
    microbe_best = _best_per_read(microbe_best.fillna(0).query('filtered == 0').drop(['filtered'], axis=1))        # :1193
    reads = read_id_list.drop_duplicates()
    gone = pandas.concat([human_ids, decoy_ids], axis=0).sort_values(['read_id', 'read_length']).drop_duplicates()
    microbe_ids = reads.merge(right=gone.set_index('read_id').rename(columns={'read_length': 'filtered'}), how='left',
                              left_on='read_id', right_index=True, suffixes=['', '_y'],
                              validate='1:1').fillna(0).query('filtered == 0').drop(['filtered'], axis=1)[['read_id', 'read_length']]
    return dict(human_best_align_list=human_best, human_read_id_list=human_ids, decoy_best_align_list=decoy_best,
                decoy_read_id_list=decoy_ids, microbe_best_align_list=microbe_best, microbe_read_id_list=microbe_ids)
