/*
 * mpn_reassign.h -- C-ABI of the MI355X-native read-reassignment pass (libmpn.so).
 *
 * Replaces the body of  /root/reference/bin/lib/reassignment.py  Reassign() :66-108
 * (counts :75-81, i-explains-j relation :27-36,:93-98, per-read relabel :38-64) and the two reductions
 * that consume it, /root/reference/bin/megapath_nano.py:1287-1289 (best row per read, aligned bp per
 * species_tax_id) and :3664-3667 (.read_count_by_name).  Strings stay on the host: the Python mirror
 * megapath_nano_amd/reassignment.py factorises read_id / name / species_tax_id into dense int32 codes,
 * groups rows by read (CSR), and calls these entry points.  The reference has no FFI here (pure pandas);
 * INTEGRATION.md section 2 shows the two-line change in megapath_nano.py that switches it over.
 *
 * The pass is split in two so that a multi-GPU run (reads sharded over ranks) can all-reduce the
 * per-name counters between the halves: counts() -> [allreduce] -> apply() ->
 * [allreduce of read_count / aligned_bp].
 *
 * All pointers are HOST pointers; arrays are caller-allocated.  Functions return 0 on success, negative
 * on HIP/runtime failure (message: mpn_last_error(), declared in mpn_ssw.h).
 */
#ifndef MPN_REASSIGN_H
#define MPN_REASSIGN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpn_reassign_plan mpn_reassign_plan;

/* Rows must be grouped by read: rows read_ptr[r] .. read_ptr[r+1]-1 belong to read r, in their original
 * relative order (that order breaks alignment_score ties inside a (read, name) group: the LAST row wins).
 *   name_idx     dense code of the (species-ised) sequence name, 0..n_names-1
 *   score        alignment_score
 *   tiebreak     alignment_score_tiebreaker (aligner.py:334-335)
 *   aligned_bp   sequence_to - sequence_from
 *   species_idx  dense code of species_tax_id, 0..n_species-1
 * Uploads everything to HBM once. */
int mpn_reassign_create(int64_t n_rows, int32_t n_reads, int32_t n_names, int32_t n_species,
                        const int64_t *read_ptr, const int32_t *name_idx, const int32_t *score,
                        const double *tiebreak, const int64_t *aligned_bp, const int32_t *species_idx,
                        mpn_reassign_plan **plan);

/* reassignment.py:73-91: dedupe (read, name) keeping the best score, then
 *   all_count[n] = rows per name, u_count[n] = rows per name among reads left with exactly one row,
 *   n_multi_reads = reads left with more than one row (0 makes the reference raise TypeError, :91). */
int mpn_reassign_counts(mpn_reassign_plan *plan, int64_t *all_count, int64_t *u_count, int64_t *n_multi_reads);

/* reassignment.py:27-36,:93-98 relation from the (global) counters, then :38-64 per-read relabel in ascending
 * name_rank of the explaining row's name, then megapath_nano.py:1287-1289,:3666 reductions.
 *   keep[row]             1 if the row survives the dedupe
 *   new_name[row]         name code after reassignment (== name_idx[row] if untouched)
 *   explainer[name]       1 if the name explains at least one other name (is_in_explain_other, :57-58)
 *   read_count_by_name    reads whose best row carries that name (n_names entries)
 *   aligned_bp_by_species sum of aligned_bp over best rows (n_species entries)
 *   n_relations           number of (i, j) pairs in the relation; 0 = the reference's early return (:100) */
int mpn_reassign_apply(mpn_reassign_plan *plan, const int64_t *all_count, const int64_t *u_count,
                       const int32_t *name_rank, double error_rate, double ratio, double as_threshold,
                       uint8_t *keep, int32_t *new_name, uint8_t *explainer, int64_t *read_count_by_name,
                       int64_t *aligned_bp_by_species, int64_t *n_relations);

void mpn_reassign_destroy(mpn_reassign_plan *plan);

#ifdef __cplusplus
}
#endif
#endif
