/*
 * mpn_abundance.h -- C-ABI of the two data-parallel steps of the output formats / abundance statistic:
 *
 *   mpn_sort_order      the coordinate sort of `samtools sort` (reference, position, strand; ties in file order) that the
 *                       reference runs on the species-placement SAM   (/root/reference/bin/lib/aligner.py:246-252)
 *   mpn_cover_by_group  `bedtools sort | bedtools merge` + the per-assembly sum of the merged lengths
 *                       (/root/reference/bin/megapath_nano.py:313-347: align_list_to_bed, bed_to_covered_bp_by_assembly_id),
 *                       i.e. the covered base pairs of every assembly; the noise-BED variant (`covered_bed.subtract(noise_bed)`,
 *                       :516-518) is two calls: |A \ N| = |A u N| - |N| per sequence (megapath_nano_amd/abundance.py)
 *
 * Both run on the GPU (csrc/interval_kernels.hip: a stable LSD radix sort of 128-bit keys, a three-phase segmented sweep);
 * all pointers are HOST pointers, results are exact integers.  Return 0, or a negative error (mpn_last_error()).
 */
#ifndef MPN_ABUNDANCE_H
#define MPN_ABUNDANCE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* order[0..n): the indices of the records in ascending (hi, lo); records with equal keys keep their input order (stable). */
int mpn_sort_order(int64_t n, const uint64_t *hi, const uint64_t *lo, int64_t *order);

/* covered[g] (g < n_groups, zeroed by the call) = sum over the sequences of group g of the length of the union of its
 * intervals [start, end); intervals that overlap or touch merge (bedtools merge, distance 0).  An interval belongs to
 * (group[i], seq[i]); 0 <= start <= end < 2^32. */
int mpn_cover_by_group(int64_t n, const int32_t *group, const int32_t *seq, const int64_t *start, const int64_t *end,
                       int32_t n_groups, int64_t *covered);

#ifdef __cplusplus
}
#endif
#endif
