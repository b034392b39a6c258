/*
 * mpn_fastq.h -- C-ABI of the read quality / length filter in front of the aligner (libmpn.so).
 *
 * Replaces the per-read arithmetic of  /root/reference/bin/tools/nanofastq.c :165-203  (the `nanofastq` child
 * process the reference starts per input file at /root/reference/bin/megapath_nano.py:1045-1057): the sum of the
 * base-call error probabilities of a read, and that sum after head/tail cropping.  The filter's decisions and its
 * `%.2f` report lines depend on the exact value of these double-precision sums, so the kernel adds in the reference's
 * order (one lane per read, sequential IEEE-754 additions, then the same subtractions): results are bit-identical to
 * the C loop.  Text parsing (kseq), log10 and formatting stay on the host: megapath_nano_amd/fastq_filter.py and
 * bin/mpn-nanofastq mirror the program's stdin/stdout/stderr contract; INTEGRATION.md section 6 shows the switch.
 *
 * All pointers are HOST pointers; arrays are caller-allocated.  Returns 0, or a negative error (mpn_last_error()).
 */
#ifndef MPN_FASTQ_H
#define MPN_FASTQ_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* qual: concatenated quality strings (ASCII, Phred+33); read i = qual[off[i] .. off[i]+len[i]).
 * table[128]: error probability per Phred score, as the caller computed it (nanofastq.c:147-149 uses pow(10, -i/10.)).
 * total[i]  = sum over the read, in order                                   (nanofastq.c:168-171)
 * cropped[i]= total[i] minus the first head_crop and the last tail_crop terms, subtracted one by one in that order
 *             (nanofastq.c:188-195); only computed (else left equal to total[i]) when
 *             len[i] - tail_crop - head_crop >= min_len                     (nanofastq.c:182)
 * status[i] = 0, or 1 if the read holds a quality character outside '!'..'~'+33 (the reference would index out of
 *             its table there); such reads get total = cropped = 0. */
int mpn_fastq_qsum_batch(int32_t n, const uint8_t *qual, const int64_t *off, const int32_t *len, int32_t head_crop,
                         int32_t tail_crop, int32_t min_len, const double *table, double *total, double *cropped,
                         uint8_t *status);

#ifdef __cplusplus
}
#endif
#endif
