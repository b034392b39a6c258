/*
 * mpn_bam.h -- C-ABI of the record encoder of the BAM writer: SAM lines -> BAM records
 * (SAMv1 section 4.2 as htslib 1.13 writes them: sam.c sam_parse1 / bam_write1), the per-record work of
 * `samtools view -b` in the pipeline the reference starts after the species placement
 * (/root/reference/bin/lib/aligner.py:246-252).  Host code, multi-threaded over the lines of a batch; BGZF blocking,
 * compression, the coordinate sort (mpn_sort_order) and the BAI index stay in megapath_nano_amd/bam.py, which the reference's
 * vendored htslib test data pin (tests/golden/htslib).
 */
#ifndef MPN_BAM_H
#define MPN_BAM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpn_bam_encoder mpn_bam_encoder;

/* an encoder for the references of one SAM header (@SQ order): RNAME / RNEXT -> refID */
mpn_bam_encoder *mpn_bam_encoder_create(const char *const *ref_names, int32_t n_ref);
void mpn_bam_encoder_destroy(mpn_bam_encoder *e);

/* n SAM lines: line i is text[line_off[i] .. line_off[i] + line_len[i]) (a trailing newline is ignored).  Record i (WITHOUT its
 * block_size word) goes to out[rec_off[i] .. rec_off[i + 1]); tid / pos0 / end0 / flag [i] receive what the index builder needs
 * (end0 = pos0 + reference length of the CIGAR, pos0 + 1 for a record without one or an unmapped one).  A CIGAR of more than
 * 65535 operations travels as <l_seq>S<ref_len>N + a CG:B,I tag (htslib bam_write1).  Returns the bytes written, -3 if out_cap is
 * too small (nothing usable is left in out), or -1 (mpn_last_error(): malformed line, unknown tag type, integer out of range). */
int64_t mpn_bam_encode(const mpn_bam_encoder *e, const char *text, const int64_t *line_off, const int32_t *line_len, int64_t n,
                       uint8_t *out, int64_t out_cap, int64_t *rec_off, int32_t *tid, int32_t *pos0, int32_t *end0, int32_t *flag);

#ifdef __cplusplus
}
#endif
#endif
