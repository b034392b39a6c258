/*
 * mpn_realign.h -- C-ABI of the MI355X-native amplicon realigner (libmpn.so), SURVEY.md section 8 row f4.
 *
 * Part 1 is the drop-in boundary: the two symbols the reference binds with ctypes in
 *   /root/reference/bin/realignment/realign_illumina_reads.py:40-43 (struct), :596-605 (realign_reads), :627-629 (free_memory)
 * with the prototypes of
 *   /root/reference/bin/realignment/realign/realigner.cpp:854-869 and the struct of realigner.h:42-46.
 * A maintainer switches over by pointing `realigner_mod` (realign_illumina_reads.py:32) at libmpn.so.
 * The k-mer seeded Hamming placement of reads on haplotypes (ReAligner::FastAlignReadsToHaplotypes, realigner.cpp:147-230)
 * and every Smith-Waterman alignment (haplotype -> reference, read -> haplotype; ssw_cpp.cpp:268-300 over ssw.c) run as
 * HIP kernels (csrc/realign.hip, csrc/ssw_kernels.hip).  There is no CPU fallback: on a HIP failure realign_reads
 * prints the error and returns NULL.
 *
 * Part 2 is the batched form the reference does not have (it calls realign_reads once per window from a Python loop,
 * realign_illumina_reads.py:533-629): many windows per call, one launch per stage over all of them.
 *
 * Preconditions (the reference's behaviour is undefined outside them): every haplotype has >= 32 bases, at most 1000
 * reads per window through Part 1 (the reference's fixed arrays), 7-bit ASCII sequences.
 */
#ifndef MPN_REALIGN_H
#define MPN_REALIGN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Part 1: reference-compatible symbols -------------------------------------------- */
typedef struct {              /* realigner.h:42-46 */
    int position[1000];
    char *cigar_string[1000];
} struct_str_arr;

/* haplotypes: whitespace-separated sequences (realigner.cpp:787-792).  Returns a new struct_str_arr whose first
 * read_size entries are filled (cigar strings individually allocated); release it with free_memory(ptr, read_size). */
struct_str_arr *realign_reads(char *seqs[], int *positions, char *cigars[], char *reference, char *haplotypes,
                              int ref_start, int ref_prefix, int ref_suffix, int read_size);
void free_memory(struct_str_arr *pointer, int size);

/* ---- Part 2: batched form ------------------------------------------------------------- */
typedef struct {
    int32_t n_reads;
    const char *const *seqs;       /* n_reads NUL-terminated reads */
    const int32_t *positions;      /* n_reads */
    const char *const *cigars;     /* n_reads NUL-terminated CIGAR strings (returned unchanged for reads left alone) */
    const char *reference;         /* ref_prefix + window + ref_suffix */
    int32_t n_haps;
    const char *const *haplotypes; /* n_haps NUL-terminated candidate haplotypes */
    int32_t ref_start, ref_prefix, ref_suffix;
} mpn_realign_window;

/* out_position / out_cigar: one entry per read of every window, in window order.  out_cigar[i] is malloc'd; release
 * the whole array's strings with mpn_realign_free_cigars.  Returns 0, or a negative code with mpn_last_error(). */
int mpn_realign_batch(int32_t n_windows, const mpn_realign_window *windows, int32_t *out_position, char **out_cigar);
void mpn_realign_free_cigars(char **cigars, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
