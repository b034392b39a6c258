/*
 * mpn_ssw.h -- C-ABI of the MI355X-native SSW-compatible local aligner (libmpn.so).
 *
 * Part 1 is the drop-in boundary: the four symbols the reference binds with ctypes in
 *   /root/reference/bin/realignment/pyssw.py:30-48
 * with the prototypes and struct layouts of
 *   /root/reference/bin/realignment/realign/ssw.h:47-57 (s_align), :77 (ssw_init), :82 (init_destroy),
 *   :120-128 (ssw_align), :133 (align_destroy).
 * A maintainer switches the reference over by pointing pyssw.SSW(lib_path=...) at libmpn.so
 * (INTEGRATION.md section 1).  Every call runs the HIP kernels in csrc/ssw_kernels.hip; there is
 * no CPU fallback: without a usable GPU ssw_align() prints the HIP error and returns NULL.
 *
 * Part 2 is the batched entry point the reference does not have (it calls ssw_align once per
 * read, pyssw.py:137-147): many read/reference pairs per launch, one wavefront per pair.
 *
 * Domain: gap_open > gap_extend (all reference call sites use 8/2).  Outside it the
 * reference's SSE lazy-F early exits make results lane-layout dependent; libmpn refuses
 * (NULL / MPN_SSW_EDOMAIN) rather than return a different answer.  Reads of up to 131072 bases are taken
 * (MPN_SSW_ETOOLONG beyond); like the reference's 16-bit pass, scores are meaningful below 32768.
 */
#ifndef MPN_SSW_H
#define MPN_SSW_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Part 1: reference-compatible symbols -------------------------------------------- */
struct _profile;
typedef struct _profile s_profile;

typedef struct {            /* ssw.h:47-57, identical field order and widths */
    uint16_t score1;
    uint16_t score2;
    int32_t ref_begin1;
    int32_t ref_end1;
    int32_t read_begin1;
    int32_t read_end1;
    int32_t ref_end2;
    uint32_t *cigar;        /* BAM encoding len<<4|op, M=0 I=1 D=2 (ssw.h:137-161) */
    int32_t cigarLen;
} s_align;

/* read and mat are BORROWED (ssw.c:749-750): keep them alive until init_destroy. */
s_profile *ssw_init(const int8_t *read, const int32_t readLen, const int8_t *mat, const int32_t n,
                    const int8_t score_size);
void init_destroy(s_profile *p);
/* Returns a malloc'd s_align (caller frees with align_destroy) or NULL + message on stderr,
 * exactly where ssw.c:794-796,802-804,840-843 do, and additionally on HIP failure. */
s_align *ssw_align(const s_profile *prof, const int8_t *ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                   const int32_t maskLen);
void align_destroy(s_align *a);

/* ---- Part 2: batched form ------------------------------------------------------------- */
enum {
    MPN_SSW_OK = 0,
    MPN_SSW_ENULL = 1,     /* the reference would return NULL for this pair (score_size 0 and score >= 255,
                              or traceback error) */
    MPN_SSW_EUNDEF = 2,    /* the reference's behaviour is undefined for this pair (walks outside its band) */
    MPN_SSW_EDOMAIN = 3,   /* gap_open <= gap_extend */
    MPN_SSW_ETOOLONG = 4,  /* read longer than 131072 */
    MPN_SSW_ECIGAR_CAP = 5 /* cigar pool too small */
};

/* All pointers are HOST pointers.  Pair i aligns reads[read_off[i] .. +read_len[i]) (codes 0..n-1)
 * to refs[ref_off[i] .. +ref_len[i]).  Offsets may alias (one reference shared by all pairs).
 * Outputs are caller-allocated arrays of n_pairs entries; cigars go to cigar_pool
 * (cigar_cap uint32 entries) at cigar_off[i] .. +cigar_len[i].
 * Returns 0 on success (per-pair conditions are in status[i]); negative on HIP/runtime failure,
 * with the message available from mpn_last_error(). */
int mpn_ssw_align_batch(int32_t n_pairs,
                        const int8_t *reads, const int64_t *read_off, const int32_t *read_len,
                        const int8_t *refs, const int64_t *ref_off, const int32_t *ref_len,
                        const int8_t *mat, int32_t n, int8_t score_size,
                        uint8_t gap_open, uint8_t gap_extend, uint8_t flag, uint16_t filters, int32_t filterd,
                        const int32_t *mask_len,
                        uint16_t *score1, uint16_t *score2, int32_t *ref_begin1, int32_t *ref_end1,
                        int32_t *read_begin1, int32_t *read_end1, int32_t *ref_end2,
                        uint32_t *cigar_pool, int64_t cigar_cap, int64_t *cigar_off, int32_t *cigar_len,
                        int32_t *status);

const char *mpn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
