/*
 * mm2_oracle.h -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the minimap2 seed-chain-extend path that
 * MegaPath-Nano shells out to (/root/reference/bin/lib/aligner.py:187-231; options at
 * /root/reference/bin/megapath_nano.py:1124 "-x map-ont", :1270 "-N 50 -p 1 -x map-ont").
 *
 * PARITY UNPINNED: minimap2 is an unpinned conda dependency of the reference (README.md:36, Dockerfile:34),
 * its source is not under /root/reference and no binary exists in the build image, and the reference holds
 * no test or golden vector for this boundary.  This file restates the PUBLISHED
 * algorithm of minimap2 (Li 2018, Bioinformatics 34:3094; release 2.17-r941, the newest release when the
 * reference's conda environment -- python 3.6.10, parallel=20191122 -- was defined) from its documented
 * behaviour.  Known, deliberate differences from a real minimap2 binary are listed in DESIGN.md section 6.
 */
#ifndef MM2_ORACLE_H
#define MM2_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } mm128;

/* flags in anchor.y (same bit positions as minimap2's MM_SEED_*) */
#define MMO_SEED_LONG_JOIN (1ULL << 40)
#define MMO_SEED_IGNORE    (1ULL << 41)
#define MMO_SEED_TANDEM    (1ULL << 42)

typedef struct {
    int k, w;
    int32_t n_seq;
    char **name;
    int32_t *len;
    int64_t *off;       /* offset of each sequence in seq4 */
    uint8_t *seq4;      /* one code (0..4) per base */
    int64_t n_keys;     /* distinct minimizers */
    uint64_t *keys;     /* sorted hash values */
    int64_t *key_off;   /* n_keys+1 */
    uint64_t *pos;      /* rid<<32 | last_pos<<1 | strand, sorted within a key */
} mmo_idx;

typedef struct {
    /* -x map-ont defaults of minimap2 2.17 */
    float mid_occ_frac;     /* -f 2e-4 */
    int32_t mid_occ;        /* derived from the index if <= 0 */
    int32_t max_gap, bw, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
    float mask_level, pri_ratio;   /* -p */
    int32_t best_n;                /* -N */
    int32_t max_join_long, max_join_short, min_join_flank_sc;
    float min_join_flank_ratio;
    int32_t a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
    float max_clip_ratio;
    int64_t max_sw_mat;
    int32_t with_cigar;     /* -c */
    uint32_t seed;          /* 11 */
} mmo_opt;

typedef struct {
    int32_t id, cnt, rid, score, qs, qe, rs, re, parent, subsc, as, mlen, blen, n_sub, score0;
    uint32_t mapq, split, rev, inv, sam_pri, split_inv, hash;
    /* base-level extension */
    int32_t has_p, dp_score, dp_max, dp_max2, n_ambi, n_cigar;
    uint32_t *cigar;
} mmo_reg;

void mmo_opt_init(mmo_opt *o);                     /* map-ont, -N 5 -p 0.8, with -c */
mmo_idx *mmo_idx_build(int32_t n_seq, const char **names, const char **seqs, const int32_t *lens, int k, int w);
void mmo_idx_destroy(mmo_idx *mi);
int32_t mmo_idx_cal_max_occ(const mmo_idx *mi, float f);
int64_t mmo_idx_get(const mmo_idx *mi, uint64_t minier, const uint64_t **pos);

/* stage outputs are malloc'd; release with mmo_free */
void mmo_free(void *p);
int64_t mmo_sketch(const char *seq, int32_t len, int w, int k, uint32_t rid, mm128 **out);
/* seeds sorted by (x, y&0xffffffff); *rep_len as minimap2 */
int64_t mmo_collect_anchors(const mmo_idx *mi, int32_t max_occ, const mm128 *mv, int64_t n_mv, int32_t qlen,
                            mm128 **a, int32_t *rep_len);
/* chaining: consumes nothing; returns chained anchors b[] (grouped per chain) and u[] = score<<32|cnt */
int64_t mmo_chain(const mmo_opt *o, int64_t n_a, const mm128 *a, int32_t *n_u, uint64_t **u, mm128 **b);

/* whole read: returns regs (malloc'd array, cigar malloc'd per reg); *n_regs.  name may be NULL. */
mmo_reg *mmo_map_read(const mmo_idx *mi, const mmo_opt *o, const char *name, const char *seq, int32_t qlen,
                      int32_t *n_regs, int32_t *rep_len);
void mmo_free_regs(mmo_reg *r, int32_t n);
/* split index (minimap2 -I / --split-prefix): hits of one read over all parts, merged; names/lengths of all parts for the writers */
mmo_idx *mmo_idx_concat_names(int32_t n_parts, const mmo_idx **parts);
mmo_reg *mmo_map_read_split(int32_t n_parts, const mmo_idx **parts, const mmo_opt *o, const char *name, const char *seq,
                            int32_t qlen, int32_t *n_regs, int32_t *rep_len);
int64_t mmo_write_sam_q(const mmo_idx *mi, const mmo_opt *o, const char *name, int32_t qlen, const char *seq, const char *qual,
                        const mmo_reg *regs, int32_t n_regs, int32_t rep_len, char *buf, int64_t cap);
/* PAF line(s) for one read into buf (NUL terminated); returns bytes written (excluding NUL) or -1 if cap too small */
int64_t mmo_write_paf(const mmo_idx *mi, const mmo_opt *o, const char *name, int32_t qlen, const mmo_reg *regs,
                      int32_t n_regs, int32_t rep_len, char *buf, int64_t cap);
/* SAM records of one read (-a), incl. the flag-4 record of an unmapped read; same buffer contract as mmo_write_paf */
int64_t mmo_write_sam(const mmo_idx *mi, const mmo_opt *o, const char *name, int32_t qlen, const char *seq, const mmo_reg *regs,
                      int32_t n_regs, int32_t rep_len, char *buf, int64_t cap);

/* standalone banded dual-affine extension (ksw2-style, difference recurrences on anti-diagonals) */
typedef struct {
    int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar;
    uint32_t *cigar;
} mmo_ez;
#define MMO_EZ_APPROX_MAX 0x02
#define MMO_EZ_RIGHT      0x08
#define MMO_EZ_EXTZ_ONLY  0x40
#define MMO_EZ_REV_CIGAR  0x80
void mmo_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t sc_mch, int8_t sc_mis,
               int8_t sc_n, int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
               mmo_ez *ez);

#ifdef __cplusplus
}
#endif
#endif
