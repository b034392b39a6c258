/*
 * mpn_map.h -- C-ABI of the MI355X-native long-read mapper (libmpn.so): the seed-chain-extend path that
 * MegaPath-Nano runs as an external `minimap2` process.
 *
 * Reference interface replaced:  /root/reference/bin/lib/aligner.py:187-231  builds
 *     [aligner, -c, (-a), -t N, -I xG, (-N 50 -p 1), -x map-ont, <target fasta>, <query fastq>...]
 * and reads PAF (or SAM) from the child's stdout (:204-206, :225-231); options come from
 * /root/reference/bin/megapath_nano.py:1124 (human/decoy filter) and :1270 (species placement).
 * The reference has no in-process FFI for this step (it is a process boundary), so the entry points below are
 * what a binding for it needs: build/load an index from target sequences, map a batch of reads, get PAF text
 * with minimap2's `-c` tag order (NM at column 13, AS at column 15, which the awk at aligner.py:271-273
 * relies on).  The Python mirror megapath_nano_amd/aligner.py Align() wraps them behind the reference's own
 * function signature; bin/mpn-aligner is the `--aligner` executable drop-in.
 *
 * Semantics: minimap2 2.17 `-x map-ont` as restated in oracle/mm2_oracle.c (PARITY UNPINNED: minimap2 is not
 * vendored by the reference; DESIGN.md section 6 lists the deliberate differences).
 *
 * Stage entry points (mpn_sketch_batch, mpn_seed_chain_batch) exist so that the parity tests can compare every
 * GPU stage with the oracle; mpn_map_batch is the product call.
 */
#ifndef MPN_MAP_H
#define MPN_MAP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpn_index mpn_index;

typedef struct {
    /* indexing */
    int32_t k, w;                 /* -k 15 -w 10 (map-ont) */
    /* seeding / chaining */
    float mid_occ_frac;           /* -f 2e-4 */
    int32_t mid_occ;              /* > 0 overrides the quantile */
    int32_t max_gap, bw, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
    /* hit selection */
    float mask_level, pri_ratio;  /* -p */
    int32_t best_n;               /* -N */
    int32_t max_join_long, max_join_short, min_join_flank_sc;
    float min_join_flank_ratio;
    /* base-level extension (-c) */
    int32_t a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
    float max_clip_ratio;
    int64_t max_sw_mat;
    int32_t with_cigar;           /* -c; 0 = mapping_only (aligner.py:188) */
    uint32_t seed;
    int32_t host_threads;         /* threads for the host-side hit bookkeeping; 0 = all cores */
    int32_t out_sam;              /* -a: 1 = the text buffer of mpn_map_batch(_ex) receives SAM records instead of PAF lines
                                   * (aligner.py:188-192; header lines: mpn_sam_header), 2 = PAF lines AND kept SAM records
                                   * (mpn_map_batch_q).  Unmapped reads get a flag-4 record; QUAL is '*' unless the
                                   * qualities are handed over (mpn_map_batch_q, mpn_hits_finish). */
} mpn_map_opt;

/* minimap2 2.17 defaults for `-x map-ont -c` (-N 5 -p 0.8) */
void mpn_map_opt_init(mpn_map_opt *opt);

/* Build the index of n_seq target sequences (ASCII, any case; non-ACGT = ambiguous) and keep it resident in
 * HBM together with the 2-bit packed targets.  Returns NULL on failure (mpn_last_error()). */
mpn_index *mpn_index_build(int32_t n_seq, const char *const *names, const char *const *seqs, const int32_t *lens,
                           int32_t k, int32_t w);
/* The same build from targets that are already resident in HBM: d_seqs is a DEVICE pointer to the concatenated ASCII
 * targets (target i at byte seq_off[i], seq_off[0] = 0, no gaps; seq_off and lens are host arrays).  The caller keeps
 * ownership of d_seqs and may release it when the call returns.  This is how a target set that never exists as host
 * strings (a decompressor or generator writing into HBM; bench.py's synthetic RefSeq stand-in) is indexed. */
mpn_index *mpn_index_build_device(int32_t n_seq, const char *const *names, const void *d_seqs, const int64_t *seq_off,
                                  const int32_t *lens, int32_t k, int32_t w);
void mpn_index_destroy(mpn_index *idx);
/* Persistent form of a built index (the reference rebuilds its index on every run: bin/lib/aligner.py:209-221; minimap2's
 * own `-d FILE` / prebuilt-index-as-target is used at bin/megapath_nano.py:1641-1645).  save: 0 or negative error;
 * load: a resident index equal to the one saved (same keys/positions/targets), or NULL + mpn_last_error(). */
int mpn_index_save(const mpn_index *idx, const char *path);
/* "@SQ" lines of the targets + one "@PG" line (cmdline may be NULL); returns the text length, or -3 if cap is too small */
int64_t mpn_sam_header(const mpn_index *idx, const char *cmdline, char *buf, int64_t cap);
mpn_index *mpn_index_load(const char *path);
/* A target set of several index parts (minimap2 -I) in ONE file, as minimap2 -d dumps it (bin/megapath_nano.py:1641-1645):
 * save_append adds a part behind what the file holds; load_at reads the part that starts at byte `offset` and leaves in
 * *next_offset where the next one starts, or -1 after the last (next_offset may be NULL). */
int mpn_index_save_append(const mpn_index *idx, const char *path);
mpn_index *mpn_index_load_at(const char *path, int64_t offset, int64_t *next_offset);
/* names and lengths of the targets of an index (for a loaded one): name i is copied into buf (cap bytes incl. NUL) */
int32_t mpn_index_n_seq(const mpn_index *idx);
int32_t mpn_index_seq_len(const mpn_index *idx, int32_t i);
int32_t mpn_index_seq_name(const mpn_index *idx, int32_t i, char *buf, int32_t cap);
int32_t mpn_index_k(const mpn_index *idx);
int32_t mpn_index_w(const mpn_index *idx);
int64_t mpn_index_n_minimizers(const mpn_index *idx);
int64_t mpn_index_n_keys(const mpn_index *idx);
/* occurrence cut-off for a given -f (minimap2: mm_idx_cal_max_occ) */
int32_t mpn_index_mid_occ(const mpn_index *idx, float frac);
/* bases [start, start+len) of target i as upper-case ASCII (ambiguous bases come back as 'N'), decoded from the 2-bit
 * targets resident in HBM: what a checker of PAF lines needs when the target set is too large to keep on the host
 * (bench.py's correctness block).  Returns len, or a negative error (range outside the target). */
int64_t mpn_index_fetch_seq(const mpn_index *idx, int32_t i, int64_t start, int64_t len, char *out);
/* copies of index arrays for the parity tests: keys[n_keys], key_off[n_keys+1], pos[n_minimizers] */
int mpn_index_export(const mpn_index *idx, uint64_t *keys, int64_t *key_off, uint64_t *pos);

/* ---- stage: (w,k)-minimizers of a batch of sequences, one CSR row per sequence ---------------------------
 * seqs: concatenated ASCII; sequence i = seqs[seq_off[i] .. +seq_len[i]).  mz_off must hold n+1 entries.
 * Minimizers are returned as (x = hash<<8|span, y = i<<32|last_pos<<1|strand) pairs in mz (cap pairs).
 * Returns the total number of minimizers, or a negative error (-3: cap too small; mz_off is still filled). */
int64_t mpn_sketch_batch(int32_t n, const char *seqs, const int64_t *seq_off, const int32_t *seq_len, int32_t k,
                         int32_t w, int64_t *mz_off, uint64_t *mz, int64_t cap);

/* ---- stage: seeds -> sorted anchors -> chains, for a batch of reads --------------------------------------
 * Outputs (caller allocated; CSR over reads):
 *   n_anchor[i], rep_len[i]               anchors found / repetitive-minimizer span (minimap2 rl:i)
 *   chain_off[n+1], chains u[] (score<<32|cnt), achor_off[n+1], chained anchors b[] as (x, y) pairs
 * Returns 0, or negative error (-3: a capacity is too small). */
int mpn_seed_chain_batch(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *seqs,
                         const int64_t *seq_off, const int32_t *seq_len, int64_t *n_anchor, int32_t *rep_len,
                         int64_t *chain_off, uint64_t *u, int64_t u_cap, int64_t *anchor_off, uint64_t *b,
                         int64_t b_cap);

/* ---- stage: the banded dual-affine extension DP on arbitrary pairs of 0..4 code strings (parity tests) -------
 * flag bits as in ksw2: 0x02 approximate max, 0x08 right-align gaps, 0x40 extension only, 0x80 reversed CIGAR.
 * force_kernel: 0 = dispatch as mpn_map_batch does, 1 = single-wave LDS kernel, 3 = workgroup kernel, 4 = systolic strip kernel where eligible, 5 = band-in-registers kernel where the band fits 1024 slots.  out9[i*9..] = max, zdropped, max_q, max_t, mqe, mqe_t, score, reach_end, n_cigar. */
int mpn_ext_dp_batch(const mpn_map_opt *opt, int32_t n, const uint8_t *qcodes, const int64_t *q_off, const int32_t *q_len,
                     const uint8_t *tcodes, const int64_t *t_off, const int32_t *t_len, const int32_t *w, const int32_t *zdrop,
                     const int32_t *end_bonus, const int32_t *flag, int32_t force_kernel, int32_t *out9, uint32_t *cigar_pool,
                     int64_t cigar_cap, int64_t *cig_off);

/* ---- product call: map a batch of reads, PAF text out ----------------------------------------------------
 * names: n NUL-terminated read names.  paf receives the lines of all reads in input order (NUL terminated).
 * Returns the number of bytes written, or negative error (-3: paf_cap too small). */
int64_t mpn_map_batch(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *const *names,
                      const char *seqs, const int64_t *seq_off, const int32_t *seq_len, char *paf, int64_t paf_cap);

/* ---- product call, extended: device-resident reads in, alignment columns out --------------------------------
 * d_seqs/d_off/d_len (DEVICE pointers, may be NULL): the same reads already resident in HBM (bench.py uploads them
 * before the timed region); the host copies are still needed for the CIGAR bookkeeping.
 * paf may be NULL.  cols may be NULL; otherwise every reported alignment fills one row of the caller-allocated
 * arrays (cap rows each), in PAF order: the 12 PAF columns as integers plus NM, AS and the tp flag; these are
 * the fields aligner.py:291-294 keeps.  Returns PAF bytes written (0 if paf is NULL) or a negative error
 * (-3: paf_cap or cols->cap too small; cols->n_rows then holds the required number). */
typedef struct {
    int64_t cap, n_rows;
    int32_t *read_idx, *qs, *qe, *rev, *rid, *rs, *re, *mlen, *blen, *mapq, *nm, *as, *primary;
} mpn_aln_cols;

/* d_seqs (when given): the reads on the device; 4-byte aligned and padded so that the aligned 32-bit word around the
 * last base may be read (any allocation rounded up to 4 bytes will do). */
int64_t mpn_map_batch_ex(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *const *names,
                         const char *seqs, const int64_t *seq_off, const int32_t *seq_len, const void *d_seqs,
                         const int64_t *d_off, const int32_t *d_len, char *paf, int64_t paf_cap, mpn_aln_cols *cols);

/* The general product call: mpn_map_batch_ex plus the reads' base qualities (quals: concatenated like seqs, same offsets, or
 * NULL) for the QUAL column of SAM records.  opt->out_sam: 0 = PAF into paf, 1 = SAM into paf, 2 = PAF into paf AND the SAM
 * records of the same hits kept by the library for mpn_map_fetch_sam (the reference's species-placement call keeps both:
 * bin/lib/aligner.py:183-184,219-227,260-261 -- one mapping pass serves them). */
int64_t mpn_map_batch_q(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                        const char *quals, const int64_t *seq_off, const int32_t *seq_len, const void *d_seqs, const int64_t *d_off,
                        const int32_t *d_len, char *paf, int64_t paf_cap, mpn_aln_cols *cols);
int64_t mpn_map_fetch_sam(char *buf, int64_t cap);  /* like mpn_map_fetch_text, for the SAM text of out_sam == 2 */

/* ---- split index: minimap2 -I <bases> parts and the --split-prefix merge (bin/lib/aligner.py:199; bin/megapath_nano.py:
 * 4019-4022 passes -I <RAM/64>G, :1124,:1270 pass --split-prefix on every call) ----------------------------------------
 * A target set that exceeds one index is mapped part by part: the caller builds the index of part p, adds the hits of the
 * batch against it (mpn_map_batch_part), destroys it, and after the last part mpn_hits_finish merges per read exactly as
 * minimap2 merges its per-part dumps -- hits pooled with target ids shifted to the concatenated target list,
 * sub-optimal bookkeeping cleared, ranking / parent-secondary grouping / -p -N selection / SAM-primary / MAPQ
 * recomputed over the pool, repetitive-seed length = the largest over the parts -- and emits text and columns like
 * mpn_map_batch_q (column rid and the names in the text refer to the concatenated target list: mpn_hits_seq_*).
 * minimap2 takes the merge path whenever --split-prefix is given, also for a single part. */
typedef struct mpn_hits mpn_hits;
mpn_hits *mpn_hits_create(int32_t n_reads);
void mpn_hits_destroy(mpn_hits *h);
/* want_text = 0: mpn_hits_finish will be asked for the integer columns only (no PAF, no SAM), so the CIGARs of the parts' hits stay in
 * HBM -- the integer fast path of the species-placement step (bin/megapath_nano.py:1262-1299 reads columns only).  Default 1. */
void mpn_hits_set_text(mpn_hits *h, int32_t want_text);
int mpn_map_batch_part(const mpn_index *part, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                       const int64_t *seq_off, const int32_t *seq_len, const void *d_seqs, const int64_t *d_off,
                       const int32_t *d_len, mpn_hits *acc);
/* The same for several parts that are RESIDENT together (an accelerator with room for the whole target set keeps every part in
 * memory instead of streaming them): one call, the reads are uploaded once and the (sub-batch, part) pairs share one pipeline.
 * The accumulator ends up as n_parts calls of mpn_map_batch_part in the given order would leave it. */
int mpn_map_batch_parts(const mpn_index *const *parts, int32_t n_parts, const mpn_map_opt *opt, int32_t n, const char *const *names,
                        const char *seqs, const int64_t *seq_off, const int32_t *seq_len, const void *r_seqs, const int64_t *r_off,
                        const int32_t *r_len, mpn_hits *acc);
int64_t mpn_hits_finish(mpn_hits *acc, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                        const char *quals, const int64_t *seq_off, const int32_t *seq_len, char *paf, int64_t paf_cap,
                        mpn_aln_cols *cols);
int32_t mpn_hits_n_parts(const mpn_hits *h);
int32_t mpn_hits_n_seq(const mpn_hits *h);
int32_t mpn_hits_seq_len(const mpn_hits *h, int32_t i);
int32_t mpn_hits_seq_name(const mpn_hits *h, int32_t i, char *buf, int32_t cap);
int64_t mpn_hits_sam_header(const mpn_hits *h, const char *cmdline, char *buf, int64_t cap);

/* After mpn_map_batch(_ex) returned -3 the results of that call are kept by the library (process-wide, until the next
 * mapping call): fetch them with larger buffers instead of mapping the batch again.
 *   mpn_map_fetch_cols: cols->cap rows available; returns the number of rows, -3 if still too small (cols->n_rows = need),
 *                       -1 if nothing is kept.
 *   mpn_map_fetch_text: buf == NULL returns the bytes needed (incl. NUL); otherwise copies the PAF/SAM text and returns
 *                       its length, -3 if cap is too small, -1 if nothing is kept. */
int64_t mpn_map_fetch_cols(mpn_aln_cols *cols);
int64_t mpn_map_fetch_text(char *buf, int64_t cap);

/* Counters and timers of the last mpn_map_batch / mpn_seed_chain_batch on this thread (bench.py roofline line):
 *  [0] input bases  [1] read minimizers  [2] anchors  [3] chains  [4] DP jobs  [5] DP cells  [6] alignments reported
 *  [7] DP rounds    [8] second-pass (exact z-drop) jobs
 * wall-clock ns of the host phases:
 *  [16] H2D reads  [17] seed+chain stage (incl. its kernels)  [18] D2H chains  [19] host: hits from chains
 *  [20] host: plan DP windows  [21] extension stage (H2D jobs, kernels, D2H results)  [22] host: stitch
 *  [23] host: rank/MAPQ/PAF  [24] whole call
 * device ns measured with HIP events on the stream the kernels run on:
 *  [10] sketch  [11] seed lookup+fill  [12] anchor sort  [13] chain DP  [14] chain ends+backtrack
 *  [15] extension DP kernels other than the strip kernel  [25] traceback kernel  [26] z-drop test kernel
 *  inside [21], wall ns: [27] host: kernel choice + scratch layout + staging  [28] enqueue  [29] wait for the GPU
 *  [30] second pass + CIGAR download
 *  [9] device ns of the strip-kernel launches ([15] covers the other DP kernels)  [31] cells (qlen x tlen) those launches computed */
void mpn_map_last_stats(int64_t stats[32]);
/* the same with the slots added after round 1 (returns the number of slots the library has; copies min(n, that)):
 *  [32] sub-batches run (= launches of every per-sub-batch kernel)
 *  device ns of single kernels, HIP events around each launch on its stream:
 *  [33] sketch count pass  [34] sketch fill pass  [35] seed lookup  [36] seed fill  [37] chain DP kernel alone
 *  [38] strip DP <16>  [39] strip DP <32>  [40] strip DP <64>  and their cells [41] [42] [43]
 *  [44] anchors that entered the sort x effective radix passes (bytes moved by the sort = 32 x this)
 *  [45] anchors kept for chaining (segments of at least min_cnt anchors)  [46] device ns of the compaction kernels
 *  device ns of the anchor sort's kernels: [47] partition  [48] chunk sort in LDS  [49] radix passes over the large buckets */
int32_t mpn_map_last_stats_ex(int64_t *stats, int32_t n);

#ifdef __cplusplus
}
#endif
#endif
