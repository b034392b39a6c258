/*
 * mm2_oracle.c -- TEST INFRASTRUCTURE ONLY (see mm2_oracle.h: PARITY UNPINNED).
 *
 * Stages, in the order minimap2's mm_map_frag runs them for one uni-segment long read:
 *   sketch (mm_sketch) -> seed lookup + anchor sort (collect_seed_hits) -> chaining DP (mm_chain_dp)
 *   -> hits (mm_gen_regs, mm_set_parent, mm_select_sub, mm_join_long) -> base-level extension
 *   (mm_align_skeleton: mm_align1 over ksw2 dual-affine extension) -> filter / re-rank / MAPQ -> PAF.
 * Sorting is done with total orders (ties broken by the remaining key bits) where minimap2 uses an
 * unstable radix sort, so results are implementation independent.
 */
#include "mm2_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PARENT_UNSET (-1)
#define PARENT_TMP_PRI (-2)
#define NEG_INF (-0x40000000)

void mmo_free(void *p) { free(p); }

static uint8_t nt4(int c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

/* ---------------------------------------------------------------- sorting helpers (total orders) */
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
static int cmp_128xy(const void *a, const void *b)
{
    const mm128 *p = (const mm128 *)a, *q = (const mm128 *)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return p->y < q->y ? -1 : p->y > q->y;
}
static int cmp_anchor(const void *a, const void *b)
{ /* by reference position (x), then query position (low 32 bits of y) */
    const mm128 *p = (const mm128 *)a, *q = (const mm128 *)b;
    uint32_t py = (uint32_t)p->y, qy = (uint32_t)q->y;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return py < qy ? -1 : py > qy;
}

/* ---------------------------------------------------------------- sketch */
static inline uint64_t hash64m(uint64_t key, uint64_t mask)
{ /* invertible integer hash (Thomas Wang), masked to 2k bits */
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

typedef struct { mm128 *a; int64_t n, m; } v128;
static void v128_push(v128 *v, mm128 x)
{
    if (v->n == v->m) { v->m = v->m ? v->m << 1 : 256; v->a = (mm128 *)realloc(v->a, (size_t)v->m * 16); }
    v->a[v->n++] = x;
}

/* (w,k)-minimizers of one sequence; symmetric k-mers are skipped before they enter the window;
 * identical minimal k-mers in a window are all reported (minimap2 sketch.c behaviour). */
int64_t mmo_sketch(const char *seq, int32_t len, int w, int k, uint32_t rid, mm128 **out)
{
    const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t kmer[2] = {0, 0};
    int i, j, l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;
    mm128 buf[256], min = {UINT64_MAX, UINT64_MAX};
    v128 v = {0, 0, 0};
    assert(len >= 0 && w > 0 && w < 256 && k > 0 && k <= 28);
    memset(buf, 0xff, (size_t)w * 16);
    for (i = 0; i < len; ++i) {
        int c = nt4((unsigned char)seq[i]);
        mm128 info = {UINT64_MAX, UINT64_MAX};
        if (c < 4) {
            int z;
            kmer_span = l + 1 < k ? l + 1 : k;
            kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
            kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
            if (kmer[0] == kmer[1]) continue; /* strand unknown: the position does not enter the window */
            z = kmer[0] < kmer[1] ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = hash64m(kmer[z], mask) << 8 | (uint64_t)kmer_span;
                info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint32_t)z;
            }
        } else l = 0, kmer_span = 0;
        buf[buf_pos] = info;
        if (l == w + k - 1 && min.x != UINT64_MAX) { /* first full window: report k-mers identical to the min */
            for (j = buf_pos + 1; j < w; ++j)
                if (min.x == buf[j].x && buf[j].y != min.y) v128_push(&v, buf[j]);
            for (j = 0; j < buf_pos; ++j)
                if (min.x == buf[j].x && buf[j].y != min.y) v128_push(&v, buf[j]);
        }
        if (info.x <= min.x) { /* new minimum (ties go to the newest) */
            if (l >= w + k && min.x != UINT64_MAX) v128_push(&v, min);
            min = info, min_pos = buf_pos;
        } else if (buf_pos == min_pos) { /* the minimum left the window */
            if (l >= w + k - 1 && min.x != UINT64_MAX) v128_push(&v, min);
            for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
                if (min.x >= buf[j].x) min = buf[j], min_pos = j;
            for (j = 0; j <= buf_pos; ++j)
                if (min.x >= buf[j].x) min = buf[j], min_pos = j;
            if (l >= w + k - 1 && min.x != UINT64_MAX) {
                for (j = buf_pos + 1; j < w; ++j)
                    if (min.x == buf[j].x && min.y != buf[j].y) v128_push(&v, buf[j]);
                for (j = 0; j <= buf_pos; ++j)
                    if (min.x == buf[j].x && min.y != buf[j].y) v128_push(&v, buf[j]);
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    if (min.x != UINT64_MAX) v128_push(&v, min);
    *out = v.a;
    return v.n;
}

/* ---------------------------------------------------------------- index */
/* Sort by (x, y): the records are first partitioned on the top bits of x (a counting pass), then every bucket is sorted on
 * its own; (x, y) is a total order (y is unique), so the result is the one a single qsort gives.  Buckets and the
 * per-sequence sketches run on OpenMP threads: the tests build indexes of a gigabase on the GPU box's host. */
static void sort_128xy(mm128 *a, int64_t n, int key_bits)
{
    const int bb = key_bits < 12 ? key_bits : 12, shift = key_bits - bb;
    const int64_t nb = (int64_t)1 << bb;
    int64_t *cnt, i, b;
    mm128 *t;
    if (n < (1 << 16)) { qsort(a, n, 16, cmp_128xy); return; }
    cnt = (int64_t *)calloc((size_t)nb + 1, 8);
    t = (mm128 *)malloc((size_t)n * 16);
    for (i = 0; i < n; ++i) ++cnt[(a[i].x >> shift) + 1];
    for (b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
    {
        int64_t *cur = (int64_t *)malloc((size_t)nb * 8);
        memcpy(cur, cnt, (size_t)nb * 8);
        for (i = 0; i < n; ++i) t[cur[a[i].x >> shift]++] = a[i];
        free(cur);
    }
#pragma omp parallel for schedule(dynamic, 16)
    for (b = 0; b < nb; ++b)
        if (cnt[b + 1] - cnt[b] > 1) qsort(t + cnt[b], (size_t)(cnt[b + 1] - cnt[b]), 16, cmp_128xy);
    memcpy(a, t, (size_t)n * 16);
    free(t); free(cnt);
}

mmo_idx *mmo_idx_build(int32_t n_seq, const char **names, const char **seqs, const int32_t *lens, int k, int w)
{
    mmo_idx *mi = (mmo_idx *)calloc(1, sizeof(mmo_idx));
    int64_t tot = 0, i, n = 0;
    int32_t s;
    mm128 *all = 0, **sv;
    int64_t *sn;
    mi->k = k, mi->w = w, mi->n_seq = n_seq;
    mi->name = (char **)calloc(n_seq, sizeof(char *));
    mi->len = (int32_t *)calloc(n_seq, 4);
    mi->off = (int64_t *)calloc(n_seq + 1, 8);
    for (s = 0; s < n_seq; ++s) mi->off[s] = tot, tot += lens[s], mi->len[s] = lens[s], mi->name[s] = strdup(names[s]);
    mi->off[n_seq] = tot;
    mi->seq4 = (uint8_t *)malloc(tot > 0 ? tot : 1);
    sv = (mm128 **)calloc(n_seq > 0 ? n_seq : 1, sizeof(mm128 *));
    sn = (int64_t *)calloc((size_t)n_seq + 1, 8);
#pragma omp parallel for schedule(dynamic, 1)
    for (s = 0; s < n_seq; ++s) {
        int64_t j;
        for (j = 0; j < lens[s]; ++j) mi->seq4[mi->off[s] + j] = nt4((unsigned char)seqs[s][j]);
        sn[s] = mmo_sketch(seqs[s], lens[s], w, k, (uint32_t)s, &sv[s]);
    }
    for (s = 0; s < n_seq; ++s) n += sn[s];
    all = (mm128 *)malloc((size_t)(n > 0 ? n : 1) * 16);
    for (s = 0, i = 0; s < n_seq; ++s) {
        if (sn[s]) memcpy(all + i, sv[s], (size_t)sn[s] * 16);
        i += sn[s];
        free(sv[s]);
    }
    free(sv); free(sn);
    for (i = 0; i < n; ++i) all[i].x >>= 8; /* drop the span: key = hash only */
    sort_128xy(all, n, 2 * k);
    mi->keys = (uint64_t *)malloc((size_t)(n + 1) * 8);
    mi->key_off = (int64_t *)malloc((size_t)(n + 2) * 8);
    mi->pos = (uint64_t *)malloc((size_t)(n + 1) * 8);
    for (i = 0; i < n; ++i) {
        if (i == 0 || all[i].x != all[i - 1].x) mi->keys[mi->n_keys] = all[i].x, mi->key_off[mi->n_keys++] = i;
        mi->pos[i] = all[i].y;
    }
    mi->key_off[mi->n_keys] = n;
    free(all);
    return mi;
}

void mmo_idx_destroy(mmo_idx *mi)
{
    int32_t s;
    if (!mi) return;
    for (s = 0; s < mi->n_seq; ++s) free(mi->name[s]);
    free(mi->name); free(mi->len); free(mi->off); free(mi->seq4); free(mi->keys); free(mi->key_off); free(mi->pos);
    free(mi);
}

int64_t mmo_idx_get(const mmo_idx *mi, uint64_t minier, const uint64_t **pos)
{
    int64_t lo = 0, hi = mi->n_keys;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (mi->keys[mid] < minier) lo = mid + 1; else hi = mid;
    }
    if (lo == mi->n_keys || mi->keys[lo] != minier) { *pos = 0; return 0; }
    *pos = mi->pos + mi->key_off[lo];
    return mi->key_off[lo + 1] - mi->key_off[lo];
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* occurrence threshold: (1-f) quantile of the per-minimizer occurrence counts, plus one */
int32_t mmo_idx_cal_max_occ(const mmo_idx *mi, float f)
{
    uint32_t *a, thres;
    int64_t i, n = mi->n_keys, kk;
    if (f <= 0.f) return INT32_MAX;
    if (n == 0) return 1;
    a = (uint32_t *)malloc((size_t)n * 4);
    for (i = 0; i < n; ++i) a[i] = (uint32_t)(mi->key_off[i + 1] - mi->key_off[i]);
    qsort(a, n, 4, cmp_u32);
    kk = (int64_t)(uint32_t)((1. - (double)f) * (double)n);
    if (kk >= n) kk = n - 1;
    thres = a[kk] + 1;
    free(a);
    return (int32_t)thres;
}

/* ---------------------------------------------------------------- seeds -> anchors */
int64_t mmo_collect_anchors(const mmo_idx *mi, int32_t max_occ, const mm128 *mv, int64_t n_mv, int32_t qlen,
                            mm128 **a_, int32_t *rep_len)
{
    int64_t i, n_a = 0, m_a = 0;
    int rep_st = 0, rep_en = 0;
    mm128 *a = 0;
    *rep_len = 0;
    for (i = 0; i < n_mv; ++i) {
        const uint64_t *cr;
        uint32_t q_pos = (uint32_t)mv[i].y, q_span = mv[i].x & 0xff;
        int64_t t = mmo_idx_get(mi, mv[i].x >> 8, &cr), k;
        if (t >= max_occ) { /* repetitive minimizer: skipped, its query span accumulates into rep_len */
            int en = (int)(q_pos >> 1) + 1, st = en - (int)q_span;
            if (st > rep_en) { *rep_len += rep_en - rep_st; rep_st = st, rep_en = en; }
            else rep_en = en;
            continue;
        }
        {
            int tandem = 0;
            if (i > 0 && mv[i].x >> 8 == mv[i - 1].x >> 8) tandem = 1;
            if (i < n_mv - 1 && mv[i].x >> 8 == mv[i + 1].x >> 8) tandem = 1;
            if (n_a + t > m_a) { m_a = (n_a + t) * 2 + 64; a = (mm128 *)realloc(a, (size_t)m_a * 16); }
            for (k = 0; k < t; ++k) {
                uint64_t r = cr[k];
                int32_t rpos = (uint32_t)r >> 1;
                mm128 *p = &a[n_a++];
                if ((r & 1) == (q_pos & 1)) { /* same strand */
                    p->x = (r & 0xffffffff00000000ULL) | (uint64_t)rpos;
                    p->y = (uint64_t)q_span << 32 | q_pos >> 1;
                } else { /* opposite strand: query coordinate on the reverse-complemented read */
                    p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint64_t)rpos;
                    p->y = (uint64_t)q_span << 32 | (uint32_t)(qlen - ((int32_t)(q_pos >> 1) + 1 - (int32_t)q_span) - 1);
                }
                if (tandem) p->y |= MMO_SEED_TANDEM;
            }
        }
    }
    *rep_len += rep_en - rep_st;
    if (n_a > 1) qsort(a, n_a, 16, cmp_anchor);
    *a_ = a;
    return n_a;
}

/* ---------------------------------------------------------------- chaining */
static inline int ilog2_32(uint32_t v)
{
    int r = 0;
    while (v >>= 1) ++r;
    return r;
}

/* Returns the number of chained anchors; *b holds them grouped chain by chain (chains ordered by the
 * reference coordinate of their first anchor), u[i] = chain score << 32 | number of anchors. */
int64_t mmo_chain(const mmo_opt *o, int64_t n, const mm128 *a, int32_t *n_u_, uint64_t **u_, mm128 **b_)
{
    const int max_dist_x = o->max_gap, max_dist_y = o->max_gap, bw = o->bw, max_skip = o->max_chain_skip;
    const int max_iter = o->max_chain_iter, min_cnt = o->min_cnt, min_sc = o->min_chain_score;
    int32_t *f, *p, *t, *v, n_u, n_v, k;
    int64_t i, j, st = 0;
    uint64_t *u, *u2, sum_qspan = 0;
    float avg_qspan;
    mm128 *b, *w;
    *n_u_ = 0, *u_ = 0, *b_ = 0;
    if (n == 0 || a == 0) return 0;
    f = (int32_t *)malloc((size_t)n * 4); p = (int32_t *)malloc((size_t)n * 4);
    t = (int32_t *)calloc(n, 4); v = (int32_t *)malloc((size_t)n * 4);
    for (i = 0; i < n; ++i) sum_qspan += a[i].y >> 32 & 0xff;
    avg_qspan = (float)sum_qspan / n;
    for (i = 0; i < n; ++i) {
        uint64_t ri = a[i].x;
        int64_t max_j = -1;
        int32_t qi = (int32_t)a[i].y, q_span = a[i].y >> 32 & 0xff;
        int32_t max_f = q_span, n_skip = 0, min_d;
        while (st < i && ri > a[st].x + max_dist_x) ++st;
        if (i - st > max_iter) st = i - max_iter;
        for (j = i - 1; j >= st; --j) {
            int64_t dr = ri - a[j].x;
            int32_t dq = qi - (int32_t)a[j].y, dd, sc, log_dd, gap_cost;
            if (dr == 0 || dq <= 0) continue;
            if (dq > max_dist_y || dq > max_dist_x) continue;
            dd = dr > dq ? dr - dq : dq - dr;
            if (dd > bw) continue;
            min_d = dq < dr ? dq : dr;
            sc = min_d > q_span ? q_span : dq < dr ? dq : dr;
            log_dd = dd ? ilog2_32(dd) : 0;
            gap_cost = (int)(dd * .01 * avg_qspan) + (log_dd >> 1);
            sc -= gap_cost;
            sc += f[j];
            if (sc > max_f) {
                max_f = sc, max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == i) {
                if (++n_skip > max_skip) break;
            }
            if (p[j] >= 0) t[p[j]] = i;
        }
        f[i] = max_f, p[i] = max_j;
        v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f; /* peak score up to i */
    }
    /* chain ends */
    memset(t, 0, (size_t)n * 4);
    for (i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
    for (i = n_u = 0; i < n; ++i) if (t[i] == 0 && v[i] >= min_sc) ++n_u;
    if (n_u == 0) { free(f); free(p); free(t); free(v); return 0; }
    u = (uint64_t *)malloc((size_t)n_u * 8);
    for (i = n_u = 0; i < n; ++i) {
        if (t[i] == 0 && v[i] >= min_sc) {
            j = i;
            while (j >= 0 && f[j] < v[j]) j = p[j]; /* back to the peak */
            if (j < 0) j = i;
            u[n_u++] = (uint64_t)f[j] << 32 | j;
        }
    }
    qsort(u, n_u, 8, cmp_u64);
    for (i = 0; i < n_u >> 1; ++i) { uint64_t x = u[i]; u[i] = u[n_u - i - 1], u[n_u - i - 1] = x; }
    /* backtrack, best chain first; an anchor belongs to one chain only */
    memset(t, 0, (size_t)n * 4);
    for (i = n_v = k = 0; i < n_u; ++i) {
        int32_t n_v0 = n_v, k0 = k;
        j = (int32_t)u[i];
        do { v[n_v++] = j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
        if (j < 0) {
            if (n_v - n_v0 >= min_cnt) u[k++] = u[i] >> 32 << 32 | (n_v - n_v0);
        } else if ((int32_t)(u[i] >> 32) - f[j] >= min_sc) {
            if (n_v - n_v0 >= min_cnt) u[k++] = ((u[i] >> 32) - f[j]) << 32 | (n_v - n_v0);
        }
        if (k0 == k) n_v = n_v0;
    }
    n_u = k;
    free(f); free(p); free(t);
    if (n_u == 0) { free(u); free(v); return 0; }
    b = (mm128 *)malloc((size_t)n_v * 16);
    for (i = 0, k = 0; i < n_u; ++i) {
        int32_t k0 = k, ni = (int32_t)u[i];
        for (j = 0; j < ni; ++j) b[k] = a[v[k0 + (ni - j - 1)]], ++k;
    }
    free(v);
    /* order chains by the reference position of their first anchor (needed by the long-join step) */
    w = (mm128 *)malloc((size_t)n_u * 16);
    for (i = k = 0; i < n_u; ++i) { w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | i; k += (int32_t)u[i]; }
    qsort(w, n_u, 16, cmp_128xy);
    u2 = (uint64_t *)malloc((size_t)n_u * 8);
    {
        mm128 *c = (mm128 *)malloc((size_t)n_v * 16);
        for (i = k = 0; i < n_u; ++i) {
            int32_t jj = (int32_t)w[i].y, nn = (int32_t)u[jj];
            u2[i] = u[jj];
            memcpy(&c[k], &b[w[i].y >> 32], (size_t)nn * 16);
            k += nn;
        }
        free(b);
        b = c;
    }
    free(u); free(w);
    *n_u_ = n_u, *u_ = u2, *b_ = b;
    return n_v;
}

/* ---------------------------------------------------------------- hits */
static inline uint64_t hash64(uint64_t key)
{
    key = ~key + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}
static inline uint32_t wang32(uint32_t key)
{
    key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
    key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
    return key;
}
static inline uint32_t x31_hash(const char *s)
{
    uint32_t h = (uint32_t)*s;
    if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
    return h;
}

static void cal_fuzzy_len(mmo_reg *r, const mm128 *a)
{
    int i;
    r->mlen = r->blen = 0;
    if (r->cnt <= 0) return;
    r->mlen = r->blen = a[r->as].y >> 32 & 0xff;
    for (i = r->as + 1; i < r->as + r->cnt; ++i) {
        int span = a[i].y >> 32 & 0xff;
        int tl = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        int ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        r->blen += tl > ql ? tl : ql;
        r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
}

static void reg_set_coor(mmo_reg *r, int32_t qlen, const mm128 *a)
{
    int32_t k = r->as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
    r->rev = a[k].x >> 63;
    r->rid = a[k].x << 1 >> 33;
    r->rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r->re = (int32_t)a[k + r->cnt - 1].x + 1;
    if (!r->rev) {
        r->qs = (int32_t)a[k].y + 1 - q_span;
        r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
    } else {
        r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1);
        r->qe = qlen - ((int32_t)a[k].y + 1 - q_span);
    }
    cal_fuzzy_len(r, a);
}

static mmo_reg *gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const mm128 *a)
{
    mm128 *z;
    mmo_reg *r;
    int i, k;
    if (n_u == 0) return 0;
    z = (mm128 *)malloc((size_t)n_u * 16);
    for (i = k = 0; i < n_u; ++i) {
        uint32_t h = (uint32_t)hash64((hash64(a[k].x) + hash64(a[k].y)) ^ hash);
        z[i].x = u[i] ^ h;
        z[i].y = (uint64_t)k << 32 | (int32_t)u[i];
        k += (int32_t)u[i];
    }
    qsort(z, n_u, 16, cmp_128xy);
    r = (mmo_reg *)calloc(n_u, sizeof(mmo_reg));
    for (i = 0; i < n_u; ++i) { /* larger score first */
        mmo_reg *ri = &r[i];
        const mm128 *zi = &z[n_u - 1 - i];
        ri->id = i;
        ri->parent = PARENT_UNSET;
        ri->score = ri->score0 = zi->x >> 32;
        ri->hash = (uint32_t)zi->x;
        ri->cnt = (int32_t)zi->y;
        ri->as = zi->y >> 32;
        reg_set_coor(ri, qlen, a);
    }
    free(z);
    return r;
}

static void set_parent(float mask_level, int n, mmo_reg *r, int sub_diff)
{
    int i, j, k, *w;
    uint64_t *cov;
    if (n <= 0) return;
    for (i = 0; i < n; ++i) r[i].id = i;
    cov = (uint64_t *)malloc((size_t)n * 8);
    w = (int *)malloc((size_t)n * sizeof(int));
    w[0] = 0, r[0].parent = 0;
    for (i = 1, k = 1; i < n; ++i) {
        mmo_reg *ri = &r[i];
        int si = ri->qs, ei = ri->qe, n_cov = 0, uncov_len = 0;
        for (j = 0; j < k; ++j) { /* primary hits overlapping on the query */
            mmo_reg *rp = &r[w[j]];
            int sj = rp->qs, ej = rp->qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)sj << 32 | ej;
        }
        if (n_cov > 0) { /* length of the hit not covered by primaries */
            int x = si;
            qsort(cov, n_cov, 8, cmp_u64);
            for (j = 0; j < n_cov; ++j) {
                if ((int)(cov[j] >> 32) > x) uncov_len += (cov[j] >> 32) - x;
                x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
            }
            if (ei > x) uncov_len += ei - x;
            for (j = 0; j < k; ++j) {
                mmo_reg *rp = &r[w[j]];
                int sj = rp->qs, ej = rp->qe, min, max, ol;
                if (ej <= si || sj >= ei) continue;
                min = ej - sj < ei - si ? ej - sj : ei - si;
                max = ej - sj > ei - si ? ej - sj : ei - si;
                ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / min - (float)uncov_len / max > mask_level) {
                    int cnt_sub = 0;
                    ri->parent = rp->parent;
                    rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
                    if (ri->cnt >= rp->cnt) cnt_sub = 1;
                    if (rp->has_p && ri->has_p &&
                        (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != min)) {
                        rp->dp_max2 = rp->dp_max2 > ri->dp_max ? rp->dp_max2 : ri->dp_max;
                        if (rp->dp_max - ri->dp_max <= sub_diff) cnt_sub = 1;
                    }
                    if (cnt_sub) ++rp->n_sub;
                    break;
                }
            }
        } else j = k;
        if (j == k) w[k++] = i, ri->parent = i, ri->n_sub = 0;
    }
    free(cov); free(w);
}

static void set_sam_pri(int n, mmo_reg *r)
{
    int i, n_pri = 0;
    for (i = 0; i < n; ++i)
        if (r[i].id == r[i].parent) { ++n_pri; r[i].sam_pri = (n_pri == 1); }
        else r[i].sam_pri = 0;
}

static void sync_regs(int n_regs, mmo_reg *regs)
{
    int *tmp, i, max_id = -1, n_tmp;
    if (n_regs <= 0) return;
    for (i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
    n_tmp = max_id + 1;
    tmp = (int *)malloc((size_t)(n_tmp > 0 ? n_tmp : 1) * sizeof(int));
    for (i = 0; i < n_tmp; ++i) tmp[i] = -1;
    for (i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
    for (i = 0; i < n_regs; ++i) {
        mmo_reg *r = &regs[i];
        r->id = i;
        if (r->parent == PARENT_TMP_PRI) r->parent = i;
        else if (r->parent >= 0 && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
        else r->parent = PARENT_UNSET;
    }
    free(tmp);
    set_sam_pri(n_regs, regs);
}

static void drop_reg(mmo_reg *r) { if (r->cigar) free(r->cigar); r->cigar = 0; r->has_p = 0; }

static void select_sub(float pri_ratio, int min_diff, int best_n, int *n_, mmo_reg *r)
{
    if (pri_ratio > 0.0f && *n_ > 0) {
        int i, k, n = *n_, n_2nd = 0;
        for (i = k = 0; i < n; ++i) {
            int p = r[i].parent;
            if (p == i || r[i].inv) r[k++] = r[i];
            else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
                if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs &&
                      r[i].re == r[p].re))
                    r[k++] = r[i], ++n_2nd;
                else drop_reg(&r[i]);
            } else drop_reg(&r[i]);
        }
        if (k != n) sync_regs(k, r);
        *n_ = k;
    }
}

static void filter_regs(const mmo_opt *opt, int qlen, int *n_regs, mmo_reg *regs)
{
    int i, k;
    for (i = k = 0; i < *n_regs; ++i) {
        mmo_reg *r = &regs[i];
        int flt = 0;
        if (!r->inv && r->cnt < opt->min_cnt) flt = 1;
        if (r->has_p) {
            if (r->mlen < opt->min_chain_score) flt = 1;
            else if (r->dp_max < opt->min_dp_max) flt = 1;
            else if (r->qs > qlen * opt->max_clip_ratio && qlen - r->qe > qlen * opt->max_clip_ratio) flt = 1;
            if (flt) drop_reg(r);
        }
        if (!flt) { if (k < i) regs[k++] = regs[i]; else ++k; }
    }
    *n_regs = k;
}

typedef struct { uint64_t key; int idx; } aux_t;
static int cmp_aux(const void *a, const void *b)
{
    const aux_t *p = (const aux_t *)a, *q = (const aux_t *)b;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    return p->idx < q->idx ? -1 : p->idx > q->idx;
}

static int squeeze_a(int n_regs, mmo_reg *regs, mm128 *a)
{ /* drop anchors no hit refers to */
    int i, as = 0;
    aux_t *aux = (aux_t *)malloc((size_t)(n_regs > 0 ? n_regs : 1) * sizeof(aux_t));
    for (i = 0; i < n_regs; ++i) aux[i].key = (uint64_t)regs[i].as, aux[i].idx = i;
    qsort(aux, n_regs, sizeof(aux_t), cmp_aux);
    for (i = 0; i < n_regs; ++i) {
        mmo_reg *r = &regs[aux[i].idx];
        if (r->as != as) { memmove(&a[as], &a[r->as], (size_t)r->cnt * 16); r->as = as; }
        as += r->cnt;
    }
    free(aux);
    return as;
}

static void join_long(const mmo_opt *opt, int qlen, int *n_regs_, mmo_reg *regs, mm128 *a)
{
    int i, n_aux, n_regs = *n_regs_, n_drop = 0;
    aux_t *aux;
    if (n_regs < 2) return;
    squeeze_a(n_regs, regs, a);
    aux = (aux_t *)malloc((size_t)n_regs * sizeof(aux_t));
    for (i = n_aux = 0; i < n_regs; ++i)
        if (regs[i].parent == i || regs[i].parent < 0) aux[n_aux].key = (uint64_t)regs[i].as, aux[n_aux++].idx = i;
    qsort(aux, n_aux, sizeof(aux_t), cmp_aux);
    for (i = n_aux - 1; i >= 1; --i) {
        mmo_reg *r0 = &regs[aux[i - 1].idx], *r1 = &regs[aux[i].idx];
        mm128 *a0e, *a1s;
        int max_gap, min_gap, sc_thres, min_flank_len;
        if (r0->as + r0->cnt != r1->as) continue;
        if (r0->rid != r1->rid || r0->rev != r1->rev) continue;
        a0e = &a[r0->as + r0->cnt - 1];
        a1s = &a[r1->as];
        if (a1s->x <= a0e->x || (int32_t)a1s->y <= (int32_t)a0e->y) continue;
        max_gap = min_gap = (int32_t)a1s->y - (int32_t)a0e->y;
        max_gap = max_gap > (int64_t)(a1s->x - a0e->x) ? max_gap : (int)(a1s->x - a0e->x);
        min_gap = min_gap < (int64_t)(a1s->x - a0e->x) ? min_gap : (int)(a1s->x - a0e->x);
        if (max_gap > opt->max_join_long || min_gap > opt->max_join_short) continue;
        sc_thres = (int)((float)opt->min_join_flank_sc / opt->max_join_long * max_gap + .499);
        if (r0->score < sc_thres || r1->score < sc_thres) continue;
        min_flank_len = (int)(max_gap * opt->min_join_flank_ratio);
        if (r0->re - r0->rs < min_flank_len || r0->qe - r0->qs < min_flank_len) continue;
        if (r1->re - r1->rs < min_flank_len || r1->qe - r1->qs < min_flank_len) continue;
        a[r1->as].y |= MMO_SEED_LONG_JOIN;
        r0->cnt += r1->cnt, r0->score += r1->score;
        reg_set_coor(r0, qlen, a);
        r1->cnt = 0;
        r1->parent = r0->id;
        ++n_drop;
    }
    free(aux);
    if (n_drop > 0) {
        for (i = 0; i < n_regs; ++i) {
            mmo_reg *r = &regs[i];
            if (r->parent >= 0 && r->id != r->parent)
                if (regs[r->parent].parent >= 0 && regs[r->parent].parent != r->parent)
                    r->parent = regs[r->parent].parent;
        }
        filter_regs(opt, qlen, n_regs_, regs);
        sync_regs(*n_regs_, regs);
    }
}

static void hit_sort(int *n_regs, mmo_reg *r)
{
    int32_t i, n_aux, n = *n_regs;
    mm128 *aux;
    mmo_reg *t;
    if (n <= 1) return;
    aux = (mm128 *)malloc((size_t)n * 16);
    t = (mmo_reg *)malloc((size_t)n * sizeof(mmo_reg));
    for (i = n_aux = 0; i < n; ++i) {
        if (r[i].inv || r[i].cnt > 0) {
            int score = r[i].has_p ? r[i].dp_max : r[i].score;
            aux[n_aux].x = (uint64_t)score << 32 | r[i].hash;
            aux[n_aux++].y = i;
        } else drop_reg(&r[i]);
    }
    qsort(aux, n_aux, 16, cmp_128xy);
    for (i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = r[aux[i].y];
    memcpy(r, t, sizeof(mmo_reg) * n_aux);
    *n_regs = n_aux;
    free(aux); free(t);
}

static void set_mapq(int n_regs, mmo_reg *regs, int min_chain_sc, int match_sc, int rep_len)
{
    static const float q_coef = 40.0f;
    int64_t sum_sc = 0;
    float uniq_ratio;
    int i;
    if (n_regs == 0) return;
    for (i = 0; i < n_regs; ++i) if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
    uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
    for (i = 0; i < n_regs; ++i) {
        mmo_reg *r = &regs[i];
        if (r->inv) r->mapq = 0;
        else if (r->parent == r->id) {
            int mapq, subsc;
            float pen_s1 = (r->score > 100 ? 1.0f : 0.01f * r->score) * uniq_ratio;
            float pen_cm = r->cnt > 10 ? 1.0f : 0.1f * r->cnt;
            pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
            subsc = r->subsc > min_chain_sc ? r->subsc : min_chain_sc;
            if (r->has_p && r->dp_max2 > 0 && r->dp_max > 0) {
                float identity = (float)r->mlen / r->blen;
                float x = (float)r->dp_max2 * subsc / r->dp_max / r->score0;
                int mapq_alt;
                mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r->dp_max / match_sc));
                mapq_alt = (int)(6.02f * identity * identity * (r->dp_max - r->dp_max2) / match_sc + .499f);
                mapq = mapq < mapq_alt ? mapq : mapq_alt;
            } else {
                float x = (float)subsc / r->score0;
                if (r->has_p) {
                    float identity = (float)r->mlen / r->blen;
                    mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r->dp_max / match_sc));
                } else mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r->score));
            }
            mapq -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
            mapq = mapq > 0 ? mapq : 0;
            r->mapq = mapq < 60 ? mapq : 60;
            if (r->has_p && r->dp_max > r->dp_max2 && r->mapq == 0) r->mapq = 1;
        } else r->mapq = 0;
    }
}

static void split_reg(mmo_reg *r, mmo_reg *r2, int n, int qlen, const mm128 *a)
{
    if (n <= 0 || n >= r->cnt) return;
    *r2 = *r;
    r2->id = -1;
    r2->sam_pri = 0;
    r2->has_p = 0, r2->cigar = 0, r2->n_cigar = 0, r2->dp_score = r2->dp_max = r2->dp_max2 = r2->n_ambi = 0;
    r2->split_inv = 0;
    r2->cnt = r->cnt - n;
    r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
    r2->as = r->as + n;
    if (r->parent == r->id) r2->parent = PARENT_TMP_PRI;
    reg_set_coor(r2, qlen, a);
    r->cnt -= r2->cnt;
    r->score -= r2->score;
    reg_set_coor(r, qlen, a);
    r->split |= 1, r2->split |= 2;
}

/* ---------------------------------------------------------------- ksw2-style dual affine extension */
static void ez_reset(mmo_ez *ez)
{
    ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
    ez->max = 0, ez->score = ez->mqe = ez->mte = NEG_INF;
    ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

static int apply_zdrop(mmo_ez *ez, int32_t H, int r, int t, int zdrop, int e)
{
    if (H > ez->max) { ez->max = H, ez->max_t = t, ez->max_q = r - t; }
    else if (t >= ez->max_t && r - t >= ez->max_q) {
        int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
        l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
    }
    return 0;
}

static void push_cigar(mmo_ez *ez, int *m, uint32_t op, int len)
{
    if (ez->n_cigar == 0 || op != (ez->cigar[ez->n_cigar - 1] & 0xf)) {
        if (ez->n_cigar == *m) { *m = *m ? *m << 1 : 16; ez->cigar = (uint32_t *)realloc(ez->cigar, (size_t)*m * 4); }
        ez->cigar[ez->n_cigar++] = (uint32_t)len << 4 | op;
    } else ez->cigar[ez->n_cigar - 1] += (uint32_t)len << 4;
}

static void backtrack(mmo_ez *ez, int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col,
                      int i0, int j0)
{
    int i = i0, j = j0, r, state = 0, m = 0, k;
    free(ez->cigar); ez->cigar = 0; ez->n_cigar = 0;
    while (i >= 0 && j >= 0) {
        int force_state = -1, tmp;
        r = i + j;
        if (i < off[r]) force_state = 2;
        if (i > off_end[r]) force_state = 1;
        tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) push_cigar(ez, &m, 0, 1), --i, --j;
        else if (state == 1 || state == 3) push_cigar(ez, &m, 2, 1), --i;
        else push_cigar(ez, &m, 1, 1), --j;
    }
    if (i >= 0) push_cigar(ez, &m, 2, i + 1);
    if (j >= 0) push_cigar(ez, &m, 1, j + 1);
    if (!is_rev)
        for (k = 0; k < ez->n_cigar >> 1; ++k) {
            uint32_t t = ez->cigar[k];
            ez->cigar[k] = ez->cigar[ez->n_cigar - 1 - k], ez->cigar[ez->n_cigar - 1 - k] = t;
        }
}

/* Difference recurrences of Suzuki-Kasahara as formulated in ksw2 (u,v,x,y + second gap x2,y2), evaluated
 * anti-diagonal by anti-diagonal in plain ints.  Cells outside the band of the previous anti-diagonal read
 * as freshly opened gaps (-q-e / -q2-e2). */
void mmo_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t sc_mch, int8_t sc_mis,
               int8_t sc_n, int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
               mmo_ez *ez)
{
    int r, t, qe, qe2, n_col, *off, *off_end, last_st0 = -1, last_en0 = -1;
    int long_thres, long_diff, approx_max = !!(flag & MMO_EZ_APPROX_MAX);
    int32_t *u, *v, *x, *y, *x2, *y2, *H = 0, H0 = 0, last_H0_t = 0;
    uint8_t *p;
    ez_reset(ez);
    if (qlen <= 0 || tlen <= 0) return;
    if (q2 + e2 < q + e) { int8_t tt = q; q = q2, q2 = tt; tt = e, e = e2, e2 = tt; }
    qe = q + e, qe2 = q2 + e2;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    n_col = qlen < tlen ? qlen : tlen;
    n_col = (n_col < w + 1 ? n_col : w + 1) + 1;
    if (-sc_mis > 2 * (q + e)) return;
    long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    u = (int32_t *)malloc((size_t)tlen * 4 * 6);
    v = u + tlen, x = v + tlen, y = x + tlen, x2 = y + tlen, y2 = x2 + tlen;
    for (t = 0; t < tlen; ++t) u[t] = v[t] = x[t] = y[t] = -qe, x2[t] = y2[t] = -qe2;
    if (!approx_max) { H = (int32_t *)malloc((size_t)tlen * 4); for (t = 0; t < tlen; ++t) H[t] = NEG_INF; }
    p = (uint8_t *)calloc((size_t)(qlen + tlen - 1) * n_col, 1);
    off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
    off_end = off + qlen + tlen - 1;

    for (r = 0; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1, x1, x21, v1;
        uint8_t *pr = p + (size_t)r * n_col;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) { ez->zdropped = 1; break; }
        /* boundary of the anti-diagonal */
        if (st > 0) {
            if (st - 1 >= last_st0 && st - 1 <= last_en0) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
            else x1 = -qe, x21 = -qe2, v1 = -qe;
        } else {
            x1 = -qe, x21 = -qe2;
            v1 = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
        }
        if (en >= r) {
            y[r] = -qe, y2[r] = -qe2;
            u[r] = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
        } else if (en > last_en0 && last_en0 >= 0) {
            /* the band grew on the target side past what the previous anti-diagonal computed */
            for (t = last_en0 + 1; t <= en; ++t) y[t] = -qe, y2[t] = -qe2, u[t] = -qe;
        }
        off[r] = st, off_end[r] = en;
        for (t = st; t <= en; ++t) {
            int sq = target[t], sr = query[r - t];
            int z = (sq == 4 || sr == 4) ? sc_n : sq == sr ? sc_mch : sc_mis;
            int ut = u[t], a = x1 + v1, b = y[t] + ut, a2 = x21 + v1, b2 = y2[t] + ut, vt_next = v[t];
            int x_next = x[t], x2_next = x2[t], d, tmp;
            if (!(flag & MMO_EZ_RIGHT)) { /* gaps left-aligned: the diagonal wins ties */
                d = a > z ? 1 : 0; z = z > a ? z : a;
                d = b > z ? 2 : d; z = z > b ? z : b;
                d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
                d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
            } else {
                d = z > a ? 0 : 1; z = z > a ? z : a;
                d = z > b ? d : 2; z = z > b ? z : b;
                d = z > a2 ? d : 3; z = z > a2 ? z : a2;
                d = z > b2 ? d : 4; z = z > b2 ? z : b2;
            }
            if (z > sc_mch) z = sc_mch;
            u[t] = z - v1;
            v[t] = z - ut;
            tmp = z - q; a -= tmp; b -= tmp;
            tmp = z - q2; a2 -= tmp; b2 -= tmp;
            if (!(flag & MMO_EZ_RIGHT)) {
                x[t] = (a > 0 ? a : 0) - qe;   d |= a > 0 ? 0x08 : 0;
                y[t] = (b > 0 ? b : 0) - qe;   d |= b > 0 ? 0x10 : 0;
                x2[t] = (a2 > 0 ? a2 : 0) - qe2; d |= a2 > 0 ? 0x20 : 0;
                y2[t] = (b2 > 0 ? b2 : 0) - qe2; d |= b2 > 0 ? 0x40 : 0;
            } else {
                x[t] = (a > 0 ? a : 0) - qe;   d |= a >= 0 ? 0x08 : 0;
                y[t] = (b > 0 ? b : 0) - qe;   d |= b >= 0 ? 0x10 : 0;
                x2[t] = (a2 > 0 ? a2 : 0) - qe2; d |= a2 >= 0 ? 0x20 : 0;
                y2[t] = (b2 > 0 ? b2 : 0) - qe2; d |= b2 >= 0 ? 0x40 : 0;
            }
            pr[t - st] = (uint8_t)d;
            x1 = x_next, x21 = x2_next, v1 = vt_next; /* values of the previous anti-diagonal at t */
        }
        if (!approx_max) {
            int32_t max_H, max_t;
            if (r > 0) {
                int en1 = st + (en - st) / 4 * 4, i;
                int32_t HH[4], tt[4];
                max_H = H[en] = en > 0 ? H[en - 1] + u[en] : H[en] + v[en];
                max_t = en;
                for (i = 0; i < 4; ++i) HH[i] = NEG_INF, tt[i] = -1;
                for (t = st; t < en1; t += 4)
                    for (i = 0; i < 4; ++i) {
                        H[t + i] += v[t + i];
                        if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t + i;
                    }
                for (i = 0; i < 4; ++i) if (max_H < HH[i]) max_H = HH[i], max_t = tt[i];
                for (; t < en; ++t) { H[t] += v[t]; if (H[t] > max_H) max_H = H[t], max_t = t; }
            } else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
            if (en == tlen - 1 && H[en] > ez->mte) ez->mte = H[en], ez->mte_q = r - en;
            if (r - st == qlen - 1 && H[st] > ez->mqe) ez->mqe = H[st], ez->mqe_t = st;
            if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en == tlen - 1) ez->score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st && last_H0_t <= en && last_H0_t + 1 >= st && last_H0_t + 1 <= en) {
                    int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
                } else if (last_H0_t >= st && last_H0_t <= en) H0 += v[last_H0_t];
                else ++last_H0_t, H0 += u[last_H0_t];
            } else H0 = v[0] - qe, last_H0_t = 0;
            if (r == qlen + tlen - 2 && en == tlen - 1) ez->score = H0;
        }
        last_st0 = st, last_en0 = en;
    }
    {
        int rev_cigar = !!(flag & MMO_EZ_REV_CIGAR);
        int r_done = r < qlen + tlen - 1 ? r : qlen + tlen - 2;
        for (t = r_done + 1; t < qlen + tlen - 1; ++t) off[t] = 0x3fffffff, off_end[t] = -1;
        if (!ez->zdropped && !(flag & MMO_EZ_EXTZ_ONLY)) backtrack(ez, rev_cigar, p, off, off_end, n_col, tlen - 1, qlen - 1);
        else if (!ez->zdropped && (flag & MMO_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
            ez->reach_end = 1;
            backtrack(ez, rev_cigar, p, off, off_end, n_col, ez->mqe_t, qlen - 1);
        } else if (ez->max_t >= 0 && ez->max_q >= 0) backtrack(ez, rev_cigar, p, off, off_end, n_col, ez->max_t, ez->max_q);
    }
    free(u); free(H); free(p); free(off);
}

/* ---------------------------------------------------------------- base-level alignment of one hit */
static void append_cigar(mmo_reg *r, int n_cigar, const uint32_t *cigar)
{
    if (n_cigar == 0) return;
    r->cigar = (uint32_t *)realloc(r->cigar, (size_t)(r->n_cigar + n_cigar) * 4);
    r->has_p = 1;
    if (r->n_cigar > 0 && (r->cigar[r->n_cigar - 1] & 0xf) == (cigar[0] & 0xf)) {
        r->cigar[r->n_cigar - 1] += (cigar[0] >> 4) << 4;
        if (n_cigar > 1) memcpy(r->cigar + r->n_cigar, cigar + 1, (size_t)(n_cigar - 1) * 4);
        r->n_cigar += n_cigar - 1;
    } else {
        memcpy(r->cigar + r->n_cigar, cigar, (size_t)n_cigar * 4);
        r->n_cigar += n_cigar;
    }
}

static void fix_cigar(mmo_reg *r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
    int32_t toff = 0, qoff = 0, to_shrink = 0;
    int k;
    *qshift = *tshift = 0;
    if (r->n_cigar <= 1) return;
    for (k = 0; k < r->n_cigar; ++k) { /* left-align indels */
        uint32_t op = r->cigar[k] & 0xf, len = r->cigar[k] >> 4;
        if (len == 0) to_shrink = 1;
        if (op == 0) toff += len, qoff += len;
        else if (op == 1 || op == 2) {
            if (k > 0 && k < r->n_cigar - 1 && (r->cigar[k - 1] & 0xf) == 0 && (r->cigar[k + 1] & 0xf) == 0) {
                int l, prev_len = r->cigar[k - 1] >> 4;
                if (op == 1) { for (l = 0; l < prev_len; ++l) if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break; }
                else { for (l = 0; l < prev_len; ++l) if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break; }
                if (l > 0) r->cigar[k - 1] -= l << 4, r->cigar[k + 1] += l << 4, qoff -= l, toff -= l;
                if (l == prev_len) to_shrink = 1;
            }
            if (op == 1) qoff += len; else toff += len;
        }
    }
    for (k = 0; k < r->n_cigar - 2; ++k) { /* runs such as 5I6D7I become one I and one D */
        if ((r->cigar[k] & 0xf) > 0 && (r->cigar[k] & 0xf) + (r->cigar[k + 1] & 0xf) == 3) {
            uint32_t l, s[3] = {0, 0, 0};
            for (l = k; l < (uint32_t)r->n_cigar; ++l) {
                uint32_t op = r->cigar[l] & 0xf;
                if (op == 1 || op == 2 || r->cigar[l] >> 4 == 0) s[op] += r->cigar[l] >> 4;
                else break;
            }
            if (s[1] > 0 && s[2] > 0 && l - k > 2) {
                r->cigar[k] = s[1] << 4 | 1;
                r->cigar[k + 1] = s[2] << 4 | 2;
                for (k += 2; k < (int)l; ++k) r->cigar[k] &= 0xf;
                to_shrink = 1;
            }
            k = l;
        }
    }
    if (to_shrink) {
        int32_t l = 0;
        for (k = 0; k < r->n_cigar; ++k) if (r->cigar[k] >> 4 != 0) r->cigar[l++] = r->cigar[k];
        r->n_cigar = l;
        for (k = l = 0; k < r->n_cigar; ++k)
            if (k == r->n_cigar - 1 || (r->cigar[k] & 0xf) != (r->cigar[k + 1] & 0xf)) r->cigar[l++] = r->cigar[k];
            else r->cigar[k + 1] += r->cigar[k] >> 4 << 4;
        r->n_cigar = l;
    }
    if ((r->cigar[0] & 0xf) == 1 || (r->cigar[0] & 0xf) == 2) { /* no leading I/D */
        int32_t l = r->cigar[0] >> 4;
        if ((r->cigar[0] & 0xf) == 1) { if (r->rev) r->qe -= l; else r->qs += l; *qshift = l; }
        else r->rs += l, *tshift = l;
        --r->n_cigar;
        memmove(r->cigar, r->cigar + 1, (size_t)r->n_cigar * 4);
    }
}

static void update_extra(mmo_reg *r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e)
{
    int k;
    uint32_t l;
    int32_t s = 0, max = 0, qshift, tshift, toff = 0, qoff = 0;
    if (!r->has_p) return;
    fix_cigar(r, qseq, tseq, &qshift, &tshift);
    qseq += qshift, tseq += tshift;
    r->blen = r->mlen = 0;
    for (k = 0; k < r->n_cigar; ++k) {
        uint32_t op = r->cigar[k] & 0xf, len = r->cigar[k] >> 4;
        if (op == 0) {
            int n_ambi = 0, n_diff = 0;
            for (l = 0; l < len; ++l) {
                int cq = qseq[qoff + l], ct = tseq[toff + l];
                if (ct > 3 || cq > 3) ++n_ambi;
                else if (ct != cq) ++n_diff;
                s += mat[ct * 5 + cq];
                if (s < 0) s = 0; else max = max > s ? max : s;
            }
            r->blen += len - n_ambi, r->mlen += len - (n_ambi + n_diff), r->n_ambi += n_ambi;
            toff += len, qoff += len;
        } else if (op == 1) {
            int n_ambi = 0;
            for (l = 0; l < len; ++l) if (qseq[qoff + l] > 3) ++n_ambi;
            r->blen += len - n_ambi, r->n_ambi += n_ambi;
            s -= q + e * len;
            if (s < 0) s = 0;
            qoff += len;
        } else if (op == 2) {
            int n_ambi = 0;
            for (l = 0; l < len; ++l) if (tseq[toff + l] > 3) ++n_ambi;
            r->blen += len - n_ambi, r->n_ambi += n_ambi;
            s -= q + e * len;
            if (s < 0) s = 0;
            toff += len;
        }
    }
    r->dp_max = max;
}

static void fix_bad_ends(const mmo_reg *r, const mm128 *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
    int32_t i, l, m;
    *as = r->as, *cnt = r->cnt;
    if (r->cnt < 3) return;
    m = l = a[r->as].y >> 32 & 0xff;
    for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
        int32_t lq, lr, min, max, q_span = a[i].y >> 32 & 0xff;
        if (a[i].y & MMO_SEED_LONG_JOIN) break;
        lr = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
        if (max - min > l >> 1) *as = i;
        l += min;
        m += min < q_span ? min : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
    }
    *cnt = r->as + r->cnt - *as;
    m = l = a[r->as + r->cnt - 1].y >> 32 & 0xff;
    for (i = r->as + r->cnt - 2; i > *as; --i) {
        int32_t lq, lr, min, max, q_span = a[i + 1].y >> 32 & 0xff;
        if (a[i + 1].y & MMO_SEED_LONG_JOIN) break;
        lr = (int32_t)a[i + 1].x - (int32_t)a[i].x;
        lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        min = lr < lq ? lr : lq, max = lr > lq ? lr : lq;
        if (max - min > l >> 1) *cnt = i + 1 - *as;
        l += min;
        m += min < q_span ? min : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
    }
}

static void filter_bad_seeds(int as1, int cnt1, mm128 *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
    int max_st, max_en, n, i, k, max, *K;
    for (i = 1, n = 0; i < cnt1; ++i) {
        int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        if (gap < -min_gap || gap > min_gap) ++n;
    }
    if (n <= 1) return;
    K = (int *)malloc((size_t)n * sizeof(int));
    for (i = 1, n = 0; i < cnt1; ++i) {
        int gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        if (gap < -min_gap || gap > min_gap) K[n++] = i;
    }
    max = 0, max_st = max_en = -1;
    for (k = 0;; ++k) {
        int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
        if (k == n || k >= max_en) {
            if (max_en > 0) for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= MMO_SEED_IGNORE;
            max = 0, max_st = max_en = -1;
            if (k == n) break;
        }
        i = K[k];
        gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
        if (gap > 0) n_ins += gap; else n_del += -gap;
        qs = (int32_t)a[as1 + i - 1].y;
        rs = (int32_t)a[as1 + i - 1].x;
        for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            int j = K[l], diff;
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            if (gap > 0) n_ins += gap; else n_del += -gap;
            diff = n_ins + n_del - abs(n_ins - n_del);
            if (max_diff < diff) max_diff = diff, max_diff_l = l;
        }
        if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
    }
    free(K);
}

static void getseq(const mmo_idx *mi, int rid, int st, int en, uint8_t *out)
{
    if (en > st) memcpy(out, mi->seq4 + mi->off[rid] + st, (size_t)(en - st));
}
static void seq_rev(int len, uint8_t *s)
{
    int i;
    for (i = 0; i < len >> 1; ++i) { uint8_t t = s[i]; s[i] = s[len - 1 - i], s[len - 1 - i] = t; }
}

static void align_pair(const mmo_opt *opt, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq, int w,
                       int end_bonus, int zdrop, int flag, mmo_ez *ez)
{
    if (opt->max_sw_mat > 0 && (int64_t)tlen * qlen > opt->max_sw_mat) { ez_reset(ez); ez->zdropped = 1; return; }
    mmo_extd2(qlen, qseq, tlen, tseq, (int8_t)opt->a, (int8_t)-opt->b, (int8_t)-opt->sc_ambi, (int8_t)opt->q,
              (int8_t)opt->e, (int8_t)opt->q2, (int8_t)opt->e2, w, zdrop, end_bonus, flag, ez);
}

/* Local alignment score with affine gaps (a gap of length L costs q + L * e), as minimap2's ksw_ll_i16 computes it for the
 * inversion code: plain recurrences; the reported end is the first target column (scanning left to right) whose column maximum
 * exceeds everything before it, and in that column the smallest query index holding the maximum (ksw's own tie rule). */
static int ll_local(int ql, const uint8_t *q, int tl, const uint8_t *t, const int8_t *mat, int gapo, int gape, int *qe, int *te)
{
    int i, j, gmax = 0, *H, *E;
    *qe = *te = -1;
    if (ql <= 0 || tl <= 0) return 0;
    H = (int *)calloc((size_t)ql + 1, sizeof(int));
    E = (int *)calloc((size_t)ql + 1, sizeof(int));
    for (i = 0; i < tl; ++i) {
        int f = 0, diag = 0, cmax = 0, cq = -1;
        for (j = 0; j < ql; ++j) {
            int h = diag + mat[t[i] * 5 + q[j]], e = E[j + 1];
            diag = H[j + 1];
            if (h < e) h = e;
            if (h < f) h = f;
            if (h < 0) h = 0;
            H[j + 1] = h;
            if (h > cmax) cmax = h, cq = j;
            e -= gape; if (e < h - gapo - gape) e = h - gapo - gape; if (e < 0) e = 0;
            f -= gape; if (f < h - gapo - gape) f = h - gapo - gape; if (f < 0) f = 0;
            E[j + 1] = e;
        }
        if (cmax > gmax) gmax = cmax, *te = i, *qe = cq;
    }
    free(H); free(E);
    return gmax;
}

/* score drop along the alignment path: 1 = exceeds zdrop (the exact second pass is run), 2 = the region of the largest drop
 * aligns to its own reverse complement well enough to be an inversion (second pass with zdrop_inv, the remainder of the hit is
 * marked split_inv) -- mm_test_zdrop */
static int test_zdrop(const mmo_opt *opt, const uint8_t *qseq, const uint8_t *tseq, int n_cigar, const uint32_t *cigar,
                      const int8_t *mat)
{
    int k, pos[2][2] = {{-1, -1}, {-1, -1}}, q_len, t_len;
    int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
#define MMO_UPD(I, J) do { \
        if (score < max) { \
            int li = (I) - max_i, lj = (J) - max_j, diff = li > lj ? li - lj : lj - li; \
            int z = max - score - diff * opt->e; \
            if (z > max_zdrop) { max_zdrop = z; pos[0][0] = max_i, pos[0][1] = max_j, pos[1][0] = (I), pos[1][1] = (J); } \
        } else max = score, max_i = (I), max_j = (J); \
    } while (0)
    for (k = 0; k < n_cigar; ++k) {
        uint32_t l, op = cigar[k] & 0xf, len = cigar[k] >> 4;
        if (op == 0) {
            for (l = 0; l < len; ++l) {
                score += mat[tseq[i + l] * 5 + qseq[j + l]];
                MMO_UPD(i + (int)l, j + (int)l);
            }
            i += len, j += len;
        } else if (op == 1 || op == 2) {
            score -= opt->q + opt->e * len;
            if (op == 1) j += len; else i += len;
            MMO_UPD(i, j);
        }
    }
#undef MMO_UPD
    q_len = pos[1][1] - pos[0][1], t_len = pos[1][0] - pos[0][0];
    if (max_zdrop > opt->zdrop_inv && q_len < opt->max_gap && t_len < opt->max_gap && q_len > 0 && t_len > 0) {
        uint8_t *qseq2 = (uint8_t *)malloc((size_t)q_len);
        int q_off, t_off, sc;
        for (k = 0; k < q_len; ++k) { int c = qseq[pos[1][1] - k - 1]; qseq2[k] = c >= 4 ? 4 : 3 - c; }
        sc = ll_local(q_len, qseq2, t_len, tseq + pos[0][0], mat, opt->q, opt->e, &q_off, &t_off);
        free(qseq2);
        if (sc >= opt->min_chain_score * opt->a && sc >= opt->min_dp_max) return 2;
    }
    return max_zdrop > opt->zdrop ? 1 : 0;
}

static void align1(const mmo_opt *opt, const mmo_idx *mi, int qlen, uint8_t *qseq0[2], mmo_reg *r, mmo_reg *r2,
                   int n_a, mm128 *a, mmo_ez *ez)
{
    int32_t rid = a[r->as].x << 1 >> 33, rev = a[r->as].x >> 63, as1, cnt1;
    uint8_t *tseq, *qseq;
    int32_t i, l, bw, dropped = 0, rs0, re0, qs0, qe0;
    int32_t rs, re, qs, qe, rs1, qs1, re1, qe1;
    const int32_t tlen_all = mi->len[rid], kh = mi->k >> 1;
    int8_t mat[25];

    r2->cnt = 0;
    if (r->cnt == 0) return;
    for (i = 0; i < 4; ++i) { int j; for (j = 0; j < 4; ++j) mat[i * 5 + j] = i == j ? opt->a : -opt->b; mat[i * 5 + 4] = -opt->sc_ambi; }
    for (i = 0; i < 5; ++i) mat[20 + i] = -opt->sc_ambi;
    bw = (int)(opt->bw * 1.5 + 1.);

    fix_bad_ends(r, a, opt->bw, opt->min_chain_score * 2, &as1, &cnt1);
    filter_bad_seeds(as1, cnt1, a, 10, 40, opt->max_gap >> 1, 10);
    rs = (int32_t)a[as1].x - kh, qs = (int32_t)a[as1].y - kh;                     /* k-mer centres */
    re = (int32_t)a[as1 + cnt1 - 1].x - kh, qe = (int32_t)a[as1 + cnt1 - 1].y - kh;

    /* region allowed for the two end extensions */
    rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
    qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    for (i = r->as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r->as].x >> 32; --i) {
        int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        if (x < rs0 && y < qs0) {
            if (++l > opt->min_cnt) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l, qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < opt->max_gap ? qs : opt->max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * opt->a > opt->q ? (l * opt->a - opt->q) / opt->e : 0;
        l = l < opt->max_gap ? l : opt->max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else rs0 = rs, qs0 = qs;
    re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
    qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
    re1 = tlen_all, qe1 = qlen;
    for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x >> 32 == a[r->as].x >> 32; ++i) {
        int32_t x = (int32_t)a[i].x + 1, y = (int32_t)a[i].y + 1;
        if (x > re0 && y > qe0) {
            if (++l > opt->min_cnt) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l, qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < tlen_all) {
        l = qlen - qe < opt->max_gap ? qlen - qe : opt->max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * opt->a > opt->q ? (l * opt->a - opt->q) / opt->e : 0;
        l = l < opt->max_gap ? l : opt->max_gap;
        l = l < tlen_all - re ? l : tlen_all - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else re0 = re, qe0 = qe;

    tseq = (uint8_t *)malloc((size_t)(re0 - rs0 > 0 ? re0 - rs0 : 1));

    if (qs > 0 && rs > 0) { /* left extension on reversed sequences, gaps right-aligned */
        qseq = &qseq0[rev][qs0];
        getseq(mi, rid, rs0, rs, tseq);
        seq_rev(qs - qs0, qseq);
        seq_rev(rs - rs0, tseq);
        align_pair(opt, qs - qs0, qseq, rs - rs0, tseq, bw, opt->end_bonus, r->split_inv ? opt->zdrop_inv : opt->zdrop,
                   MMO_EZ_EXTZ_ONLY | MMO_EZ_RIGHT | MMO_EZ_REV_CIGAR, ez);
        if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->dp_score += ez->max; }
        rs1 = rs - (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
        qs1 = qs - (ez->reach_end ? qs - qs0 : ez->max_q + 1);
        seq_rev(qs - qs0, qseq);
    } else rs1 = rs, qs1 = qs;
    re1 = rs, qe1 = qs;

    for (i = 1; i < cnt1; ++i) { /* fill between anchors */
        if ((a[as1 + i].y & (MMO_SEED_IGNORE | MMO_SEED_TANDEM)) && i != cnt1 - 1) continue;
        re = (int32_t)a[as1 + i].x - kh, qe = (int32_t)a[as1 + i].y - kh;
        re1 = re, qe1 = qe;
        if (i == cnt1 - 1 || (a[as1 + i].y & MMO_SEED_LONG_JOIN) || (qe - qs >= opt->min_ksw_len && re - rs >= opt->min_ksw_len)) {
            int j, bw1 = bw;
            if (a[as1 + i].y & MMO_SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
            qseq = &qseq0[rev][qs];
            getseq(mi, rid, rs, re, tseq);
            int zdrop_code;
            align_pair(opt, qe - qs, qseq, re - rs, tseq, bw1, -1, opt->zdrop, MMO_EZ_APPROX_MAX, ez);
            if ((zdrop_code = test_zdrop(opt, qseq, tseq, ez->n_cigar, ez->cigar, mat)) != 0)
                align_pair(opt, qe - qs, qseq, re - rs, tseq, bw1, -1, zdrop_code == 2 ? opt->zdrop_inv : opt->zdrop, 0, ez);
            if (ez->n_cigar > 0) append_cigar(r, ez->n_cigar, ez->cigar);
            if (ez->zdropped) { /* the alignment broke: keep the left part, hand the rest back as a new hit */
                r->has_p = 1;
                for (j = i - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
                dropped = 1;
                if (j < 0) j = 0;
                r->dp_score += ez->max;
                re1 = rs + (ez->max_t + 1);
                qe1 = qs + (ez->max_q + 1);
                if (cnt1 - (j + 1) >= opt->min_cnt) {
                    split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
                    if (zdrop_code == 2) r2->split_inv = 1;
                }
                break;
            } else r->dp_score += ez->score;
            rs = re, qs = qe;
        }
    }

    if (!dropped && qe < qe0 && re < re0) { /* right extension */
        qseq = &qseq0[rev][qe];
        getseq(mi, rid, re, re0, tseq);
        align_pair(opt, qe0 - qe, qseq, re0 - re, tseq, bw, opt->end_bonus, opt->zdrop, MMO_EZ_EXTZ_ONLY, ez);
        if (ez->n_cigar > 0) { append_cigar(r, ez->n_cigar, ez->cigar); r->dp_score += ez->max; }
        re1 = re + (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
        qe1 = qe + (ez->reach_end ? qe0 - qe : ez->max_q + 1);
    }

    r->rs = rs1, r->re = re1;
    if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
    else r->qs = qs1, r->qe = qe1;
    if (r->has_p) {
        free(tseq);
        tseq = (uint8_t *)malloc((size_t)(re1 - rs1 > 0 ? re1 - rs1 : 1));
        getseq(mi, rid, rs1, re1, tseq);
        update_extra(r, &qseq0[r->rev][qs1], tseq, mat, (int8_t)opt->q, (int8_t)opt->e);
    }
    free(tseq);
}

/* the inverted segment between the two halves of a hit that was split at an inversion (mm_align1_inv): a local alignment of
 * the reverse strand of the read's gap against the target's gap locates its start, an extension from there gives the hit */
static int align1_inv(const mmo_opt *opt, const mmo_idx *mi, int qlen, uint8_t *qseq0[2], const mmo_reg *r1, const mmo_reg *r2,
                      mmo_reg *r_inv, mmo_ez *ez)
{
    int tl, ql, score, ret = 0, q_off, t_off, i;
    uint8_t *tseq, *qseq;
    int8_t mat[25];
    memset(r_inv, 0, sizeof(mmo_reg));
    if (!(r1->split & 1) || !(r2->split & 2)) return 0;
    if (r1->id != r1->parent && r1->parent != PARENT_TMP_PRI) return 0;
    if (r2->id != r2->parent && r2->parent != PARENT_TMP_PRI) return 0;
    if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
    ql = r1->rev ? r1->qs - r2->qe : r2->qs - r1->qe;
    tl = r2->rs - r1->re;
    if (ql < opt->min_chain_score || ql > opt->max_gap) return 0;
    if (tl < opt->min_chain_score || tl > opt->max_gap) return 0;
    for (i = 0; i < 4; ++i) { int j; for (j = 0; j < 4; ++j) mat[i * 5 + j] = i == j ? opt->a : -opt->b; mat[i * 5 + 4] = -opt->sc_ambi; }
    for (i = 0; i < 5; ++i) mat[20 + i] = -opt->sc_ambi;
    tseq = (uint8_t *)malloc((size_t)tl);
    getseq(mi, r1->rid, r1->re, r2->rs, tseq);
    qseq = r1->rev ? &qseq0[0][r2->qe] : &qseq0[1][qlen - r2->qs];
    seq_rev(ql, qseq);
    seq_rev(tl, tseq);
    score = ll_local(ql, qseq, tl, tseq, mat, opt->q, opt->e, &q_off, &t_off);
    seq_rev(ql, qseq);
    seq_rev(tl, tseq);
    if (score < opt->min_dp_max) goto end_inv;
    q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
    align_pair(opt, ql - q_off, qseq + q_off, tl - t_off, tseq + t_off, (int)(opt->bw * 1.5), -1, opt->zdrop, MMO_EZ_EXTZ_ONLY, ez);
    if (ez->n_cigar == 0) goto end_inv;
    append_cigar(r_inv, ez->n_cigar, ez->cigar);
    r_inv->dp_score = ez->max;
    r_inv->id = -1;
    r_inv->parent = PARENT_UNSET;
    r_inv->inv = 1;
    r_inv->rev = !r1->rev;
    r_inv->rid = r1->rid;
    if (r_inv->rev == 0) {
        r_inv->qs = r2->qe + q_off;
        r_inv->qe = r_inv->qs + ez->max_q + 1;
    } else {
        r_inv->qe = r2->qs - q_off;
        r_inv->qs = r_inv->qe - (ez->max_q + 1);
    }
    r_inv->rs = r1->re + t_off;
    r_inv->re = r_inv->rs + ez->max_t + 1;
    update_extra(r_inv, &qseq[q_off], &tseq[t_off], mat, (int8_t)opt->q, (int8_t)opt->e);
    ret = 1;
end_inv:
    free(tseq);
    return ret;
}

static mmo_reg *insert_reg(const mmo_reg *r, int i, int *n_regs, mmo_reg *regs)
{
    regs = (mmo_reg *)realloc(regs, (size_t)(*n_regs + 1) * sizeof(mmo_reg));
    if (i + 1 != *n_regs) memmove(&regs[i + 2], &regs[i + 1], sizeof(mmo_reg) * (size_t)(*n_regs - i - 1));
    regs[i + 1] = *r;
    ++*n_regs;
    return regs;
}

static mmo_reg *align_skeleton(const mmo_opt *opt, const mmo_idx *mi, int qlen, const char *qstr, int *n_regs_,
                               mmo_reg *regs, mm128 *a)
{
    int32_t i, n_regs = *n_regs_, n_a;
    uint8_t *qseq0[2];
    mmo_ez ez;
    qseq0[0] = (uint8_t *)malloc((size_t)qlen * 2);
    qseq0[1] = qseq0[0] + qlen;
    for (i = 0; i < qlen; ++i) {
        qseq0[0][i] = nt4((unsigned char)qstr[i]);
        qseq0[1][qlen - 1 - i] = qseq0[0][i] < 4 ? 3 - qseq0[0][i] : 4;
    }
    n_a = squeeze_a(n_regs, regs, a);
    memset(&ez, 0, sizeof(ez));
    for (i = 0; i < n_regs; ++i) {
        mmo_reg r2;
        align1(opt, mi, qlen, qseq0, &regs[i], &r2, n_a, a, &ez);
        if (r2.cnt > 0) regs = insert_reg(&r2, i, &n_regs, regs);
        if (i > 0 && regs[i].split_inv) {
            if (align1_inv(opt, mi, qlen, qseq0, &regs[i - 1], &regs[i], &r2, &ez)) {
                regs = insert_reg(&r2, i, &n_regs, regs);
                ++i; /* skip the inserted INV alignment */
            }
        }
    }
    *n_regs_ = n_regs;
    free(qseq0[0]);
    free(ez.cigar);
    filter_regs(opt, qlen, n_regs_, regs);
    hit_sort(n_regs_, regs);
    return regs;
}

/* ---------------------------------------------------------------- driver */
void mmo_opt_init(mmo_opt *o)
{
    memset(o, 0, sizeof(*o));
    o->mid_occ_frac = 2e-4f;
    o->min_cnt = 3, o->min_chain_score = 40, o->bw = 500, o->max_gap = 5000;
    o->max_chain_skip = 25, o->max_chain_iter = 5000;
    o->mask_level = 0.5f, o->pri_ratio = 0.8f, o->best_n = 5;
    o->max_join_long = 20000, o->max_join_short = 2000, o->min_join_flank_sc = 1000, o->min_join_flank_ratio = 0.5f;
    o->a = 2, o->b = 4, o->q = 4, o->e = 2, o->q2 = 24, o->e2 = 1;
    o->sc_ambi = 1, o->zdrop = 400, o->zdrop_inv = 200, o->end_bonus = -1;
    o->min_dp_max = o->min_chain_score * o->a;
    o->min_ksw_len = 200;
    o->max_clip_ratio = 1.0f;
    o->max_sw_mat = 100000000;
    o->with_cigar = 1;
    o->seed = 11;
}

mmo_reg *mmo_map_read(const mmo_idx *mi, const mmo_opt *o, const char *name, const char *seq, int32_t qlen,
                      int32_t *n_regs_, int32_t *rep_len_)
{
    mm128 *mv = 0, *a = 0, *b = 0;
    uint64_t *u = 0;
    int64_t n_mv, n_a;
    int32_t n_u = 0, rep_len = 0, n_regs;
    int32_t mid_occ = o->mid_occ > 0 ? o->mid_occ : mmo_idx_cal_max_occ(mi, o->mid_occ_frac);
    uint32_t hash;
    mmo_reg *regs;
    *n_regs_ = 0;
    if (rep_len_) *rep_len_ = 0;
    if (qlen <= 0) return 0;
    n_mv = mmo_sketch(seq, qlen, mi->w, mi->k, 0, &mv);
    n_a = mmo_collect_anchors(mi, mid_occ, mv, n_mv, qlen, &a, &rep_len);
    free(mv);
    if (rep_len_) *rep_len_ = rep_len;
    mmo_chain(o, n_a, a, &n_u, &u, &b);
    free(a);
    if (n_u == 0) { free(u); free(b); return 0; }
    hash = name ? x31_hash(name) : 0;
    hash ^= wang32((uint32_t)qlen) + wang32(o->seed);
    hash = wang32(hash);
    regs = gen_regs(hash, qlen, n_u, u, b);
    n_regs = n_u;
    free(u);
    set_parent(o->mask_level, n_regs, regs, o->a * 2 + o->b);
    select_sub(o->pri_ratio, mi->k * 2, o->best_n, &n_regs, regs);
    join_long(o, qlen, &n_regs, regs, b);
    if (o->with_cigar) {
        regs = align_skeleton(o, mi, qlen, seq, &n_regs, regs, b);
        set_parent(o->mask_level, n_regs, regs, o->a * 2 + o->b);
        select_sub(o->pri_ratio, mi->k * 2, o->best_n, &n_regs, regs);
        set_sam_pri(n_regs, regs);
    }
    set_mapq(n_regs, regs, o->min_chain_score, o->a, rep_len);
    free(b);
    *n_regs_ = n_regs;
    return regs;
}

/* ---- split index: minimap2 -I parts + --split-prefix (map.c: mm_split_merge / merge_hits), restated ------------------
 * Every part is mapped on its own (its own mid-occ cut-off); the hits of a read from all parts are then pooled with the
 * target ids shifted to the concatenated target list, the sub-optimal bookkeeping (subsc, n_sub, dp_max2) is cleared,
 * and ranking, parent/secondary grouping, the -p/-N selection, the SAM-primary flag and MAPQ are computed again over
 * the pool.  The repetitive-seed length of the read is the largest over the parts.  PARITY UNPINNED (see header). */
mmo_idx *mmo_idx_concat_names(int32_t n_parts, const mmo_idx **parts)
{
    mmo_idx *mi = (mmo_idx *)calloc(1, sizeof(mmo_idx));
    int32_t p, s, n = 0;
    for (p = 0; p < n_parts; ++p) n += parts[p]->n_seq;
    mi->k = parts[0]->k, mi->w = parts[0]->w, mi->n_seq = n;
    mi->name = (char **)calloc(n > 0 ? n : 1, sizeof(char *));
    mi->len = (int32_t *)calloc(n > 0 ? n : 1, 4);
    mi->off = (int64_t *)calloc((size_t)n + 1, 8);
    for (p = 0, n = 0; p < n_parts; ++p)
        for (s = 0; s < parts[p]->n_seq; ++s, ++n) mi->name[n] = strdup(parts[p]->name[s]), mi->len[n] = parts[p]->len[s];
    return mi;  /* names and lengths only: for the writers */
}

mmo_reg *mmo_map_read_split(int32_t n_parts, const mmo_idx **parts, const mmo_opt *o, const char *name, const char *seq,
                            int32_t qlen, int32_t *n_regs_, int32_t *rep_len_)
{
    mmo_reg *all = 0;
    int32_t p, i, n_all = 0, rep_len = 0, rid0 = 0;
    for (p = 0; p < n_parts; ++p) {
        int32_t n = 0, rep = 0;
        mmo_opt op = *o;
        mmo_reg *r;
        op.mid_occ = 0; /* every part has its own cut-off (-f quantile of ITS keys) */
        r = mmo_map_read(parts[p], &op, name, seq, qlen, &n, &rep);
        if (n > 0) {
            all = (mmo_reg *)realloc(all, (size_t)(n_all + n) * sizeof(mmo_reg));
            for (i = 0; i < n; ++i) { all[n_all + i] = r[i]; all[n_all + i].rid += rid0; }
            n_all += n;
            free(r); /* (the cigars moved) */
        }
        if (rep > rep_len) rep_len = rep;
        rid0 += parts[p]->n_seq;
    }
    for (i = 0; i < n_all; ++i) { all[i].subsc = 0; all[i].n_sub = 0; if (all[i].has_p) all[i].dp_max2 = 0; }
    if (n_all > 0) {
        hit_sort(&n_all, all);
        set_parent(o->mask_level, n_all, all, o->a * 2 + o->b);
        select_sub(o->pri_ratio, parts[0]->k * 2, o->best_n, &n_all, all);
        set_sam_pri(n_all, all);
        set_mapq(n_all, all, o->min_chain_score, o->a, rep_len);
    }
    *n_regs_ = n_all;
    if (rep_len_) *rep_len_ = rep_len;
    return all;
}

void mmo_free_regs(mmo_reg *r, int32_t n)
{
    int32_t i;
    if (!r) return;
    for (i = 0; i < n; ++i) free(r[i].cigar);
    free(r);
}

static double event_identity(const mmo_reg *r)
{
    int32_t i, n_gapo = 0, n_gap = 0;
    for (i = 0; i < r->n_cigar; ++i) {
        int32_t op = r->cigar[i] & 0xf, len = r->cigar[i] >> 4;
        if (op == 1 || op == 2) ++n_gapo, n_gap += len;
    }
    return (double)r->mlen / (r->blen + r->n_ambi - n_gap + n_gapo);
}

int64_t mmo_write_paf(const mmo_idx *mi, const mmo_opt *o, const char *name, int32_t qlen, const mmo_reg *regs,
                      int32_t n_regs, int32_t rep_len, char *buf, int64_t cap)
{
    int64_t n = 0;
    int32_t i, k;
    for (i = 0; i < n_regs; ++i) {
        const mmo_reg *r = &regs[i];
        int type = r->id == r->parent ? (r->inv ? 'I' : 'P') : (r->inv ? 'i' : 'S');
        int64_t need = 512 + (int64_t)strlen(name) + (int64_t)strlen(mi->name[r->rid]) + (int64_t)r->n_cigar * 12;
        if (n + need > cap) return -1;
        n += sprintf(buf + n, "%s\t%d\t%d\t%d\t%c\t%s\t%d\t%d\t%d\t%d\t%d\t%d", name, qlen, r->qs, r->qe, "+-"[r->rev],
                     mi->name[r->rid], mi->len[r->rid], r->rs, r->re, r->mlen, r->blen, r->mapq);
        if (r->has_p)
            n += sprintf(buf + n, "\tNM:i:%d\tms:i:%d\tAS:i:%d\tnn:i:%d", r->blen - r->mlen + r->n_ambi, r->dp_max,
                         r->dp_score, r->n_ambi);
        n += sprintf(buf + n, "\ttp:A:%c\tcm:i:%d\ts1:i:%d", type, r->cnt, r->score);
        if (r->parent == r->id) n += sprintf(buf + n, "\ts2:i:%d", r->subsc);
        if (r->has_p) {
            double div = 1.0 - event_identity(r);
            if (div == 0.0) n += sprintf(buf + n, "\tde:f:0");
            else n += sprintf(buf + n, "\tde:f:%.4f", div);
        }
        if (r->split) n += sprintf(buf + n, "\tzd:i:%d", r->split);
        n += sprintf(buf + n, "\trl:i:%d", rep_len);
        if (r->has_p && o->with_cigar) {
            n += sprintf(buf + n, "\tcg:Z:");
            for (k = 0; k < r->n_cigar; ++k) n += sprintf(buf + n, "%d%c", r->cigar[k] >> 4, "MIDNSH"[r->cigar[k] & 0xf]);
        }
        buf[n++] = '\n';
    }
    buf[n] = 0;
    return n;
}


/* ---- SAM records (minimap2 2.17 -a: format.c mm_write_sam3 / write_sam_cigar / sam_write_sq), single-segment reads,
 * no read group, qualities not carried ('*').  Unmapped reads get a flag-4 record. ---- */
static char sam_comp(char c)
{
    static const char *from = "ACGTUNRYKMSWBDHVacgtunrykmswbdhv", *to = "TGCAANYRMKSWVHDBtgcaanyrmkswvhdb";
    const char *p = strchr(from, c);
    return p && c ? to[p - from] : c;
}

static int64_t sam_seq(char *buf, const char *seq, int32_t st, int32_t en, int rev)
{
    int32_t i, n = 0;
    if (!rev) for (i = st; i < en; ++i) buf[n++] = seq[i];
    else for (i = en - 1; i >= st; --i) buf[n++] = sam_comp(seq[i]);
    return n;
}

/* QUAL of a record whose SEQ is seq[st, en) on strand rev (mm_write_sam3 prints the qualities in the orientation of SEQ;
 * '*' when the read has none) */
static int64_t sam_qual(char *buf, const char *qual, int32_t st, int32_t en, int rev)
{
    int32_t i, n = 0;
    buf[n++] = '\t';
    if (!qual) { buf[n++] = '*'; return n; }
    if (!rev) for (i = st; i < en; ++i) buf[n++] = qual[i];
    else for (i = en - 1; i >= st; --i) buf[n++] = qual[i];
    return n;
}

int64_t mmo_write_sam(const mmo_idx *mi, const mmo_opt *o, const char *name, int32_t qlen, const char *seq, const mmo_reg *regs,
                      int32_t n_regs, int32_t rep_len, char *buf, int64_t cap)
{
    return mmo_write_sam_q(mi, o, name, qlen, seq, 0, regs, n_regs, rep_len, buf, cap);
}

int64_t mmo_write_sam_q(const mmo_idx *mi, const mmo_opt *o, const char *name, int32_t qlen, const char *seq, const char *qual,
                        const mmo_reg *regs, int32_t n_regs, int32_t rep_len, char *buf, int64_t cap)
{
    int64_t n = 0;
    int32_t i, j, k;
    (void)o;
    if (n_regs == 0) {
        if ((int64_t)strlen(name) + 2 * (int64_t)qlen + 64 > cap) return -1;
        n += sprintf(buf + n, "%s\t4\t*\t0\t0\t*\t*\t0\t0\t", name);
        n += sam_seq(buf + n, seq, 0, qlen, 0);
        n += sam_qual(buf + n, qual, 0, qlen, 0);
        n += sprintf(buf + n, "\trl:i:%d\n", rep_len);
        buf[n] = 0;
        return n;
    }
    for (i = 0; i < n_regs; ++i) {
        const mmo_reg *r = &regs[i];
        int flag = 0, type = r->id == r->parent ? (r->inv ? 'I' : 'P') : (r->inv ? 'i' : 'S');
        int64_t need = 1024 + (int64_t)strlen(name) + (int64_t)strlen(mi->name[r->rid]) + (int64_t)r->n_cigar * 12 + 2 * (int64_t)qlen + (int64_t)n_regs * 128;
        if (n + need > cap) return -1;
        if (r->rev) flag |= 0x10;
        if (r->parent != r->id) flag |= 0x100;
        else if (!r->sam_pri) flag |= 0x800;
        n += sprintf(buf + n, "%s\t%d\t%s\t%d\t%d\t", name, flag, mi->name[r->rid], r->rs + 1, r->mapq);
        if (!r->has_p) buf[n++] = '*';
        else {
            int32_t clip0 = r->rev ? qlen - r->qe : r->qs, clip1 = r->rev ? r->qs : qlen - r->qe;
            int clip_char = (flag & 0x800) ? 'H' : 'S';
            if (clip0) n += sprintf(buf + n, "%d%c", clip0, clip_char);
            for (k = 0; k < r->n_cigar; ++k) n += sprintf(buf + n, "%d%c", r->cigar[k] >> 4, "MIDNSH"[r->cigar[k] & 0xf]);
            if (clip1) n += sprintf(buf + n, "%d%c", clip1, clip_char);
        }
        n += sprintf(buf + n, "\t*\t0\t0\t");
        if ((flag & 0x900) == 0) { n += sam_seq(buf + n, seq, 0, qlen, r->rev); n += sam_qual(buf + n, qual, 0, qlen, r->rev); }
        else if (flag & 0x100) n += sprintf(buf + n, "*\t*");
        else { n += sam_seq(buf + n, seq, r->qs, r->qe, r->rev); n += sam_qual(buf + n, qual, r->qs, r->qe, r->rev); }
        if (r->has_p)
            n += sprintf(buf + n, "\tNM:i:%d\tms:i:%d\tAS:i:%d\tnn:i:%d", r->blen - r->mlen + r->n_ambi, r->dp_max, r->dp_score, r->n_ambi);
        n += sprintf(buf + n, "\ttp:A:%c\tcm:i:%d\ts1:i:%d", type, r->cnt, r->score);
        if (r->parent == r->id) n += sprintf(buf + n, "\ts2:i:%d", r->subsc);
        if (r->has_p) {
            double div = 1.0 - event_identity(r);
            if (div == 0.0) n += sprintf(buf + n, "\tde:f:0");
            else n += sprintf(buf + n, "\tde:f:%.4f", div);
        }
        if (r->split) n += sprintf(buf + n, "\tzd:i:%d", r->split);
        if (r->parent == r->id && r->has_p && n_regs > 1) {  /* SA: the other non-secondary hits with a CIGAR */
            int n_sa = 0;
            for (j = 0; j < n_regs; ++j) if (j != i && regs[j].parent == regs[j].id && regs[j].has_p) ++n_sa;
            if (n_sa > 0) {
                n += sprintf(buf + n, "\tSA:Z:");
                for (j = 0; j < n_regs; ++j) {
                    const mmo_reg *q = &regs[j];
                    int32_t l_M, l_I = 0, l_D = 0, c5, c3;
                    if (j == i || q->parent != q->id || !q->has_p) continue;
                    if (q->qe - q->qs < q->re - q->rs) { l_M = q->qe - q->qs; l_D = (q->re - q->rs) - l_M; }
                    else { l_M = q->re - q->rs; l_I = (q->qe - q->qs) - l_M; }
                    c5 = q->rev ? qlen - q->qe : q->qs; c3 = q->rev ? q->qs : qlen - q->qe;
                    n += sprintf(buf + n, "%s,%d,%c,", mi->name[q->rid], q->rs + 1, "+-"[q->rev]);
                    if (c5) n += sprintf(buf + n, "%dS", c5);
                    if (l_M) n += sprintf(buf + n, "%dM", l_M);
                    if (l_I) n += sprintf(buf + n, "%dI", l_I);
                    if (l_D) n += sprintf(buf + n, "%dD", l_D);
                    if (c3) n += sprintf(buf + n, "%dS", c3);
                    n += sprintf(buf + n, ",%d,%d;", q->mapq, q->blen - q->mlen + q->n_ambi);
                }
            }
        }
        n += sprintf(buf + n, "\trl:i:%d\n", rep_len);
    }
    buf[n] = 0;
    return n;
}
