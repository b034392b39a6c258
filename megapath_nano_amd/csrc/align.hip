// mpn_map_batch: chains -> hits -> base-level extension -> MAPQ -> PAF  (include/mpn_map.h).
//
// Device work: seed/chain kernels (map_kernels.h) and the extension kernels (ext_kernels.h).
// Host work (this file, multi-threaded over reads): the per-read hit bookkeeping of minimap2's hit.c/align.c --
// ranking chains, primary/secondary marking, long-join, planning the DP windows of every hit, stitching the
// CIGARs the GPU returns, MAPQ (float logf, evaluated on the host so that it is libm-exact) and PAF text.
// All DP windows of a batch are planned up front and run as independent GPU jobs (one wavefront each); a hit
// whose alignment z-drops is split and its remainder goes through another round.
#include "map_types.h"
#include "ext_kernels.h"
#include "fin_kernels.h"
#include "stitch_kernels.h"
#include "plan_kernels.h"
#include "hit_kernels.h"
#include "mapper_internal.h"
#include "../../include/mpn_map.h"
#include "../../include/mpn_ssw.h"

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <thread>
#include <pthread.h>
#include <time.h>
#include <atomic>
#include <functional>
#include <mutex>
#include <stdlib.h>
#include <string>
#include <vector>

namespace mpn {

int upload_seqs(int32_t n, const char *seqs, const int64_t *seq_off, const int32_t *seq_len, DevBuf<uint8_t> &d_seqs,
                DevBuf<int64_t> &d_off, DevBuf<int32_t> &d_len, int64_t *total_bases, hipStream_t st);
int seed_chain_device(const mpn_index *idx, const mpn_map_opt *opt, int n, const uint8_t *d_seqs, const int64_t *d_off,
                      const int32_t *d_len, const int32_t *h_len, SeedChainOut &o, hipStream_t st, const ReadSketch *pre);
int sketch_reads(int k, int w, int n, const uint8_t *d_seqs, const int64_t *d_off, const int32_t *d_len, const int32_t *h_len, ReadSketch &sk,
                 hipStream_t st);
int download_chains(int n, SeedChainOut &o, HostChains &h, PoolBuf &pin_u, PoolBuf &pin_b, hipStream_t st, int mode);
int download_chain_records(SeedChainOut &o, HostChains &h, PoolBuf &pin_u, hipStream_t st);

const char *get_error();

static bool g_cpu_on = false;                 // MPN_DEBUG_CPU=1: thread CPU time of every parallel region, by tag
static std::atomic<long long> g_cpu_ns[32];
static const int PARENT_UNSET = -1, PARENT_TMP_PRI = -2;
static const uint64_t SEED_LONG_JOIN = 1ULL << 40, SEED_IGNORE = 1ULL << 41, SEED_TANDEM = 1ULL << 42;

struct Reg {
    int32_t id = 0, cnt = 0, rid = 0, score = 0, qs = 0, qe = 0, rs = 0, re = 0, parent = PARENT_UNSET, subsc = 0, as = 0;
    int32_t mlen = 0, blen = 0, n_sub = 0, score0 = 0;
    uint32_t mapq = 0, split = 0, rev = 0, inv = 0, sam_pri = 0, split_inv = 0, hash = 0;
    int32_t has_p = 0, dp_score = 0, dp_max = 0, dp_max2 = 0, n_ambi = 0;
    std::vector<uint32_t> cigar;
    int32_t aligned = 0;  // base-level extension already done (or not needed)
    int32_t fin_qs = 0, fin_rs = 0, fin_ql = 0, fin_tl = 0;   // an inversion hit before its extension: the window to extend
    int32_t fin_idx = -1;  // >= 0: stitched this round (its index among the round's hits): CIGAR fix-up and statistics are due
    // The hit's first and last chained anchor (the anchors themselves stay in HBM: chain_backtrack_kernel's ChainRec) and, until
    // the squeeze, where its chain lies in the chain stage's pool (relative to the read's b_pos) and its segment of the squeeze
    uint64_t fx = 0, fy = 0, lx = 0, ly = 0;
    int64_t src = 0;
    int32_t seg = -1;
};

static inline uint64_t hash64(uint64_t key) {
    key = ~key + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}
static inline uint32_t wang32(uint32_t key) {
    key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
    key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
    return key;
}
static inline uint32_t x31_hash(const char *s) {
    uint32_t h = (uint32_t)(unsigned char)*s;
    if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)(unsigned char)*s;
    return h;
}

// mm_reg_set_coor from the hit's first and last anchor (the approximate lengths of mm_cal_fuzzy_len come with the chain
// record, or are combined from two hits' when they are joined)
static void reg_set_coor(Reg &r, int32_t qlen) {
    const int32_t q_span = (int32_t)(r.fy >> 32 & 0xff);
    r.rev = r.fx >> 63;
    r.rid = r.fx << 1 >> 33;
    r.rs = (int32_t)r.fx + 1 > q_span ? (int32_t)r.fx + 1 - q_span : 0;
    r.re = (int32_t)r.lx + 1;
    if (!r.rev) { r.qs = (int32_t)r.fy + 1 - q_span; r.qe = (int32_t)r.ly + 1; }
    else { r.qs = qlen - ((int32_t)r.ly + 1); r.qe = qlen - ((int32_t)r.fy + 1 - q_span); }
}

// Per-thread scratch of the per-read hit bookkeeping: a few hundred thousand reads per step each need a handful of small
// arrays; allocating them per read costs more than the work on them.
struct HitScratch {
    std::vector<uint64_t> cov;
    std::vector<int> w, idx;
    std::vector<std::pair<uint64_t, int>> aux;
    std::vector<int32_t> order;
    std::vector<int64_t> src;
};
static thread_local HitScratch tl_hs;

// Chain c of a read in the order of the first anchors (the order minimap2's chaining leaves them in, HostChains::chain_order):
// its (score, count) word, its record and the start of its anchors relative to the read's slice of the chain pool
struct ChainIn { uint64_t u; const ChainRec *rec; int64_t src; };

static void gen_regs(uint32_t hash, int qlen, int n_u, const ChainIn *c, std::vector<Reg> &regs) {
    struct Z { uint64_t x, y; int i; };
    static thread_local std::vector<Z> z;
    z.resize((size_t)n_u);
    int k = 0;
    for (int i = 0; i < n_u; ++i) {
        const uint32_t h = (uint32_t)hash64((hash64(c[i].rec->fx) + hash64(c[i].rec->fy)) ^ hash);
        z[i].x = c[i].u ^ h;
        z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)c[i].u;
        z[i].i = i;
        k += (int32_t)c[i].u;
    }
    std::sort(z.begin(), z.end(), [](const Z &p, const Z &q) { return p.x != q.x ? p.x < q.x : p.y < q.y; });
    regs.assign(n_u, Reg());
    for (int i = 0; i < n_u; ++i) {
        Reg &ri = regs[i];
        const Z &zi = z[n_u - 1 - i];
        const ChainRec &rc = *c[zi.i].rec;
        ri.id = i;
        ri.parent = PARENT_UNSET;
        ri.score = ri.score0 = (int32_t)(zi.x >> 32);
        ri.hash = (uint32_t)zi.x;
        ri.cnt = (int32_t)zi.y;
        ri.as = (int32_t)(zi.y >> 32);
        ri.fx = rc.fx; ri.fy = rc.fy; ri.lx = rc.lx; ri.ly = rc.ly; ri.src = c[zi.i].src;
        ri.mlen = rc.mlen; ri.blen = rc.blen;
        reg_set_coor(ri, qlen);
    }
}

static void set_parent(float mask_level, std::vector<Reg> &r, int sub_diff) {
    const int n = (int)r.size();
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) r[i].id = i;
    std::vector<uint64_t> &cov = tl_hs.cov;
    std::vector<int> &w = tl_hs.w;
    if ((int)cov.size() < n) { cov.resize((size_t)n); w.resize((size_t)n); }
    w[0] = 0; r[0].parent = 0;
    int k = 1;
    for (int i = 1; i < n; ++i) {
        Reg &ri = r[i];
        const int si = ri.qs, ei = ri.qe;
        int n_cov = 0, uncov_len = 0, j;
        for (j = 0; j < k; ++j) {
            const Reg &rp = r[w[j]];
            int sj = rp.qs, ej = rp.qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)sj << 32 | (uint32_t)ej;
        }
        if (n_cov > 0) {
            int x = si;
            std::sort(cov.begin(), cov.begin() + n_cov);
            for (j = 0; j < n_cov; ++j) {
                if ((int)(cov[j] >> 32) > x) uncov_len += (int)(cov[j] >> 32) - x;
                x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
            }
            if (ei > x) uncov_len += ei - x;
            for (j = 0; j < k; ++j) {
                Reg &rp = r[w[j]];
                const int sj = rp.qs, ej = rp.qe;
                if (ej <= si || sj >= ei) continue;
                const int min = ej - sj < ei - si ? ej - sj : ei - si, max = ej - sj > ei - si ? ej - sj : ei - si;
                const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / min - (float)uncov_len / max > mask_level) {
                    int cnt_sub = 0;
                    ri.parent = rp.parent;
                    rp.subsc = rp.subsc > ri.score ? rp.subsc : ri.score;
                    if (ri.cnt >= rp.cnt) cnt_sub = 1;
                    if (rp.has_p && ri.has_p && (rp.rid != ri.rid || rp.rs != ri.rs || rp.re != ri.re || ol != min)) {
                        rp.dp_max2 = rp.dp_max2 > ri.dp_max ? rp.dp_max2 : ri.dp_max;
                        if (rp.dp_max - ri.dp_max <= sub_diff) cnt_sub = 1;
                    }
                    if (cnt_sub) ++rp.n_sub;
                    break;
                }
            }
        } else j = k;
        if (j == k) { w[k++] = i; ri.parent = i; ri.n_sub = 0; }
    }
}

static void set_sam_pri(std::vector<Reg> &r) {
    int n_pri = 0;
    for (auto &x : r)
        if (x.id == x.parent) { ++n_pri; x.sam_pri = (n_pri == 1); }
        else x.sam_pri = 0;
}

static void sync_regs(std::vector<Reg> &regs) {
    const int n_regs = (int)regs.size();
    if (n_regs <= 0) return;
    int max_id = -1;
    for (auto &r : regs) max_id = max_id > r.id ? max_id : r.id;
    std::vector<int> tmp(max_id + 1 > 0 ? max_id + 1 : 1, -1);
    for (int i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
    for (int i = 0; i < n_regs; ++i) {
        Reg &r = regs[i];
        r.id = i;
        if (r.parent == PARENT_TMP_PRI) r.parent = i;
        else if (r.parent >= 0 && r.parent <= max_id && tmp[r.parent] >= 0) r.parent = tmp[r.parent];
        else r.parent = PARENT_UNSET;
    }
    set_sam_pri(regs);
}

static void select_sub(float pri_ratio, int min_diff, int best_n, std::vector<Reg> &r) {
    if (pri_ratio > 0.0f && !r.empty()) {
        const int n = (int)r.size();
        int k = 0, n_2nd = 0;
        for (int i = 0; i < n; ++i) {
            const int p = r[i].parent;
            if (p == i || r[i].inv) { if (k != i) r[k] = r[i]; ++k; }
            else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
                if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) {
                    if (k != i) r[k] = r[i];
                    ++k; ++n_2nd;
                }
            }
        }
        // NB: r[p] above may already have been overwritten when p > k; minimap2 has the same in-place semantics
        if (k != n) { r.resize(k); sync_regs(r); }
    }
}

static void filter_regs(const mpn_map_opt *opt, int qlen, std::vector<Reg> &regs) {
    int k = 0;
    for (int i = 0; i < (int)regs.size(); ++i) {
        Reg &r = regs[i];
        int flt = 0;
        if (!r.inv && r.cnt < opt->min_cnt) flt = 1;
        if (r.has_p) {
            if (r.mlen < opt->min_chain_score) flt = 1;
            else if (r.dp_max < opt->min_dp_max) flt = 1;
            else if (r.qs > qlen * opt->max_clip_ratio && qlen - r.qe > qlen * opt->max_clip_ratio) flt = 1;
        }
        if (!flt) { if (k < i) regs[k] = regs[i]; ++k; }
    }
    regs.resize(k);
}

// mm_squeeze_a: the hits that survived selection get consecutive places in the read's anchor list, in the order of their first
// anchors.  The anchors themselves are moved on the device (anchor_squeeze_kernel) along the segments written here: every hit
// is still ONE chain at this point.  Returns the anchors the list holds.
// (the read's segments are appended to `segs`, the staging list of the calling pool thread; Reg::seg indexes it)
static int squeeze_a(std::vector<Reg> &regs, int read, int64_t pool_base, std::vector<SqueezeSeg> &segs) {
    const int n_regs = (int)regs.size();
    std::vector<std::pair<uint64_t, int>> &aux = tl_hs.aux;
    aux.resize((size_t)n_regs);
    for (int i = 0; i < n_regs; ++i) aux[i] = {(uint64_t)regs[i].as, i};
    if (n_regs > 1) std::sort(aux.begin(), aux.end());
    int as = 0;
    for (int i = 0; i < n_regs; ++i) {
        Reg &r = regs[aux[i].second];
        r.as = as;
        r.seg = (int32_t)segs.size();
        segs.push_back(SqueezeSeg{pool_base + r.src, as, r.cnt, read, 0});
        as += r.cnt;
    }
    return as;
}

// (called on a squeezed list: squeeze_a above has run)
static void join_long(const mpn_map_opt *opt, int qlen, std::vector<Reg> &regs, std::vector<SqueezeSeg> &segs) {
    const int n_regs = (int)regs.size();
    if (n_regs < 2) return;
    std::vector<std::pair<uint64_t, int>> &aux = tl_hs.aux;
    aux.clear();
    for (int i = 0; i < n_regs; ++i)
        if (regs[i].parent == i || regs[i].parent < 0) aux.push_back({(uint64_t)regs[i].as, i});
    std::sort(aux.begin(), aux.end());
    int n_drop = 0;
    for (int i = (int)aux.size() - 1; i >= 1; --i) {
        Reg &r0 = regs[aux[i - 1].second], &r1 = regs[aux[i].second];
        if (r0.as + r0.cnt != r1.as) continue;
        if (r0.rid != r1.rid || r0.rev != r1.rev) continue;
        const u128 a0e_{r0.lx, r0.ly}, a1s_{r1.fx, r1.fy};
        const u128 *a0e = &a0e_, *a1s = &a1s_;
        if (a1s->x <= a0e->x || (int32_t)a1s->y <= (int32_t)a0e->y) continue;
        int max_gap, min_gap;
        max_gap = min_gap = (int32_t)a1s->y - (int32_t)a0e->y;
        max_gap = max_gap > (int64_t)(a1s->x - a0e->x) ? max_gap : (int)(a1s->x - a0e->x);
        min_gap = min_gap < (int64_t)(a1s->x - a0e->x) ? min_gap : (int)(a1s->x - a0e->x);
        if (max_gap > opt->max_join_long || min_gap > opt->max_join_short) continue;
        const int sc_thres = (int)((float)opt->min_join_flank_sc / opt->max_join_long * max_gap + .499);
        if (r0.score < sc_thres || r1.score < sc_thres) continue;
        const int min_flank_len = (int)(max_gap * opt->min_join_flank_ratio);
        if (r0.re - r0.rs < min_flank_len || r0.qe - r0.qs < min_flank_len) continue;
        if (r1.re - r1.rs < min_flank_len || r1.qe - r1.qs < min_flank_len) continue;
        segs[(size_t)r1.seg].flag = 1;   // (a[r1.as].y |= SEED_LONG_JOIN, applied by the squeeze kernel)
        r0.cnt += r1.cnt; r0.score += r1.score;
        {   // mm_cal_fuzzy_len over the joined anchors: both parts' sums and the step across the joint
            const int span = (int)(a1s->y >> 32 & 0xff);
            const int tl = (int32_t)a1s->x - (int32_t)a0e->x, ql = (int32_t)a1s->y - (int32_t)a0e->y;
            r0.blen += r1.blen - span + (tl > ql ? tl : ql);
            r0.mlen += r1.mlen - span + (tl > span && ql > span ? span : tl < ql ? tl : ql);
        }
        r0.lx = r1.lx; r0.ly = r1.ly;
        reg_set_coor(r0, qlen);
        r1.cnt = 0;
        r1.parent = r0.id;
        ++n_drop;
    }
    if (n_drop > 0) {
        for (auto &r : regs)
            if (r.parent >= 0 && r.id != r.parent)
                if (regs[r.parent].parent >= 0 && regs[r.parent].parent != r.parent) r.parent = regs[r.parent].parent;
        filter_regs(opt, qlen, regs);
        sync_regs(regs);
    }
}

static void hit_sort(std::vector<Reg> &r) {
    const int n = (int)r.size();
    if (n <= 1) return;
    std::vector<std::pair<uint64_t, int>> &aux = tl_hs.aux;
    aux.clear();
    for (int i = 0; i < n; ++i)
        if (r[i].inv || r[i].cnt > 0) aux.push_back({(uint64_t)(uint32_t)(r[i].has_p ? r[i].dp_max : r[i].score) << 32 | r[i].hash, i});
    std::sort(aux.begin(), aux.end());
    const int m = (int)aux.size();
    if (m == n) {
        // every hit stays: permute in place along the cycles (a read of a split, strain-rich target set brings a hundred hits
        // to the merge; a second array of them costs an allocation and a construction + destruction per hit)
        std::vector<int> &from = tl_hs.idx;   // the hit that belongs at place k
        from.resize((size_t)n);
        for (int k = 0; k < n; ++k) from[(size_t)k] = aux[(size_t)(n - 1 - k)].second;
        for (int s0 = 0; s0 < n; ++s0) {
            if (from[(size_t)s0] == s0 || from[(size_t)s0] < 0) continue;
            Reg tmp = std::move(r[(size_t)s0]);
            int k = s0;
            for (;;) {
                const int f = from[(size_t)k];
                from[(size_t)k] = -1;
                if (f == s0) { r[(size_t)k] = std::move(tmp); break; }
                r[(size_t)k] = std::move(r[(size_t)f]);
                k = f;
            }
        }
        return;
    }
    std::vector<Reg> t(aux.size());
    for (int i = m - 1; i >= 0; --i) t[(size_t)(m - 1 - i)] = std::move(r[(size_t)aux[i].second]);
    r.swap(t);
}

static void set_mapq(std::vector<Reg> &regs, int min_chain_sc, int match_sc, int rep_len) {
    static const float q_coef = 40.0f;
    if (regs.empty()) return;
    int64_t sum_sc = 0;
    for (auto &r : regs) if (r.parent == r.id) sum_sc += r.score;
    const float uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
    for (auto &r : regs) {
        if (r.inv) r.mapq = 0;
        else if (r.parent == r.id) {
            int mapq;
            float pen_s1 = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio;
            float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
            pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
            const int subsc = r.subsc > min_chain_sc ? r.subsc : min_chain_sc;
            if (r.has_p && r.dp_max2 > 0 && r.dp_max > 0) {
                const float identity = (float)r.mlen / r.blen;
                const float x = (float)r.dp_max2 * subsc / r.dp_max / r.score0;
                mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r.dp_max / match_sc));
                const int mapq_alt = (int)(6.02f * identity * identity * (r.dp_max - r.dp_max2) / match_sc + .499f);
                mapq = mapq < mapq_alt ? mapq : mapq_alt;
            } else {
                const float x = (float)subsc / r.score0;
                if (r.has_p) {
                    const float identity = (float)r.mlen / r.blen;
                    mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r.dp_max / match_sc));
                } else mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r.score));
            }
            mapq -= (int)(4.343f * logf(r.n_sub + 1) + .499f);
            mapq = mapq > 0 ? mapq : 0;
            r.mapq = mapq < 60 ? mapq : 60;
            if (r.has_p && r.dp_max > r.dp_max2 && r.mapq == 0) r.mapq = 1;
        } else r.mapq = 0;
    }
}

static void split_reg(Reg &r, Reg &r2, int n, int qlen, const SplitRec *sp) {
    if (n <= 0 || n >= r.cnt || !sp) return;
    r2 = r;
    r2.id = -1;
    r2.sam_pri = 0;
    r2.has_p = 0; r2.cigar.clear(); r2.dp_score = r2.dp_max = r2.dp_max2 = r2.n_ambi = 0;
    r2.split_inv = 0;
    r2.aligned = 0;
    r2.cnt = r.cnt - n;
    r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
    r2.as = r.as + n;
    if (r.parent == r.id) r2.parent = PARENT_TMP_PRI;
    r2.fx = sp->fx; r2.fy = sp->fy;   // (its last anchor is the hit's)
    r2.mlen = sp->mlen_r; r2.blen = sp->blen_r;
    reg_set_coor(r2, qlen);
    r.cnt -= r2.cnt;
    r.score -= r2.score;
    r.lx = sp->lx_left; r.ly = sp->ly_left;
    r.mlen = sp->mlen_l; r.blen = sp->blen_l;
    reg_set_coor(r, qlen);
    r.split |= 1; r2.split |= 2;
}

// MPN_DEBUG_CPU: thread CPU time of the sections of a host phase (g_cpu_ns[16 + k])
struct CpuSect {
    timespec t;
    bool on;
    explicit CpuSect(bool o) : on(o) { if (on) clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); }
    void lap(int k);
};

// What the stitching kernel found for one hit (stitch_kernels.h) applied to the hit: coordinates, DP score, and -- when a
// gap fill z-dropped -- the split of the hit at the last anchor before the drop (mm_align1's `dropped` branch; the kernel
// has located the anchor).  returns true if a split remainder was produced in r2
static bool apply_stitch(int qlen, Reg &r, Reg &r2, const SplitRec *splits, const StitchOut &so, int fin_idx) {
    bool has_r2 = false;
    r2.cnt = 0;
    if (so.has_p) r.has_p = 1;
    r.dp_score += so.dp_score;
    if (so.split_n > 0) {
        const int old_cnt = r.cnt;
        split_reg(r, r2, so.split_n, qlen, so.split_rec >= 0 ? splits + so.split_rec : nullptr);
        has_r2 = r2.cnt > 0 && r.cnt != old_cnt;
        if (so.split_inv) r2.split_inv = 1;
    }
    r.rs = so.rs1; r.re = so.re1;
    if (r.rev) { r.qs = qlen - so.qe1; r.qe = qlen - so.qs1; }
    else { r.qs = so.qs1; r.qe = so.qe1; }
    // the CIGAR fix-up and the alignment statistics (mm_update_extra) are done for the whole round by aln_finish_wave_kernel
    r.fin_idx = r.has_p ? fin_idx : -1;
    r.aligned = 1;
    return has_r2;
}

// ------------------------------------------------------------------------------------------------------------
// Host thread pool shared by the pipeline workers.  A worker's host phases (hits from chains, DP-window planning, CIGAR
// stitching, MAPQ/text) are short bursts between GPU waits; with a private share of the cores a worker would crawl
// through its burst while the threads of the workers that wait on the GPU sleep.  Every burst is a job in one queue and
// every idle pool thread helps the oldest open job; the posting thread works on its own job too, so a job always
// advances.  fn(i, slot): slot < max_par identifies the helping thread (per-thread accumulators of the caller).
void CpuSect::lap(int k) {
    if (!on) return;
    timespec b;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &b);
    g_cpu_ns[16 + k] += (b.tv_sec - t.tv_sec) * 1000000000LL + (b.tv_nsec - t.tv_nsec);
    t = b;
}
class HostPool {
    struct Job {
        const std::function<void(int, int)> *fn;
        int n, max_par, tag = 0, chunk = 1;   // items are handed out `chunk` at a time: one atomic per item is a hot cache line
        std::atomic<int> next{0};
        int slots = 1, active = 0;  // guarded by mu (slot 0 is the posting thread)
    };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job *> jobs;
    std::vector<std::thread> threads;
    bool stop = false;

    static void drain(Job &j, int slot) {
        for (;;) {
            const int i0 = j.next.fetch_add(j.chunk);
            if (i0 >= j.n) break;
            const int i1 = std::min(j.n, i0 + j.chunk);
            for (int i = i0; i < i1; ++i) (*j.fn)(i, slot);
        }
    }
    static void run(Job &j, int slot) {
        if (!g_cpu_on) { drain(j, slot); return; }
        timespec a, b;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &a);
        drain(j, slot);
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &b);
        g_cpu_ns[j.tag & 31] += (b.tv_sec - a.tv_sec) * 1000000000LL + (b.tv_nsec - a.tv_nsec);
    }
    Job *pick(int *slot) {  // mu held
        for (Job *j : jobs)
            if (j->slots < j->max_par && j->next.load(std::memory_order_relaxed) < j->n) { *slot = j->slots++; ++j->active; return j; }
        return nullptr;
    }
    void worker() {
        pthread_setname_np(pthread_self(), "mpn-pool");
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            int slot = 0;
            Job *j = nullptr;
            cv_work.wait(lk, [&]() { return stop || (j = pick(&slot)) != nullptr; });
            if (stop) return;
            lk.unlock();
            run(*j, slot);
            lk.lock();
            if (--j->active == 0) cv_done.notify_all();
        }
    }

public:
    void ensure(int n_threads) {
        std::lock_guard<std::mutex> g(mu);
        while ((int)threads.size() < n_threads - 1) threads.emplace_back([this]() { worker(); });
    }
    // grain: items that are worth one helping thread.  The bursts of a sub-batch are a few thousand light items: waking all
    // pool threads for each of them (a futex round trip and a fight for the queue lock per thread and burst) cost more CPU
    // than the items themselves.
    void parallel_for(int n, int max_par, const std::function<void(int, int)> &fn, int tag = 0, int grain = 512) {
        max_par = std::min(max_par, 1 + n / std::max(1, grain));
        if (max_par <= 1 || n < 2) { for (int i = 0; i < n; ++i) fn(i, 0); return; }
        Job j;
        j.fn = &fn; j.n = n; j.max_par = max_par; j.tag = tag;
        j.chunk = std::max(1, std::min(32, n / (max_par * 8)));
        { std::lock_guard<std::mutex> g(mu); jobs.push_back(&j); }
        for (int k = 1; k < max_par; ++k) cv_work.notify_one();
        run(j, 0);
        std::unique_lock<std::mutex> lk(mu);
        for (auto it = jobs.begin(); it != jobs.end(); ++it) if (*it == &j) { jobs.erase(it); break; }
        cv_done.wait(lk, [&]() { return j.active == 0; });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
};
static HostPool g_pool;

static void parallel_for(int n, int n_threads, const std::function<void(int, int)> &fn, int tag = 0, int grain = 512) { g_pool.parallel_for(n, n_threads, fn, tag, grain); }

struct ReadState {
    std::vector<Reg> regs;
    int n_a = 0;           // anchors of the read's squeezed list (which lives in HBM)
    std::vector<int> pending; // reg indices aligned this round
};

static double event_identity(const Reg &r) {
    int32_t n_gapo = 0, n_gap = 0;
    for (uint32_t c : r.cigar) { const int32_t op = c & 0xf, len = c >> 4; if (op == 1 || op == 2) { ++n_gapo; n_gap += len; } }
    return (double)r.mlen / (r.blen + r.n_ambi - n_gap + n_gapo);
}

// names and lengths of the targets the hits refer to: one index, or all parts of a split index in order
struct Targets {
    const std::vector<std::string> *names;
    const std::vector<int32_t> *lens;
};

static void write_paf(const Targets mi_, const mpn_map_opt *o, const char *name, int32_t qlen, const std::vector<Reg> &regs,
                      int32_t rep_len, std::string &out) {
    const Targets *mi = &mi_;
    char buf[1024];
    for (const Reg &r : regs) {
        const int type = r.id == r.parent ? (r.inv ? 'I' : 'P') : (r.inv ? 'i' : 'S');
        out += name;
        snprintf(buf, sizeof(buf), "\t%d\t%d\t%d\t%c\t", qlen, r.qs, r.qe, "+-"[r.rev]);
        out += buf;
        out += (*mi->names)[r.rid];
        snprintf(buf, sizeof(buf), "\t%d\t%d\t%d\t%d\t%d\t%d", (*mi->lens)[r.rid], r.rs, r.re, r.mlen, r.blen, r.mapq);
        out += buf;
        if (r.has_p) {
            snprintf(buf, sizeof(buf), "\tNM:i:%d\tms:i:%d\tAS:i:%d\tnn:i:%d", r.blen - r.mlen + r.n_ambi, r.dp_max, r.dp_score, r.n_ambi);
            out += buf;
        }
        snprintf(buf, sizeof(buf), "\ttp:A:%c\tcm:i:%d\ts1:i:%d", type, r.cnt, r.score);
        out += buf;
        if (r.parent == r.id) { snprintf(buf, sizeof(buf), "\ts2:i:%d", r.subsc); out += buf; }
        if (r.has_p) {
            const double div = 1.0 - event_identity(r);
            if (div == 0.0) out += "\tde:f:0";
            else { snprintf(buf, sizeof(buf), "\tde:f:%.4f", div); out += buf; }
        }
        if (r.split) { snprintf(buf, sizeof(buf), "\tzd:i:%d", r.split); out += buf; }
        snprintf(buf, sizeof(buf), "\trl:i:%d", rep_len);
        out += buf;
        if (r.has_p && o->with_cigar) {
            out += "\tcg:Z:";
            for (uint32_t c : r.cigar) { snprintf(buf, sizeof(buf), "%d%c", c >> 4, "MIDNSH"[c & 0xf]); out += buf; }
        }
        out += '\n';
    }
}

// SAM records of one read (minimap2 2.17 -a: mm_write_sam3 / write_sam_cigar / sam_write_sq for single-segment reads
// without read group; qualities are not carried).  A read without hits gets a flag-4 record.
static char sam_comp(char c) {
    static const char from[] = "ACGTUNRYKMSWBDHVacgtunrykmswbdhv", to[] = "TGCAANYRMKSWVHDBtgcaanyrmkswvhdb";
    const char *p = c ? strchr(from, c) : nullptr;
    return p ? to[p - from] : c;
}

static void sam_seq(std::string &out, const char *seq, int st, int en, bool rev) {
    if (!rev) out.append(seq + st, (size_t)(en - st));
    else for (int i = en - 1; i >= st; --i) out += sam_comp(seq[i]);
}

// QUAL column of a record whose SEQ is seq[st, en) on strand rev: the qualities in the same orientation, or '*' without them
static void sam_qual(std::string &out, const char *qual, int st, int en, bool rev) {
    out += '\t';
    if (!qual) { out += '*'; return; }
    if (!rev) out.append(qual + st, (size_t)(en - st));
    else for (int i = en - 1; i >= st; --i) out += qual[i];
}

static void write_sam(const Targets mi_, const char *name, int32_t qlen, const char *seq, const char *qual, const std::vector<Reg> &regs,
                      int32_t rep_len, std::string &out) {
    const Targets *mi = &mi_;
    char buf[1024];
    if (regs.empty()) {
        out += name;
        out += "\t4\t*\t0\t0\t*\t*\t0\t0\t";
        sam_seq(out, seq, 0, qlen, false);
        sam_qual(out, qual, 0, qlen, false);
        snprintf(buf, sizeof(buf), "\trl:i:%d\n", rep_len);
        out += buf;
        return;
    }
    for (size_t i = 0; i < regs.size(); ++i) {
        const Reg &r = regs[i];
        const int type = r.id == r.parent ? (r.inv ? 'I' : 'P') : (r.inv ? 'i' : 'S');
        int flag = r.rev ? 0x10 : 0;
        if (r.parent != r.id) flag |= 0x100;
        else if (!r.sam_pri) flag |= 0x800;
        out += name;
        snprintf(buf, sizeof(buf), "\t%d\t", flag);
        out += buf;
        out += (*mi->names)[r.rid];
        snprintf(buf, sizeof(buf), "\t%d\t%d\t", r.rs + 1, r.mapq);
        out += buf;
        if (!r.has_p) out += '*';
        else {
            const int clip0 = r.rev ? qlen - r.qe : r.qs, clip1 = r.rev ? r.qs : qlen - r.qe;
            const char clip_char = (flag & 0x800) ? 'H' : 'S';
            if (clip0) { snprintf(buf, sizeof(buf), "%d%c", clip0, clip_char); out += buf; }
            for (uint32_t c : r.cigar) { snprintf(buf, sizeof(buf), "%d%c", c >> 4, "MIDNSH"[c & 0xf]); out += buf; }
            if (clip1) { snprintf(buf, sizeof(buf), "%d%c", clip1, clip_char); out += buf; }
        }
        out += "\t*\t0\t0\t";
        if ((flag & 0x900) == 0) { sam_seq(out, seq, 0, qlen, r.rev); sam_qual(out, qual, 0, qlen, r.rev); }
        else if (flag & 0x100) out += "*\t*";
        else { sam_seq(out, seq, r.qs, r.qe, r.rev); sam_qual(out, qual, r.qs, r.qe, r.rev); }
        if (r.has_p) {
            snprintf(buf, sizeof(buf), "\tNM:i:%d\tms:i:%d\tAS:i:%d\tnn:i:%d", r.blen - r.mlen + r.n_ambi, r.dp_max, r.dp_score, r.n_ambi);
            out += buf;
        }
        snprintf(buf, sizeof(buf), "\ttp:A:%c\tcm:i:%d\ts1:i:%d", type, r.cnt, r.score);
        out += buf;
        if (r.parent == r.id) { snprintf(buf, sizeof(buf), "\ts2:i:%d", r.subsc); out += buf; }
        if (r.has_p) {
            const double div = 1.0 - event_identity(r);
            if (div == 0.0) out += "\tde:f:0";
            else { snprintf(buf, sizeof(buf), "\tde:f:%.4f", div); out += buf; }
        }
        if (r.split) { snprintf(buf, sizeof(buf), "\tzd:i:%d", r.split); out += buf; }
        if (r.parent == r.id && r.has_p && regs.size() > 1) {  // SA: the other non-secondary hits that have a CIGAR
            bool any = false;
            for (size_t j = 0; j < regs.size(); ++j) {
                const Reg &q = regs[j];
                if (j == i || q.parent != q.id || !q.has_p) continue;
                if (!any) { out += "\tSA:Z:"; any = true; }
                int l_M, l_I = 0, l_D = 0;
                if (q.qe - q.qs < q.re - q.rs) { l_M = q.qe - q.qs; l_D = (q.re - q.rs) - l_M; }
                else { l_M = q.re - q.rs; l_I = (q.qe - q.qs) - l_M; }
                const int c5 = q.rev ? qlen - q.qe : q.qs, c3 = q.rev ? q.qs : qlen - q.qe;
                out += (*mi->names)[q.rid];
                snprintf(buf, sizeof(buf), ",%d,%c,", q.rs + 1, "+-"[q.rev]);
                out += buf;
                if (c5) { snprintf(buf, sizeof(buf), "%dS", c5); out += buf; }
                if (l_M) { snprintf(buf, sizeof(buf), "%dM", l_M); out += buf; }
                if (l_I) { snprintf(buf, sizeof(buf), "%dI", l_I); out += buf; }
                if (l_D) { snprintf(buf, sizeof(buf), "%dD", l_D); out += buf; }
                if (c3) { snprintf(buf, sizeof(buf), "%dS", c3); out += buf; }
                snprintf(buf, sizeof(buf), ",%d,%d;", q.mapq, q.blen - q.mlen + q.n_ambi);
                out += buf;
            }
        }
        snprintf(buf, sizeof(buf), "\trl:i:%d\n", rep_len);
        out += buf;
    }
}

// per-worker resources: a stream, a device arena, grow-only scratch pools and pinned staging buffers
struct Slot {
    hipStream_t st = nullptr;
    hipStream_t st2 = nullptr;       // side stream: the few long windows run beside the many short ones
    hipStream_t st3 = nullptr;       // third stream: the tiled strips (one wave per long gap fill) beside the band kernels
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr;
    Arena arena;
    PoolBuf pool_jobs, pool_P, pool_P2, pool_OFF, pool_order, pool_state, pool_CIG, pool_res, pool_redo, pool_compact, pool_used;
    PoolBuf pool_redo_ids, pool_sregs, pool_souts, pool_fin_jobs, pool_fin_out, pool_fin_cig;
    PoolBuf pool_probes, pool_sizes, pool_buckets, pool_pregs, pool_psum, pool_job_anchor, pool_splits;
    PoolBuf pin_segs{nullptr, 0, true}, pin_pregs{nullptr, 0, true}, pin_hits{nullptr, 0, true};
    PoolBuf pin_order{nullptr, 0, true}, pin_res{nullptr, 0, true};
    PoolBuf pin_chain_u{nullptr, 0, true}, pin_chain_b{nullptr, 0, true};
    PoolBuf pin_fin_cig{nullptr, 0, true}, pin_fin_out{nullptr, 0, true};
    std::vector<std::vector<SqueezeSeg>> seg_stage;   // per pool thread: the squeeze segments of the reads it handled
    size_t device_bytes() const {
        size_t h = 0;
        for (const PoolBuf *pb : {&pool_jobs, &pool_P, &pool_P2, &pool_OFF, &pool_order, &pool_state, &pool_CIG, &pool_res, &pool_redo, &pool_compact, &pool_used,
                                  &pool_redo_ids, &pool_sregs, &pool_souts, &pool_fin_jobs, &pool_fin_out, &pool_fin_cig, &pool_probes, &pool_sizes, &pool_buckets,
                                  &pool_pregs, &pool_psum, &pool_job_anchor, &pool_splits})
            h += pb->cap;
        for (const auto &c : arena.chunks) h += c.cap;
        return h;
    }
    // the slot's device memory goes back to the device (a worker that sheds, or a slot that idles in this call)
    void release_device() {
        arena.release_all();
        for (PoolBuf *pb : {&pool_jobs, &pool_P, &pool_P2, &pool_OFF, &pool_order, &pool_state, &pool_CIG, &pool_res, &pool_redo, &pool_compact, &pool_used,
                            &pool_redo_ids, &pool_sregs, &pool_souts, &pool_fin_jobs, &pool_fin_out, &pool_fin_cig, &pool_probes, &pool_sizes, &pool_buckets,
                            &pool_pregs, &pool_psum, &pool_job_anchor, &pool_splits})
            pb->release();
    }
};
static Slot g_slots[16];
static thread_local Slot *tl_slot = &g_slots[0];

static int g_force_kernel = 0;  // test hook: 0 auto, 1 single-wave LDS kernel, 3 workgroup kernel, 4 strip (else band), 5 band

static void parallel_chunks(int64_t n, int n_threads, const std::function<void(int64_t, int64_t, int)> &fn, int tag = 0) {
    if (n_threads <= 1 || n < 8192) { fn(0, n, 0); return; }
    // chunk t of n_threads equal ranges; the chunk index is what the callers use for their per-thread accumulators
    g_pool.parallel_for(n_threads, n_threads, [&](int t, int) { fn(n * t / n_threads, n * (t + 1) / n_threads, t); }, tag, 1);
}

// the second pass of a gap fill whose CIGAR failed the z-drop test: exact maximum, band (or anti-diagonal) layout
__global__ void ext_redo_patch_kernel(ExtJob *jobs, const int32_t *ids, const int32_t *layout, const int32_t *qstride,
                                      const int64_t *p_off, const int32_t *inv, int zdrop_inv, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    ExtJob &jb = jobs[ids[k]];
    jb.flag &= ~EZ_APPROX_MAX;
    if (inv[k]) { jb.flag |= EZ_INV; jb.zdrop = zdrop_inv; }   // (mm_align1: second pass with zdrop_inv when mm_test_zdrop returned 2)
    jb.layout = layout[k];
    jb.qstride = qstride[k];
    jb.p_off = p_off[k];
}

// 0..4 codes of a target interval / of a read interval on a strand, for the rare host-side steps (inversion probes)
__global__ void ref_codes_kernel(RefView rv, int64_t g0, int n, int8_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int8_t)ref_code(rv, g0 + i);
}
static int fetch_ref_codes(const RefView &rv, int64_t seq_off_rid, int start, int n, int8_t *out, hipStream_t st) {
    if (n <= 0) return 0;
    DevBuf<int8_t> d;
    if (d.alloc((size_t)n)) return -1;
    hipLaunchKernelGGL(ref_codes_kernel, dim3((n + 255) / 256), dim3(256), 0, st, rv, seq_off_rid + start, n, d.p);
    MPN_HIP_CHECK(hipGetLastError());
    MPN_HIP_CHECK(hipMemcpyAsync(out, d.p, (size_t)n, hipMemcpyDeviceToHost, st));
    MPN_HIP_CHECK(stream_sync(st));
    return 0;
}
static inline int8_t host_code(char c) { c |= 0x20; return c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : (c == 't' || c == 'u') ? 3 : 4; }
// base x of read `seq` (length rlen) on strand rev, as a 0..4 code
static inline int8_t host_qbase(const char *seq, int rlen, int rev, int x) {
    if (!rev) return host_code(seq[x]);
    const int8_t c = host_code(seq[rlen - 1 - x]);
    return c < 4 ? (int8_t)(3 - c) : (int8_t)4;
}
// local alignment scores (ksw_ll_i16 of minimap2's inversion code) of a few pairs on the GPU: the SSW kernels of this
// library compute the same recurrences with the same tie rules for the end (first reference column with a strictly larger
// column maximum, smallest read index in it); a gap of length L costs q + L * e = SSW's (q + e) + (L - 1) * e
struct LocalHit { int score, qe, te; };
static int local_scores(const mpn_map_opt *opt, const std::vector<std::vector<int8_t>> &qs, const std::vector<std::vector<int8_t>> &ts, std::vector<LocalHit> &out) {
    const int n = (int)qs.size();
    out.assign((size_t)n, LocalHit{0, -1, -1});
    if (n == 0) return 0;
    std::vector<int8_t> qb, tb;
    std::vector<int64_t> qo(n), to(n), coff(n);
    std::vector<int32_t> ql(n), tl(n), mask(n, 15), rb(n), re(n), qb1(n), qe1(n), re2(n), clen(n), status(n);
    std::vector<uint16_t> s1(n), s2(n);
    for (int i = 0; i < n; ++i) {
        qo[i] = (int64_t)qb.size(); ql[i] = (int32_t)qs[i].size(); qb.insert(qb.end(), qs[i].begin(), qs[i].end());
        to[i] = (int64_t)tb.size(); tl[i] = (int32_t)ts[i].size(); tb.insert(tb.end(), ts[i].begin(), ts[i].end());
    }
    int8_t mat[25];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[i * 5 + j] = (int8_t)(i == j ? opt->a : -opt->b); mat[i * 5 + 4] = (int8_t)-opt->sc_ambi; }
    for (int i = 0; i < 5; ++i) mat[20 + i] = (int8_t)-opt->sc_ambi;
    uint32_t dummy_cig[4];
    const int rc = mpn_ssw_align_batch(n, qb.data(), qo.data(), ql.data(), tb.data(), to.data(), tl.data(), mat, 5, 2, (uint8_t)(opt->q + opt->e), (uint8_t)opt->e,
                                       0, 0, 0, mask.data(), s1.data(), s2.data(), rb.data(), re.data(), qb1.data(), qe1.data(), re2.data(), dummy_cig, 4,
                                       coff.data(), clen.data(), status.data());
    if (rc) return -1;
    for (int i = 0; i < n; ++i) if (status[i] == 0) out[(size_t)i] = LocalHit{(int)s1[i], qe1[i], re[i]};
    return 0;
}

// Device-resident state of a round's DP windows: the job records, one result record per window, and the pool of compacted
// CIGAR operations; `used` = [operations in the pool, windows listed for the exact second pass, stitched operations].  They stay
// in HBM for the stitching kernel (valid until the worker's next round); nothing per window travels to the host.
struct DevRound {
    ExtJob *jobs = nullptr;
    ExtRes *res = nullptr;
    uint32_t *compact = nullptr;
    unsigned long long *used = nullptr;
    int64_t cig_cap = 0;   // CIGAR operations the round's windows can produce at most
    int n_jobs = 0;        // windows of the round (read back with the layout counters)
};

// Run one group of DP windows whose raw job records are on the device (dv.jobs[0..nj), as plan_kernel or the stage test wrote
// them): kernel choice, direction-matrix layout and launch lists are made on the device (plan_kernels.h) and a block of counters
// comes back; then the DP kernels, the traceback, the z-drop test and the rare exact second pass.  budget > 0: returns 1
// without launching any DP if the direction matrices need more than that (the caller then cuts the range).
// host views the rare host-side steps need (inversion probes): the sub-batch's reads and the targets' offsets; null for the
// stage test, which then skips the probes
struct HostSeqs { const char *seqs; const int64_t *seq_off; const int32_t *seq_len; const int64_t *tseq_off; };

static int run_job_group(const RefView &rv, const mpn_map_opt *opt, int nj_cap, const unsigned long long *d_nj, DevRound &dv, const uint8_t *d_reads,
                         const int64_t *d_read_off, const int32_t *d_read_len, int64_t budget, const HostSeqs *hs, hipStream_t st) {
    // nj_cap: an upper bound of the number of windows (the count itself is on the device, *d_nj: the planning kernel wrote it)
    if (nj_cap == 0) return 0;
    int nj = nj_cap;
    WallTimer wt;
    Slot &SL = *tl_slot;
    const int strip_scores = ext_strip_scores_ok(opt->a, -opt->b, -opt->sc_ambi, opt->q + opt->e, opt->q2 + opt->e2) ? 1 : 0;
    ExtParams prm;
    prm.sc_mch = (int8_t)opt->a; prm.sc_mis = (int8_t)-opt->b; prm.sc_n = (int8_t)-opt->sc_ambi;
    prm.q = (int8_t)opt->q; prm.e = (int8_t)opt->e; prm.q2 = (int8_t)opt->q2; prm.e2 = (int8_t)opt->e2; prm.zdrop_thres = opt->zdrop;
    prm.zdrop_inv = opt->zdrop_inv; prm.max_gap = opt->max_gap;
    const size_t order_cap = (size_t)nj + (size_t)N_STRIP * 8 + 16;   // (a strip list is padded to whole waves: at most 7 entries)
    if (SL.pool_sizes.ensure((size_t)nj * sizeof(JobSizes) + 16) || SL.pool_buckets.ensure((size_t)2 * N_BUCKETS * 4 + sizeof(LayoutTotals) + 16) ||
        SL.pool_order.ensure(order_cap * 4) || SL.pin_res.ensure(sizeof(LayoutTotals) + 64) ||
        SL.pool_redo_ids.ensure((size_t)nj * 4 + 16) || SL.pool_probes.ensure((size_t)nj * sizeof(InvProbe) + 16))
        return -1;
    struct { ExtJob *p; } d_jobs{dv.jobs};
    JobSizes *d_sizes = SL.pool_sizes.as<JobSizes>();
    int32_t *d_bcnt = SL.pool_buckets.as<int32_t>(), *d_bcur = d_bcnt + N_BUCKETS;
    LayoutTotals *d_tot = reinterpret_cast<LayoutTotals *>(d_bcnt + 2 * N_BUCKETS);
    struct { int32_t *p; } d_order{SL.pool_order.as<int32_t>()};
    // (bucket counters and the totals block are neighbours in one pool: one fill clears both; the padding of the strip lists
    // is written by the scan kernel)
    {   // one launch zeroes the bucket counters + totals and this group's two list counters (the other counters of the round's block
        // were zeroed with it at the start of the round)
        ZeroList z{};
        zero_list_push(z, d_bcnt, (size_t)2 * N_BUCKETS * 4 + sizeof(LayoutTotals));
        zero_list_push(z, dv.used + 1, 8);
        zero_list_push(z, dv.used + 5, 8);
        MPN_HIP_CHECK(zero_regions(z, st));
    }
    const int lay_grid = std::max(1, std::min((nj + 255) / 256, 256));   // (a block per CU: every block flushes its counters once)
    EvTimer evl(st);
    static const bool tiled_on = []() { const char *e = getenv("MPN_TILED"); return !e || atoi(e) != 0; }();
    hipLaunchKernelGGL(job_classify_kernel, dim3(lay_grid), dim3(256), 0, st, d_jobs.p, d_nj, strip_scores, prm, g_force_kernel ? g_force_kernel : (tiled_on ? 0 : 7), d_sizes,
                       d_bcnt, d_tot);
    hipLaunchKernelGGL(job_scan_kernel, dim3(1), dim3(1024), 0, st, d_sizes, d_nj, (const int32_t *)d_bcnt, d_bcur, d_tot, d_order.p);
    hipLaunchKernelGGL(job_layout_kernel, dim3(lay_grid), dim3(256), 0, st, d_jobs.p, d_nj, (const JobSizes *)d_sizes, d_bcur, d_order.p);
    MPN_HIP_CHECK(hipGetLastError());
    LayoutTotals *h_tot = SL.pin_res.as<LayoutTotals>();
    evl.mark(56);
    MPN_HIP_CHECK(hipMemcpyAsync(h_tot, d_tot, sizeof(LayoutTotals), hipMemcpyDeviceToHost, st));
    MPN_HIP_CHECK(stream_sync(st));
    evl.resolve();
    const LayoutTotals T = *h_tot;
    nj = T.n_jobs;
    dv.n_jobs = nj;
    if (nj == 0) return 0;
    if (T.too_large) { set_error("DP window too large for LDS staging (%d x %d)", T.tl_q, T.tl_t); return -4; }
    if (!dv.compact) {
        // (the compact pool holds every window's operations, and once more those of the windows that take the second pass)
        if (SL.pool_compact.ensure((size_t)T.cig_tot * 4 * 2 + 16)) return -1;
        dv.compact = SL.pool_compact.as<uint32_t>();
        dv.cig_cap = T.cig_tot;
    }
    if (budget > 0 && T.p_tot > budget && nj > 1) return 1;
    const int *cnt = T.cnt, *base = T.base;
    g_stats[4] += nj; g_stats[5] += T.cells; g_stats[31] += T.strip_cells[0] + T.strip_cells[1] + T.strip_cells[2];
    for (int c = 0; c < 3; ++c) g_stats[41 + c] += T.strip_cells[c];
    g_stats[58] += T.xstrip_cells;
    if (getenv("MPN_DEBUG_JOBS")) {
        static const char *const fam[] = {"lds", "wg", "strip", "band"};
        for (int l = 0; l < N_LISTS; ++l)
            if (cnt[l]) fprintf(stderr, "[jobs] %s list %-2d n=%d\n", fam[l < L_WG ? 0 : l < L_STRIP ? 1 : l < L_BAND ? 2 : 3], l, cnt[l]);
    }
    if (const char *dump = getenv("MPN_DUMP_STRIPS")) {   // debug: the geometry of the strip launch lists, in launch order
        const int n_ord = base[L_BAND] - base[L_STRIP];
        std::vector<int32_t> ord(n_ord);
        std::vector<ExtJob> hj(nj);
        MPN_HIP_CHECK(hipMemcpy(ord.data(), d_order.p + base[L_STRIP], (size_t)n_ord * 4, hipMemcpyDeviceToHost));
        MPN_HIP_CHECK(hipMemcpy(hj.data(), d_jobs.p, (size_t)nj * sizeof(ExtJob), hipMemcpyDeviceToHost));
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        if (const char *all = getenv("MPN_DUMP_JOBS"))   // ... and every window of the group: list, sizes, band, flags
            if (FILE *f = fopen(all, "ab")) {
                for (int j = 0; j < nj; ++j) {
                    const int32_t rec[6] = {hj[j].cls, hj[j].qlen, hj[j].tlen, hj[j].w, hj[j].flag, hj[j].n_col};
                    fwrite(rec, 4, 6, f);
                }
                fclose(f);
            }
        if (FILE *f = fopen(dump, "ab")) {
            for (int l = L_STRIP; l < L_BAND; ++l)
                for (int k = base[l]; k < base[l + 1]; ++k) {
                    const int jid = ord[k - base[L_STRIP]];
                    const int32_t rec[4] = {strip_glc_of_list(l), jid >= 0 ? hj[jid].qlen : 0, jid >= 0 ? hj[jid].tlen : 0, jid >= 0 ? hj[jid].strip_s : 0};
                    fwrite(rec, 4, 4, f);
                }
            fclose(f);
        }
    }
    if (SL.pool_P.ensure((size_t)T.p_tot + 16) || SL.pool_OFF.ensure((size_t)T.row_tot * 2 * 4 + 16) ||
        SL.pool_state.ensure((size_t)T.state_tot + 16) || SL.pool_CIG.ensure((size_t)T.cig_tot * 4 + 16))
        return -1;
    wt.stop_into(g_stats[27]);
    struct { uint8_t *p; } P{SL.pool_P.as<uint8_t>()};
    struct { int32_t *p; } OFF{SL.pool_OFF.as<int32_t>()};
    struct { int8_t *p; } gstate{SL.pool_state.as<int8_t>()};
    struct { uint32_t *p; } CIG{SL.pool_CIG.as<uint32_t>()};
    struct { ExtRes *p; } d_res{dv.res};
    uint32_t *d_compact = dv.compact;
    unsigned long long *d_used = dv.used;   // [0] operations in the compact pool (the whole round), [1] windows listed for the second pass (this group)
    int32_t *d_redo_ids = SL.pool_redo_ids.as<int32_t>();
    InvProbe *d_probes = SL.pool_probes.as<InvProbe>();
    // one launch of launch list `l` over ord[0..n)
    auto launch_list = [&](int l, const int32_t *ord, int n, hipStream_t s) -> int {
        if (n == 0) return 0;
        if (l < L_WG) {
            const size_t lds = std::max<size_t>((size_t)T.lds_need[l - L_LDS], 64);
            if (lds > 64 * 1024) MPN_HIP_CHECK(hipFuncSetAttribute((const void *)ext_dp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(ext_dp_kernel, dim3(n), dim3(64), lds, s, d_jobs.p, ord, n, prm, d_reads, d_read_off, d_read_len, rv, P.p, OFF.p,
                               gstate.p, d_res.p);
        } else if (l < L_STRIP) {
            const int ntc = (l - L_WG) / 5;
            const size_t lds = std::max<size_t>((size_t)T.lds_need[(l - L_WG) % 5], 64);
#define MPN_WG_LAUNCH(NT)                                                                                                             \
            do {                                                                                                                      \
                if (lds > 64 * 1024) MPN_HIP_CHECK(hipFuncSetAttribute((const void *)ext_dp_wg_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL(ext_dp_wg_kernel<NT>, dim3(n), dim3(NT), lds, s, d_jobs.p, ord, n, prm, d_reads, d_read_off, d_read_len, \
                                   rv, P.p, OFF.p, gstate.p, d_res.p);                                                                \
            } while (0)
            if (ntc == 0) MPN_WG_LAUNCH(256); else if (ntc == 1) MPN_WG_LAUNCH(512); else MPN_WG_LAUNCH(1024);
#undef MPN_WG_LAUNCH
        } else if (l < L_BAND) {  // called once per variant family (l = its first list): all its lane-group classes in ONE launch
            const int fam = (l - L_STRIP) / 48;   // 0: gap fills (approximate maximum); 1: exact (left- and right-aligned gaps)
            StripSegs segs;
            segs.n = 0;
            int blocks = 0;
            size_t lds = 0;
            const int c_lo = fam == 0 ? 0 : 3, c_hi = fam == 0 ? 3 : N_STRIP_CLASS;
            const int pair = fam == 0 ? 1 : 0;   // gap fills: two windows per lane group (ext_strip_pair)
            for (int glc = 2; glc >= 0; --glc)   // wide lane groups (the long windows) first
                for (int sclass = c_lo + glc; sclass < c_hi; sclass += 3) {
                    const int l0 = L_STRIP + 16 * sclass, nl = base[l0 + 16] - base[l0], per = (pair ? 8 : 4) >> glc;
                    if (nl == 0) continue;
                    const int stride = (std::max(T.strip_lds[sclass], 16) + 3) & ~3;
                    // exact variants: a slot per anti-diagonal and the E4 table per window (+ 16: the rows that pad the last strip)
                    const int nr_stride = fam ? T.strip_nr[sclass] + 16 : 0;
                    lds = std::max(lds, (size_t)stride * per + STRIP_TAB_BYTES + (size_t)per * 8 * nr_stride);
                    segs.s[segs.n++] = StripSeg{blocks, nl, base[l0], stride, nr_stride, glc, sclass >= 6 ? 1 : 0, pair};
                    blocks += nl / per;
                }
            if (blocks == 0) return 0;
#define MPN_STRIP_LAUNCH(EX)                                                                                                          \
            do {                                                                                                                      \
                if (lds > 64 * 1024) MPN_HIP_CHECK(hipFuncSetAttribute((const void *)ext_dp_strip_kernel<EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL(ext_dp_strip_kernel<EX>, dim3(blocks), dim3(64), lds, s, d_jobs.p, (const int32_t *)d_order.p, segs, prm, d_reads, \
                                   d_read_off, d_read_len, rv, P.p, d_res.p);                                                         \
            } while (0)
            if (fam == 0) MPN_STRIP_LAUNCH(false); else MPN_STRIP_LAUNCH(true);
#undef MPN_STRIP_LAUNCH
        } else if (l == L_TILE) {
            const size_t lds = (size_t)std::max(T.tile_lds, 64) + 64;   // (ext_tile_lds_bytes of the list's largest window)
            if (lds > 64 * 1024) MPN_HIP_CHECK(hipFuncSetAttribute((const void *)ext_dp_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(ext_dp_tile_kernel, dim3(n), dim3(64), lds, s, d_jobs.p, ord, n, prm, d_reads, d_read_off, d_read_len, rv, P.p, gstate.p, d_res.p);
        } else {
            const int bvar = (l - L_BAND) / 4;
            const size_t lds = std::max<size_t>((size_t)T.band_lds[(l - L_BAND) % 4], 64);
#define MPN_BAND_LAUNCH(NW, TT)                                                                                                       \
            do {                                                                                                                      \
                if (lds > 64 * 1024) MPN_HIP_CHECK(hipFuncSetAttribute((const void *)ext_dp_band_kernel<NW, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL((ext_dp_band_kernel<NW, TT>), dim3(n), dim3(NW * 64), lds, s, d_jobs.p, ord, n, prm, d_reads, d_read_off, \
                                   d_read_len, rv, P.p, d_res.p);                                                                    \
            } while (0)
            // (two slots per thread: what bounds these kernels is the latency of an anti-diagonal, i.e. the cells a thread computes in a row)
            if (bvar == 0) MPN_BAND_LAUNCH(1, 2); else if (bvar == 1) MPN_BAND_LAUNCH(2, 2); else if (bvar == 2) MPN_BAND_LAUNCH(4, 2); else MPN_BAND_LAUNCH(8, 2);
#undef MPN_BAND_LAUNCH
        }
        MPN_HIP_CHECK(hipGetLastError());
        return 0;
    };
    EvTimer ev(st);
    // the few large windows are the long pole: they go to a side stream and overlap the short windows below
    if (!SL.st2) {
        MPN_HIP_CHECK(hipStreamCreateWithFlags(&SL.st2, hipStreamNonBlocking));
        MPN_HIP_CHECK(hipEventCreateWithFlags(&SL.ev_a, hipEventDisableTiming));
        MPN_HIP_CHECK(hipEventCreateWithFlags(&SL.ev_b, hipEventDisableTiming));
        MPN_HIP_CHECK(hipStreamCreateWithFlags(&SL.st3, hipStreamNonBlocking));
        MPN_HIP_CHECK(hipEventCreateWithFlags(&SL.ev_c, hipEventDisableTiming));
    }
    MPN_HIP_CHECK(hipEventRecord(SL.ev_a, st));
    MPN_HIP_CHECK(hipStreamWaitEvent(SL.st2, SL.ev_a, 0));
    auto on_side = [](int l) { return (l >= L_WG && l < L_STRIP) || l >= L_BAND + 8; };  // workgroup windows, 512- and 1024-slot bands, tiled strips
    const bool have_tiles = cnt[L_TILE] > 0;
    if (have_tiles) {   // (a wave per window for milliseconds: on a stream of its own it overlaps the band kernels instead of preceding them)
        MPN_HIP_CHECK(hipStreamWaitEvent(SL.st3, SL.ev_a, 0));
        if (launch_list(L_TILE, d_order.p + base[L_TILE], cnt[L_TILE], SL.st3)) return -1;
    }
    for (int l = L_TILE - 1; l >= 0; --l)
        if (on_side(l) && launch_list(l, d_order.p + base[l], cnt[l], SL.st2)) return -1;
    // The long windows are also the long pole of the two lane-per-window kernels (traceback, z-drop test): those of the side
    // lists run on the side stream as soon as their DP is done, beside the strip DP of the main stream.
    const int side_lo[2] = {base[L_WG], base[L_BAND + 8]}, side_hi[2] = {base[L_STRIP], base[L_TILE]};
    const int tile_lo[2] = {base[L_TILE], 0}, tile_hi[2] = {base[N_LISTS], 0};
    const int main_lo[2] = {0, base[L_STRIP]}, main_hi[2] = {base[L_WG], base[L_BAND + 8]};
    auto bt_ztest = [&](const int *lo, const int *hi, hipStream_t s, bool timed) -> int {
        for (int k = 0; k < 2; ++k) {
            const int n = hi[k] - lo[k];
            if (n > 0) hipLaunchKernelGGL(ext_bt_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d_jobs.p, d_order.p + lo[k], n, P.p, OFF.p, CIG.p, d_compact, d_used, d_res.p);
        }
        MPN_HIP_CHECK(hipGetLastError());
        if (timed) ev.mark(25);
        // z-drop test of the gap-fill CIGARs (the kernel skips the other windows); flagged ones are recomputed with the exact maximum
        for (int k = 0; k < 2; ++k) {
            const int n = hi[k] - lo[k];
            if (n > 0) hipLaunchKernelGGL(ext_ztest_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d_jobs.p, d_order.p + lo[k], n, prm, d_reads, d_read_off, d_read_len, rv, CIG.p, d_res.p,
                                          d_redo_ids, d_used + 1, d_probes, d_used + 5);
        }
        MPN_HIP_CHECK(hipGetLastError());
        if (timed) ev.mark(26);
        return 0;
    };
    if (have_tiles) {
        if (bt_ztest(tile_lo, tile_hi, SL.st3, false)) return -1;
        MPN_HIP_CHECK(hipEventRecord(SL.ev_c, SL.st3));
    }
    if (bt_ztest(side_lo, side_hi, SL.st2, false)) return -1;
    MPN_HIP_CHECK(hipEventRecord(SL.ev_b, SL.st2));
    for (int l = N_LISTS - 1; l >= 0; --l) {  // wide before narrow, strips (the bulk) in the middle
        if (on_side(l)) continue;
        if (l == L_BAND - 1) ev.mark(15);  // the strip launches are timed on their own ([9]): the roofline kernel of bench.py
        if (l >= L_STRIP && l < L_BAND) {  // one launch per variant family: its lists are contiguous
            if (l == L_STRIP + 48) { if (launch_list(l, nullptr, 1, st)) return -1; ev.mark(57); }        // the exact variants (end extensions)
            else if (l == L_STRIP) { if (launch_list(l, nullptr, 1, st)) return -1; ev.mark(9, 38); }    // the gap fills: the roofline kernel of bench.py
        } else if (launch_list(l, d_order.p + base[l], cnt[l], st)) return -1;
    }
    ev.mark(15);
    if (bt_ztest(main_lo, main_hi, st, true)) return -1;
    MPN_HIP_CHECK(hipStreamWaitEvent(st, SL.ev_b, 0));
    if (have_tiles) MPN_HIP_CHECK(hipStreamWaitEvent(st, SL.ev_c, 0));
    unsigned long long *h_used = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(SL.pin_res.p) + ((sizeof(LayoutTotals) + 15) & ~(size_t)15));
    MPN_HIP_CHECK(hipMemcpyAsync(h_used, d_used, 48, hipMemcpyDeviceToHost, st));
    wt.stop_into(g_stats[28]);
    MPN_HIP_CHECK(stream_sync(st));
    wt.stop_into(g_stats[29]);
    // second pass: the windows whose CIGAR failed the z-drop test were listed by the test kernel (in no particular order), and
    // the windows whose largest drop may hide an inversion (mm_test_zdrop's probe: rare; decided here)
    std::vector<int32_t> redo((size_t)h_used[1]);
    const int n_probe = (int)h_used[5];
    if (!redo.empty() || n_probe) {
        // (rare: the job records of the group come to the host for it)
        std::vector<ExtJob> jobs((size_t)nj);
        std::vector<InvProbe> probes((size_t)n_probe);
        if (SL.pin_order.ensure(redo.size() * 4 + 16)) return -1;
        if (!redo.empty()) MPN_HIP_CHECK(hipMemcpyAsync(SL.pin_order.p, d_redo_ids, redo.size() * 4, hipMemcpyDeviceToHost, st));
        MPN_HIP_CHECK(hipMemcpyAsync(jobs.data(), d_jobs.p, (size_t)nj * sizeof(ExtJob), hipMemcpyDeviceToHost, st));
        if (n_probe) MPN_HIP_CHECK(hipMemcpyAsync(probes.data(), d_probes, (size_t)n_probe * sizeof(InvProbe), hipMemcpyDeviceToHost, st));
        MPN_HIP_CHECK(stream_sync(st));
        if (!redo.empty()) memcpy(redo.data(), SL.pin_order.p, redo.size() * 4);
        std::vector<uint8_t> is_inv((size_t)nj, 0);
        if (n_probe) {
            std::sort(probes.begin(), probes.end(), [](const InvProbe &x, const InvProbe &y) { return x.jid < y.jid; });
            std::vector<LocalHit> hits((size_t)n_probe, LocalHit{0, -1, -1});
            if (hs) {
                // the drop's query region on the opposite strand against the drop's target region (ksw_ll_i16 in minimap2)
                std::vector<std::vector<int8_t>> qv((size_t)n_probe), tv((size_t)n_probe);
                for (int k = 0; k < n_probe; ++k) {
                    const InvProbe &pb = probes[(size_t)k];
                    const ExtJob &jb = jobs[(size_t)pb.jid];
                    const int q_len = pb.q1 - pb.q0, t_len = pb.t1 - pb.t0, rlen = hs->seq_len[jb.read];
                    const char *rd = hs->seqs + hs->seq_off[jb.read];
                    qv[(size_t)k].resize((size_t)q_len);
                    for (int x = 0; x < q_len; ++x) { const int8_t c = host_qbase(rd, rlen, jb.rev, jb.qs + pb.q1 - x - 1); qv[(size_t)k][(size_t)x] = c >= 4 ? (int8_t)4 : (int8_t)(3 - c); }
                    tv[(size_t)k].resize((size_t)t_len);
                    if (fetch_ref_codes(rv, hs->tseq_off[jb.rid], jb.ts + pb.t0, t_len, tv[(size_t)k].data(), st)) return -1;
                }
                if (local_scores(opt, qv, tv, hits)) return -1;
            }
            for (int k = 0; k < n_probe; ++k) {
                const InvProbe &pb = probes[(size_t)k];
                const bool inv = hits[(size_t)k].score >= opt->min_chain_score * opt->a && hits[(size_t)k].score >= opt->min_dp_max;
                if (inv) is_inv[(size_t)pb.jid] = 1;
                if (inv || pb.over) redo.push_back(pb.jid);
            }
        }
        g_stats[8] += (int64_t)redo.size();
        const int nr = (int)redo.size();
        auto list_of = [&](int j) { return jobs[(size_t)j].cls & 0xff; };
        auto redo_list_of = [&](int j) { return jobs[(size_t)j].cls >> 8 & 0xff; };
        auto band_of = [&](int j) { return (jobs[(size_t)j].cls >> 16 & 0xff) - 1; };
        std::sort(redo.begin(), redo.end(), [&](int x, int y) { return redo_list_of(x) != redo_list_of(y) ? redo_list_of(x) < redo_list_of(y) : x < y; });
        // [ids | layout | qstride | inversion flag | p_off (int64)]: a strip window's second pass needs a band / anti-diagonal
        // matrix, which comes from a pool of its own (offsets are relative to the main pool's base: one flat address space)
        std::vector<int32_t> pack((size_t)nr * 6 + 2);
        const size_t off_p = ((size_t)nr * 4 + 1) & ~(size_t)1;
        int64_t *pack_off = reinterpret_cast<int64_t *>(pack.data() + off_p);
        int64_t p2_tot = 0;
        for (int k = 0; k < nr; ++k) {
            const int j = redo[k];
            const ExtJob &jb = jobs[(size_t)j];
            const int bv = band_of(j);
            pack[k] = j; pack[nr + k] = bv >= 0 ? 2 : 0; pack[2 * nr + k] = 128 << std::max(bv, 0); pack[3 * nr + k] = is_inv[(size_t)j];
            if ((list_of(j) >= L_STRIP && list_of(j) < L_BAND) || list_of(j) == L_TILE) {   // (strip and tiled layouts are sized for themselves)
                const int64_t n_r = (int64_t)jb.qlen + jb.tlen - 1;
                pack_off[k] = -1 - p2_tot;  // resolved below, once the pool address is known
                p2_tot += ((bv >= 0 ? n_r * (128 << bv) : n_r * jb.n_col) + 15) & ~(int64_t)15;
            } else pack_off[k] = jb.p_off;
        }
        if (nr > 0) {
            if (SL.pool_P2.ensure((size_t)p2_tot + 16)) return -1;
            const int64_t p2_base = (int64_t)(SL.pool_P2.as<uint8_t>() - P.p);
            for (int k = 0; k < nr; ++k) if (pack_off[k] < 0) pack_off[k] = p2_base + (-1 - pack_off[k]);
            if (SL.pool_redo.ensure(pack.size() * 4)) return -1;
            struct { int32_t *p; } d_redo{SL.pool_redo.as<int32_t>()};
            MPN_HIP_CHECK(hipMemcpyAsync(d_redo.p, pack.data(), pack.size() * 4, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(ext_redo_patch_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, d_jobs.p, d_redo.p, d_redo.p + nr, d_redo.p + 2 * nr,
                               reinterpret_cast<const int64_t *>(d_redo.p + off_p), d_redo.p + 3 * nr, opt->zdrop_inv, nr);
            MPN_HIP_CHECK(hipGetLastError());
            ev.skip();
            for (int lo = 0; lo < nr;) {
                int hi = lo;
                while (hi < nr && redo_list_of(redo[hi]) == redo_list_of(redo[lo])) ++hi;
                if (launch_list(redo_list_of(redo[lo]), d_redo.p + lo, hi - lo, st)) return -1;
                lo = hi;
            }
            ev.mark(15);
            hipLaunchKernelGGL(ext_bt_kernel, dim3((nr + 63) / 64), dim3(64), 0, st, d_jobs.p, d_redo.p, nr, P.p, OFF.p, CIG.p, d_compact, d_used, d_res.p);
            MPN_HIP_CHECK(hipGetLastError());
            ev.mark(25);
            MPN_HIP_CHECK(stream_sync(st));
        }
    }
    ev.resolve();
    wt.stop_into(g_stats[30]);
    return 0;
}

// Run the DP windows whose raw job records the caller has put at the start of the worker's job pool (SL.pool_jobs), in groups
// whose direction scratch stays under the budget (MPN_DP_BUDGET bytes, for tests).  The job records, results and compacted
// CIGARs of ALL groups stay in the worker's device pools (out); a group only borrows the direction-matrix scratch.
// block_zeroed: the caller has zeroed the whole counter block of the round (and the planning kernel has counted into [3] since)
static int run_jobs(const RefView &rv, const mpn_map_opt *opt, int nj_cap, const uint8_t *d_reads, const int64_t *d_read_off,
                    const int32_t *d_read_len, DevRound &out, const HostSeqs *hs, hipStream_t st, bool block_zeroed = false) {
    static const int64_t budget = []() { const char *e = getenv("MPN_DP_BUDGET"); return e ? std::max<int64_t>(1 << 20, atoll(e)) : (int64_t)40 << 30; }();
    Slot &SL = *tl_slot;
    if (SL.pool_res.ensure((size_t)nj_cap * sizeof(ExtRes) + 16) || SL.pool_used.ensure(128)) return -1;
    out.jobs = SL.pool_jobs.as<ExtJob>(); out.res = SL.pool_res.as<ExtRes>(); out.compact = nullptr;
    out.used = SL.pool_used.as<unsigned long long>();   // [0] compacted ops, [1] second-pass windows, [2] stitched ops, [3] windows of the round, [4] of a sub-range
    if (!block_zeroed) {   // [0..2] and [5..7]; [3], [4] carry the window counts
        MPN_HIP_CHECK(hipMemsetAsync(out.used, 0, 24, st));
        MPN_HIP_CHECK(hipMemsetAsync(out.used + 5, 0, 24, st));
    }
    const int rc = run_job_group(rv, opt, nj_cap, out.used + 3, out, d_reads, d_read_off, d_read_len, budget, hs, st);
    if (rc != 1) return rc;
    // over the budget: cut the range where the direction matrices (their offsets are in the size table) fill it
    const int nj = out.n_jobs;
    std::vector<JobSizes> sz((size_t)nj);
    MPN_HIP_CHECK(hipMemcpy(sz.data(), SL.pool_sizes.p, (size_t)nj * sizeof(JobSizes), hipMemcpyDeviceToHost));
    std::vector<int> cuts{0};
    for (int j = 1; j < nj; ++j) if (sz[(size_t)j].p - sz[(size_t)cuts.back()].p > budget) cuts.push_back(j);
    cuts.push_back(nj);
    for (size_t g = 0; g + 1 < cuts.size(); ++g) {
        DevRound dv = out;
        dv.jobs += cuts[g]; dv.res += cuts[g];
        const unsigned long long cnt = (unsigned long long)(cuts[g + 1] - cuts[g]);
        MPN_HIP_CHECK(hipMemcpy(out.used + 4, &cnt, 8, hipMemcpyHostToDevice));
        const int r2 = run_job_group(rv, opt, (int)cnt, out.used + 4, dv, d_reads, d_read_off, d_read_len, 0, hs, st);
        if (r2) return r2 < 0 ? r2 : -1;
    }
    out.n_jobs = nj;
    return 0;
}

}  // namespace mpn

using namespace mpn;

// Stitching + CIGAR fix-up + statistics of every hit aligned in this round, on the device (stitch_kernels.h, fin_kernels.h):
// the windows' CIGARs are concatenated per hit in HBM, fixed and measured there; the host gets 56 + 32 bytes per hit, and the
// fixed CIGARs only when the caller wants text (need_cigar).
static bool g_need_cigar = true;   // set per mapping call (the calls take turns: g_call_mu)

// n_sr hits of the round; their StitchReg / PlanReg / PlanSum records and the windows' anchor indices are on the device (written
// by plan_kernel); sr_base[i] = index of read i's first hit of the round, S.pending its hits.
struct RoundDev {
    const StitchReg *sregs; const PlanReg *pregs; const PlanSum *psum; const int32_t *job_anchor; const u128 *anchors;
};

static int stitch_and_finish(const mpn_index *idx, const mpn_map_opt *opt, ReadState *rs, int n, const int32_t *seq_len, const DevRound &dv,
                             const RoundDev &rd, const std::vector<int32_t> &sr_base, const uint8_t *d_seqs, const int64_t *d_off, const int32_t *d_len,
                             int n_threads, hipStream_t st) {
    const int n_sr = sr_base[(size_t)n];
    if (n_sr == 0) return 0;
    Slot &SL = *tl_slot;
    if (SL.pin_fin_out.ensure((size_t)n_sr * (sizeof(StitchOut) + sizeof(FinOut) + sizeof(SplitRec)) + 128) ||
        SL.pool_souts.ensure((size_t)n_sr * sizeof(StitchOut) + 16) || SL.pool_splits.ensure((size_t)n_sr * sizeof(SplitRec) + 16) ||
        SL.pool_fin_jobs.ensure((size_t)n_sr * sizeof(FinJob) + 16) || SL.pool_fin_out.ensure((size_t)n_sr * sizeof(FinOut) + 16) ||
        SL.pool_fin_cig.ensure((size_t)dv.cig_cap * 4 + 16))
        return -1;
    StitchOut *d_so = SL.pool_souts.as<StitchOut>();
    FinJob *d_fj = SL.pool_fin_jobs.as<FinJob>();
    FinOut *d_fo = SL.pool_fin_out.as<FinOut>();
    uint32_t *d_cig = SL.pool_fin_cig.as<uint32_t>();
    unsigned long long *d_out_used = dv.used + 2;
    StitchOut *h_so = SL.pin_fin_out.as<StitchOut>();
    FinOut *h_fo = reinterpret_cast<FinOut *>(h_so + n_sr);
    unsigned long long *h_out_used = reinterpret_cast<unsigned long long *>(h_fo + n_sr);   // the counter block as read back
    SplitRec *h_splits = reinterpret_cast<SplitRec *>(h_out_used + 8);
    SplitRec *d_splits = SL.pool_splits.as<SplitRec>();
    unsigned long long *d_n_splits = dv.used + 6;
    EvTimer ev(st);
    ev.skip();
    hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)std::min(n_sr, 256 * 32)), dim3(64), 0, st, rd.sregs, n_sr, (const ExtJob *)dv.jobs,
                       (const ExtRes *)dv.res, (const uint32_t *)dv.compact, d_cig, d_out_used, d_so, d_fj, rd.pregs, rd.psum, rd.job_anchor, rd.anchors,
                       opt->min_cnt, d_splits, d_n_splits);
    MPN_HIP_CHECK(hipGetLastError());
    ev.mark(54);
    MPN_HIP_CHECK(hipMemcpyAsync(h_so, d_so, (size_t)n_sr * sizeof(StitchOut), hipMemcpyDeviceToHost, st));
    MPN_HIP_CHECK(hipMemcpyAsync(h_out_used, dv.used, 64, hipMemcpyDeviceToHost, st));   // the round's counter block: [2] stitched ops, [6] cut hits
    MPN_HIP_CHECK(stream_sync(st));
    const int64_t n_ops = (int64_t)h_out_used[2];
    const unsigned long long n_cut = h_out_used[6];
    if (n_cut) {   // (the cut hits' records: rare)
        MPN_HIP_CHECK(hipMemcpyAsync(h_splits, d_splits, (size_t)n_cut * sizeof(SplitRec), hipMemcpyDeviceToHost, st));
        MPN_HIP_CHECK(stream_sync(st));
    }
    // hits: coordinates, score, splits at z-drops
    parallel_for(n, n_threads, [&](int i, int) {
        ReadState &S = rs[i];
        if (S.pending.empty()) return;
        int shift = 0;
        for (size_t pi = 0; pi < S.pending.size(); ++pi) {
            const int k = S.pending[pi] + shift, fi = sr_base[(size_t)i] + (int)pi;
            Reg r2;
            const bool has = apply_stitch(seq_len[i], S.regs[(size_t)k], r2, h_splits, h_so[fi], fi);
            if (has) { S.regs.insert(S.regs.begin() + k + 1, r2); ++shift; }
        }
    }, 4);
    // launch lists of the finishing kernel by LDS need (CIGAR, shift table, both code arrays): three LDS classes, the rest works
    // in global scratch (a fourth class of 152 KB for the longest alignments runs them faster but blocks whole CUs: -2 % under
    // the full pipeline).  Hits without operations have nothing to fix (what update_extra leaves for an empty CIGAR).
    constexpr int NC = 3;
    static const size_t kLds[NC] = {(size_t)16 << 10, (size_t)32 << 10, (size_t)64 << 10};
    std::vector<int32_t> lists[NC + 1];
    std::vector<int64_t> code_offs;
    int64_t code_bytes = 0;
    for (int j = 0; j < n_sr; ++j) {
        const StitchOut &so = h_so[j];
        if (so.n_ops == 0 || !so.has_p) continue;
        const size_t codes = (size_t)((std::max(0, so.qe1 - so.qs1) + 3) & ~3) + (size_t)((std::max(0, so.re1 - so.rs1) + 3) & ~3);
        const size_t need = (size_t)so.n_ops * 8 + codes + 16;
        int c = 0;
        while (c < NC && need > kLds[c]) ++c;
        if (c == NC) { code_offs.push_back(code_bytes); code_bytes += (int64_t)codes; }
        lists[c].push_back(j);
    }
    std::vector<int32_t> order;
    int base[NC + 1];
    for (int c = 0; c <= NC; ++c) { base[c] = (int)order.size(); order.insert(order.end(), lists[c].begin(), lists[c].end()); }
    if (!order.empty()) {
        DevBuf<uint32_t> d_aux;
        DevBuf<uint8_t> d_codes;
        DevBuf<int64_t> d_code_offs;
        DevBuf<int32_t> d_list;
        if (!lists[NC].empty() && (d_aux.alloc((size_t)n_ops + 1) || d_codes.alloc((size_t)code_bytes + 16) || d_code_offs.upload(code_offs.data(), code_offs.size(), st)))
            return -1;
        if (d_list.upload(order.data(), order.size(), st)) return -1;
        FinParams prm;
        for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) prm.mat[i * 5 + j] = (int8_t)(i == j ? opt->a : -opt->b); prm.mat[i * 5 + 4] = (int8_t)-opt->sc_ambi; }
        for (int i = 0; i < 5; ++i) prm.mat[20 + i] = (int8_t)-opt->sc_ambi;
        prm.q = (int8_t)opt->q; prm.e = (int8_t)opt->e;
        const RefView rvw{idx->d_seq2.p, idx->d_seq_off.p, idx->d_nrun_s.p, idx->d_nrun_e.p, idx->n_nruns};
        ev.skip();
        MPN_HIP_CHECK(hipFuncSetAttribute((const void *)aln_finish_wave_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds[NC - 1]));
        for (int c = 0; c < NC; ++c)
            if (!lists[c].empty())
                hipLaunchKernelGGL(aln_finish_wave_kernel<true>, dim3((unsigned)std::min<size_t>(lists[c].size(), 256 * 64)), dim3(64), kLds[c], st, (const FinJob *)d_fj,
                                   (const int32_t *)d_list.p + base[c], (int)lists[c].size(), d_cig, (uint32_t *)nullptr, (uint8_t *)nullptr, d_seqs, d_off, d_len, rvw, prm, d_fo,
                                   (const int64_t *)nullptr);
        if (!lists[NC].empty())
            hipLaunchKernelGGL(aln_finish_wave_kernel<false>, dim3((unsigned)std::min<size_t>(lists[NC].size(), 256 * 64)), dim3(64), 0, st, (const FinJob *)d_fj,
                               (const int32_t *)d_list.p + base[NC], (int)lists[NC].size(), d_cig, d_aux.p, d_codes.p, d_seqs, d_off, d_len, rvw, prm, d_fo,
                               (const int64_t *)d_code_offs.p);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(52);
        MPN_HIP_CHECK(hipMemcpyAsync(h_fo, d_fo, (size_t)n_sr * sizeof(FinOut), hipMemcpyDeviceToHost, st));
        if (g_need_cigar) {
            if (SL.pin_fin_cig.ensure((size_t)n_ops * 4 + 16)) return -1;
            MPN_HIP_CHECK(hipMemcpyAsync(SL.pin_fin_cig.p, d_cig, (size_t)n_ops * 4, hipMemcpyDeviceToHost, st));
        }
        MPN_HIP_CHECK(stream_sync(st));
    }
    ev.resolve();
    g_stats[53] += n_ops;
    const uint32_t *h_cig = SL.pin_fin_cig.as<uint32_t>();
    parallel_for(n, n_threads, [&](int i, int) {
        for (Reg &r : rs[i].regs) {
            if (r.fin_idx < 0) continue;
            const StitchOut &so = h_so[r.fin_idx];
            if (so.n_ops == 0) { r.blen = r.mlen = 0; r.dp_max = 0; r.fin_idx = -1; continue; }   // (what update_extra leaves for an empty CIGAR)
            const FinOut &f = h_fo[r.fin_idx];
            if (g_need_cigar) r.cigar.assign(h_cig + so.cig_off, h_cig + so.cig_off + f.n_cigar);
            if (f.qshift) { if (r.rev) r.qe -= f.qshift; else r.qs += f.qshift; }
            r.rs += f.tshift;
            r.blen = f.blen; r.mlen = f.mlen; r.n_ambi += f.n_ambi; r.dp_max = f.dp_max;
            r.fin_idx = -1;
        }
    }, 9);
    return 0;
}

// One contiguous range [lo, hi) of the batch through the whole path, on the calling worker's stream and arena.
// Fills rs[lo..hi) (the final hits of every read) and rep_len[lo..hi).
static int map_range(const mpn_index *idx, const mpn_map_opt *opt, const char *const *names, const char *seqs,
                     const int64_t *seq_off_all, const int32_t *seq_len_all, const uint8_t *d_seqs_p, const int64_t *d_off_all,
                     const int32_t *d_len_all, const uint32_t *d_name_hash_all, int lo, int hi, int n_threads, hipStream_t st,
                     std::vector<ReadState> &rs_all, std::vector<int32_t> &rep_len_all, const ReadSketch *sketch) {
    const int n = hi - lo;
    if (n <= 0) return 0;
    const int64_t *seq_off = seq_off_all + lo;
    const int32_t *seq_len = seq_len_all + lo;
    struct { const uint8_t *p; } d_seqs{d_seqs_p};
    struct { const int64_t *p; } d_off{d_off_all + lo};
    struct { const int32_t *p; } d_len{d_len_all + lo};
    ReadState *rs = rs_all.data() + lo;
    WallTimer wt;
    HostChains h;
    SeedChainOut o;   // (the chained anchors stay in its device pool until the squeeze below)
    bool gpu_hits = true;
    {
        // The seed + sort + chain stage is bound by HBM traffic and latency, the extension stage by VALU issue: workers in
        // different stages share the GPU well, workers in the same memory-bound stage only queue on HBM.  At most
        // MPN_SEED_SLOTS workers are inside this stage at a time (which also keeps the workers out of lock-step).
        struct StageGate {
            std::mutex mu; std::condition_variable cv; int free_slots;
            StageGate() { const char *e = getenv("MPN_SEED_SLOTS"); free_slots = e ? std::max(1, atoi(e)) : 8; }
            void enter() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return free_slots > 0; }); --free_slots; }
            void leave() { { std::lock_guard<std::mutex> g(mu); ++free_slots; } cv.notify_one(); }
        };
        static StageGate gate;
        struct Hold { StageGate &g; Hold(StageGate &x) : g(x) { g.enter(); } ~Hold() { g.leave(); } } hold(gate);
        if (seed_chain_device(idx, opt, n, d_seqs.p, d_off.p, d_len.p, seq_len, o, st, sketch)) return -1;
        wt.stop_into(g_stats[17]);
        // MPN_HOST_HITS=1 (tests): hits from chains on the host for every read, from the downloaded chain records
        static const bool host_hits = []() { const char *e = getenv("MPN_HOST_HITS"); return e && atoi(e) != 0; }();
        gpu_hits = !host_hits;
        if (download_chains(n, o, h, tl_slot->pin_chain_u, tl_slot->pin_chain_b, st, gpu_hits ? 0 : 1)) return -1;
        wt.stop_into(g_stats[18]);
    }
    g_stats[3] += h.chain_off[n];
    for (int i = 0; i < n; ++i) rep_len_all[lo + i] = h.rep_len[i];
    // Hits from chains.  hit_select_kernel (hit_kernels.h) makes them on the GPU from the chain records and leaves the squeeze
    // segments there too; the host receives the selected hits (72 bytes each) and 24 bytes per read.  The chained anchors never
    // leave HBM.  Reads with more chains than the kernel stages (and every read under MPN_HOST_HITS=1) take the host path: the same
    // functions of hit.c on the downloaded records, their segments uploaded behind the kernel's.
    Slot &SL = *tl_slot;
    const int64_t n_chains_all = h.chain_off[(size_t)n];
    if (SL.pin_segs.ensure((size_t)n_chains_all * sizeof(SqueezeSeg) + (size_t)(n + 1) * 8 + 64)) return -1;
    int64_t *sq_off = SL.pin_segs.as<int64_t>();                                     // [n + 1] start of every read's squeezed list
    SqueezeSeg *h_segs = reinterpret_cast<SqueezeSeg *>(sq_off + n + 1);
    DevBuf<HitRec> d_hregs;
    DevBuf<SqueezeSeg> d_hsegs;
    DevBuf<unsigned char> d_hreads;   // [counters (4 x 8 bytes) | HitRead[n]]
    const HitRead *h_reads = nullptr;
    const HitRec *h_hregs = nullptr;
    unsigned long long hit_counters[4] = {0, 0, 0, 0};
    if (d_hsegs.alloc((size_t)n_chains_all + 1)) return -1;
    if (gpu_hits && n_chains_all > 0) {
        const size_t reads_bytes = 32 + (size_t)n * sizeof(HitRead);
        if (d_hregs.alloc((size_t)n_chains_all + 1) || d_hreads.alloc(reads_bytes) || SL.pin_hits.ensure(reads_bytes + (size_t)n_chains_all * sizeof(HitRec) + 64)) return -1;
        MPN_HIP_CHECK(hipMemsetAsync(d_hreads.p, 0, 32, st));
        HitSelParams hp;
        hp.mask_level = opt->mask_level; hp.pri_ratio = opt->pri_ratio; hp.min_join_flank_ratio = opt->min_join_flank_ratio;
        hp.min_diff = idx->k * 2; hp.best_n = opt->best_n; hp.max_join_long = opt->max_join_long; hp.max_join_short = opt->max_join_short;
        hp.min_join_flank_sc = opt->min_join_flank_sc; hp.min_cnt = opt->min_cnt; hp.with_cigar = opt->with_cigar; hp.seed_mix = wang32(opt->seed);
        static const int hit_max = []() { const char *e = getenv("MPN_HIT_MAX_CHAINS"); return e ? std::max(0, std::min(HIT_MAX_CHAINS, atoi(e))) : HIT_MAX_CHAINS; }();
        hp.max_chains = hit_max;
        EvTimer evh(st);
        // two instantiations: reads with few chains (nearly all reads of a random target set) need little LDS, so many waves per CU
#define MPN_HIT_LAUNCH(NN, LO, GRID)                                                                                                      \
        hipLaunchKernelGGL(hit_select_kernel<NN>, dim3((unsigned)std::max(1, std::min(n, GRID))), dim3(64), 0, st, hp, LO, n, (const int32_t *)o.n_chain.p, \
                           (const int64_t *)o.u_pos.p, (const int64_t *)o.b_pos.p, (const uint64_t *)o.u_compact.p, (const ChainRec *)o.recs.p, d_len.p, \
                           d_name_hash_all + lo, d_hregs.p, d_hsegs.p, reinterpret_cast<unsigned long long *>(d_hreads.p),                \
                           reinterpret_cast<HitRead *>(d_hreads.p + 32))
        MPN_HIT_LAUNCH(HIT_SMALL_CHAINS, 0, 256 * 16);
        MPN_HIT_LAUNCH(HIT_MAX_CHAINS, HIT_SMALL_CHAINS, 256 * 3);
#undef MPN_HIT_LAUNCH
        MPN_HIP_CHECK(hipGetLastError());
        evh.mark(62);
        unsigned char *pin = SL.pin_hits.as<unsigned char>();
        MPN_HIP_CHECK(hipMemcpyAsync(pin, d_hreads.p, reads_bytes, hipMemcpyDeviceToHost, st));
        MPN_HIP_CHECK(stream_sync(st));
        evh.resolve();
        memcpy(hit_counters, pin, 32);
        h_reads = reinterpret_cast<const HitRead *>(pin + 32);
        HitRec *dst = reinterpret_cast<HitRec *>(pin + ((reads_bytes + 15) & ~(size_t)15));
        if (hit_counters[0]) MPN_HIP_CHECK(hipMemcpyAsync(dst, d_hregs.p, (size_t)hit_counters[0] * sizeof(HitRec), hipMemcpyDeviceToHost, st));
        if (hit_counters[2] && download_chain_records(o, h, SL.pin_chain_u, st)) return -1;   // reads left to the host
        MPN_HIP_CHECK(stream_sync(st));
        h_hregs = dst;
        g_stats[63] += (int64_t)hit_counters[2];
    }
    // (every pool thread stages the segments of its reads in a list of its own: a shared cursor is a hot cache line)
    std::vector<std::vector<SqueezeSeg>> &stage = SL.seg_stage;
    if ((int)stage.size() < std::max(1, n_threads)) stage.resize((size_t)std::max(1, n_threads));
    for (auto &v : stage) v.clear();
    parallel_for(n, n_threads, [&](int i, int slot) {
        ReadState &S = rs[i];
        const int nc = h.n_chain[i];
        if (nc == 0) return;
        const int qlen = seq_len[i];
        if (h_reads && h_reads[i].n_regs >= 0) {
            // the kernel's hits: the rest of the record follows from the first and last anchor (mm_reg_set_coor)
            const HitRead &hr = h_reads[i];
            const HitRec *src = h_hregs + hr.reg_pos;
            S.regs.assign((size_t)hr.n_regs, Reg());
            for (int k = 0; k < hr.n_regs; ++k) {
                Reg &r = S.regs[(size_t)k];
                const HitRec &x = src[k];
                r.id = k; r.parent = x.parent; r.score = x.score; r.score0 = x.score0; r.hash = x.hash; r.cnt = x.cnt; r.as = x.as;
                r.subsc = x.subsc; r.n_sub = x.n_sub; r.mlen = x.mlen; r.blen = x.blen;
                r.fx = x.fx; r.fy = x.fy; r.lx = x.lx; r.ly = x.ly;
                reg_set_coor(r, qlen);
            }
            if (hr.flags & 1) set_sam_pri(S.regs);
            S.n_a = hr.n_a;
            return;
        }
        CpuSect sect(g_cpu_on);
        static thread_local std::vector<ChainIn> cin;
        std::vector<int32_t> &order = tl_hs.order;
        std::vector<int64_t> &src = tl_hs.src;
        order.resize((size_t)nc); src.resize((size_t)nc); cin.resize((size_t)nc);
        h.chain_order(i, order.data(), src.data());
        for (int c = 0; c < nc; ++c) cin[c] = ChainIn{h.u_all[h.u_pos[i] + order[c]], h.rec_all + h.u_pos[i] + order[c], src[c]};
        sect.lap(3);
        uint32_t hash = names && names[lo + i] ? x31_hash(names[lo + i]) : 0;
        hash ^= wang32((uint32_t)qlen) + wang32(opt->seed);
        hash = wang32(hash);
        gen_regs(hash, qlen, nc, cin.data(), S.regs);
        set_parent(opt->mask_level, S.regs, opt->a * 2 + opt->b);
        select_sub(opt->pri_ratio, idx->k * 2, opt->best_n, S.regs);
        sect.lap(4);
        std::vector<SqueezeSeg> &segs = stage[(size_t)slot % stage.size()];
        const size_t seg0 = segs.size();
        S.n_a = squeeze_a(S.regs, i, h.b_pos[i], segs);
        join_long(opt, qlen, S.regs, segs);
        if (!opt->with_cigar) segs.resize(seg0);
        sect.lap(5);
    }, 1, 128);
    int64_t n_host_segs = 0;
    for (auto &v : stage) { if (!v.empty()) memcpy(h_segs + n_host_segs, v.data(), v.size() * sizeof(SqueezeSeg)); n_host_segs += (int64_t)v.size(); }
    const int64_t n_dev_segs = (int64_t)hit_counters[1], n_segs = n_dev_segs + n_host_segs;
    wt.stop_into(g_stats[19]);
    sq_off[0] = 0;
    for (int i = 0; i < n; ++i) sq_off[i + 1] = sq_off[i] + rs[i].n_a;
    const int64_t n_sq = sq_off[n];
    if (opt->with_cigar && n_sq > 0) {
        DevBuf<u128> d_a;
        DevBuf<int64_t> d_sq_off;
        if (d_a.alloc((size_t)n_sq) || d_sq_off.alloc((size_t)n + 1)) return -1;
        MPN_HIP_CHECK(hipMemcpyAsync(d_sq_off.p, sq_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));
        // (the host's segments go behind the kernel's in the device list)
        if (n_host_segs) MPN_HIP_CHECK(hipMemcpyAsync(d_hsegs.p + n_dev_segs, h_segs, (size_t)n_host_segs * sizeof(SqueezeSeg), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(anchor_squeeze_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_segs + 3) / 4, 256 * 16))), dim3(256), 0, st,
                           (const SqueezeSeg *)d_hsegs.p, (int)n_segs, (const u128 *)o.chained.p, (const int64_t *)d_sq_off.p, d_a.p);
        MPN_HIP_CHECK(hipGetLastError());
        g_stats[59] += n_sq;
        PlanOpt po;
        po.bw = opt->bw; po.bw15 = (int)(opt->bw * 1.5 + 1.); po.min_chain_score = opt->min_chain_score; po.max_gap = opt->max_gap;
        po.min_cnt = opt->min_cnt; po.a = opt->a; po.q = opt->q; po.e = opt->e; po.zdrop = opt->zdrop; po.zdrop_inv = opt->zdrop_inv;
        po.end_bonus = opt->end_bonus; po.min_ksw_len = opt->min_ksw_len; po.k = idx->k; po.pad = 0; po.max_sw_mat = opt->max_sw_mat;
        const RefView rv{idx->d_seq2.p, idx->d_seq_off.p, idx->d_nrun_s.p, idx->d_nrun_e.p, idx->n_nruns};
        const HostSeqs hseqs{seqs, seq_off, seq_len, idx->seq_off.data()};
        std::vector<int32_t> sr_base((size_t)n + 1, 0);
        for (int round = 0; round < 64; ++round) {
            wt.stop_into(g_stats[23]);
            // the hits to align in this round: 40 bytes each go up; their windows are planned, laid out, computed, traced back and
            // stitched on the device
            parallel_for(n, n_threads, [&](int i, int) {
                ReadState &S = rs[i];
                S.pending.clear();
                for (int k = 0; k < (int)S.regs.size(); ++k) {
                    Reg &r = S.regs[(size_t)k];
                    if (r.aligned) continue;
                    if (r.cnt == 0) { r.aligned = 1; continue; }
                    S.pending.push_back(k);
                }
            }, 2);
            for (int i = 0; i < n; ++i) sr_base[(size_t)i + 1] = sr_base[(size_t)i] + (int32_t)rs[i].pending.size();
            const int n_sr = sr_base[(size_t)n];
            if (n_sr == 0) { wt.stop_into(g_stats[20]); break; }
            std::vector<int64_t> cap_t((size_t)std::max(1, n_threads), 0);
            if (SL.pin_pregs.ensure((size_t)n_sr * sizeof(PlanReg) + 64) || SL.pool_pregs.ensure((size_t)n_sr * sizeof(PlanReg) + 16) ||
                SL.pool_psum.ensure((size_t)n_sr * sizeof(PlanSum) + 16) || SL.pool_sregs.ensure((size_t)n_sr * sizeof(StitchReg) + 16) || SL.pool_used.ensure(128))
                return -1;
            PlanReg *h_pr = SL.pin_pregs.as<PlanReg>();
            parallel_for(n, n_threads, [&](int i, int slot) {
                const ReadState &S = rs[i];
                for (size_t pi = 0; pi < S.pending.size(); ++pi) {
                    const Reg &r = S.regs[(size_t)S.pending[pi]];
                    h_pr[(size_t)sr_base[(size_t)i] + pi] = PlanReg{sq_off[i], S.n_a, r.as, r.cnt, r.mlen, i, seq_len[i], (int32_t)r.split_inv, 0};
                    cap_t[(size_t)slot % cap_t.size()] += r.cnt + 2;   // a hit of cnt anchors has at most cnt + 1 windows
                }
            }, 3);
            int64_t nj_cap64 = 0;
            for (int64_t c : cap_t) nj_cap64 += c;
            if (nj_cap64 > 0x7fffffff) { set_error("too many DP windows in one round"); return -1; }
            const int nj_cap = (int)nj_cap64;
            if (SL.pool_jobs.ensure((size_t)nj_cap * sizeof(ExtJob) + 16) || SL.pool_job_anchor.ensure((size_t)nj_cap * 4 + 16)) return -1;
            PlanReg *d_pr = SL.pool_pregs.as<PlanReg>();
            PlanSum *d_ps = SL.pool_psum.as<PlanSum>();
            StitchReg *d_sr = SL.pool_sregs.as<StitchReg>();
            ExtJob *d_jobs = SL.pool_jobs.as<ExtJob>();
            int32_t *d_janchor = SL.pool_job_anchor.as<int32_t>();
            unsigned long long *d_used = SL.pool_used.as<unsigned long long>();
            MPN_HIP_CHECK(hipMemsetAsync(d_used, 0, 64, st));   // every counter of the round in one fill
            MPN_HIP_CHECK(hipMemcpyAsync(d_pr, h_pr, (size_t)n_sr * sizeof(PlanReg), hipMemcpyHostToDevice, st));
            EvTimer evp(st);
            hipLaunchKernelGGL(plan_kernel, dim3((unsigned)std::min(n_sr, 256 * 4)), dim3(64), 0, st, po, (const PlanReg *)d_pr, n_sr, d_a.p,
                               (const int32_t *)idx->d_lens.p, d_used + 3, d_jobs, d_janchor, d_sr, d_ps);
            MPN_HIP_CHECK(hipGetLastError());
            evp.mark(55);
            wt.stop_into(g_stats[20]);
            ++g_stats[7];
            DevRound dv;
            if (run_jobs(rv, opt, nj_cap, d_seqs.p, d_off.p, d_len.p, dv, &hseqs, st, true)) return -1;
            evp.resolve();
            if (!dv.compact) {   // (a round without any window: the stitching kernel still sets the hits' coordinates)
                if (SL.pool_compact.ensure(64)) return -1;
                dv.compact = SL.pool_compact.as<uint32_t>();
            }
            wt.stop_into(g_stats[21]);
            const RoundDev rd{d_sr, d_pr, d_ps, d_janchor, d_a.p};
            if (stitch_and_finish(idx, opt, rs, n, seq_len, dv, rd, sr_base, d_seqs.p, d_off.p, d_len.p, n_threads, st)) return -1;
            wt.stop_into(g_stats[22]);
        }
        // ---- inversions (mm_align1_inv): where a hit was cut at an inversion, the read's gap between the two pieces is aligned
        // to the target's gap on the opposite strand.  Rare; the candidates are gathered on the host, the local alignment that
        // locates the inverted segment runs on the SSW kernels, its extension is one more (tiny) round of the DP pipeline.
        struct InvCand { int read, at, ql, tl, rev, qstart, tstart, rid; };
        std::vector<InvCand> cand;
        for (int i = 0; i < n; ++i) {
            const std::vector<Reg> &R = rs[i].regs;
            for (int k = 1; k < (int)R.size(); ++k) {
                if (!R[(size_t)k].split_inv) continue;
                const Reg &r1 = R[(size_t)k - 1], &r2 = R[(size_t)k];
                if (!(r1.split & 1) || !(r2.split & 2)) continue;
                if (r1.id != r1.parent && r1.parent != PARENT_TMP_PRI) continue;
                if (r2.id != r2.parent && r2.parent != PARENT_TMP_PRI) continue;
                if (r1.rid != r2.rid || r1.rev != r2.rev) continue;
                const int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
                if (ql < opt->min_chain_score || ql > opt->max_gap || tl < opt->min_chain_score || tl > opt->max_gap) continue;
                // the read's gap on the strand opposite to the hit: forward coordinates from r2.qe if the hit is on the reverse
                // strand, reverse-strand coordinates from qlen - r2.qs otherwise
                cand.push_back(InvCand{i, k, ql, tl, r1.rev ? 0 : 1, r1.rev ? r2.qe : seq_len[i] - r2.qs, r1.re, r1.rid});
            }
        }
        if (!cand.empty()) {
            const int nc = (int)cand.size();
            std::vector<std::vector<int8_t>> qv((size_t)nc), tv((size_t)nc);
            for (int c = 0; c < nc; ++c) {   // both sequences reversed: the best local alignment's END there is its START here
                const InvCand &ic = cand[(size_t)c];
                const char *rd_ = seqs + seq_off[ic.read];
                qv[(size_t)c].resize((size_t)ic.ql);
                for (int x = 0; x < ic.ql; ++x) qv[(size_t)c][(size_t)x] = host_qbase(rd_, seq_len[ic.read], ic.rev, ic.qstart + ic.ql - 1 - x);
                std::vector<int8_t> t((size_t)ic.tl);
                if (fetch_ref_codes(rv, idx->seq_off[(size_t)ic.rid], ic.tstart, ic.tl, t.data(), st)) return -1;
                tv[(size_t)c].assign(t.rbegin(), t.rend());
            }
            std::vector<LocalHit> lh;
            if (local_scores(opt, qv, tv, lh)) return -1;
            std::vector<ExtJob> jobs;
            std::vector<StitchReg> sregs;
            for (int i = 0; i < n; ++i) rs[i].pending.clear();
            // placeholders go in from the back of every read's list, so that the positions of the earlier candidates stay valid
            for (int c = nc - 1; c >= 0; --c) {
                const InvCand &ic = cand[(size_t)c];
                if (lh[(size_t)c].score < opt->min_dp_max) continue;
                const int q_off = ic.ql - (lh[(size_t)c].qe + 1), t_off = ic.tl - (lh[(size_t)c].te + 1);
                Reg ri;
                ri.id = -1; ri.parent = PARENT_UNSET; ri.inv = 1; ri.rev = (uint32_t)ic.rev; ri.rid = ic.rid; ri.cnt = 0; ri.aligned = 0;
                ri.fin_qs = ic.qstart + q_off; ri.fin_rs = ic.tstart + t_off; ri.fin_ql = ic.ql - q_off; ri.fin_tl = ic.tl - t_off;
                rs[ic.read].regs.insert(rs[ic.read].regs.begin() + ic.at + 1, ri);
            }
            for (int i = 0; i < n; ++i) {
                std::vector<Reg> &R = rs[i].regs;
                for (int k = 0; k < (int)R.size(); ++k) {
                    Reg &r = R[(size_t)k];
                    if (!r.inv || r.aligned) continue;
                    rs[i].pending.push_back(k);
                    ExtJob j;
                    memset(&j, 0, sizeof(j));
                    j.read = i; j.rid = r.rid; j.rev = (int32_t)r.rev; j.qs = r.fin_qs; j.qlen = r.fin_ql; j.ts = r.fin_rs; j.tlen = r.fin_tl; j.reversed = 0;
                    j.w = (int)(opt->bw * 1.5); j.zdrop = opt->zdrop; j.end_bonus = -1; j.flag = EZ_EXTZ_ONLY; j.strip_s = 1; j.cls = -1;
                    sregs.push_back(StitchReg{(int32_t)jobs.size(), 1, j.qs, j.ts, j.qs, j.ts, j.qs, j.qs + j.qlen, i, r.rid, (int32_t)r.rev, 0});
                    jobs.push_back(j);
                }
            }
            if (!jobs.empty()) {
                const int nj = (int)jobs.size();
                for (int i = 0; i < n; ++i) sr_base[(size_t)i + 1] = sr_base[(size_t)i] + (int32_t)rs[i].pending.size();
                if (SL.pool_jobs.ensure((size_t)nj * sizeof(ExtJob) + 16) || SL.pool_sregs.ensure((size_t)nj * sizeof(StitchReg) + 16) ||
                    SL.pool_pregs.ensure(64) || SL.pool_psum.ensure(64) || SL.pool_job_anchor.ensure(64) || SL.pool_used.ensure(128))
                    return -1;
                const unsigned long long cnt = (unsigned long long)nj;
                MPN_HIP_CHECK(hipMemcpyAsync(SL.pool_jobs.p, jobs.data(), (size_t)nj * sizeof(ExtJob), hipMemcpyHostToDevice, st));
                MPN_HIP_CHECK(hipMemcpyAsync(SL.pool_sregs.p, sregs.data(), (size_t)nj * sizeof(StitchReg), hipMemcpyHostToDevice, st));
                MPN_HIP_CHECK(hipMemcpyAsync(SL.pool_used.as<unsigned long long>() + 3, &cnt, 8, hipMemcpyHostToDevice, st));
                MPN_HIP_CHECK(stream_sync(st));
                DevRound dv;
                if (run_jobs(rv, opt, nj, d_seqs.p, d_off.p, d_len.p, dv, &hseqs, st)) return -1;
                if (!dv.compact) { if (SL.pool_compact.ensure(64)) return -1; dv.compact = SL.pool_compact.as<uint32_t>(); }
                const RoundDev rd{SL.pool_sregs.as<StitchReg>(), SL.pool_pregs.as<PlanReg>(), SL.pool_psum.as<PlanSum>(), SL.pool_job_anchor.as<int32_t>(), d_a.p};
                if (stitch_and_finish(idx, opt, rs, n, seq_len, dv, rd, sr_base, d_seqs.p, d_off.p, d_len.p, n_threads, st)) return -1;
            }
            // an extension that produced nothing yields no inversion hit (mm_align1_inv returns 0)
            for (int i = 0; i < n; ++i) {
                std::vector<Reg> &R = rs[i].regs;
                R.erase(std::remove_if(R.begin(), R.end(), [](const Reg &r) { return r.inv && !r.has_p; }), R.end());
            }
        }
    }
    wt.stop_into(g_stats[23]);
    // rank, MAPQ, text
    std::atomic<int64_t> n_aln(0);
    parallel_for(n, n_threads, [&](int i, int) {
        ReadState &S = rs[i];
        const int qlen = seq_len[i];
        if (!S.regs.empty()) {
            if (opt->with_cigar) {
                filter_regs(opt, qlen, S.regs);
                hit_sort(S.regs);
                set_parent(opt->mask_level, S.regs, opt->a * 2 + opt->b);
                select_sub(opt->pri_ratio, idx->k * 2, opt->best_n, S.regs);
                set_sam_pri(S.regs);
            }
            set_mapq(S.regs, opt->min_chain_score, opt->a, h.rep_len[i]);
        }
        n_aln += (int64_t)S.regs.size();
    }, 5);
    g_stats[6] += n_aln;
    wt.stop_into(g_stats[23]);
    return 0;
}

// Hits of a batch accumulated over the parts of a split index (minimap2 -I / --split-prefix): per read the hits of every
// part with target ids shifted to the concatenated target table, and the largest repetitive-seed span.
struct mpn_hits {
    int32_t n_reads = 0, n_parts = 0, k = 15;
    int32_t want_text = 1;   // 0: mpn_hits_finish will be asked for columns only: the CIGARs need not leave the GPU (mpn_hits_set_text)
    std::vector<std::vector<Reg>> regs;
    std::vector<int32_t> rep_len;
    std::vector<std::string> names;
    std::vector<int32_t> lens;
};

// Threads of the host pool: what the caller asks for, else the cores this process may actually use -- the container's CPU
// quota (cgroup v2 cpu.max / v1 cfs quota) counts, not just the visible cores: with 256 visible cores and a quota of 16
// a pool sized by the visible cores is throttled -- capped at 32.
static int default_host_threads(const mpn_map_opt *opt) {
    if (const char *e = getenv("MPN_HOST_THREADS")) return std::max(1, atoi(e));
    if (opt->host_threads > 0) return opt->host_threads;
    static const int detected = []() {
        long long cores = std::max(1, (int)std::thread::hardware_concurrency());
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            if (fscanf(f, "%63s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
        }
        if (quota > 0 && period > 0) cores = std::min(cores, std::max(1LL, (quota + period - 1) / period));
        // one process per GPU: the ranks of a node share its cores (bench.py / the launcher export the rank count)
        int ranks = 1;
        if (const char *e = getenv("MPN_RANKS_ON_NODE")) ranks = std::max(1, atoi(e));
        else if (const char *e2 = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e2));
        // twice this rank's share of the cores, at most 32: the pool's threads mostly serve short bursts between GPU waits, and a
        // burst waiting for a free pool thread costs more than the quota's throttling (measured on a 16-CPU quota, one rank:
        // 16 threads 84.7, 24: 86.4, 32: 88.2, 48: 84.1 Gbp/min); never fewer than 4 (a rank's host phases need ~4 cores)
        const int n = (int)std::max(4LL, std::min(32LL, 2 * cores / ranks));
        return n;
    }();
    return detected;
}

// The mapping itself: every read's final hits against every one of n_parts RESIDENT indexes -> rs[part], rep_len[part] (no text).
// One index is the plain case.  Several are the parts of a target set that exceeds one index (minimap2 -I): the work items of
// the pipeline are (sub-batch, part) pairs, so the reads go up once, and the pipeline neither drains nor refills between parts.
static int map_batch_core(const mpn_index *const *parts, int n_parts, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                          const int64_t *seq_off, const int32_t *seq_len, const void *r_seqs, const int64_t *r_off, const int32_t *r_len,
                          std::vector<std::vector<ReadState>> &rs, std::vector<std::vector<int32_t>> &rep_len, int *n_threads_out) {
    hipStream_t st0 = 0;
    struct Borrowed {  // device views of the reads: owned uploads or the caller's resident buffers
        DevBuf<uint8_t> seqs; DevBuf<int64_t> off; DevBuf<int32_t> len;
        const uint8_t *p = nullptr; const int64_t *po = nullptr; const int32_t *pl = nullptr;
    } dv;
    int64_t bases = 0;
    WallTimer wt;
    if (r_seqs && r_off && r_len) {
        dv.p = (const uint8_t *)r_seqs; dv.po = r_off; dv.pl = r_len;
        for (int i = 0; i < n; ++i) bases += seq_len[i];
    } else {
        if (upload_seqs(n, seqs, seq_off, seq_len, dv.seqs, dv.off, dv.len, &bases, st0)) return -1;
        dv.p = dv.seqs.p; dv.po = dv.off.p; dv.pl = dv.len.p;
    }
    // per read: the x31 hash of its name (what minimap2 mixes into the order of equal-scoring chains), for hit_select_kernel
    DevBuf<uint32_t> d_name_hash;
    {
        std::vector<uint32_t> nh((size_t)n, 0u);
        if (names) for (int i = 0; i < n; ++i) nh[(size_t)i] = names[i] ? x31_hash(names[i]) : 0u;
        if (d_name_hash.upload(nh.data(), (size_t)n, st0)) return -1;
        MPN_HIP_CHECK(hipStreamSynchronize(st0));
    }
    MPN_HIP_CHECK(hipStreamSynchronize(st0));
    wt.stop_into(g_stats[16]);
    int n_threads = default_host_threads(opt);
    *n_threads_out = n_threads;
    g_pool.ensure(n_threads);
    // sub-batches of ~32 Mbp run through a small pool of workers (12 by default), each with its own HIP streams and
    // device arena, so that the host phases of one sub-batch overlap the GPU phases of the others and the
    // latency-bound kernels (chain DP, long extensions) of one overlap the throughput-bound ones of another
    int n_workers = 12;
    if (const char *e = getenv("MPN_PIPE_WORKERS")) n_workers = std::max(1, std::min(16, atoi(e)));
    // Sub-batches: equal-sized, their count a multiple of the worker count so that no worker idles in the last round.  Every worker
    // holds its own scratch (direction matrices above all): ~400 bytes per base of a sub-batch on a random target set, several times
    // that on a strain-rich one -- what the workers of the previous calls took per sub-batch base is remembered.  When the free HBM
    // (plus what the slots hold) does not cover all workers at the default size -- several index parts resident -- the sub-batches
    // shrink (not below 11 Mbp) before workers are given up: 12 workers on 11 Mbp run 3 % faster than 6 on 22 Mbp (profiles/r04).
    static double obs_bytes_per_bp = 0.0;   // (the mapping calls take turns: g_call_mu)
    static int64_t obs_sub_bp = 0;
    const bool env_target = getenv("MPN_SUB_BATCH_BP") != nullptr;
    const int W_all = n_workers;
    std::vector<int> cut;
    auto make_cuts = [&](int64_t target) {
        cut.assign(1, 0);
        std::vector<int64_t> sizes;
        int64_t n_cut = std::max<int64_t>(1, (bases + target - 1) / target);
        if (n_cut > W_all) n_cut = (n_cut + W_all - 1) / W_all * W_all;
        for (int64_t i = 0; i < n_cut; ++i) sizes.push_back(std::max<int64_t>(1, (bases + n_cut - 1) / n_cut));
        size_t si = 0;
        int64_t acc = 0;
        for (int i = 0; i < n; ++i) {
            acc += seq_len[i];
            if (si + 1 < sizes.size() && acc >= sizes[si] && i + 1 < n) { cut.push_back(i + 1); acc = 0; ++si; }
        }
        cut.push_back(n);
    };
    int64_t target = 32000000;
    if (env_target) target = std::max<int64_t>(1000, atoll(getenv("MPN_SUB_BATCH_BP")));
    make_cuts(target);
    int n_sub = (int)cut.size() - 1;
    int64_t largest = n_sub > 0 ? (bases + n_sub - 1) / n_sub : bases;
    {
        size_t free_b = 0, total_b = 0, held = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            for (const Slot &S : g_slots) held += S.device_bytes();
            const double avail = 0.9 * (double)(free_b + held);
            // (no call has been observed yet: the need of a strain-rich multi-part target set, 850 bytes per base, is assumed -- a first call
            // sized for a random target set ran out of memory on it, shed workers and took five times a steady-state call)
            const double bytes_per_bp = obs_bytes_per_bp > 0 ? std::max(400.0, 1.15 * obs_bytes_per_bp) : 900.0;
            auto per_worker = [&](int64_t sub_bp) { return bytes_per_bp * (double)std::max<int64_t>(sub_bp, 1) + 2e9; };
            int fit = (int)std::max(1.0, avail / per_worker(largest));
            if (fit < std::min(W_all, n_sub) && !env_target && largest > 11000000) {
                // what fits W_all workers, at least 11 Mbp
                const double room = avail / W_all - 2e9;
                const int64_t t2 = std::max<int64_t>(11000000, (int64_t)(room / bytes_per_bp));
                if (t2 < largest) {
                    make_cuts(t2);
                    n_sub = (int)cut.size() - 1;
                    largest = n_sub > 0 ? (bases + n_sub - 1) / n_sub : bases;
                    fit = (int)std::max(1.0, avail / per_worker(largest));
                }
            }
            // pools that were grown for much larger sub-batches would keep the memory the smaller ones free: give them back once
            if (obs_sub_bp > 0 && largest < obs_sub_bp * 3 / 4) {
                MPN_HIP_CHECK(hipDeviceSynchronize());
                for (Slot &S : g_slots) S.release_device();
            }
            n_workers = std::min(n_workers, fit);
        }
    }
    const int n_items = n_sub;   // a work item is a sub-batch: its worker maps it against every part in turn, on ONE sketch of its reads
    n_workers = std::max(1, std::min(n_workers, n_items));
    // slots that idle in this call give their scratch back: the running workers may need it (a call with fewer workers than the
    // last one -- less free memory since another index part became resident, or MPN_PIPE_WORKERS lowered)
    for (int wdx = n_workers; wdx < 16; ++wdx)
        if (g_slots[wdx].device_bytes()) { MPN_HIP_CHECK(hipDeviceSynchronize()); g_slots[wdx].release_device(); }
    int dev = 0;
    MPN_HIP_CHECK(hipGetDevice(&dev));
    rs.assign((size_t)n_parts, std::vector<ReadState>());
    rep_len.assign((size_t)n_parts, std::vector<int32_t>());
    for (int p = 0; p < n_parts; ++p) { rs[(size_t)p].assign((size_t)n, ReadState()); rep_len[(size_t)p].assign((size_t)n, 0); }
    std::atomic<int> next(0), failed(0);
    std::mutex mu;
    std::deque<int> retry;          // sub-batches given back by a worker that ran out of device memory (guarded by mu)
    int live_workers = n_workers, live_workers_busy = 0, n_shed = 0;   // (guarded by mu)
    int64_t tot_stats[MPN_NSTATS] = {0};
    std::string err;
    const bool dbg_workers = getenv("MPN_DEBUG_WORKERS") != nullptr;
    g_phase_log.on = getenv("MPN_DEBUG_PHASES") != nullptr;
    g_cpu_on = getenv("MPN_DEBUG_CPU") != nullptr;
    if (g_cpu_on) { for (auto &c : g_cpu_ns) c = 0; for (auto &c : g_worker_cpu_ns) c = 0; }
    g_worker_cpu_on = g_cpu_on;
    if (g_phase_log.on) { g_phase_log.recs.clear(); g_phase_log.origin = std::chrono::steady_clock::now(); }
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    auto worker = [&](int wid) {
        pthread_setname_np(pthread_self(), "mpn-work");
        if (hipSetDevice(dev) != hipSuccess) { failed = 1; return; }
        Slot &S = g_slots[wid];
        if (!S.st && hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking) != hipSuccess) { failed = 1; return; }
        tl_slot = &S;
        tl_arena = &S.arena;
        tl_worker_id = wid;
        memset(g_stats, 0, sizeof(g_stats));
        for (;;) {
            int sb = -1;
            { std::lock_guard<std::mutex> g(mu); if (!retry.empty()) { sb = retry.front(); retry.pop_front(); } }
            if (sb < 0) sb = next.fetch_add(1);
            if (sb >= n_items || failed) {
                // (a sub-batch may still come back from a worker that is shedding: wait for those before leaving)
                std::unique_lock<std::mutex> lk(mu);
                if (failed || (retry.empty() && live_workers_busy == 0)) break;
                lk.unlock();
                timespec ts{0, 200000};
                nanosleep(&ts, nullptr);
                continue;
            }
            { std::lock_guard<std::mutex> g(mu); ++live_workers_busy; }
            struct Busy { std::mutex &m; int &c; ~Busy() { std::lock_guard<std::mutex> g(m); --c; } } busy_{mu, live_workers_busy};
            S.arena.reset();
            tl_oom = false;
            int64_t stats_before[MPN_NSTATS];
            memcpy(stats_before, g_stats, sizeof(stats_before));
            const double t_in = since();
            struct Out { bool on; int wid, sb; double t_in; decltype(since) &f; ~Out() { if (on) fprintf(stderr, "[worker %d] sub-batch %d: %.1f -> %.1f ms\n", wid, sb, t_in, f()); } } out_{dbg_workers, wid, sb, t_in, since};
            int rc_item = 0;
            {
                ReadSketch sk;
                const int sbi = sb, nr = cut[sbi + 1] - cut[sbi];
                if (n_parts > 1 && nr > 0) {
                    WallTimer wts;
                    rc_item = sketch_reads(parts[0]->k, parts[0]->w, nr, dv.p, dv.po + cut[sbi], dv.pl + cut[sbi], seq_len + cut[sbi], sk, S.st);
                    wts.stop_into(g_stats[17]);
                }
                const Arena::Mark after_sketch = S.arena.mark();
                for (int prt = 0; prt < n_parts && !rc_item; ++prt) {
                    S.arena.rewind(after_sketch);   // (what the previous part took is dead; the sketch stays)
                    rc_item = map_range(parts[prt], opt, names, seqs, seq_off, seq_len, dv.p, dv.po, dv.pl, d_name_hash.p, cut[sbi], cut[sbi + 1], n_threads, S.st,
                                        rs[(size_t)prt], rep_len[(size_t)prt], sk.valid ? &sk : nullptr);
                }
            }
            if (rc_item) {
                std::lock_guard<std::mutex> g(mu);
                // Out of device memory: the workers' scratch grows with what the target set throws at them (a strain-rich index
                // yields tens of times the anchors of a random one), and the up-front estimate can be too low.  This worker gives
                // its memory back and leaves; its sub-batch goes to the others.  The last worker standing reports the failure.
                if (tl_oom && live_workers > 1) {   // (the flag of this thread's failed device allocation, not the error text)
                    --live_workers;
                    ++n_shed;
                    memcpy(g_stats, stats_before, sizeof(stats_before));
                    (void)hipGetLastError();
                    (void)hipStreamSynchronize(S.st);
                    S.release_device();
                    retry.push_back(sb);
                    if (dbg_workers) fprintf(stderr, "[worker %d] out of device memory at sub-batch %d: leaving, %d workers go on\n", wid, sb, live_workers);
                    break;
                }
                err = get_error();
                failed = 1;
                break;
            }
        }
        tl_arena = nullptr;
        std::lock_guard<std::mutex> g(mu);
        for (int k = 0; k < MPN_NSTATS; ++k) tot_stats[k] += g_stats[k];
    };
    {
        std::vector<std::thread> th;
        for (int wdx = 0; wdx < n_workers; ++wdx) th.emplace_back(worker, wdx);
        for (auto &t : th) t.join();
    }
    if (dbg_workers) {
        fprintf(stderr, "[call] workers joined at %.1f ms\n", since());
        for (int wdx = 0; wdx < n_workers; ++wdx) {
            const Slot &S = g_slots[wdx];
            size_t arena = 0;
            for (const auto &c : S.arena.chunks) arena += c.cap;
            const size_t pools = S.pool_jobs.cap + S.pool_P.cap + S.pool_P2.cap + S.pool_OFF.cap + S.pool_order.cap + S.pool_state.cap + S.pool_CIG.cap +
                                 S.pool_res.cap + S.pool_redo.cap + S.pool_compact.cap + S.pool_used.cap;
            const size_t pinned = S.pin_order.cap + S.pin_res.cap + S.pin_chain_u.cap + S.pin_chain_b.cap + S.pin_segs.cap + S.pin_pregs.cap + S.pin_fin_cig.cap + S.pin_fin_out.cap;
            fprintf(stderr, "[slot %d] arena %.2f GB, pools %.2f GB (P %.2f, CIG %.2f, compact %.2f), pinned host %.2f GB\n", wdx, arena / 1e9,
                    pools / 1e9, S.pool_P.cap / 1e9, S.pool_CIG.cap / 1e9, S.pool_compact.cap / 1e9, pinned / 1e9);
        }
    }
    if (g_phase_log.on) {
        for (const auto &r : g_phase_log.recs) fprintf(stderr, "[phase] %d %d %lld %lld\n", r.worker, r.slot, (long long)r.t0, (long long)r.t1);
        fprintf(stderr, "[phase-end]\n");
    }
    if (g_cpu_on) {
        static const char *const nm[] = {"other", "hits", "plan", "plan-copy", "stitch", "final", "dp-group-A", "strip-sort", "job-copy", "finish-copy", "accumulate", "merge"};
        fprintf(stderr, "[cpu] thread CPU ms in parallel regions:");
        for (int k = 0; k < 12; ++k) fprintf(stderr, " %s %.0f", nm[k], g_cpu_ns[k].load() / 1e6);
        fprintf(stderr, "\n[cpu] sections: stitch-append %.0f", g_cpu_ns[16].load() / 1e6);
        for (int k = 3; k < 16; ++k) if (g_cpu_ns[16 + k].load() > 500000) fprintf(stderr, " s%d %.0f", k, g_cpu_ns[16 + k].load() / 1e6);
        fprintf(stderr, "\n[cpu] worker-thread CPU ms by phase slot:");
        for (int k = 0; k < 64; ++k) if (g_worker_cpu_ns[k].load() > 500000) fprintf(stderr, " [%d] %.0f", k, g_worker_cpu_ns[k].load() / 1e6);
        fprintf(stderr, "\n");
    }
    if (failed) { set_error("%s", err.empty() ? "worker failed" : err.c_str()); return -1; }
    {
        const int64_t h2d = g_stats[16];
        memcpy(g_stats, tot_stats, sizeof(tot_stats));
        g_stats[16] = h2d;
        g_stats[0] = bases;
        g_stats[60] = n_shed; g_stats[61] = n_workers;
        {   // what a worker took per sub-batch base in this call (the next call sizes its sub-batches and workers by it)
            size_t held_max = 0;
            for (const Slot &S : g_slots) held_max = std::max(held_max, S.device_bytes());
            if (largest > 0 && held_max > 0) {
                const double r = (double)held_max / (double)largest;
                obs_bytes_per_bp = obs_sub_bp == largest ? std::max(obs_bytes_per_bp, r) : r;
                obs_sub_bp = largest;
            }
        }
    }
    return 0;
}

// Text and columns of a batch whose hits are final.  opt->out_sam: 0 = PAF into text, 1 = SAM into text, 2 = PAF into text and
// SAM kept for mpn_map_fetch_sam.  Results that do not fit the caller's buffers are kept (mpn_map_fetch_cols / _text)
// so that the caller fetches them with larger buffers instead of mapping the batch again.  Returns bytes of text or -3.
static struct Kept { std::vector<int32_t> cols; int64_t n_rows = -1; std::string text, sam; bool has_text = false, has_sam = false; } g_kept;

static int64_t emit_batch(const Targets tg, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs, const char *quals,
                          const int64_t *seq_off, const int32_t *seq_len, std::vector<std::vector<Reg>*> &regs, const std::vector<int32_t> &rep_len,
                          int n_threads, char *text, int64_t text_cap, mpn_aln_cols *cols) {
    const bool want_paf = text && opt->out_sam != 1, want_sam = opt->out_sam == 2 || (text && opt->out_sam == 1);
    std::vector<std::string> paf_lines(want_paf ? n : 0), sam_lines(want_sam ? n : 0);
    int64_t n_aln = 0;
    for (int i = 0; i < n; ++i) n_aln += (int64_t)regs[i]->size();
    if (want_paf || want_sam)
        parallel_for(n, n_threads, [&](int i, int) {
            const char *nm = names && names[i] ? names[i] : "*";
            if (want_paf && !regs[i]->empty()) write_paf(tg, opt, nm, seq_len[i], *regs[i], rep_len[i], paf_lines[i]);
            if (want_sam) write_sam(tg, nm, seq_len[i], seqs + seq_off[i], quals ? quals + seq_off[i] : nullptr, *regs[i], rep_len[i], sam_lines[i]);
        }, 0, 16);   // (text is heavy per read)
    bool short_cols = false, short_text = false;
    g_kept.cols.clear(); g_kept.text.clear(); g_kept.sam.clear(); g_kept.n_rows = -1; g_kept.has_text = g_kept.has_sam = false;
    if (cols) {
        cols->n_rows = n_aln;
        short_cols = n_aln > cols->cap;
        int32_t *dst[13] = {cols->read_idx, cols->qs, cols->qe, cols->rev, cols->rid, cols->rs, cols->re, cols->mlen, cols->blen, cols->mapq,
                            cols->nm, cols->as, cols->primary};
        if (short_cols) {
            g_kept.cols.resize((size_t)n_aln * 13);
            g_kept.n_rows = n_aln;
            for (int c = 0; c < 13; ++c) dst[c] = g_kept.cols.data() + (size_t)c * (size_t)n_aln;
        }
        int64_t k = 0;
        for (int i = 0; i < n; ++i)
            for (const Reg &r : *regs[i]) {
                dst[0][k] = i; dst[1][k] = r.qs; dst[2][k] = r.qe; dst[3][k] = (int32_t)r.rev; dst[4][k] = r.rid;
                dst[5][k] = r.rs; dst[6][k] = r.re; dst[7][k] = r.mlen; dst[8][k] = r.blen; dst[9][k] = (int32_t)r.mapq;
                dst[10][k] = r.has_p ? r.blen - r.mlen + r.n_ambi : -1;
                dst[11][k] = r.has_p ? r.dp_score : -1;
                dst[12][k] = r.id == r.parent;
                ++k;
            }
    }
    int64_t w = 0;
    if (text) {
        std::vector<std::string> &lines = opt->out_sam == 1 ? sam_lines : paf_lines;
        int64_t tot = 0;
        for (auto &l : lines) tot += (int64_t)l.size();
        short_text = tot + 1 > text_cap;
        if (short_text) {
            g_kept.text.reserve((size_t)tot);
            for (auto &l : lines) g_kept.text += l;
            g_kept.has_text = true;
        } else {
            for (auto &l : lines) { memcpy(text + w, l.data(), l.size()); w += (int64_t)l.size(); }
            text[w] = 0;
        }
    }
    if (opt->out_sam == 2) {
        size_t tot = 0;
        for (auto &l : sam_lines) tot += l.size();
        g_kept.sam.reserve(tot);
        for (auto &l : sam_lines) g_kept.sam += l;
        g_kept.has_sam = true;
    }
    parallel_chunks(n, n_threads, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) { if (want_paf) std::string().swap(paf_lines[i]); if (want_sam) std::string().swap(sam_lines[i]); }
    });
    g_stats[6] = n_aln;
    return (short_cols || short_text) ? -3 : w;
}

static std::mutex g_call_mu;  // the worker slots (streams, arenas, pools) are process-wide: concurrent callers take turns

extern "C" int64_t mpn_map_batch_q(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *const *names,
                                   const char *seqs, const char *quals, const int64_t *seq_off, const int32_t *seq_len, const void *r_seqs,
                                   const int64_t *r_off, const int32_t *r_len, char *paf, int64_t paf_cap, mpn_aln_cols *cols) {
    std::lock_guard<std::mutex> call_guard(g_call_mu);
    memset(g_stats, 0, sizeof(g_stats));
    if (cols) cols->n_rows = 0;
    if (n <= 0) { if (paf && paf_cap > 0) paf[0] = 0; g_kept = Kept(); if (opt->out_sam == 2) g_kept.has_sam = true; return 0; }
    WallTimer whole;
    std::vector<std::vector<ReadState>> rs_p;
    std::vector<std::vector<int32_t>> rep_p;
    int n_threads = 1;
    g_need_cigar = paf != nullptr || opt->out_sam != 0;   // columns only: the CIGARs never leave the GPU
    if (map_batch_core(&idx, 1, opt, n, names, seqs, seq_off, seq_len, r_seqs, r_off, r_len, rs_p, rep_p, &n_threads)) return -1;
    std::vector<ReadState> &rs = rs_p[0];
    std::vector<int32_t> &rep_len = rep_p[0];
    std::vector<std::vector<Reg>*> regs((size_t)n);
    for (int i = 0; i < n; ++i) regs[(size_t)i] = &rs[(size_t)i].regs;
    const int64_t w = emit_batch(Targets{&idx->names, &idx->lens}, opt, n, names, seqs, quals, seq_off, seq_len, regs, rep_len, n_threads, paf, paf_cap, cols);
    // the per-read state (regs with their CIGARs) is a few hundred thousand small allocations: free them in parallel
    parallel_chunks(n, n_threads, [&](int64_t lo, int64_t hi, int) { for (int64_t i = lo; i < hi; ++i) rs[(size_t)i] = ReadState(); });
    whole.stop_into(g_stats[24]);
    return w;
}

extern "C" int64_t mpn_map_batch_ex(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *const *names,
                                    const char *seqs, const int64_t *seq_off, const int32_t *seq_len, const void *r_seqs,
                                    const int64_t *r_off, const int32_t *r_len, char *paf, int64_t paf_cap, mpn_aln_cols *cols) {
    return mpn_map_batch_q(idx, opt, n, names, seqs, nullptr, seq_off, seq_len, r_seqs, r_off, r_len, paf, paf_cap, cols);
}

// ---- split index (minimap2 -I parts + --split-prefix merge) ---------------------------------------------------------
extern "C" mpn_hits *mpn_hits_create(int32_t n_reads) {
    if (n_reads < 0) { set_error("mpn_hits_create: negative read count"); return nullptr; }
    mpn_hits *h = new mpn_hits();
    h->n_reads = n_reads;
    h->regs.resize((size_t)n_reads);
    h->rep_len.assign((size_t)n_reads, 0);
    return h;
}
extern "C" void mpn_hits_destroy(mpn_hits *h) { delete h; }
extern "C" void mpn_hits_set_text(mpn_hits *h, int32_t want_text) { if (h) h->want_text = want_text ? 1 : 0; }
extern "C" int32_t mpn_hits_n_seq(const mpn_hits *h) { return (int32_t)h->lens.size(); }
extern "C" int32_t mpn_hits_n_parts(const mpn_hits *h) { return h->n_parts; }
extern "C" int32_t mpn_hits_seq_len(const mpn_hits *h, int32_t i) { return i >= 0 && (size_t)i < h->lens.size() ? h->lens[(size_t)i] : -1; }
extern "C" int32_t mpn_hits_seq_name(const mpn_hits *h, int32_t i, char *buf, int32_t cap) {
    if (i < 0 || (size_t)i >= h->names.size() || !buf || cap <= 0) return -1;
    const std::string &nm = h->names[(size_t)i];
    const int32_t l = (int32_t)std::min<size_t>(nm.size(), (size_t)cap - 1);
    memcpy(buf, nm.data(), (size_t)l);
    buf[l] = 0;
    return (int32_t)nm.size();
}

// several RESIDENT parts in one call (288 GB of HBM hold what a CPU host streams through -I): the pipeline's work items are
// (sub-batch, part) pairs; the hits land in the accumulator part by part, in the order given, as repeated
// mpn_map_batch_part calls would leave them
extern "C" int mpn_map_batch_parts(const mpn_index *const *parts, int32_t n_parts, const mpn_map_opt *opt, int32_t n, const char *const *names,
                                   const char *seqs, const int64_t *seq_off, const int32_t *seq_len, const void *r_seqs, const int64_t *r_off,
                                   const int32_t *r_len, mpn_hits *acc) {
    if (!acc || n != acc->n_reads) { set_error("mpn_map_batch_parts: the accumulator was created for another batch"); return -1; }
    if (!parts || n_parts <= 0) { set_error("mpn_map_batch_parts: no index part"); return -1; }
    for (int p = 0; p < n_parts; ++p)
        if (!parts[p] || parts[p]->k != parts[0]->k) { set_error("mpn_map_batch_parts: null part or parts built with different k"); return -1; }
    std::lock_guard<std::mutex> call_guard(g_call_mu);
    memset(g_stats, 0, sizeof(g_stats));
    if (n > 0) {
        std::vector<std::vector<ReadState>> rs;
        std::vector<std::vector<int32_t>> rep_len;
        int n_threads = 1;
        g_need_cigar = acc->want_text != 0;   // (text is asked for at mpn_hits_finish: the accumulator says whether it will be)
        if (map_batch_core(parts, n_parts, opt, n, names, seqs, seq_off, seq_len, r_seqs, r_off, r_len, rs, rep_len, &n_threads)) return -1;
        std::vector<int32_t> rid0((size_t)n_parts);
        int32_t r0 = (int32_t)acc->lens.size();
        for (int p = 0; p < n_parts; ++p) { rid0[(size_t)p] = r0; r0 += parts[p]->n_seq; }
        parallel_chunks(n, n_threads, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                size_t more = 0;
                for (int p = 0; p < n_parts; ++p) more += rs[(size_t)p][(size_t)i].regs.size();
                acc->regs[(size_t)i].reserve(acc->regs[(size_t)i].size() + more);
                for (int p = 0; p < n_parts; ++p) {
                    for (Reg &r : rs[(size_t)p][(size_t)i].regs) { r.rid += rid0[(size_t)p]; acc->regs[(size_t)i].push_back(std::move(r)); }
                    acc->rep_len[(size_t)i] = std::max(acc->rep_len[(size_t)i], rep_len[(size_t)p][(size_t)i]);
                    rs[(size_t)p][(size_t)i] = ReadState();
                }
            }
        }, 10);
    }
    for (int p = 0; p < n_parts; ++p) {
        acc->names.insert(acc->names.end(), parts[p]->names.begin(), parts[p]->names.end());
        acc->lens.insert(acc->lens.end(), parts[p]->lens.begin(), parts[p]->lens.end());
        acc->k = parts[p]->k;
        ++acc->n_parts;
    }
    return 0;
}

extern "C" int mpn_map_batch_part(const mpn_index *part, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                                  const int64_t *seq_off, const int32_t *seq_len, const void *r_seqs, const int64_t *r_off,
                                  const int32_t *r_len, mpn_hits *acc) {
    return mpn_map_batch_parts(&part, 1, opt, n, names, seqs, seq_off, seq_len, r_seqs, r_off, r_len, acc);
}

// minimap2's merge of the per-part hits of a read (mm_split_merge): the sub-optimal bookkeeping is reset, the hits are
// ranked, grouped and trimmed again over all parts, and MAPQ is recomputed
static void merge_regs(const mpn_map_opt *opt, int k, std::vector<Reg> &regs, int32_t rep_len) {
    if (regs.empty()) return;
    for (Reg &r : regs) { r.subsc = 0; r.n_sub = 0; if (r.has_p) r.dp_max2 = 0; }
    hit_sort(regs);
    set_parent(opt->mask_level, regs, opt->a * 2 + opt->b);
    select_sub(opt->pri_ratio, k * 2, opt->best_n, regs);
    set_sam_pri(regs);
    set_mapq(regs, opt->min_chain_score, opt->a, rep_len);
}

extern "C" int64_t mpn_hits_finish(mpn_hits *acc, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                                   const char *quals, const int64_t *seq_off, const int32_t *seq_len, char *paf, int64_t paf_cap,
                                   mpn_aln_cols *cols) {
    if (!acc || n != acc->n_reads) { set_error("mpn_hits_finish: the accumulator was created for another batch"); return -1; }
    std::lock_guard<std::mutex> call_guard(g_call_mu);
    if (cols) cols->n_rows = 0;
    if (n <= 0) { if (paf && paf_cap > 0) paf[0] = 0; g_kept = Kept(); if (opt->out_sam == 2) g_kept.has_sam = true; return 0; }
    const int n_threads = default_host_threads(opt);
    g_pool.ensure(n_threads);
    if (!acc->n_parts) { set_error("mpn_hits_finish: no part was mapped"); return -1; }
    if (!acc->want_text && (paf || opt->out_sam != 0)) { set_error("mpn_hits_finish: text asked of an accumulator that was told there would be none (mpn_hits_set_text)"); return -1; }
    if (g_cpu_on) { g_cpu_ns[11] = 0; g_cpu_ns[0] = 0; }
    struct Report { ~Report() { if (g_cpu_on) fprintf(stderr, "[cpu] finish: merge %.0f ms, text/other %.0f ms (thread CPU in parallel regions)\n", g_cpu_ns[11].load() / 1e6, g_cpu_ns[0].load() / 1e6); } } report_;
    parallel_for(n, n_threads, [&](int i, int) { merge_regs(opt, acc->k, acc->regs[(size_t)i], acc->rep_len[(size_t)i]); }, 11, 128);
    std::vector<std::vector<Reg>*> regs((size_t)n);
    for (int i = 0; i < n; ++i) regs[(size_t)i] = &acc->regs[(size_t)i];
    return emit_batch(Targets{&acc->names, &acc->lens}, opt, n, names, seqs, quals, seq_off, seq_len, regs, acc->rep_len, n_threads, paf, paf_cap, cols);
}

extern "C" int64_t mpn_map_fetch_cols(mpn_aln_cols *cols) {
    if (!cols || g_kept.n_rows < 0) { set_error("mpn_map_fetch_cols: no columns kept from the last call"); return -1; }
    cols->n_rows = g_kept.n_rows;
    if (cols->cap < g_kept.n_rows) return -3;
    int32_t *dst[13] = {cols->read_idx, cols->qs, cols->qe, cols->rev, cols->rid, cols->rs, cols->re, cols->mlen, cols->blen, cols->mapq,
                        cols->nm, cols->as, cols->primary};
    for (int c = 0; c < 13; ++c)
        if (g_kept.n_rows) memcpy(dst[c], g_kept.cols.data() + (size_t)c * (size_t)g_kept.n_rows, (size_t)g_kept.n_rows * 4);
    const int64_t r = g_kept.n_rows;
    std::vector<int32_t>().swap(g_kept.cols);
    g_kept.n_rows = -1;
    return r;
}

static int64_t fetch_kept(std::string &txt, bool &has, const char *what, char *buf, int64_t cap) {
    if (!has) { set_error("%s: no text kept from the last call", what); return -1; }
    const int64_t need = (int64_t)txt.size();
    if (!buf) return need + 1;  // size query
    if (cap < need + 1) return -3;
    memcpy(buf, txt.data(), (size_t)need);
    buf[need] = 0;
    std::string().swap(txt);
    has = false;
    return need;
}
extern "C" int64_t mpn_map_fetch_text(char *buf, int64_t cap) { return fetch_kept(g_kept.text, g_kept.has_text, "mpn_map_fetch_text", buf, cap); }
extern "C" int64_t mpn_map_fetch_sam(char *buf, int64_t cap) { return fetch_kept(g_kept.sam, g_kept.has_sam, "mpn_map_fetch_sam", buf, cap); }

static int64_t sam_header_of(const std::vector<std::string> &names, const std::vector<int32_t> &lens, const char *cmdline, char *buf, int64_t cap);

extern "C" int64_t mpn_sam_header(const mpn_index *idx, const char *cmdline, char *buf, int64_t cap) {
    return sam_header_of(idx->names, idx->lens, cmdline, buf, cap);
}
extern "C" int64_t mpn_hits_sam_header(const mpn_hits *h, const char *cmdline, char *buf, int64_t cap) {
    return sam_header_of(h->names, h->lens, cmdline, buf, cap);
}

static int64_t sam_header_of(const std::vector<std::string> &names, const std::vector<int32_t> &lens, const char *cmdline, char *buf, int64_t cap) {
    std::string h;
    char line[64];
    for (size_t i = 0; i < names.size(); ++i) {
        h += "@SQ\tSN:"; h += names[i];
        snprintf(line, sizeof(line), "\tLN:%d\n", lens[i]);
        h += line;
    }
    h += "@PG\tID:mpn-aligner\tPN:mpn-aligner\tVN:r02";
    if (cmdline && cmdline[0]) { h += "\tCL:"; h += cmdline; }
    h += '\n';
    if ((int64_t)h.size() + 1 > cap) return -3;
    memcpy(buf, h.data(), h.size());
    buf[h.size()] = 0;
    return (int64_t)h.size();
}

extern "C" int64_t mpn_map_batch(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *const *names, const char *seqs,
                                 const int64_t *seq_off, const int32_t *seq_len, char *paf, int64_t paf_cap) {
    return mpn_map_batch_ex(idx, opt, n, names, seqs, seq_off, seq_len, nullptr, nullptr, nullptr, paf, paf_cap, nullptr);
}

// stage entry point for the parity tests: the DP kernels on arbitrary (query, target) code pairs
extern "C" int mpn_ext_dp_batch(const mpn_map_opt *opt, int32_t n, const uint8_t *qcodes, const int64_t *q_off, const int32_t *q_len,
                                const uint8_t *tcodes, const int64_t *t_off, const int32_t *t_len, const int32_t *w,
                                const int32_t *zdrop, const int32_t *end_bonus, const int32_t *flag, int32_t force_kernel,
                                int32_t *out9, uint32_t *cigar_pool, int64_t cigar_cap, int64_t *cig_off) {
    hipStream_t st = 0;
    if (n <= 0) return 0;
    int64_t qtot = 0, ttot = 0;
    for (int i = 0; i < n; ++i) { qtot = std::max<int64_t>(qtot, q_off[i] + q_len[i]); ttot = std::max<int64_t>(ttot, t_off[i] + t_len[i]); }
    std::vector<uint8_t> ascii((size_t)qtot + 16, 'N');
    for (int64_t i = 0; i < qtot; ++i) ascii[(size_t)i] = (uint8_t)"ACGTN"[qcodes[i] > 4 ? 4 : qcodes[i]];
    DevBuf<uint8_t> d_reads;
    DevBuf<uint32_t> d_ref;
    DevBuf<int64_t> d_qoff, d_toff, d_ns, d_ne;
    std::vector<uint32_t> words;
    std::vector<int64_t> ns, ne;
    pack_2bit(tcodes, ttot, words, ns, ne);
    DevBuf<int32_t> d_qlen;
    if (d_reads.upload(ascii.data(), ascii.size(), st) || d_ref.upload(words.data(), words.size(), st) || d_ns.upload(ns.data(), ns.size(), st) ||
        d_ne.upload(ne.data(), ne.size(), st) || d_qoff.upload(q_off, n, st) ||
        d_qlen.upload(q_len, n, st) || d_toff.upload(t_off, n, st))
        return -1;
    std::vector<ExtJob> jobs(n);
    for (int i = 0; i < n; ++i) {
        ExtJob &j = jobs[i];
        memset(&j, 0, sizeof(j));
        j.read = i; j.rid = i; j.rev = 0; j.qs = 0; j.qlen = q_len[i]; j.ts = 0; j.tlen = t_len[i]; j.reversed = 0;
        j.w = w[i]; j.zdrop = zdrop[i]; j.end_bonus = end_bonus[i]; j.flag = flag[i];
    }
    g_force_kernel = force_kernel;
    // no second pass here: the caller asks for exactly one DP per pair
    mpn_map_opt o2 = *opt;
    o2.zdrop = 0x3fffffff;
    DevRound dv;
    for (ExtJob &j : jobs) { j.strip_s = 1; j.cls = -1; }
    if (tl_slot->pool_jobs.ensure((size_t)n * sizeof(ExtJob) + 16)) return -1;
    MPN_HIP_CHECK(hipMemcpy(tl_slot->pool_jobs.p, jobs.data(), (size_t)n * sizeof(ExtJob), hipMemcpyHostToDevice));
    {
        const unsigned long long cnt = (unsigned long long)n;
        if (tl_slot->pool_used.ensure(64)) return -1;
        MPN_HIP_CHECK(hipMemcpy(tl_slot->pool_used.as<unsigned long long>() + 3, &cnt, 8, hipMemcpyHostToDevice));
    }
    const int rc = run_jobs(RefView{d_ref.p, d_toff.p, d_ns.p, d_ne.p, (int32_t)ns.size()}, &o2, n, d_reads.p, d_qoff.p, d_qlen.p, dv, nullptr, st);
    g_force_kernel = 0;
    if (rc) return rc;
    // (the product keeps these on the device for the stitching kernel; the stage test reads them back)
    std::vector<ExtRes> res((size_t)n);
    unsigned long long h_used = 0;
    MPN_HIP_CHECK(hipMemcpy(res.data(), dv.res, (size_t)n * sizeof(ExtRes), hipMemcpyDeviceToHost));
    MPN_HIP_CHECK(hipMemcpy(&h_used, dv.used, 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> cig((size_t)h_used + 1);
    if (h_used) MPN_HIP_CHECK(hipMemcpy(cig.data(), dv.compact, (size_t)h_used * 4, hipMemcpyDeviceToHost));
    int64_t used = 0;
    for (int i = 0; i < n; ++i) {
        const ExtRes &e = res[i];
        int32_t *o = out9 + (size_t)i * 9;
        o[0] = e.max; o[1] = e.zdropped; o[2] = e.max_q; o[3] = e.max_t; o[4] = e.mqe; o[5] = e.mqe_t; o[6] = e.score; o[7] = e.reach_end;
        o[8] = e.n_cigar;
        cig_off[i] = used;
        if (used + e.n_cigar > cigar_cap) return -3;
        if (e.n_cigar) memcpy(cigar_pool + used, cig.data() + e.cig_pos, (size_t)e.n_cigar * 4);
        used += e.n_cigar;
    }
    return 0;
}
