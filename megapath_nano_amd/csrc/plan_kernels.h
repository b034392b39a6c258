// Planning of the base-level extension on the GPU: the first half of minimap2's mm_align1 (mm_fix_bad_ends, mm_filter_bad_seeds,
// the extension limits, the cut of a chain into gap-fill windows) for every hit of a round, followed by the choice of a DP kernel
// and of the direction-matrix layout for every window and by the launch lists.  The host hands over 40 bytes per hit and
// reads back a few counters; no job record is built, copied or sorted on the host.
//
//   plan_kernel<false / true>   one LANE per hit (the logic is a sequential walk over the hit's anchors with early exits): count
//                               its windows / write their job records + the hit's stitching record
//   job_classify_kernel         one lane per window: band width, strip / band / LDS kernel, matrix layout, scratch needs, list
//   job_layout_kernel           scratch offsets from the scans, scatter of the windows into their launch lists (strip lists
//                               bucketed by query length, so that the windows that share a wave need a similar number of steps)
#pragma once
#include "stitch_kernels.h"

namespace mpn {

constexpr uint64_t PK_SEED_LONG_JOIN = 1ULL << 40, PK_SEED_IGNORE = 1ULL << 41, PK_SEED_TANDEM = 1ULL << 42;

struct PlanOpt {
    int32_t bw, bw15, min_chain_score, max_gap, min_cnt, a, q, e, zdrop, zdrop_inv, end_bonus, min_ksw_len, k, pad;
    int64_t max_sw_mat;
};

__device__ __forceinline__ int32_t pk_x(const u128 &v) { return (int32_t)v.x; }
__device__ __forceinline__ int32_t pk_y(const u128 &v) { return (int32_t)v.y; }
__device__ __forceinline__ int32_t pk_span(const u128 &v) { return (int32_t)(v.y >> 32 & 0xff); }

// The read's anchors as the planning walks see them: the hit's own anchors [lo, hi) are staged in LDS by the wave (a walk is a
// chain of dependent loads: ~30 ns each from LDS, ~1 us from HBM), the few neighbours outside come from global memory.  Flags
// are written to both copies (the global one carries them to the next round).
struct AnchorView {
    u128 *g;          // the read's anchors in global memory
    u128 *l;          // LDS copy of [lo, hi)
    int lo, hi;
    const unsigned long long *kb;   // bit r: the gap between anchors lo + r - 1 and lo + r exceeds mm_filter_bad_seeds' min_gap (set by the wave)
    int kb_n;                       // anchors the bits cover (r < kb_n)
    __device__ __forceinline__ u128 operator[](int i) const { return (i >= lo && i < hi) ? l[i - lo] : g[i]; }
    __device__ __forceinline__ void or_y(int i, uint64_t f) { g[i].y |= f; if (i >= lo && i < hi) l[i - lo].y |= f; }
};

// mm_filter_bad_seeds without its index array: K = the anchors whose gap to their predecessor exceeds min_gap is walked as a
// virtual sequence (an entry is found by scanning forward), K-indices and anchor indices are tracked side by side
__device__ inline void pk_filter_bad_seeds(int as1, int cnt1, AnchorView &a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) {
    auto gap_at = [&](int i) { return (pk_y(a[as1 + i]) - pk_y(a[as1 + i - 1])) - (pk_x(a[as1 + i]) - pk_x(a[as1 + i - 1])); };
    auto is_k = [&](int i) { const int g = gap_at(i); return g < -min_gap || g > min_gap; };
    // anchor index of the next K entry (cnt1: none): a scan of the bits the wave prepared (a walk over the anchors beyond them)
    const int off = as1 - a.lo;
    auto next_k = [&](int i) {
        ++i;
        int r = i + off;
        while (r < a.kb_n && i < cnt1) {
            const unsigned long long w = a.kb[r >> 6] >> (r & 63);
            if (w) { const int s_ = __builtin_ctzll(w); return i + s_ < cnt1 ? i + s_ : cnt1; }
            const int step = 64 - (r & 63);
            r += step; i += step;
        }
        for (; i < cnt1; ++i) if (is_k(i)) return i;
        return cnt1;
    };
    int first = next_k(0);
    if (first >= cnt1 || next_k(first) >= cnt1) return;   // fewer than two entries
    int max = 0, max_st_k = -1, max_en_k = -1, max_st_i = -1, max_en_i = -1;
    int ik = first;   // anchor index of K[k]
    for (int k = 0;; ++k) {
        const bool at_end = ik >= cnt1;
        if (at_end || k >= max_en_k) {
            if (max_en_k > 0) for (int i = max_st_i; i < max_en_i; ++i) a.or_y(as1 + i, PK_SEED_IGNORE);
            max = 0; max_st_k = max_en_k = -1;
            if (at_end) break;
        }
        int gap = gap_at(ik), n_ins = 0, n_del = 0, max_diff = 0, max_diff_l = -1, max_diff_i = -1;
        if (gap > 0) n_ins += gap; else n_del += -gap;
        const int qs = pk_y(a[as1 + ik - 1]), rs = pk_x(a[as1 + ik - 1]);
        int jl = ik;
        for (int l = k + 1; l <= k + max_ext_cnt; ++l) {
            jl = next_k(jl);
            if (jl >= cnt1) break;
            if (pk_y(a[as1 + jl]) - qs > max_ext_len || pk_x(a[as1 + jl]) - rs > max_ext_len) break;
            gap = gap_at(jl);
            if (gap > 0) n_ins += gap; else n_del += -gap;
            const int d = n_ins - n_del, diff = n_ins + n_del - (d < 0 ? -d : d);
            if (max_diff < diff) { max_diff = diff; max_diff_l = l; max_diff_i = jl; }
        }
        if (max_diff > diff_thres && max_diff > max) { max = max_diff; max_st_k = k; max_en_k = max_diff_l; max_st_i = ik; max_en_i = max_diff_i; }
        ik = next_k(ik);
    }
    (void)max_st_k;
}

// minimap2's mm_squeeze_a on the device: the chains that survived hit selection are gathered, in the order of their first
// anchors, into the read's squeezed anchor list (the planning walks look at a hit's neighbours in THAT list).  A wave per
// segment and pass; the first anchor of a chain that join_long attached to its predecessor carries SEED_LONG_JOIN.
__global__ __launch_bounds__(256) void anchor_squeeze_kernel(const SqueezeSeg *__restrict__ segs, int n_segs, const u128 *__restrict__ src,
                                                             const int64_t *__restrict__ sq_off, u128 *__restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int s = wave; s < n_segs; s += n_waves) {
        const SqueezeSeg sg = segs[s];
        const u128 *from = src + sg.src;
        u128 *to = dst + sq_off[sg.read] + sg.dst;
        for (int j = lane; j < sg.cnt; j += 64) {
            u128 v = from[j];
            if (j == 0 && sg.flag) v.y |= PK_SEED_LONG_JOIN;
            to[j] = v;
        }
    }
}

// The planning half of mm_align1 for one hit.  FILL = false counts the windows; FILL = true writes them (jobs, job_anchor) and
// the hit's stitching record.  The counting pass sets the SEED_IGNORE flags (the filter is the most expensive walk), the other reads them.
template <bool FILL>
__device__ inline int plan_hit(const PlanOpt &o, const PlanReg &pr, AnchorView &a, const int32_t *__restrict__ tlens, int32_t ri, int32_t first_job,
                               ExtJob *__restrict__ jobs, int32_t *__restrict__ job_anchor, StitchReg *__restrict__ sregs, PlanSum *__restrict__ psum) {
    const int32_t rid = (int32_t)(a[pr.as].x << 1 >> 33), rev = (int32_t)(a[pr.as].x >> 63);
    const int32_t tlen_all = tlens[rid], kh = o.k >> 1, qlen = pr.qlen;
    const int bw = o.bw15;
    int nj = 0;
    auto emit = [&](int qs_, int ql, int ts_, int tl, int reversed, int w, int zdrop, int end_bonus, int flag, int anchor_i) {
        if constexpr (FILL) {
            ExtJob j;
            j.read = pr.read; j.rid = rid; j.rev = rev; j.qs = qs_; j.qlen = ql; j.ts = ts_; j.tlen = tl; j.reversed = reversed;
            j.w = w; j.zdrop = zdrop; j.end_bonus = end_bonus; j.flag = flag;
            j.p_off = 0; j.row_off = 0; j.cig_off = 0; j.n_col = 0; j.state_mode = 0; j.state_off = 0; j.layout = 0; j.qstride = 0; j.strip_s = 1; j.cls = -1;
            jobs[first_job + nj] = j;
            job_anchor[first_job + nj] = anchor_i;
        }
        ++nj;
    };
    auto too_big = [&](int ql, int tl) { return o.max_sw_mat > 0 && (int64_t)tl * ql > o.max_sw_mat; };
    // ---- mm_fix_bad_ends ----
    int32_t as1 = pr.as, cnt1 = pr.cnt;
    if (pr.cnt >= 3) {
        const int min_match = o.min_chain_score * 2;
        int32_t m, l;
        m = l = pk_span(a[pr.as]);
        for (int i = pr.as + 1; i < pr.as + pr.cnt - 1; ++i) {
            const int32_t q_span = pk_span(a[i]);
            if (a[i].y & PK_SEED_LONG_JOIN) break;
            const int32_t lr = pk_x(a[i]) - pk_x(a[i - 1]), lq = pk_y(a[i]) - pk_y(a[i - 1]);
            const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
            if (mx - mn > l >> 1) as1 = i;
            l += mn;
            m += mn < q_span ? mn : q_span;
            if (l >= o.bw << 1 || (m >= min_match && m >= o.bw) || m >= pr.mlen >> 1) break;
        }
        cnt1 = pr.as + pr.cnt - as1;
        m = l = pk_span(a[pr.as + pr.cnt - 1]);
        for (int i = pr.as + pr.cnt - 2; i > as1; --i) {
            const int32_t q_span = pk_span(a[i + 1]);
            if (a[i + 1].y & PK_SEED_LONG_JOIN) break;
            const int32_t lr = pk_x(a[i + 1]) - pk_x(a[i]), lq = pk_y(a[i + 1]) - pk_y(a[i]);
            const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
            if (mx - mn > l >> 1) cnt1 = i + 1 - as1;
            l += mn;
            m += mn < q_span ? mn : q_span;
            if (l >= o.bw << 1 || (m >= min_match && m >= o.bw) || m >= pr.mlen >> 1) break;
        }
    }
    if constexpr (!FILL) pk_filter_bad_seeds(as1, cnt1, a, 10, 40, o.max_gap >> 1, 10);   // (the counting pass has set the flags)
    // ---- limits of the two extensions ----
    int32_t rs = pk_x(a[as1]) - kh, qs = pk_y(a[as1]) - kh;
    int32_t re = pk_x(a[as1 + cnt1 - 1]) - kh, qe = pk_y(a[as1 + cnt1 - 1]) - kh;
    int32_t rs0, qs0, re0, qe0, rs1 = 0, qs1 = 0, re1, qe1, l, i;
    rs0 = pk_x(a[pr.as]) + 1 - pk_span(a[pr.as]);
    qs0 = pk_y(a[pr.as]) + 1 - pk_span(a[pr.as]);
    if (rs0 < 0) rs0 = 0;
    for (i = pr.as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[pr.as].x >> 32; --i) {
        const int32_t x = pk_x(a[i]) + 1 - pk_span(a[i]), y = pk_y(a[i]) + 1 - pk_span(a[i]);
        if (x < rs0 && y < qs0) {
            if (++l > o.min_cnt) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l; qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < o.max_gap ? qs : o.max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else { rs0 = rs; qs0 = qs; }
    re0 = pk_x(a[pr.as + pr.cnt - 1]) + 1;
    qe0 = pk_y(a[pr.as + pr.cnt - 1]) + 1;
    re1 = tlen_all; qe1 = qlen;
    for (i = pr.as + pr.cnt, l = 0; i < pr.n_a && a[i].x >> 32 == a[pr.as].x >> 32; ++i) {
        const int32_t x = pk_x(a[i]) + 1, y = pk_y(a[i]) + 1;
        if (x > re0 && y > qe0) {
            if (++l > o.min_cnt) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l; qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < tlen_all) {
        l = qlen - qe < o.max_gap ? qlen - qe : o.max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < tlen_all - re ? l : tlen_all - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else { re0 = re; qe0 = qe; }
    const int32_t first_qs = qs, first_rs = rs;
    // ---- the windows: left extension, gap fills, right extension ----
    if (qs > 0 && rs > 0) {
        if (!too_big(qs - qs0, rs - rs0))
            emit(qs0, qs - qs0, rs0, rs - rs0, 1, bw, pr.split_inv ? o.zdrop_inv : o.zdrop, o.end_bonus, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR, -1);
        else emit(qs0, 0, rs0, 0, 1, 0, 0, 0, EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR | EZ_REFUSED, -1);
    }
    for (i = 1; i < cnt1; ++i) {
        const uint64_t fl = a[as1 + i].y;
        if ((fl & (PK_SEED_IGNORE | PK_SEED_TANDEM)) && i != cnt1 - 1) continue;
        re = pk_x(a[as1 + i]) - kh; qe = pk_y(a[as1 + i]) - kh;
        if (i == cnt1 - 1 || (fl & PK_SEED_LONG_JOIN) || (qe - qs >= o.min_ksw_len && re - rs >= o.min_ksw_len)) {
            int bw1 = bw;
            if (fl & PK_SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
            if (!too_big(qe - qs, re - rs) && qe - qs > 0 && re - rs > 0) emit(qs, qe - qs, rs, re - rs, 0, bw1, o.zdrop, -1, EZ_APPROX_MAX, i);
            else emit(qs, 0, rs, 0, 0, 0, 0, 0, EZ_APPROX_MAX | EZ_REFUSED, i);
            rs = re; qs = qe;
        }
    }
    if (qe < qe0 && re < re0) {
        if (!too_big(qe0 - qe, re0 - re)) emit(qe, qe0 - qe, re, re0 - re, 0, bw, o.zdrop, o.end_bonus, EZ_EXTZ_ONLY, -1);
        else emit(qe, 0, re, 0, 0, 0, 0, 0, EZ_EXTZ_ONLY | EZ_REFUSED, -1);
    }
    if constexpr (FILL) {
        sregs[ri] = StitchReg{first_job, nj, first_qs, first_rs, qe, re, qs0, qe0, pr.read, rid, rev, 0};
        psum[ri] = PlanSum{as1, cnt1};
    }
    return nj;
}

// One wave per hit: the 64 lanes stage the hit's anchors in LDS, lane 0 counts the hit's windows, reserves their slots in the
// job array (one atomic per hit; the windows of a hit are contiguous, hits come in no particular order) and writes them.
constexpr int PLAN_LDS_ANCHORS = 1024;   // 16 KB: the rest of a longer hit (reads beyond ~50 kb) is walked in global memory
constexpr int PLAN_KBITS = 8192;         // anchors of a hit whose "large gap" flags (mm_filter_bad_seeds' K list) the wave prepares as bits
__global__ __launch_bounds__(64) void plan_kernel(PlanOpt o, const PlanReg *__restrict__ pregs, int n_regs, u128 *__restrict__ A,
                                                  const int32_t *__restrict__ tlens, unsigned long long *__restrict__ n_jobs_total,
                                                  ExtJob *__restrict__ jobs, int32_t *__restrict__ job_anchor, StitchReg *__restrict__ sregs,
                                                  PlanSum *__restrict__ psum) {
    __shared__ u128 lds_a[PLAN_LDS_ANCHORS];
    __shared__ unsigned long long lds_kb[PLAN_KBITS / 64];
    const int lane = threadIdx.x;
    for (int ri = blockIdx.x; ri < n_regs; ri += gridDim.x) {
        const PlanReg pr = pregs[ri];
        AnchorView av;
        av.g = A + pr.a_off; av.l = lds_a; av.lo = pr.as; av.hi = pr.as + (pr.cnt < PLAN_LDS_ANCHORS ? pr.cnt : PLAN_LDS_ANCHORS);
        av.kb = lds_kb; av.kb_n = pr.cnt < PLAN_KBITS ? pr.cnt : PLAN_KBITS;
        for (int i = lane; i < av.hi - av.lo; i += 64) lds_a[i] = av.g[av.lo + i];
        __syncthreads();
        // the K list of mm_filter_bad_seeds as bits: 64 anchors per step instead of one per dependent load of lane 0's walk
        for (int r0 = 0; r0 < av.kb_n; r0 += 64) {
            const int r = r0 + lane;
            bool big = false;
            if (r >= 1 && r < av.kb_n) {
                const u128 cur = av[av.lo + r], prev = av[av.lo + r - 1];
                const int gp = (pk_y(cur) - pk_y(prev)) - (pk_x(cur) - pk_x(prev));
                big = gp < -10 || gp > 10;   // (min_gap of the call in plan_hit)
            }
            const unsigned long long m = __ballot(big);
            if (lane == 0) lds_kb[r0 >> 6] = m;
        }
        __syncthreads();
        if (lane == 0) {
            const int nj = plan_hit<false>(o, pr, av, tlens, ri, 0, jobs, job_anchor, sregs, psum);
            const int32_t first = (int32_t)atomicAdd(n_jobs_total, (unsigned long long)nj);
            plan_hit<true>(o, pr, av, tlens, ri, first, jobs, job_anchor, sregs, psum);
        }
        __syncthreads();
    }
}

// An exact tiled window runs on ONE wave at ~50 instructions per cell: 4 ms for 3000 anti-diagonals, 26 ms for a 5000 x 5000
// extension -- twice the band kernel's eight waves.  The pipeline is sensitive to that pole (-17 % on the strain-rich headline with
// every extension tiled), so the longest extensions stay on the band kernel.
constexpr int TILE_EXACT_MAX_NR = 4096;

// ---- kernel choice and direction-matrix layout of every window --------------------------------------------------------------
// launch lists: every DP window belongs to exactly one
// (strip lists: kernel variant (gap fill with approximate maximum / exact / exact with right-aligned gaps) x lane-group class
// (16/32/64 lanes per window) x strip height 1..16; each is padded to whole waves)
constexpr int N_STRIP_CLASS = 9;   // variant * 3 + lane-group class
enum { L_LDS = 0, L_WG = 5, L_STRIP = 20, N_STRIP = 16 * N_STRIP_CLASS, L_BAND = L_STRIP + N_STRIP, L_TILE = L_BAND + 16, N_LISTS = L_TILE + 1 };
constexpr int STRIP_QB = 64;                                  // query-length buckets inside a strip list (longest first)
constexpr int N_BUCKETS = N_LISTS + N_STRIP * (STRIP_QB - 1);  // scatter buckets: a strip list is STRIP_QB consecutive buckets
__host__ __device__ inline int strip_glc_of_list(int l) { return ((l - L_STRIP) / 16) % 3; }
// (the gap-fill lists -- variant 0 -- are padded to whole waves of PAIRED windows: two per lane group, ext_strip_pair)
__host__ __device__ inline int strip_windows_per_wave(int l) { return (l < L_STRIP + 48 ? 8 : 4) >> strip_glc_of_list(l); }
__host__ __device__ inline int bucket_of_list(int l) {       // first bucket of list l
    return l < L_STRIP ? l : l < L_BAND ? L_STRIP + (l - L_STRIP) * STRIP_QB : L_STRIP + N_STRIP * STRIP_QB + (l - L_BAND);
}

struct LayoutTotals {          // read back by the host after job_layout_kernel
    long long p_tot, row_tot, cig_tot, state_tot, cells, strip_cells[3], xstrip_cells;
    int lds_need[5], strip_lds[N_STRIP_CLASS], band_lds[4], strip_nr[N_STRIP_CLASS];   // (strip_nr: anti-diagonals of the longest exact window)
    int tile_lds, pad_;                                                                  // longest query of the tiled-strip list
    int too_large, tl_q, tl_t, n_jobs;
    int cnt[N_LISTS], base[N_LISTS + 1];   // launch lists in the flat order array (strip lists padded to whole waves)
};

struct JobSizes { long long p, row, cig, st; };   // scratch needs of a window (scanned into offsets)

__global__ __launch_bounds__(256) void job_classify_kernel(ExtJob *__restrict__ jobs, const unsigned long long *__restrict__ nj_p, int strip_scores,
                                                           ExtParams prm, int force_kernel, JobSizes *__restrict__ sizes, int32_t *__restrict__ bucket_cnt,
                                                           LayoutTotals *__restrict__ tot) {
    const int nj = (int)*nj_p;   // (written by plan_kernel, or by the host for the stage test)
    const bool no_tile = force_kernel == 7;   // 7: automatic choice without the tiled strips (MPN_TILED=0)
    if (no_tile) force_kernel = 0;
    const int lds_cap[4] = {8 << 10, 24 << 10, 64 << 10, 150 << 10};
    // counters and maxima are gathered per block in LDS and leave with one atomic per block and slot: a hundred thousand
    // windows adding to ONE global address serialise at the memory side (DESIGN.md lesson 4)
    __shared__ int s_bucket[N_BUCKETS];
    constexpr int M_STRIP = 5, M_BAND = M_STRIP + N_STRIP_CLASS, M_NR = M_BAND + 4, M_TILE = M_NR + N_STRIP_CLASS, M_END = M_TILE + 1;
    __shared__ int s_max[M_END];   // lds_need[5] | strip_lds[9] | band_lds[4] | strip_nr[9] | tile_lds
    for (int k = threadIdx.x; k < N_BUCKETS; k += blockDim.x) s_bucket[k] = 0;
    if (threadIdx.x < M_END) s_max[threadIdx.x] = 0;
    __syncthreads();
    long long cells = 0, scells[3] = {0, 0, 0}, xcells = 0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        ExtJob jb = jobs[j];
        JobSizes sz{0, 0, 0, 0};
        if (jb.flag & EZ_REFUSED) { jb.cls = -1; jobs[j].cls = -1; sizes[j] = sz; continue; }
        const int w = jb.w < 0 ? max(jb.tlen, jb.qlen) : jb.w;
        int n_col = min(jb.qlen, jb.tlen);
        n_col = min(n_col, w + 1) + 1;
        jb.n_col = n_col;
        const long long n_r = (long long)jb.qlen + jb.tlen - 1;
        // strip kernel: lane-group class by target rows (16 x 16, 32 x 16, 64 x 16) and by what the queries of one wave may
        // take in LDS (1024 / 2048 / 4096 bases per window)
        // (the exact variants also take the end extensions and exact fills; they keep H in 16 bits: ext_strip_exact_ok)
        int glc = -1, variant = 0;
        if (w >= max(jb.qlen, jb.tlen) && strip_scores && (force_kernel == 0 || force_kernel == 4)) {
            bool fits = false;
            if (jb.flag & EZ_APPROX_MAX) fits = !(jb.flag & (EZ_EXTZ_ONLY | EZ_RIGHT)) && !jb.reversed;
            else {
                fits = ext_strip_exact_ok(prm.sc_mch, prm.q, prm.e, prm.q2, prm.e2, jb.qlen, jb.tlen);
                variant = (jb.flag & EZ_RIGHT) ? 2 : 1;
            }
            for (int c = 0; c < 3 && fits && glc < 0; ++c)
                if (jb.tlen <= (256 << c) && jb.qlen <= (1024 << c)) glc = c;
        }
        const bool strip = glc >= 0;
        // tiled strips: the gap fills the strip kernel cannot take (target beyond 1024 rows, or a band that clips)
        // ... and the end extensions (exact maximum, z-drop) beyond the exact strip variants' reach
        const bool tile_exact = !(jb.flag & EZ_APPROX_MAX);
        const bool tiled = !strip && strip_scores && !no_tile && (force_kernel == 0 || force_kernel == 6) &&
                           (tile_exact ? (ext_tile_exact_ok(jb.qlen, jb.tlen, w) && (force_kernel == 6 || jb.qlen + jb.tlen <= TILE_EXACT_MAX_NR))
                                       : (!(jb.flag & (EZ_EXTZ_ONLY | EZ_RIGHT)) && !jb.reversed && ext_tile_ok(jb.qlen, jb.tlen, w)));
        const int sclass = variant * 3 + max(glc, 0);
        const int seqb = ((jb.qlen + 3) & ~3) + ((jb.tlen + 3) & ~3);
        // band kernel: the band (n_col - 1 cells at most) plus the stale left neighbour must fit the slots
        int bv = n_col <= 128 ? 0 : n_col <= 256 ? 1 : n_col <= 512 ? 2 : n_col <= 1024 ? 3 : -1;
        if (seqb > lds_cap[3] || !(force_kernel == 0 || force_kernel == 4 || force_kernel == 5)) bv = -1;
        int bc = 3;
        for (int c = 0; c < 4; ++c) if (seqb <= lds_cap[c]) { bc = c; break; }
        if (bv >= 0) atomicMax(&s_max[M_BAND + bc], seqb);   // (also for strip and tiled windows: their exact second pass runs on the band kernel)
        jb.layout = strip ? 1 : tiled ? 3 : bv >= 0 ? 2 : 0;
        const int strip_gl = 16 << max(glc, 0);
        jb.strip_s = max(1, min(16, (jb.tlen + strip_gl - 1) / strip_gl));   // strip height: the window's rows over its lane group
        const int strip_lanes = (jb.tlen + jb.strip_s - 1) / jb.strip_s;
        jb.qstride = strip ? strip_lanes * jb.strip_s : 128 << max(bv, 0);   // row width of the direction matrix (layouts 1, 2)
        const long long strip_bytes = (long long)(jb.qlen + strip_lanes - 1) * (strip_lanes * jb.strip_s);
        // (the rare exact second pass of a strip window gets its direction matrix from a pool of its own)
        sz.p = ((strip ? strip_bytes : tiled ? (long long)tile_matrix_bytes(jb.qlen, jb.tlen, w) : bv >= 0 ? n_r * (128 << bv) : n_r * n_col) + 15) & ~15LL;
        cells += n_r * n_col;
        if (strip) { if (variant == 0) scells[glc] += (long long)jb.qlen * jb.tlen; else xcells += (long long)jb.qlen * jb.tlen; }
        const int stateb = ((6 * jb.tlen + 3) & ~3) + 4 * jb.tlen;
        int cls = 4;
        for (int c = 0; c < 4; ++c) if (seqb + stateb <= lds_cap[c]) { cls = c; break; }
        jb.state_mode = 0;
        const bool use_wg = force_kernel == 3 || (force_kernel != 1 && n_col - 1 > 128);
        const int wg_nt = n_col - 1 <= 256 ? 0 : n_col - 1 <= 512 ? 1 : 2;
        const int redo_list = bv >= 0 ? L_BAND + bv * 4 + bc : use_wg ? L_WG + wg_nt * 5 + cls : L_LDS + cls;
        int lid;
        if (strip) {
            lid = L_STRIP + sclass * 16 + (16 - jb.strip_s);   // tall strips (the waves with the most cells) first: they must not start last
            atomicMax(&s_max[M_STRIP + sclass], (jb.qlen + 15) & ~15);
            if (variant) atomicMax(&s_max[M_NR + sclass], (int)n_r);
        }
        else if (tiled) {
            lid = L_TILE;
            atomicMax(&s_max[M_TILE], ext_tile_lds_bytes(jb.qlen, jb.tlen, tile_exact));
            // the boundary between tiles: two buffers of 12 bytes per query column; exact: + H of the last query column per target row
            sz.st = ((long long)24 * jb.qlen + (tile_exact ? (long long)4 * jb.tlen : 0) + 15) & ~15LL;
        }
        else if (bv >= 0) lid = L_BAND + bv * 4 + bc;
        else lid = redo_list;
        if (bv < 0) {   // the LDS-state kernels may run this window (now or in the second pass)
            if (cls == 4) {
                if (seqb > lds_cap[3]) { tot->too_large = 1; tot->tl_q = jb.qlen; tot->tl_t = jb.tlen; }
                jb.state_mode = 1; sz.st = max(sz.st, (long long)((stateb + 15) & ~15));   // (a tiled window keeps its boundary buffers there too)
                atomicMax(&s_max[4], seqb);
            } else atomicMax(&s_max[cls], seqb + stateb);
            sz.row = n_r;   // band limits are stored only by the LDS-state kernels
        }
        sz.cig = jb.qlen + jb.tlen + 2;
        jb.cls = lid | redo_list << 8 | (bv + 1) << 16;
        jobs[j] = jb;
        sizes[j] = sz;
        // bucket: a strip list is split by query length, longest first
        int b = bucket_of_list(lid);
        if (strip) b += STRIP_QB - 1 - min(STRIP_QB - 1, jb.qlen >> (4 + glc));
        atomicAdd(&s_bucket[b], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N_BUCKETS; k += blockDim.x) if (s_bucket[k]) atomicAdd(&bucket_cnt[k], s_bucket[k]);
    if (threadIdx.x < M_END && s_max[threadIdx.x]) {
        const int k = threadIdx.x;
        atomicMax(k < M_STRIP ? &tot->lds_need[k] : k < M_BAND ? &tot->strip_lds[k - M_STRIP] : k < M_NR ? &tot->band_lds[k - M_BAND] :
                  k < M_TILE ? &tot->strip_nr[k - M_NR] : &tot->tile_lds, s_max[k]);
    }
    // (per-block reduction of the cell counters, one atomic per block)
    __shared__ long long red[5];
    if (threadIdx.x < 5) red[threadIdx.x] = 0;
    __syncthreads();
    long long v[5] = {cells, scells[0], scells[1], scells[2], xcells};
    for (int q = 0; q < 5; ++q) {
        for (int d = 32; d; d >>= 1) v[q] += __shfl_xor(v[q], d);
        if ((threadIdx.x & 63) == 0 && v[q]) atomicAdd((unsigned long long *)&red[q], (unsigned long long)v[q]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (red[0]) atomicAdd((unsigned long long *)&tot->cells, (unsigned long long)red[0]);
        for (int c = 0; c < 3; ++c) if (red[1 + c]) atomicAdd((unsigned long long *)&tot->strip_cells[c], (unsigned long long)red[1 + c]);
        if (red[4]) atomicAdd((unsigned long long *)&tot->xstrip_cells, (unsigned long long)red[4]);
    }
}

// one block: exclusive scans of the four scratch sizes (in place: sizes[j] becomes the window's offsets), of the bucket counts
// (bucket_cur = first slot of every bucket in the flat order array, strip lists padded to whole waves) and the list table
__global__ __launch_bounds__(1024) void job_scan_kernel(JobSizes *__restrict__ sizes, const unsigned long long *__restrict__ nj_p,
                                                        const int32_t *__restrict__ bucket_cnt, int32_t *__restrict__ bucket_cur,
                                                        LayoutTotals *__restrict__ tot, int32_t *__restrict__ order) {
    __shared__ long long part[1024][4];
    __shared__ int list_cnt[N_LISTS], list_base[N_LISTS + 1];
    const int nj = (int)*nj_p;
    if (threadIdx.x == 0) tot->n_jobs = nj;
    const int t = threadIdx.x, per = (nj + 1023) / 1024, lo = min(nj, t * per), hi = min(nj, lo + per);
    long long s[4] = {0, 0, 0, 0};
    for (int j = lo; j < hi; ++j) { s[0] += sizes[j].p; s[1] += sizes[j].row; s[2] += sizes[j].cig; s[3] += sizes[j].st; }
    for (int q = 0; q < 4; ++q) part[t][q] = s[q];
    __syncthreads();
    if (t < 4) {
        long long acc = 0;
        for (int k = 0; k < 1024; ++k) { const long long v = part[k][t]; part[k][t] = acc; acc += v; }
        if (t == 0) tot->p_tot = acc; else if (t == 1) tot->row_tot = acc; else if (t == 2) tot->cig_tot = acc; else tot->state_tot = acc;
    }
    __syncthreads();
    long long o[4] = {part[t][0], part[t][1], part[t][2], part[t][3]};
    for (int j = lo; j < hi; ++j) {
        const JobSizes z = sizes[j];
        // (cig: the window's END offset in the CIGAR scratch, as the traceback kernel expects)
        sizes[j] = JobSizes{o[0], o[1], o[2] + z.cig, o[3]};
        o[0] += z.p; o[1] += z.row; o[2] += z.cig; o[3] += z.st;
    }
    // launch lists: a thread per list adds up its buckets, thread 0 places the lists, then every list's thread places its buckets
    if (t < N_LISTS) {
        const int b0 = bucket_of_list(t), nb = (t >= L_STRIP && t < L_BAND) ? STRIP_QB : 1;
        int c = 0;
        for (int b = 0; b < nb; ++b) c += bucket_cnt[b0 + b];
        list_cnt[t] = c;
    }
    __syncthreads();
    if (t == 0) {
        int pos = 0;
        for (int l = 0; l < N_LISTS; ++l) {
            int c = list_cnt[l];
            list_base[l] = pos;
            if (l >= L_STRIP && l < L_BAND) { const int pw = strip_windows_per_wave(l); c = (c + pw - 1) / pw * pw; }
            pos += c;
        }
        list_base[N_LISTS] = pos;
    }
    __syncthreads();
    if (t < N_LISTS) {
        const int b0 = bucket_of_list(t), nb = (t >= L_STRIP && t < L_BAND) ? STRIP_QB : 1;
        int pos = list_base[t];
        for (int b = 0; b < nb; ++b) { bucket_cur[b0 + b] = pos; pos += bucket_cnt[b0 + b]; }
        tot->cnt[t] = list_cnt[t]; tot->base[t] = list_base[t];
        // the padding of a strip list up to whole waves: entries of -1 (the scatter fills the slots before them)
        for (int k = list_base[t] + list_cnt[t]; k < list_base[t + 1]; ++k) order[k] = -1;
    }
    if (t == 0) tot->base[N_LISTS] = list_base[N_LISTS];
}

__global__ __launch_bounds__(256) void job_layout_kernel(ExtJob *__restrict__ jobs, const unsigned long long *__restrict__ nj_p,
                                                         const JobSizes *__restrict__ offs, int32_t *__restrict__ bucket_cur, int32_t *__restrict__ order) {
    const int nj = (int)*nj_p;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        ExtJob &jb = jobs[j];
        if (jb.cls < 0) continue;
        const JobSizes z = offs[j];
        jb.p_off = z.p; jb.row_off = z.row; jb.cig_off = z.cig; jb.state_off = z.st;
        const int lid = jb.cls & 0xff;
        int b = bucket_of_list(lid);
        if (lid >= L_STRIP && lid < L_BAND) b += STRIP_QB - 1 - min(STRIP_QB - 1, jb.qlen >> (4 + strip_glc_of_list(lid)));
        order[atomicAdd(&bucket_cur[b], 1)] = j;
    }
}

}  // namespace mpn
