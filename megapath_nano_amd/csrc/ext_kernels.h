// Base-level extension kernels (gfx950): banded dual-affine-gap DP with z-drop (the ksw2 formulation minimap2 uses
// for `-c`), traceback, and the z-drop test of a finished CIGAR.  Integer VALU + LDS work: no MFMA.
//
//   ext_dp_strip_kernel<EXACT> windows whose band never clips (the bulk of the cells): systolic, several windows per wave,
//                             all state in VGPRs, 1 B/cell of direction codes in a step-major matrix
//   ext_dp_band_kernel<NW,T>  every other window whose band fits 1024 slots (end extensions with exact max + z-drop,
//                             clipped fills, the exact second pass): band in registers, one barrier per anti-diagonal
//   ext_dp_kernel, ext_dp_wg_kernel<NT>   fallbacks (wider bands, windows beyond LDS): Suzuki-Kasahara states u,v,x,y,x2,y2
//                             and the H row in LDS or global scratch, one wave or one workgroup per window
//   ext_bt_kernel, ext_ztest_kernel        one LANE per window: traceback is a serial pointer chase, the z-drop test a CIGAR walk
#pragma once
#include "mpn_common.h"
#include "map_types.h"

namespace mpn {

enum { EZ_APPROX_MAX = 0x02, EZ_RIGHT = 0x08, EZ_EXTZ_ONLY = 0x40, EZ_REV_CIGAR = 0x80 };

struct ExtJob {
    int32_t read;        // read index in the batch
    int32_t rid;         // target sequence
    int32_t rev;         // hit strand
    int32_t qs, qlen;    // query window [qs, qs+qlen) in strand coordinates
    int32_t ts, tlen;    // target window
    int32_t reversed;    // 1: both windows are read back to front (left extension)
    int32_t w, zdrop, end_bonus, flag;
    int64_t p_off;       // direction scratch offset
    int64_t row_off;     // anti-diagonal row offset (band start/end scratch)
    int64_t cig_off;     // cigar scratch END offset (ops are written back to front unless REV_CIGAR)
    int32_t n_col;
    int32_t state_mode;  // 0: state arrays in LDS, 1: in global scratch
    int64_t state_off;
    int32_t layout;      // direction matrix: 0 = [anti-diagonal][t - band start]; 1 = strip kernel: cell (t, j) at [j + t/S][t];
                         // 2 = band kernel: [anti-diagonal][t mod SL]; 3 = tiled strips (tile_geom)
    int32_t qstride;     // layout 1: row width W = n_lanes * S bytes; layout 2: SL
    int32_t strip_s;     // layout 1: S
    int32_t cls;         // launch list | second-pass list << 8 | (band variant + 1) << 16 (job_classify_kernel); -1: placeholder without DP
};

struct ExtRes {
    int32_t max, zdropped, max_q, max_t, mqe, mqe_t, score, reach_end, n_cigar, r_done, bt_i, bt_j, do_bt, zcode;
    int64_t cig_pos;  // start of the ops in the compact pool (forward order)
};

struct ExtParams {
    int8_t sc_mch, sc_mis, sc_n, q, e, q2, e2;
    int32_t zdrop_thres;  // opt->zdrop for the path test
    int32_t zdrop_inv, max_gap;   // inversion probe of the path test (mm_test_zdrop): a drop above zdrop_inv over a region shorter than max_gap
};

// a gap fill whose largest score drop may hide an inversion: the region of the drop [t0, t1) x [q0, q1) in window coordinates
// (the host runs the local alignment of the region's reverse complement, rare) and whether the drop also exceeds zdrop
struct InvProbe { int32_t jid, t0, q0, t1, q1, over; };

__device__ __forceinline__ uint8_t ext_qbase(const uint8_t *__restrict__ reads, int64_t roff, int32_t rlen, int rev, int x) {
    // base x of the read on the hit's strand, as a 0..4 code
    if (!rev) return (uint8_t)nt4_code(reads[roff + x]);
    const int c = nt4_code(reads[roff + (rlen - 1 - x)]);
    return (uint8_t)(c < 4 ? 3 - c : 4);
}

struct ExtApply {  // running z-drop state (ksw_extz_t subset)
    int32_t max, max_t, max_q, zdropped;
};

__device__ __forceinline__ bool ext_apply_zdrop(ExtApply &ez, int32_t H, int r, int t, int zdrop, int e) {
    if (H > ez.max) { ez.max = H; ez.max_t = t; ez.max_q = r - t; }
    else if (t >= ez.max_t && r - t >= ez.max_q) {
        const int tl = t - ez.max_t, ql = (r - t) - ez.max_q;
        const int l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && ez.max - H > zdrop + l * e) { ez.zdropped = 1; return true; }
    }
    return false;
}

__global__ __launch_bounds__(64) void ext_dp_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_jobs,
                                                    ExtParams prm, const uint8_t *__restrict__ reads,
                                                    const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                    RefView rv,
                                                    uint8_t *__restrict__ P, int32_t *__restrict__ OFF, int8_t *__restrict__ gstate,
                                                    ExtRes *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x;
    const int jid = order[blockIdx.x];
    const ExtJob jb = jobs[jid];
    const int qlen = jb.qlen, tlen = jb.tlen;
    int q = prm.q, e = prm.e, q2 = prm.q2, e2 = prm.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    ExtRes out;
    out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = NEG_INF;
    out.reach_end = 0; out.n_cigar = 0; out.r_done = -1; out.bt_i = out.bt_j = -1; out.do_bt = 0; out.zcode = 0;
    if (qlen <= 0 || tlen <= 0 || -prm.sc_mis > 2 * (q + e)) { if (lane == 0) res[jid] = out; return; }
    int w = jb.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int n_col = jb.n_col;
    // LDS carve: qseq[qlen] tseq[tlen] (padded to 4) then state
    uint8_t *qs_ = smem;
    uint8_t *ts_ = smem + ((qlen + 3) & ~3);
    int8_t *sbase = jb.state_mode ? gstate + jb.state_off : (int8_t *)(ts_ + ((tlen + 3) & ~3));
    int8_t *u = sbase, *v = u + tlen, *x = v + tlen, *y = x + tlen, *x2 = y + tlen, *y2 = x2 + tlen;
    int32_t *H = (int32_t *)(sbase + (((size_t)6 * tlen + 3) & ~(size_t)3));
    const bool approx = (jb.flag & EZ_APPROX_MAX) != 0;
    {
        const int64_t roff = read_off[jb.read];
        const int32_t rlen = read_len[jb.read];
        for (int i = lane; i < qlen; i += 64) qs_[i] = ext_qbase(reads, roff, rlen, jb.rev, jb.qs + (jb.reversed ? qlen - 1 - i : i));
        const int64_t g0 = rv.seq_off[jb.rid] + jb.ts;
        for (int i = lane; i < tlen; i += 64) ts_[i] = (uint8_t)ref_code(rv, g0 + (jb.reversed ? tlen - 1 - i : i));
        for (int i = lane; i < tlen; i += 64) {
            u[i] = v[i] = x[i] = y[i] = (int8_t)-qe;
            x2[i] = y2[i] = (int8_t)-qe2;
            if (!approx) H[i] = NEG_INF;
        }
    }
    __syncthreads();
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    uint8_t *p = P + jb.p_off;
    const int n_r = qlen + tlen - 1;
    int32_t *off = OFF + 2 * jb.row_off;  // [r] band start, [n_r + r] band end
    int32_t *off_end = off + n_r;
    ExtApply ez; ez.max = 0; ez.max_t = ez.max_q = -1; ez.zdropped = 0;
    int32_t mqe = NEG_INF, mqe_t = -1, score = NEG_INF, H0 = 0, last_H0_t = 0;
    int last_st = -1, last_en = -1, r;
    const bool right = (jb.flag & EZ_RIGHT) != 0;
    for (r = 0; r < n_r; ++r) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) { ez.zdropped = 1; break; }
        int cx1, cx21, cv1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { cx1 = x[st - 1]; cx21 = x2[st - 1]; cv1 = v[st - 1]; }
            else { cx1 = -qe; cx21 = -qe2; cv1 = -qe; }
        } else {
            cx1 = -qe; cx21 = -qe2;
            cv1 = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
        }
        if (en >= r) {
            if (lane == 0) {
                y[r] = (int8_t)-qe; y2[r] = (int8_t)-qe2;
                u[r] = (int8_t)(r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
            }
            __syncthreads();
        }
        if (lane == 0) { off[r] = st; off_end[r] = en; }
        uint8_t *pr = p + (int64_t)r * n_col;
        // exact-max bookkeeping needs OLD H[en-1]
        int32_t Hen_new = 0;
        if (!approx && r > 0) Hen_new = en > 0 ? H[en - 1] : H[en];
        int32_t bestH = NEG_INF, bestKey = 0x7fffffff;
        const int en1 = st + (en - st) / 4 * 4;
        for (int c0 = st; c0 <= en; c0 += 64) {
            const int t = c0 + lane;
            const bool act = t <= en;
            int ut = 0, vt = 0, xt = 0, x2t = 0, yt = 0, y2t = 0, sc = 0;
            if (act) {
                ut = u[t]; vt = v[t]; xt = x[t]; x2t = x2[t]; yt = y[t]; y2t = y2[t];
                const int sq = ts_[t], sr = qs_[r - t];
                sc = (sq == 4 || sr == 4) ? prm.sc_n : sq == sr ? prm.sc_mch : prm.sc_mis;
            }
            const int v1 = wave_shr1(vt, cv1), x1 = wave_shr1(xt, cx1), x21 = wave_shr1(x2t, cx21);
            cv1 = __builtin_amdgcn_readlane(vt, 63); cx1 = __builtin_amdgcn_readlane(xt, 63); cx21 = __builtin_amdgcn_readlane(x2t, 63);
            int z = sc, a = x1 + v1, b = yt + ut, a2 = x21 + v1, b2 = y2t + ut, d;
            if (!right) {
                d = a > z ? 1 : 0; z = max(z, a);
                d = b > z ? 2 : d; z = max(z, b);
                d = a2 > z ? 3 : d; z = max(z, a2);
                d = b2 > z ? 4 : d; z = max(z, b2);
            } else {
                d = z > a ? 0 : 1; z = max(z, a);
                d = z > b ? d : 2; z = max(z, b);
                d = z > a2 ? d : 3; z = max(z, a2);
                d = z > b2 ? d : 4; z = max(z, b2);
            }
            z = min(z, (int)prm.sc_mch);
            const int nu = z - v1, nv = z - ut;
            int tmp = z - q; a -= tmp; b -= tmp;
            tmp = z - q2; a2 -= tmp; b2 -= tmp;
            if (!right) {
                d |= a > 0 ? 0x08 : 0; d |= b > 0 ? 0x10 : 0; d |= a2 > 0 ? 0x20 : 0; d |= b2 > 0 ? 0x40 : 0;
            } else {
                d |= a >= 0 ? 0x08 : 0; d |= b >= 0 ? 0x10 : 0; d |= a2 >= 0 ? 0x20 : 0; d |= b2 >= 0 ? 0x40 : 0;
            }
            if (act) {
                u[t] = (int8_t)nu; v[t] = (int8_t)nv;
                x[t] = (int8_t)(max(a, 0) - qe); y[t] = (int8_t)(max(b, 0) - qe);
                x2[t] = (int8_t)(max(a2, 0) - qe2); y2[t] = (int8_t)(max(b2, 0) - qe2);
                pr[t - st] = (uint8_t)d;
                if (!approx && r > 0 && t < en) {
                    const int32_t h = H[t] + nv;
                    H[t] = h;
                    const int key = (t < en1 ? 1 + ((t - st) & 3) : 5) << 24 | t;
                    if (h > bestH || (h == bestH && key < bestKey)) { bestH = h; bestKey = key; }
                }
            }
        }
        __syncthreads();
        if (!approx) {
            int32_t max_H, max_t;
            if (r > 0) {
                const int32_t hen = Hen_new + (en > 0 ? (int)u[en] : (int)v[en]);
                if (lane == 0) H[en] = hen;
                // en wins every tie; then (t-st)&3 class order for t < en1; then the tail
                const int32_t m = wave_reduce_max(bestH);
                int kk = (bestH == m && m > NEG_INF) ? bestKey : 0x7fffffff;
                kk = wave_reduce_min(kk);
                if (m > hen) { max_H = m; max_t = kk & 0xffffff; }
                else { max_H = hen; max_t = en; }
            } else {
                const int32_t h0 = (int)v[0] - qe;
                if (lane == 0) H[0] = h0;
                max_H = h0; max_t = 0;
            }
            __syncthreads();
            if (r - st == qlen - 1) { const int32_t hs = H[st]; if (hs > mqe) { mqe = hs; mqe_t = st; } }
            if (ext_apply_zdrop(ez, max_H, r, max_t, jb.zdrop, e2)) break;
            if (r == n_r - 1 && en == tlen - 1) score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st && last_H0_t <= en && last_H0_t + 1 >= st && last_H0_t + 1 <= en) {
                    const int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0; else { H0 += d1; ++last_H0_t; }
                } else if (last_H0_t >= st && last_H0_t <= en) H0 += v[last_H0_t];
                else { ++last_H0_t; H0 += u[last_H0_t]; }
            } else { H0 = (int)v[0] - qe; last_H0_t = 0; }
            if (r == n_r - 1 && en == tlen - 1) score = H0;
        }
        last_st = st; last_en = en;
    }
    out.max = ez.max; out.max_t = ez.max_t; out.max_q = ez.max_q; out.zdropped = ez.zdropped;
    out.mqe = mqe; out.mqe_t = mqe_t; out.score = score;
    out.r_done = r < n_r ? r : n_r - 1;
    if (!ez.zdropped && !(jb.flag & EZ_EXTZ_ONLY)) { out.do_bt = 1; out.bt_i = tlen - 1; out.bt_j = qlen - 1; }
    else if (!ez.zdropped && (jb.flag & EZ_EXTZ_ONLY) && mqe + jb.end_bonus > ez.max) { out.reach_end = 1; out.do_bt = 1; out.bt_i = mqe_t; out.bt_j = qlen - 1; }
    else if (ez.max_t >= 0 && ez.max_q >= 0) { out.do_bt = 1; out.bt_i = ez.max_t; out.bt_j = ez.max_q; }
    if (lane == 0) res[jid] = out;
}

// Workgroup-per-window variant for windows the band kernel below cannot take (band wider than 1024 cells, or
// sequences that do not fit LDS): the band of one anti-diagonal is spread over NT threads instead of being walked by one
// wave tile after tile.  States live in LDS (or the
// global scratch for huge windows).  Tiles are processed from the highest target position down: a tile reads its own
// cells and its left neighbour (t-1, old values), a barrier follows, then it writes; lower tiles are still untouched.
template <int NT>
__global__ __launch_bounds__(NT) void ext_dp_wg_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_jobs,
                                                       ExtParams prm, const uint8_t *__restrict__ reads,
                                                       const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                       RefView rv,
                                                       uint8_t *__restrict__ P, int32_t *__restrict__ OFF, int8_t *__restrict__ gstate,
                                                       ExtRes *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int red_h[NT / 64], red_k[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int jid = order[blockIdx.x];
    const ExtJob jb = jobs[jid];
    const int qlen = jb.qlen, tlen = jb.tlen;
    int q = prm.q, e = prm.e, q2 = prm.q2, e2 = prm.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    ExtRes out;
    out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = NEG_INF;
    out.reach_end = 0; out.n_cigar = 0; out.r_done = -1; out.bt_i = out.bt_j = -1; out.do_bt = 0; out.zcode = 0; out.cig_pos = 0;
    if (qlen <= 0 || tlen <= 0 || -prm.sc_mis > 2 * (q + e)) { if (tid == 0) res[jid] = out; return; }
    int w = jb.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int n_col = jb.n_col;
    uint8_t *qs_ = smem;
    uint8_t *ts_ = smem + ((qlen + 3) & ~3);
    int8_t *sbase = jb.state_mode ? gstate + jb.state_off : (int8_t *)(ts_ + ((tlen + 3) & ~3));
    int8_t *u = sbase, *v = u + tlen, *x = v + tlen, *y = x + tlen, *x2 = y + tlen, *y2 = x2 + tlen;
    int32_t *H = (int32_t *)(sbase + (((size_t)6 * tlen + 3) & ~(size_t)3));
    const bool approx = (jb.flag & EZ_APPROX_MAX) != 0;
    {
        const int64_t roff = read_off[jb.read];
        const int32_t rlen = read_len[jb.read];
        for (int i = tid; i < qlen; i += NT) qs_[i] = ext_qbase(reads, roff, rlen, jb.rev, jb.qs + (jb.reversed ? qlen - 1 - i : i));
        const int64_t g0 = rv.seq_off[jb.rid] + jb.ts;
        for (int i = tid; i < tlen; i += NT) {
            ts_[i] = (uint8_t)ref_code(rv, g0 + (jb.reversed ? tlen - 1 - i : i));
            u[i] = v[i] = x[i] = y[i] = (int8_t)-qe;
            x2[i] = y2[i] = (int8_t)-qe2;
            if (!approx) H[i] = NEG_INF;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    uint8_t *p = P + jb.p_off;
    const int n_r = qlen + tlen - 1;
    int32_t *off = OFF + 2 * jb.row_off, *off_end = off + n_r;
    ExtApply ez; ez.max = 0; ez.max_t = ez.max_q = -1; ez.zdropped = 0;
    int32_t mqe = NEG_INF, mqe_t = -1, score = NEG_INF, H0 = 0, last_H0_t = 0;
    int last_st = -1, last_en = -1, r;
    const bool right = (jb.flag & EZ_RIGHT) != 0;
    for (r = 0; r < n_r; ++r) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) { ez.zdropped = 1; break; }
        const bool left_known = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
        int bv1 = -qe;
        if (st == 0) bv1 = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
        const int bnd_u = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
        if (tid == 0) { off[r] = st; off_end[r] = en; }
        uint8_t *pr = p + (int64_t)r * n_col;
        const int ncell = en - st + 1, ntile = (ncell + NT - 1) / NT;
        const int en1 = st + (en - st) / 4 * 4;
        int32_t bestH = NEG_INF, bestKey = 0x7fffffff, hen_old = 0;
        if (!approx && r > 0) hen_old = en > 0 ? H[en - 1] : H[en];
        int32_t hen = 0;
        for (int tile = ntile - 1; tile >= 0; --tile) {
            const int t = st + tile * NT + tid;
            const bool act = t <= en;
            int nu = 0, nv = 0, nx = 0, ny = 0, nx2 = 0, ny2 = 0, d = 0;
            if (act) {
                int ut = u[t], yt = y[t], y2t = y2[t];
                if (en >= r && t == r) { yt = -qe; y2t = -qe2; ut = bnd_u; }
                int v1, x1, x21;
                if (t == st && !left_known) { v1 = bv1; x1 = -qe; x21 = -qe2; }
                else { v1 = v[t - 1]; x1 = x[t - 1]; x21 = x2[t - 1]; }
                const int sq = ts_[t], sr = qs_[r - t];
                const int sc = (sq == 4 || sr == 4) ? prm.sc_n : sq == sr ? prm.sc_mch : prm.sc_mis;
                int z = sc, a = x1 + v1, b = yt + ut, a2 = x21 + v1, b2 = y2t + ut;
                if (!right) {
                    d = a > z ? 1 : 0; z = max(z, a);
                    d = b > z ? 2 : d; z = max(z, b);
                    d = a2 > z ? 3 : d; z = max(z, a2);
                    d = b2 > z ? 4 : d; z = max(z, b2);
                } else {
                    d = z > a ? 0 : 1; z = max(z, a);
                    d = z > b ? d : 2; z = max(z, b);
                    d = z > a2 ? d : 3; z = max(z, a2);
                    d = z > b2 ? d : 4; z = max(z, b2);
                }
                z = min(z, (int)prm.sc_mch);
                nu = z - v1; nv = z - ut;
                int tmp = z - q; a -= tmp; b -= tmp;
                tmp = z - q2; a2 -= tmp; b2 -= tmp;
                if (!right) { d |= a > 0 ? 0x08 : 0; d |= b > 0 ? 0x10 : 0; d |= a2 > 0 ? 0x20 : 0; d |= b2 > 0 ? 0x40 : 0; }
                else { d |= a >= 0 ? 0x08 : 0; d |= b >= 0 ? 0x10 : 0; d |= a2 >= 0 ? 0x20 : 0; d |= b2 >= 0 ? 0x40 : 0; }
                nx = max(a, 0) - qe; ny = max(b, 0) - qe; nx2 = max(a2, 0) - qe2; ny2 = max(b2, 0) - qe2;
            }
            __syncthreads();  // every read of this tile (incl. its left neighbour) is done
            if (act) {
                u[t] = (int8_t)nu; v[t] = (int8_t)nv; x[t] = (int8_t)nx; y[t] = (int8_t)ny; x2[t] = (int8_t)nx2; y2[t] = (int8_t)ny2;
                pr[t - st] = (uint8_t)d;
                if (!approx && r > 0) {
                    if (t < en) {
                        const int32_t h = H[t] + nv;
                        H[t] = h;
                        const int key = (t < en1 ? 1 + ((t - st) & 3) : 5) << 24 | t;
                        if (h > bestH || (h == bestH && key < bestKey)) { bestH = h; bestKey = key; }
                    } else {
                        hen = hen_old + (en > 0 ? nu : nv);
                        H[en] = hen;
                    }
                }
                if (!approx && r == 0) H[0] = nv - qe;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (!approx) {
            int32_t max_H, max_t;
            if (r > 0) {
                const int32_t hen_all = H[en];
                const int32_t m = wave_reduce_max(bestH);
                int kk = (bestH == m && m > NEG_INF) ? bestKey : 0x7fffffff;
                kk = wave_reduce_min(kk);
                if (lane == 0) { red_h[wv] = m; red_k[wv] = kk; }
                __syncthreads();
                int32_t mm = NEG_INF; int mk = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < NT / 64; ++i) {
                    const int32_t hh = red_h[i]; const int k2 = red_k[i];
                    if (hh > mm || (hh == mm && k2 < mk)) { mm = hh; mk = k2; }
                }
                if (mm > hen_all) { max_H = mm; max_t = mk & 0xffffff; }
                else { max_H = hen_all; max_t = en; }
                __syncthreads();
            } else { max_H = H[0]; max_t = 0; }
            if (r - st == qlen - 1) { const int32_t hs = H[st]; if (hs > mqe) { mqe = hs; mqe_t = st; } }
            if (ext_apply_zdrop(ez, max_H, r, max_t, jb.zdrop, e2)) break;
            if (r == n_r - 1 && en == tlen - 1) score = H[tlen - 1];
        } else {
            if (r > 0) {
                if (last_H0_t >= st && last_H0_t <= en && last_H0_t + 1 >= st && last_H0_t + 1 <= en) {
                    const int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0; else { H0 += d1; ++last_H0_t; }
                } else if (last_H0_t >= st && last_H0_t <= en) H0 += v[last_H0_t];
                else { ++last_H0_t; H0 += u[last_H0_t]; }
            } else { H0 = (int)v[0] - qe; last_H0_t = 0; }
            if (r == n_r - 1 && en == tlen - 1) score = H0;
        }
        last_st = st; last_en = en;
    }
    out.max = ez.max; out.max_t = ez.max_t; out.max_q = ez.max_q; out.zdropped = ez.zdropped;
    out.mqe = mqe; out.mqe_t = mqe_t; out.score = score;
    out.r_done = r < n_r ? r : n_r - 1;
    if (!ez.zdropped && !(jb.flag & EZ_EXTZ_ONLY)) { out.do_bt = 1; out.bt_i = tlen - 1; out.bt_j = qlen - 1; }
    else if (!ez.zdropped && (jb.flag & EZ_EXTZ_ONLY) && mqe + jb.end_bonus > ez.max) { out.reach_end = 1; out.do_bt = 1; out.bt_i = mqe_t; out.bt_j = qlen - 1; }
    else if (ez.max_t >= 0 && ez.max_q >= 0) { out.do_bt = 1; out.bt_i = ez.max_t; out.bt_j = ez.max_q; }
    if (tid == 0) res[jid] = out;
}

// Band-in-registers variant: the general kernel for windows whose band is at most NW*64*T cells wide (any tlen, both
// the approximate-max gap fills and the exact-max / z-drop end extensions).  Target position t lives in slot
// t mod SL (SL = NW*64*T, a power of two); thread `tid` owns the T consecutive slots tid*T..tid*T+T-1 for the whole job,
// so the six difference states and the H row never leave VGPRs while the band slides along the target.  The left
// neighbour t-1 is the thread's own previous slot, the previous lane's last slot (one DPP wave_shr per state) or, for
// lane 0, the last slot of the previous wave, handed over through LDS.  One barrier per anti-diagonal: boundary
// states, per-wave maxima and the few single-owner values (H[en], H[st], the approximate-max cell) are published
// into parity-indexed LDS slots at the end of anti-diagonal r and consumed at the start of r+1, which also moves the
// z-drop decision for r there (nothing of r+1 is computed before it).  Direction codes are stored by slot
// ([r][t mod SL], one aligned 32-bit store per four cells); the traceback recomputes the band limits from r, so no
// band-start/end arrays are written.
// APPROX: the bookkeeping of ksw2's approximate maximum (gap fills) or of the exact one (end extensions, exact fills): a workgroup
// runs one window, so the kernel branches once per window into the instantiation that carries only its own bookkeeping.  A wave
// none of whose slots lies in the band of an anti-diagonal (the band is at most 752 wide, the slots 1024) skips the cells and
// only publishes its unchanged boundary: its issue slots go to the other kernels on the CU.
template <int NW, int T, bool APPROX>
__device__ __forceinline__ void ext_dp_band_body(const ExtJob &jb, const int jid, const ExtParams &prm, const uint8_t *__restrict__ reads,
                                                 const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                 const RefView &rv, uint8_t *__restrict__ P, ExtRes *__restrict__ res, uint8_t *smem,
                                                 int (*bnd)[NW][4], int (*red)[NW][2], int (*pub)[4]) {
    constexpr int NT = NW * 64, SL = NT * T;
    constexpr bool approx = APPROX;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qlen = jb.qlen, tlen = jb.tlen;
    int q = prm.q, e = prm.e, q2 = prm.q2, e2 = prm.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    ExtRes out;
    out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = NEG_INF;
    out.reach_end = 0; out.n_cigar = 0; out.r_done = -1; out.bt_i = out.bt_j = -1; out.do_bt = 0; out.zcode = 0; out.cig_pos = 0;
    if (qlen <= 0 || tlen <= 0 || -prm.sc_mis > 2 * (q + e)) { if (tid == 0) res[jid] = out; return; }
    int w = jb.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    uint8_t *qs_ = smem;
    uint8_t *ts_ = smem + ((qlen + 3) & ~3);
    {
        const int64_t roff = read_off[jb.read];
        const int32_t rlen = read_len[jb.read];
        for (int i = tid; i < qlen; i += NT) qs_[i] = ext_qbase(reads, roff, rlen, jb.rev, jb.qs + (jb.reversed ? qlen - 1 - i : i));
        const int64_t g0 = rv.seq_off[jb.rid] + jb.ts;
        for (int i = tid; i < tlen; i += NT) ts_[i] = (uint8_t)ref_code(rv, g0 + (jb.reversed ? tlen - 1 - i : i));
    }
    int U[T], V[T], X[T], Y[T], X2[T], Y2[T], H[APPROX ? 1 : T];
#pragma unroll
    for (int k = 0; k < T; ++k) { U[k] = V[k] = X[k] = Y[k] = -qe; X2[k] = Y2[k] = -qe2; if constexpr (!APPROX) H[k] = NEG_INF; }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    uint8_t *p = P + jb.p_off + tid * T;
    const int n_r = qlen + tlen - 1;
    ExtApply ez; ez.max = 0; ez.max_t = ez.max_q = -1; ez.zdropped = 0;
    int32_t mqe = NEG_INF, mqe_t = -1, score = NEG_INF, H0 = 0, last_H0_t = 0;
    int last_st = -1, last_en = -1, r, r_done = n_r - 1;
    const bool right = (jb.flag & EZ_RIGHT) != 0;
    const int slot0 = tid * T;
    for (r = 0;; ++r) {
        const int par = r & 1;
        __syncthreads();  // publications of anti-diagonal r-1 (and, for r = 0, the staged sequences) are visible
        if (r > 0) {      // close anti-diagonal r-1
            const int pr_ = r - 1, pp = par ^ 1;
            if (!approx) {
                const int32_t hen = pub[pp][0];
                int32_t max_H = hen, max_t = last_en;
                if (pr_ > 0) {
                    int32_t mm = NEG_INF; int mk = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < NW; ++i) {
                        const int32_t hh = red[pp][i][0]; const int k2 = red[pp][i][1];
                        if (hh > mm || (hh == mm && k2 < mk)) { mm = hh; mk = k2; }
                    }
                    if (mm > hen) { max_H = mm; max_t = mk & 0xffffff; }
                }
                if (pr_ - last_st == qlen - 1) { const int32_t hs = pub[pp][1]; if (hs > mqe) { mqe = hs; mqe_t = last_st; } }
                if (ext_apply_zdrop(ez, max_H, pr_, max_t, jb.zdrop, e2)) { r_done = pr_; break; }
                if (pr_ == n_r - 1 && last_en == tlen - 1) score = hen;
            } else {
                if (pr_ > 0) {
                    const bool in0 = last_H0_t >= last_st && last_H0_t <= last_en, in1 = last_H0_t + 1 >= last_st && last_H0_t + 1 <= last_en;
                    const int32_t d0 = pub[pp][2], d1 = pub[pp][3];
                    if (in0 && in1) { if (d0 > d1) H0 += d0; else { H0 += d1; ++last_H0_t; } }
                    else if (in0) H0 += d0;
                    else { ++last_H0_t; H0 += d1; }
                } else { H0 = pub[pp][2] - qe; last_H0_t = 0; }
                if (pr_ == n_r - 1 && last_en == tlen - 1) score = H0;
            }
        }
        if (r == n_r) break;
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
        if (en > (r + w) >> 1) en = (r + w) >> 1;
        if (st > en) { ez.zdropped = 1; r_done = r; break; }
        const bool left_known = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
        const int bnd_u = r == 0 ? -qe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
        const int bv1 = st == 0 ? bnd_u : -qe;
        const bool en_new = en > last_en;
        // old state of the slot to the left of this thread's first slot
        const int pw = (wv + NW - 1) % NW;
        int cv = 0, cx = 0, cx2 = 0, ch = 0;
        if (r > 0 && lane == 0) { cv = bnd[par ^ 1][pw][0]; cx = bnd[par ^ 1][pw][1]; cx2 = bnd[par ^ 1][pw][2]; if constexpr (!APPROX) ch = bnd[par ^ 1][pw][3]; }
        const int lv = wave_shr1(V[T - 1], cv), lx = wave_shr1(X[T - 1], cx), lx2 = wave_shr1(X2[T - 1], cx2);
        int lh = 0;
        if constexpr (!APPROX) lh = wave_shr1(H[T - 1], ch);
        // does the band touch this wave's slots at all?  (its slots from the band's first one, modulo SL: [wa, wa + 64 T))
        const int wa = (wv * 64 * T - st) & (SL - 1);
        const bool wave_act = wa < en - st + 1 || wa + 64 * T > SL;
        const int en1 = st + (en - st) / 4 * 4;
        int32_t bestH = NEG_INF, bestKey = 0x7fffffff;
        uint32_t dw[(T + 3) / 4];
#pragma unroll
        for (int k = 0; k < (T + 3) / 4; ++k) dw[k] = 0;
        // single-owner values of this anti-diagonal, published once after the cell loop (selects, no branches, inside it)
        int32_t pv_en = 0, pv_st = 0, pv_h0v = 0, pv_h0u = 0;
        bool has_en = false, has_st = false, has_h0v = false, has_h0u = false, any = false;
        const bool first = r == 0, en_pos = en > 0;
        if (wave_act) {
#pragma unroll
        for (int k = T - 1; k >= 0; --k) {
            // Cells outside the band compute on clamped operands and their results are never read: a slot that has not
            // entered the band yet is overridden when it does (`fresh`), one that left it is read at most once more, as
            // the left neighbour of the band start, and that read sees the values from before this anti-diagonal.
            const int t = st + ((slot0 + k - st) & (SL - 1));
            const bool act = t <= en;
            any |= act;
            const bool is_en = t == en, is_st = t == st;
            const bool fresh = is_en && en_new;
            const bool col0 = is_en && en >= r;  // t == r: first query column
            const int ut = col0 ? bnd_u : fresh ? -qe : U[k], yt = fresh ? -qe : Y[k], y2t = fresh ? -qe2 : Y2[k];
            const bool edge = is_st && !left_known;
            const int v1 = edge ? bv1 : k > 0 ? V[k > 0 ? k - 1 : 0] : lv;
            const int x1 = edge ? -qe : k > 0 ? X[k > 0 ? k - 1 : 0] : lx;
            const int x21 = edge ? -qe2 : k > 0 ? X2[k > 0 ? k - 1 : 0] : lx2;
            int hl = lh;
            if constexpr (!APPROX) hl = k > 0 ? H[k > 0 ? k - 1 : 0] : lh;
            const int sq = ts_[min(t, tlen - 1)], sr = qs_[min(max(r - t, 0), qlen - 1)];
            const int sc = (sq == 4 || sr == 4) ? prm.sc_n : sq == sr ? prm.sc_mch : prm.sc_mis;
            int z = sc, a = x1 + v1, b = yt + ut, a2 = x21 + v1, b2 = y2t + ut, d;
            if (!right) {
                d = a > z ? 1 : 0; z = max(z, a);
                d = b > z ? 2 : d; z = max(z, b);
                d = a2 > z ? 3 : d; z = max(z, a2);
                d = b2 > z ? 4 : d; z = max(z, b2);
            } else {
                d = z > a ? 0 : 1; z = max(z, a);
                d = z > b ? d : 2; z = max(z, b);
                d = z > a2 ? d : 3; z = max(z, a2);
                d = z > b2 ? d : 4; z = max(z, b2);
            }
            z = min(z, (int)prm.sc_mch);
            const int nu = z - v1, nv = z - ut;
            int tmp = z - q; a -= tmp; b -= tmp;
            tmp = z - q2; a2 -= tmp; b2 -= tmp;
            if (!right) { d |= a > 0 ? 0x08 : 0; d |= b > 0 ? 0x10 : 0; d |= a2 > 0 ? 0x20 : 0; d |= b2 > 0 ? 0x40 : 0; }
            else { d |= a >= 0 ? 0x08 : 0; d |= b >= 0 ? 0x10 : 0; d |= a2 >= 0 ? 0x20 : 0; d |= b2 >= 0 ? 0x40 : 0; }
            U[k] = nu; V[k] = nv;
            X[k] = max(a, 0) - qe; Y[k] = max(b, 0) - qe;
            X2[k] = max(a2, 0) - qe2; Y2[k] = max(b2, 0) - qe2;
            dw[k >> 2] |= (uint32_t)d << (8 * (k & 3));
            if constexpr (!APPROX) {   // exact mode: the H row
                const int32_t h = first ? nv - qe : (is_en && en_pos) ? hl + nu : H[k] + nv;
                H[k] = act ? h : H[k];  // (H alone can be read stale: H[en-1] of a one-cell band)
                const int key = (t < en1 ? 1 + ((t - st) & 3) : 5) << 24 | t;
                const bool better = t < en && (h > bestH || (h == bestH && key < bestKey));
                bestH = better ? h : bestH; bestKey = better ? key : bestKey;
                pv_en = is_en ? h : pv_en; has_en |= is_en;
                pv_st = is_st ? h : pv_st; has_st |= is_st;
            } else {                   // approximate mode: the tracked cell
                const bool is0 = act && t == last_H0_t, is1 = act && t == last_H0_t + 1;
                pv_h0v = is0 ? nv : pv_h0v; has_h0v |= is0;
                pv_h0u = is1 ? nu : pv_h0u; has_h0u |= is1;
            }
        }
        }
        if (any) {
            uint8_t *dst = p + (int64_t)r * SL;
            if constexpr (T == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)dw[0];
            else {
#pragma unroll
                for (int k = 0; k < T / 4; ++k) reinterpret_cast<uint32_t *>(dst)[k] = dw[k];
            }
        }
        if (has_en) pub[par][0] = pv_en;
        if (has_st) pub[par][1] = pv_st;
        if (has_h0v) pub[par][2] = pv_h0v;
        if (has_h0u) pub[par][3] = pv_h0u;
        if (lane == 63) { bnd[par][wv][0] = V[T - 1]; bnd[par][wv][1] = X[T - 1]; bnd[par][wv][2] = X2[T - 1]; if constexpr (!APPROX) bnd[par][wv][3] = H[T - 1]; }
        if (!approx && r > 0) {
            const int32_t m = wave_reduce_max(bestH);
            int kk = (bestH == m && m > NEG_INF) ? bestKey : 0x7fffffff;
            kk = wave_reduce_min(kk);
            if (lane == 0) { red[par][wv][0] = m; red[par][wv][1] = kk; }
        }
        last_st = st; last_en = en;
    }
    out.max = ez.max; out.max_t = ez.max_t; out.max_q = ez.max_q; out.zdropped = ez.zdropped;
    out.mqe = mqe; out.mqe_t = mqe_t; out.score = score;
    out.r_done = r_done;
    if (!ez.zdropped && !(jb.flag & EZ_EXTZ_ONLY)) { out.do_bt = 1; out.bt_i = tlen - 1; out.bt_j = qlen - 1; }
    else if (!ez.zdropped && (jb.flag & EZ_EXTZ_ONLY) && mqe + jb.end_bonus > ez.max) { out.reach_end = 1; out.do_bt = 1; out.bt_i = mqe_t; out.bt_j = qlen - 1; }
    else if (ez.max_t >= 0 && ez.max_q >= 0) { out.do_bt = 1; out.bt_i = ez.max_t; out.bt_j = ez.max_q; }
    if (tid == 0) res[jid] = out;
}

template <int NW, int T>
__global__ __launch_bounds__(NW * 64) void ext_dp_band_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_jobs,
                                                              ExtParams prm, const uint8_t *__restrict__ reads,
                                                              const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                              RefView rv, uint8_t *__restrict__ P, ExtRes *__restrict__ res) {
    static_assert(T % 4 == 0 || T == 2, "T must be 2 or a multiple of 4");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int bnd[2][NW][4];   // last slot of each wave after the anti-diagonal: v, x, x2, H
    __shared__ int red[2][NW][2];   // per-wave best H and its tie-break key
    __shared__ int pub[2][4];       // H[en], H[st], v[last_H0_t], u[last_H0_t + 1]
    const int jid = order[blockIdx.x];
    const ExtJob jb = jobs[jid];
    if (jb.flag & EZ_APPROX_MAX) ext_dp_band_body<NW, T, true>(jb, jid, prm, reads, read_off, read_len, rv, P, res, smem, bnd, red, pub);
    else ext_dp_band_body<NW, T, false>(jb, jid, prm, reads, read_off, read_len, rv, P, res, smem, bnd, red, pub);
}

// Systolic strip variant for gap-fill windows whose band never clips (w >= max(qlen, tlen), tlen <= 1024, gaps
// left-aligned).  A window runs on a group of GL = 16, 32 or 64 lanes (see ext_dp_strip_kernel below): lane l owns the S consecutive target rows t = l*S .. l*S+S-1 and walks the query columns j = step - l,
// so after a ramp of n_lanes-1 steps every lane computes S cells per step -- no partially filled anti-diagonal tiles.
// The left neighbour (t, j-1) of a cell is the lane's own previous step (u, y, y2 kept per row in VGPRs), the upper
// neighbour (t-1, j) is the previous row of the same step or, for the first row of a strip, the bottom row lane l-1
// finished one step earlier (one DPP wave_shr per state); query bases ride the same shift.
// This kernel runs at the VALU issue limit, so the cell is written for instruction count:
//  * one exec mask per step (j in range) instead of a predicate per cell;
//  * the substitution score is one byte permute: the query base travels as the word of its four scores (against A, C, G, T) and
//    each row keeps the selector of its target base;
//  * the constants of the recurrences (gap open/extend offsets, candidate ranks, a bias that keeps the candidates positive) are
//    folded into the stored states, so a candidate is ONE packed add and a new gap state ONE saturating packed subtract;
//  * the direction is "first operand that equals the maximum": the states are pre-scaled by 8 and the candidates carry their rank in
//    the low bits, so one max gives value and direction (stored as the rank, decoded by the traceback); the continuation flags come
//    from the new gap states;
//  * the corner score is summed along row 0 and then down the last column, which costs one add per step.
// Same recurrences, boundary rules and direction codes as ext_dp_kernel.  Directions are stored step-major: cell
// (t, j) lives at [j + t/S][t], so the S bytes a lane produces in one step are contiguous and the whole wave writes one
// contiguous row of n_lanes*S bytes per step (row-major dword stores cost 7x their bytes in HBM writes, measured).
__host__ __device__ inline bool ext_strip_scores_ok(int mch, int mis, int amb, int qe, int qe2) {
    // 8 * score + rank + 128 is a byte; with gap costs up to 127 every biased state and sum stays a positive 16-bit number
    return mch >= -16 && mch <= 15 && mis >= -16 && mis <= 15 && amb >= -16 && amb <= 15 && qe >= 0 && qe <= 127 && qe2 >= 0 && qe2 <= 127;
}
constexpr int STRIP_TAB_BYTES = 32;   // the query score words, behind the groups' queries in LDS
// EXACT windows (end extensions, exact global fills): is 8 H + 32768 a 16-bit number for every cell (see the H bookkeeping in
// ext_strip_pack)?  -(gap of t+1) - (gap of j+1) <= H(t, j) <= match * min(t+1, j+1); the tie-break key holds t in 10 bits.
__host__ __device__ inline bool ext_strip_exact_ok(int mch, int q, int e, int q2, int e2, int qlen, int tlen) {
    const int L = (qlen > tlen ? qlen : tlen) + 1;
    const int c1 = q + e * L, c2 = q2 + e2 * L;
    return tlen <= 1024 && 16 * (c1 < c2 ? c1 : c2) < 32000 && 8 * mch * L < 32000;
}

// EXACT = the ksw2 "exact" bookkeeping of the end extensions (and exact global fills): the maximum of every anti-diagonal with
// ksw2's tie order, the z-drop rule over the anti-diagonals in order, the best score of the last query column.  The systolic
// array does not visit the cells in anti-diagonal order, so the rule is applied AFTER the matrix is done:
//  * every cell carries its H (one packed add of the horizontal difference per cell; H(t, -1) is the closed form of the
//    first-column boundary) and merges (H, tie-break key) into its anti-diagonal's slot in LDS with one ds_max_u32: the word is
//    (8 H + 32768) << 10 | (8191 - key), key = ksw2's lane class << 10 | t, the band's last cell (which wins every tie) 8191;
//  * the last query column is what the lane's H registers hold when it stops;
//  * then the lanes of the group walk the anti-diagonals in chunks: running maximum by a prefix scan over the lanes, the first
//    anti-diagonal that z-drops by a minimum over the lanes, the state before it by a second scan.  Cells past the z-drop were
//    computed for nothing (an end-extension window is small); the traceback starts where ksw2's would.
// RIGHT = right-aligned gaps (KSW_EZ_RIGHT, the left extension): the later operand wins a tie -- the ranks of the five
// candidates are reversed (the traceback reads the operand index as the rank itself) -- and a gap continues when its state
// is >= 0, not > 0: the new gap state is computed one unit (8) high, flagged, and brought down by a second saturating subtract.
template <int S, int GL, bool EXACT, bool RIGHT>
__device__ __forceinline__ void ext_strip_pack(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, const int first,
                                               const int n_list, const ExtParams &prm, const uint8_t *__restrict__ reads,
                                               const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                               const RefView &rv, uint8_t *__restrict__ P, ExtRes *__restrict__ res, uint8_t *smem,
                                               const int lds_stride, const int nr_stride) {
    constexpr int NG = 64 / GL;  // windows per wave: each takes a group of GL lanes
    const int lane = threadIdx.x, g = lane / GL, gl = lane % GL;
    const int jid = first + g < n_list ? order[first + g] : -1;  // -1: padding at the end of a launch list
    int q = prm.q, e = prm.e, q2 = prm.q2, e2 = prm.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    // the group's window, one copy per lane
    int qlen = 0, tlen = 0, W = 0, rd = 0, rid = 0, rev = 0, qs = 0, ts = 0, back = 0, zdrop = -1, end_bonus = 0, jflag = 0;
    int64_t p_off = 0;
    if (jid >= 0) {
        const ExtJob &jb = jobs[jid];
        qlen = jb.qlen; tlen = jb.tlen; W = jb.qstride; rd = jb.read; rid = jb.rid; rev = jb.rev; qs = jb.qs; ts = jb.ts; p_off = jb.p_off;
        back = jb.reversed; zdrop = jb.zdrop; end_bonus = jb.end_bonus; jflag = jb.flag;
    }
    const bool ok = jid >= 0 && qlen > 0 && tlen > 0 && !(-prm.sc_mis > 2 * (q + e));
    if (!ok) { qlen = 0; tlen = 0; }
    // ranks of the five candidates in the low three bits: the max then prefers the earlier operand (score, a, b, a2, b2) on a
    // tie, or the later one for right-aligned gaps
    constexpr int RS = RIGHT ? 0 : 4, RA = RIGHT ? 1 : 3, RB = 2, RA2 = RIGHT ? 3 : 1, RB2 = RIGHT ? 4 : 0;
    // Scores travel as BYTES sb(s) = 8 s + rank + 128 (the candidate "diagonal" of a cell, biased to be unsigned): a
    // query base is the word of its four scores against target A, C, G, T (tab[base]; an ambiguous base scores sc_n against
    // everything), and a row picks its byte with one v_perm whose selector is the row's constant.
    const uint32_t sb_mch = (uint32_t)(8 * prm.sc_mch + RS + 128) & 0xff, sb_mis = (uint32_t)(8 * prm.sc_mis + RS + 128) & 0xff,
                   sb_n = (uint32_t)(8 * prm.sc_n + RS + 128) & 0xff;
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem + NG * lds_stride);   // (lds_stride is a multiple of 4)
    if (lane < 5) tab[lane] = lane == 4 ? sb_n * 0x01010101u : (sb_mis * 0x01010101u) ^ ((sb_mch ^ sb_mis) << (8 * lane));
    // queries -> LDS as 4 * base code (the byte offset of the base's word in tab), one region per group, staged by the whole wave
#pragma unroll
    for (int g2 = 0; g2 < NG; ++g2) {
        const int ql = __builtin_amdgcn_readlane(qlen, g2 * GL);
        if (ql > 0) {
            const int r2 = __builtin_amdgcn_readlane(rd, g2 * GL), rv2 = __builtin_amdgcn_readlane(rev, g2 * GL),
                      qs2 = __builtin_amdgcn_readlane(qs, g2 * GL), bk2 = __builtin_amdgcn_readlane(back, g2 * GL);
            const int64_t roff = read_off[r2];
            const int32_t rlen = read_len[r2];
            for (int i = lane; i < ql; i += 64)
                smem[g2 * lds_stride + i] = (uint8_t)(4 * ext_qbase(reads, roff, rlen, rv2, qs2 + (bk2 ? ql - 1 - i : i)));
        }
    }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    // All states are kept PRE-SCALED by 8: the low three bits of the five candidates of a cell then carry their rank, so ONE
    // max yields both the cell's value and which operand won (the direction) -- no compare/select chain.  Differences stay
    // far below 2^12, so the 16-bit halves do not overflow.
#define MPN_BND(R) (8 * ((R) == 0 ? -qe : (R) < long_thres ? -e : (R) == long_thres ? long_diff : -e2))
    // Difference states as pairs of 16-bit lanes of one register (they are small integers): the two gap types of a cell go
    // through the packed 16-bit instructions together.  The candidates of the max all carry the bias BETA (so that they are
    // positive: the new gap states max(0, c - z - e) are then ONE unsigned saturating subtraction each), and the constants of
    // the recurrences are folded into the stored states:
    //   UL: u of the previous column + CU, both halves the same u;   YL: (y + (q+e) | y2 + (q2+e2)) of the previous column
    //   Vp: v of the row above + CV;                                 Xp: (x + (q+e) | x2 + (q2+e2)), both start at 0
    // with CV = (-8(q+e) + rank a | -8(q2+e2) + rank a2) + BETA and CU = (-8(q+e) + rank b | -8(q2+e2) + rank b2) + BETA, so that
    // the candidates are plain sums: 8 (a | a2) + rank + BETA = Xp + Vp and 8 (b | b2) + rank + BETA = YL + UL.
    // Every half of every state and candidate is a POSITIVE 16-bit number (the bias), and so is every sum or difference the
    // recurrences form: a plain 32-bit add or subtract of two registers therefore adds or subtracts the halves exactly -- no carry
    // or borrow crosses bit 16 -- and v_add_u32 / v_sub_u32 / v_and_b32 issue in 2.7 cycles per wave on gfx950 where the packed
    // 16-bit and all three-operand instructions take 4.4 (profiles/r03/valu_microbench2.txt).  Only max, min and the saturating
    // subtract, which need the halves kept apart, are packed instructions.  pk(lo, hi) is the register with the given halves,
    // formed as a signed sum so that a negative constant in the low half borrows from the high one as the 32-bit add expects.
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pk = [](int lo, int hi) { return (uint32_t)(hi * 65536 + lo); };
    constexpr int BETA = 0x2000 + 128;   // the score byte's 128 and the 0x20 the perm puts above it
    const int cv_lo = -8 * qe + RA + BETA, cv_hi = -8 * qe2 + RA2 + BETA, cu_lo = -8 * qe + RB + BETA, cu_hi = -8 * qe2 + RB2 + BETA;
    const uint32_t CV = pk(cv_lo, cv_hi), CU = pk(cu_lo, cu_hi);
    uint32_t UL[S], YL[S];
    uint32_t TSEL[S];
    uint32_t Hh[EXACT ? S : 1];   // EXACT: low half = 8 H(t, j) + (j + 1) CV.lo of the lane's current column (mod 2^16)
    const int64_t g0 = ok ? rv.seq_off[rid] + ts : 0;
    const int t0 = gl * S;
    const uint32_t KONST = sb_n | 0x2000u;   // byte 0: the score against an ambiguous target base; byte 1: the high byte of every score
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int t = t0 + k;
        const int sq = t < tlen ? ref_code(rv, g0 + (back ? tlen - 1 - t : t)) : 4;
        TSEL[k] = 0x0c0c0100u | (sq < 4 ? 4u + (uint32_t)sq : 0u);   // v_perm(QT, KONST): byte 0 = QT[sq] or KONST[0], byte 1 = KONST[1]
        UL[k] = pk(MPN_BND(t), MPN_BND(t)) + CU;   // left of column 0: the first-column boundary (u of anti-diagonal r = t)
        YL[k] = 0;
        if constexpr (EXACT) {
            // H(t, -1): the sum of the boundary differences above
            const int h = t < long_thres ? -(q + e * (t + 1)) : -(q2 + e2 * (t + 1));
            Hh[k] = (uint32_t)(8 * h) & 0xffffu;
        }
    }
    // EXACT: per group, a slot per anti-diagonal (BEST) and the table E4[r] = (cells of anti-diagonal r - 1) & ~3, the part of the
    // band ksw2 covers with whole 4-lane vectors; E4's space holds the last query column afterwards
    uint32_t *BEST = nullptr, *E4T = nullptr;
    const int n_r = ok ? qlen + tlen - 1 : 0;
    if constexpr (EXACT) {
        BEST = reinterpret_cast<uint32_t *>(smem + NG * lds_stride + STRIP_TAB_BYTES) + (size_t)g * 2 * nr_stride;
        E4T = BEST + nr_stride;
        const int mn = (qlen < tlen ? qlen : tlen) - 1;
        for (int r = gl; r < nr_stride; r += GL) {
            const int c = min(min(r, n_r - 1 - r), mn);
            BEST[r] = 0;
            E4T[r] = c > 0 ? (uint32_t)c & ~3u : 0u;
        }
    }
    __syncthreads();
    const int n_lanes = (tlen + S - 1) / S;
    const int max_steps = wave_reduce_max(ok ? qlen + n_lanes - 1 : 0);
    const uint8_t *qrow = smem + g * lds_stride;
    uint8_t *prow = P + p_off + t0;
    int out_v = 0, out_x = 0;
    uint32_t qt = 0;   // the score word of the query base this lane works on
    int32_t row0 = 0;  // first lane of a group: sum of the horizontal differences of row 0
    // z + CU + CV (what the new u and v are subtracted from), and z + e + the offset of the candidate a gap state comes from
    // (right-aligned gaps: one unit lower, so that the flag sees state >= 0)
    constexpr int RU = RIGHT ? 8 : 0;
    const uint32_t KZZ = pk(-16 * qe + RA + RB + BETA, -16 * qe2 + RA2 + RB2 + BETA);
    const uint32_t KEA = pk(8 * e - 8 * qe + RA - RU, 8 * e2 - 8 * qe2 + RA2 - RU), KEB = pk(8 * e - 8 * qe + RB - RU, 8 * e2 - 8 * qe2 + RB2 - RU);
    const uint32_t EIGHT = 0x00080008u;
    const uint32_t RANK_CLR = 0xfff8fff8u;
    const uint32_t MCH7 = (uint32_t)(8 * prm.sc_mch + 7 + BETA) * 0x00010001u;
    const bool head = gl == 0;
    const int tlm1 = tlen - 1;
    // the head lane's query words, fetched two steps ahead (base code, then its word): two dependent LDS reads off the critical path
    auto qcode = [&](int s_) -> uint32_t { return s_ < qlen ? (uint32_t)qrow[s_] : 16u; };
    uint32_t qt_next = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + qcode(0));
    uint32_t qc_next = qcode(1);
    for (int step = 0; step < max_steps; ++step) {
        // query bases and bottom-row states move one lane to the right; the first lane of a group takes the boundary
        const uint32_t q_in = qt_next;
        qt_next = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + qc_next);
        qc_next = qcode(step + 2);
        const uint32_t qt_s = (uint32_t)wave_shr1_zero((int)qt);
        const int v_s = wave_shr1_zero(out_v), x_s = wave_shr1_zero(out_x);
        const int j = step - gl;
        const int bj = MPN_BND(step);  // first lane: j = step
        qt = head ? q_in : qt_s;
        uint32_t Vp = head ? pk(bj, bj) + CV : (uint32_t)v_s, Xp = head ? 0u : (uint32_t)x_s;
        if (j >= 0 && j < qlen && gl < n_lanes) {
            uint32_t dw[(S + 3) / 4], ecell[4] = {0, 0, 0, 0};
            int nv0 = 0;
            // EXACT: what turns a lane's H register into 8 H + 32768 in this column, t - st of the rows below the query's end,
            // the anti-diagonal slots of the lane's first row
            const uint32_t offj = (uint32_t)((j + 1) * cv_lo - 32768);
            const int qm1j = qlen - 1 - j;
            uint32_t *bslot = EXACT ? BEST + (t0 + j) : nullptr;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const uint32_t sc16 = __builtin_amdgcn_perm(qt, KONST, TSEL[k]);   // 8 s + rank + BETA in the low half
                const uint32_t Up = UL[k];
                const uint32_t A = Xp + Vp, B = YL[k] + Up;      // 8 (a | a2) + ranks + BETA, 8 (b | b2) + ranks + BETA
                const uint32_t M = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, A), __builtin_bit_cast(s16x2, B)));
                // max of M's two halves (a packed max of M with its halves swapped) and the score (the two-operand 16-bit max is a
                // fast instruction; the three-operand one takes 8.3 cycles); the result is the low 16 bits, everything after reads those
                uint32_t m2, z16;
                asm("v_pk_max_i16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m2) : "v"(M));
                asm("v_max_i16 %0, %1, %2" : "=v"(z16) : "v"(m2), "v"(sc16));
                // both halves = min(z, match score) (the packed min reads z's low half for both), rank bits cleared
                uint32_t zc;
                asm("v_pk_min_i16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(zc) : "v"(z16), "v"(MCH7));
                const uint32_t Zc = zc & RANK_CLR;               // 8 z + BETA
                const uint32_t ZZ = Zc + KZZ;
                // new v = ZZ - (old u), new u = ZZ - (v from above), with the new u written over the old one (one asm statement
                // says so: left alone, the compiler puts the new v there and copies all S new u back at the end of every step)
                uint32_t nu = Up, nv;
                asm("v_sub_u32 %1, %2, %0\n\tv_sub_u32 %0, %2, %3" : "+v"(nu), "=&v"(nv) : "v"(ZZ), "v"(Vp));
                // new gap states max(0, candidate - z - e): the candidates and z + e (+ the candidate's offset) are positive
                u16x2 An = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, A), __builtin_bit_cast(u16x2, Zc + KEA));
                u16x2 Bn = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, B), __builtin_bit_cast(u16x2, Zc + KEB));
                // Continuation flags (x > 0, or x >= 0 for right-aligned gaps: there An, Bn are still one unit high) of the four
                // gap states, which are multiples of 8: min(x, 8) as unsigned leaves bit 3
                // of each half (written as the instruction: the compiler turns the expression into compares and selects).
                // F: bit 3 a, 4 b, 19 a2, 20 b2; the cell's byte is rank | F | F >> 14, assembled four cells at a time below.
                uint32_t HA, HB;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(HA) : "v"(__builtin_bit_cast(uint32_t, An)), "v"(EIGHT));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(HB) : "v"(__builtin_bit_cast(uint32_t, Bn)), "v"(EIGHT));
                if constexpr (RIGHT) {
                    An = __builtin_elementwise_sub_sat(An, __builtin_bit_cast(u16x2, EIGHT));
                    Bn = __builtin_elementwise_sub_sat(Bn, __builtin_bit_cast(u16x2, EIGHT));
                }
                // rank of the winner (the traceback reads the operand index as 4 - rank, or as the rank for right-aligned gaps)
                // and F.  (Written as the two instructions: left to itself the compiler spreads the shifts of this and of the
                // packing below over more of them.)
                uint32_t Fw;
                asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(Fw) : "v"(HB), "v"(HA));
                asm("v_and_or_b32 %0, %1, 7, %2" : "=v"(ecell[k & 3]) : "v"(z16), "v"(Fw));
                UL[k] = nu; YL[k] = __builtin_bit_cast(uint32_t, Bn);
                Vp = nv; Xp = __builtin_bit_cast(uint32_t, An);
                if constexpr (EXACT) {
                    Hh[k] += nv;
                    const uint32_t hb = (Hh[k] - offj) & 0xffffu;            // 8 H(t, j) + 32768
                    const int t = t0 + k, m = min(t, qm1j);                  // m = t - (first cell of the anti-diagonal)
                    uint32_t inv = (m < (int)E4T[t0 + j + k] ? ~((uint32_t)m << 10) & 0xc00u : (uint32_t)-1024) + (uint32_t)(4095 - t);
                    inv = (j == 0 || t == tlm1) ? 8191u : inv;                // the band's last cell
                    atomicMax(bslot + k, t <= tlm1 ? hb * 1024u + inv : 0u);  // (0: a row that pads the last strip)
                }
                if ((k & 3) == 3 || k == S - 1) {
                    // bytes 0 (rank, a, b) and bytes 2 (a2, b2) of up to four cells -> one word each, then hi << 2 joins lo
                    uint32_t e01 = ecell[0], e23 = (k & 3) >= 2 ? ecell[2] : 0u;
                    if ((k & 3) >= 1) asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(e01) : "v"(ecell[1]), "v"(ecell[0]));
                    if ((k & 3) == 3) asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(e23) : "v"(ecell[3]), "v"(ecell[2]));
                    const uint32_t lo = __builtin_amdgcn_perm(e23, e01, 0x05040100u), hi = __builtin_amdgcn_perm(e23, e01, 0x07060302u);
                    asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(dw[k >> 2]) : "v"(hi), "v"(lo));
                }
                if (k == 0) nv0 = (int)(nv & 0xffffu);
            }
            out_v = (int)Vp; out_x = (int)Xp;
            row0 += nv0 - cv_lo;
            uint8_t *dst = prow + (int64_t)step * W;  // p_off is 16-aligned, W and t0 are multiples of S
            if constexpr (S == 4) *reinterpret_cast<uint32_t *>(dst) = dw[0];
            else if constexpr (S == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(dw[0], dw[1]);
            else if constexpr (S == 16) *reinterpret_cast<uint4 *>(dst) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            else {  // other strip heights: 32-bit pieces at the natural alignment of lane*S (global memory takes unaligned stores)
#pragma unroll
                for (int c = 0; c < S / 4; ++c) __builtin_memcpy(dst + 4 * c, &dw[c], 4);
                if constexpr ((S & 2) != 0) { const uint16_t h = (uint16_t)dw[S / 4]; __builtin_memcpy(dst + (S & ~3), &h, 2); }
                if constexpr ((S & 1) != 0) dst[S - 1] = (uint8_t)(dw[S / 4] >> ((S & 2) ? 16 : 0));
            }
        }
    }
#undef MPN_BND
    if constexpr (!EXACT) {
        // H(tlen-1, qlen-1) = H(0,-1) + sum_j v(0,j) + sum_{t>0} u(t, qlen-1); a lane's UL froze at its last column
        // (the sums are of pre-scaled differences, exact multiples of 8)
        int32_t tot = head ? row0 - 8 * qe : 0;
#pragma unroll
        for (int k = 0; k < S; ++k) tot += (t0 + k > 0 && t0 + k < tlen) ? (int)(UL[k] & 0xffffu) - cu_lo : 0;
#pragma unroll
        for (int dlt = GL / 2; dlt; dlt >>= 1) tot += __shfl_xor(tot, dlt);
        tot >>= 3;
        if (head && jid >= 0) {
            ExtRes out;
            out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = ok ? tot : NEG_INF;
            out.reach_end = 0; out.n_cigar = 0; out.r_done = ok ? qlen + tlen - 2 : -1; out.zcode = 0; out.cig_pos = 0;
            out.do_bt = ok ? 1 : 0; out.bt_i = ok ? tlen - 1 : -1; out.bt_j = ok ? qlen - 1 : -1;
            res[jid] = out;
        }
    } else {
        // ---- the z-drop rule over the anti-diagonals, in order ----
        __syncthreads();   // every slot has its maximum
        // the last query column: a lane's H registers froze there.  HL[t] = H(t, qlen - 1) + 4096, over the E4 table
        uint32_t *HL = E4T;
        {
            const uint32_t offl = (uint32_t)(qlen * cv_lo - 32768);
#pragma unroll
            for (int k = 0; k < S; ++k) if (t0 + k < tlen) HL[t0 + k] = ((Hh[k] - offl) & 0xffffu) >> 3;
        }
        __syncthreads();
        struct Best { int m, t, r; };   // a maximum, its target position and its anti-diagonal (its place in walking order)
        // first strict maximum in walking order over the lanes of the group: inclusive scan, the later lane wins only when greater
        auto scan_first_max = [&](Best x) {
#pragma unroll
            for (int d = 1; d < GL; d <<= 1) {
                const Best o{__shfl_up(x.m, d, GL), __shfl_up(x.t, d, GL), __shfl_up(x.r, d, GL)};
                if (gl >= d && !(x.m > o.m)) x = o;
            }
            return x;
        };
        auto diag = [&](int r, int &H, int &mt) {
            const uint32_t pv = BEST[r];
            const uint32_t inv = pv & 8191u;
            H = (int)(pv >> 13) - 4096;
            mt = inv == 8191u ? min(tlm1, r) : (int)((8191u - inv) & 1023u);
        };
        const int C = (n_r + GL - 1) / GL, rlo = min(n_r, gl * C), rhi = min(n_r, rlo + C);
        // (1) the running maximum that enters every lane's chunk
        auto chunk_best = [&](int lim) {
            Best b{NEG_INF, -1, -1};
            for (int r = rlo; r < rhi && r <= lim; ++r) { int H, mt; diag(r, H, mt); if (H > b.m) b = Best{H, mt, r}; }
            return b;
        };
        Best incl = scan_first_max(chunk_best(n_r));
        Best in{__shfl_up(incl.m, 1, GL), __shfl_up(incl.t, 1, GL), __shfl_up(incl.r, 1, GL)};
        ExtApply ez; ez.max = 0; ez.max_t = ez.max_q = -1; ez.zdropped = 0;
        if (gl > 0 && in.m > 0) { ez.max = in.m; ez.max_t = in.t; ez.max_q = in.r - in.t; }
        // (2) the first anti-diagonal that z-drops
        int r_break = n_r;
        for (int r = rlo; r < rhi; ++r) {
            int H, mt; diag(r, H, mt);
            if (ext_apply_zdrop(ez, H, r, mt, zdrop, e2)) { r_break = r; break; }
        }
#pragma unroll
        for (int dlt = GL / 2; dlt; dlt >>= 1) r_break = min(r_break, __shfl_xor(r_break, dlt));
        const bool dropped = r_break < n_r;
        // (3) the state when the walk stops: maximum over the anti-diagonals up to there, best cell of the last query column
        // (H(st, qlen - 1) of the anti-diagonals qlen - 1 .. r_break: target positions 0 .. r_break - qlen + 1)
        Best fin = scan_first_max(chunk_best(r_break));
        fin = Best{__shfl(fin.m, GL - 1, GL), __shfl(fin.t, GL - 1, GL), __shfl(fin.r, GL - 1, GL)};
        const int t_lim = min(tlm1, r_break - (qlen - 1));
        const int Ct = (tlen + GL - 1) / GL, tlo = min(tlen, gl * Ct), thi = min(tlen, tlo + Ct);
        Best me{NEG_INF, -1, -1};
        for (int t = tlo; t < thi && t <= t_lim; ++t) { const int h = (int)HL[t] - 4096; if (h > me.m) me = Best{h, t, t}; }
        me = scan_first_max(me);
        me = Best{__shfl(me.m, GL - 1, GL), __shfl(me.t, GL - 1, GL), 0};
        if (head && jid >= 0) {
            ExtRes out;
            out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = NEG_INF;
            out.reach_end = 0; out.n_cigar = 0; out.r_done = -1; out.bt_i = out.bt_j = -1; out.do_bt = 0; out.zcode = 0; out.cig_pos = 0;
            if (ok) {
                if (fin.m > 0) { out.max = fin.m; out.max_t = fin.t; out.max_q = fin.r - fin.t; }
                out.zdropped = dropped ? 1 : 0;
                if (me.m > NEG_INF) { out.mqe = me.m; out.mqe_t = me.t; }
                if (!dropped) out.score = (int)HL[tlm1] - 4096;
                out.r_done = dropped ? r_break : n_r - 1;
                if (!dropped && !(jflag & EZ_EXTZ_ONLY)) { out.do_bt = 1; out.bt_i = tlen - 1; out.bt_j = qlen - 1; }
                else if (!dropped && (jflag & EZ_EXTZ_ONLY) && out.mqe + end_bonus > out.max) { out.reach_end = 1; out.do_bt = 1; out.bt_i = out.mqe_t; out.bt_j = qlen - 1; }
                else if (out.max_t >= 0 && out.max_q >= 0) { out.do_bt = 1; out.bt_i = out.max_t; out.bt_j = out.max_q; }
            }
            res[jid] = out;
        }
    }
}

// TWO gap-fill windows per lane group (round 4): the 16-bit halves of a register hold the same state of window A (low) and window B
// (high) instead of the two gap types of one window, so every packed instruction -- and the score / maximum / clamp / difference
// chain, which is scalar in the one-window cell -- works on two cells: 16.5 instead of 21.5 instructions per cell.  The two gap
// types live in registers of their own (X1, X2 from above; Y1, Y2 per row); u and v carry the constants of type 1 and the type-2
// candidates come out too large by the constant D = 8 (qe2 - qe) + 2 (rank a - rank a2 = rank b - rank b2 = 2), which is taken off
// their maximum once and folded into the constants of their saturating subtractions.  The windows of a pair are neighbours in
// the launch list (same strip height, queries within one length bucket); each keeps its own matrix, stride and result.  A half
// whose window is shorter keeps computing (its query continues with ambiguous bases, its target with ambiguous rows: a valid DP of
// bounded differences, so nothing carries into the other half); what a window needs from its last column and row 0 is taken when
// the lane is there.  Gap fills only (approximate maximum, left-aligned gaps).
template <int S, int GL>
__device__ __forceinline__ void ext_strip_pair(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, const int first,
                                               const int n_list, const ExtParams &prm, const uint8_t *__restrict__ reads,
                                               const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                               const RefView &rv, uint8_t *__restrict__ P, ExtRes *__restrict__ res, uint8_t *smem,
                                               const int lds_stride) {
    constexpr int NG = 64 / GL;  // pairs per wave
    const int lane = threadIdx.x, g = lane / GL, gl = lane % GL;
    int q = prm.q, e = prm.e, q2 = prm.q2, e2 = prm.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    // the pair's windows, one copy per lane ([0] = A, [1] = B)
    int jid[2], qlen[2], tlen[2], W[2], rd[2], rid[2], rev[2], qs[2], ts[2], back[2];
    int64_t p_off[2];
    bool ok[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int slot = first + 2 * g + w;
        jid[w] = slot < n_list ? order[slot] : -1;  // -1: padding at the end of a launch list
        qlen[w] = tlen[w] = W[w] = rd[w] = rid[w] = rev[w] = qs[w] = ts[w] = back[w] = 0; p_off[w] = 0;
        if (jid[w] >= 0) {
            const ExtJob &jb = jobs[jid[w]];
            qlen[w] = jb.qlen; tlen[w] = jb.tlen; W[w] = jb.qstride; rd[w] = jb.read; rid[w] = jb.rid; rev[w] = jb.rev; qs[w] = jb.qs; ts[w] = jb.ts;
            p_off[w] = jb.p_off; back[w] = jb.reversed;
        }
        ok[w] = jid[w] >= 0 && qlen[w] > 0 && tlen[w] > 0 && !(-prm.sc_mis > 2 * (q + e));
        if (!ok[w]) { qlen[w] = 0; tlen[w] = 0; }
    }
    constexpr int RS = 4, RA = 3, RB = 2, RA2 = 1, RB2 = 0;
    const uint32_t sb_mch = (uint32_t)(8 * prm.sc_mch + RS + 128) & 0xff, sb_mis = (uint32_t)(8 * prm.sc_mis + RS + 128) & 0xff,
                   sb_n = (uint32_t)(8 * prm.sc_n + RS + 128) & 0xff;
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem + 2 * NG * lds_stride);   // (lds_stride is a multiple of 4)
    if (lane < 5) tab[lane] = lane == 4 ? sb_n * 0x01010101u : (sb_mis * 0x01010101u) ^ ((sb_mch ^ sb_mis) << (8 * lane));
    if (lane == 5) tab[5] = 0;   // what the lanes that are not the first of their group read (code 20): see the query words below
    // queries -> LDS as 4 * base code, one region per window, staged by the whole wave
#pragma unroll
    for (int g2 = 0; g2 < NG; ++g2)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int ql = __builtin_amdgcn_readlane(qlen[w], g2 * GL);
            if (ql > 0) {
                const int r2 = __builtin_amdgcn_readlane(rd[w], g2 * GL), rv2 = __builtin_amdgcn_readlane(rev[w], g2 * GL),
                          qs2 = __builtin_amdgcn_readlane(qs[w], g2 * GL), bk2 = __builtin_amdgcn_readlane(back[w], g2 * GL);
                const int64_t roff = read_off[r2];
                const int32_t rlen = read_len[r2];
                for (int i = lane; i < ql; i += 64)
                    smem[(2 * g2 + w) * lds_stride + i] = (uint8_t)(4 * ext_qbase(reads, roff, rlen, rv2, qs2 + (bk2 ? ql - 1 - i : i)));
            }
        }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
#define MPN_BND(R) (8 * ((R) == 0 ? -qe : (R) < long_thres ? -e : (R) == long_thres ? long_diff : -e2))
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    constexpr int BETA = 0x2000 + 128;
    const int cv1 = -8 * qe + RA + BETA, cu1 = -8 * qe + RB + BETA;
    const uint32_t both = 0x00010001u;
    const uint32_t CV = (uint32_t)cv1 * both, CU = (uint32_t)cu1 * both;
    const int Dk = 8 * (qe2 - qe) + (RA - RA2);    // (RB - RB2 is the same 2)
    const uint32_t DD = (uint32_t)Dk * both;
    const uint32_t KZZ = (uint32_t)(-16 * qe + RA + RB + BETA) * both;
    const uint32_t KEA1 = (uint32_t)(8 * e - 8 * qe + RA) * both, KEB1 = (uint32_t)(8 * e - 8 * qe + RB) * both;
    const uint32_t KEA2 = (uint32_t)(8 * e2 - 8 * qe2 + RA2 + Dk) * both, KEB2 = (uint32_t)(8 * e2 - 8 * qe2 + RB2 + Dk) * both;
    const uint32_t EIGHT = 0x00080008u, RANK_CLR = 0xfff8fff8u, SEVEN = 0x00070007u;
    const uint32_t MCH7 = (uint32_t)(8 * prm.sc_mch + 7 + BETA) * both;
    uint32_t U[S], Y1[S], Y2[S], T1[S], T2[S];
    const int t0 = gl * S;
    const uint32_t KONST = sb_n | 0x2000u;   // byte 0: the score against an ambiguous target base; byte 1: the high byte of every score
    {
        const int64_t ga = ok[0] ? rv.seq_off[rid[0]] + ts[0] : 0, gb = ok[1] ? rv.seq_off[rid[1]] + ts[1] : 0;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int t = t0 + k;
            const int sa = t < tlen[0] ? ref_code(rv, ga + (back[0] ? tlen[0] - 1 - t : t)) : 4;
            const int sb_ = t < tlen[1] ? ref_code(rv, gb + (back[1] ? tlen[1] - 1 - t : t)) : 4;
            T1[k] = 0x0c0c0100u | (sa < 4 ? 4u + (uint32_t)sa : 0u);                 // -> low half: byte 0 the score, byte 1 its high byte
            T2[k] = 0x01000c0cu | (sb_ < 4 ? 4u + (uint32_t)sb_ : 0u) << 16;         // -> high half
            U[k] = (uint32_t)MPN_BND(t) * both + CU;   // left of column 0: the first-column boundary (the same for both windows)
            Y1[k] = 0; Y2[k] = 0;
        }
    }
    __syncthreads();
    const int nl0 = (tlen[0] + S - 1) / S, nl1 = (tlen[1] + S - 1) / S;
    const int qmax = max(qlen[0], qlen[1]), nlmax = max(nl0, nl1);
    const int max_steps = wave_reduce_max(qmax > 0 ? qmax + nlmax - 1 : 0);
    const uint8_t *qrow0 = smem + (2 * g) * lds_stride, *qrow1 = qrow0 + lds_stride;
    uint8_t *prow0 = P + p_off[0] + t0, *prow1 = P + p_off[1] + t0;
    int out_v = 0, out_x1 = 0, out_x2 = 0;
    uint32_t qta = 0, qtb = 0;       // the score words of the query bases this lane works on
    int32_t row0a = 0, row0b = 0;    // first lane of a group: sums of the horizontal differences of row 0
    int32_t tota = 0, totb = 0;      // sum of the vertical differences of the lane's rows in the window's last column
    const bool head = gl == 0;
    const int qa1 = qlen[0] - 1, qb1 = qlen[1] - 1;
    // The first lane of a group takes the query words from LDS and the boundary states, the others what the lane before them hands
    // on: as masks instead of selects (v_cndmask issues in ~17 cycles on gfx950, profiles/r03/valu_microbench2.txt) -- the other lanes
    // read a zero word where the first reads its query word, and the shifted states are and-ed with the not-first mask.
    const uint32_t nh = head ? 0u : 0xffffffffu;
    const int qh0 = head ? qlen[0] : 0, qh1 = head ? qlen[1] : 0;
    const uint32_t past = head ? 16u : 20u;
    auto code0 = [&](int s_) -> uint32_t { return s_ < qh0 ? (uint32_t)qrow0[s_] : past; };
    auto code1 = [&](int s_) -> uint32_t { return s_ < qh1 ? (uint32_t)qrow1[s_] : past; };
    auto word = [&](uint32_t c) -> uint32_t { return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + c); };
    uint32_t qa_next = word(code0(0)), qb_next = word(code1(0));
    uint32_t ca_next = code0(1), cb_next = code1(1);
    for (int step = 0; step < max_steps; ++step) {
        const uint32_t qa_in = qa_next, qb_in = qb_next;
        qa_next = word(ca_next); qb_next = word(cb_next);
        ca_next = code0(step + 2); cb_next = code1(step + 2);
        const uint32_t qa_s = (uint32_t)wave_shr1_zero((int)qta), qb_s = (uint32_t)wave_shr1_zero((int)qtb);
        const int v_s = wave_shr1_zero(out_v), x1_s = wave_shr1_zero(out_x1), x2_s = wave_shr1_zero(out_x2);
        const int j = step - gl;
        const int bj = MPN_BND(step);  // first lane: j = step
        qta = (qa_s & nh) | qa_in; qtb = (qb_s & nh) | qb_in;
        uint32_t Vp = ((uint32_t)v_s & nh) | (((uint32_t)bj * both + CV) & ~nh), X1 = (uint32_t)x1_s & nh, X2 = (uint32_t)x2_s & nh;
        if (j >= 0 && j < qmax && gl < nlmax) {
            uint32_t dwa[(S + 3) / 4], dwb[(S + 3) / 4], ecell[4] = {0, 0, 0, 0};
            uint32_t nv0 = 0;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const uint32_t sc = __builtin_amdgcn_perm(qta, KONST, T1[k]) | __builtin_amdgcn_perm(qtb, KONST, T2[k]);   // 8 s + rank + BETA per half
                const uint32_t Up = U[k];
                const uint32_t A1 = X1 + Vp, A2 = X2 + Vp, B1 = Y1[k] + Up, B2 = Y2[k] + Up;   // A2, B2: D too large
                const uint32_t M1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, A1), __builtin_bit_cast(s16x2, B1)));
                const uint32_t M2 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, A2), __builtin_bit_cast(s16x2, B2))) - DD;
                const uint32_t M = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, M1), __builtin_bit_cast(s16x2, M2)));
                const uint32_t z = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, M), __builtin_bit_cast(s16x2, sc)));
                const uint32_t zc = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, z), __builtin_bit_cast(s16x2, MCH7)));
                const uint32_t Zc = zc & RANK_CLR;               // 8 z + BETA
                const uint32_t ZZ = Zc + KZZ;
                uint32_t nu = Up, nv;
                asm("v_sub_u32 %1, %2, %0\n\tv_sub_u32 %0, %2, %3" : "+v"(nu), "=&v"(nv) : "v"(ZZ), "v"(Vp));
                const u16x2 An1 = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, A1), __builtin_bit_cast(u16x2, Zc + KEA1));
                const u16x2 An2 = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, A2), __builtin_bit_cast(u16x2, Zc + KEA2));
                const u16x2 Bn1 = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, B1), __builtin_bit_cast(u16x2, Zc + KEB1));
                const u16x2 Bn2 = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, B2), __builtin_bit_cast(u16x2, Zc + KEB2));
                // continuation flags: min(state, 8) leaves bit 3 of each half; a -> bit 3, b -> 4, a2 -> 5, b2 -> 6; rank in bits 0..2
                uint32_t HA1, HB1, HA2, HB2, F;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(HA1) : "v"(__builtin_bit_cast(uint32_t, An1)), "v"(EIGHT));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(HB1) : "v"(__builtin_bit_cast(uint32_t, Bn1)), "v"(EIGHT));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(HA2) : "v"(__builtin_bit_cast(uint32_t, An2)), "v"(EIGHT));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(HB2) : "v"(__builtin_bit_cast(uint32_t, Bn2)), "v"(EIGHT));
                asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(F) : "v"(HB1), "v"(HA1));
                asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(F) : "v"(HA2), "v"(F));
                asm("v_lshl_or_b32 %0, %1, 3, %2" : "=v"(F) : "v"(HB2), "v"(F));
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ecell[k & 3]) : "v"(z), "v"(SEVEN), "v"(F));
                U[k] = nu; Y1[k] = __builtin_bit_cast(uint32_t, Bn1); Y2[k] = __builtin_bit_cast(uint32_t, Bn2);
                Vp = nv; X1 = __builtin_bit_cast(uint32_t, An1); X2 = __builtin_bit_cast(uint32_t, An2);
                if ((k & 3) == 3 || k == S - 1) {
                    // a cell word holds window A's byte in byte 0 and window B's in byte 2: four rows -> one word per window
                    const uint32_t p01 = (k & 3) >= 1 ? __builtin_amdgcn_perm(ecell[1], ecell[0], 0x06020400u) : __builtin_amdgcn_perm(0u, ecell[0], 0x0c020c00u);
                    const uint32_t p23 = (k & 3) == 3 ? __builtin_amdgcn_perm(ecell[3], ecell[2], 0x06020400u)
                                       : (k & 3) == 2 ? __builtin_amdgcn_perm(0u, ecell[2], 0x0c020c00u) : 0u;
                    dwa[k >> 2] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    dwb[k >> 2] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
                }
                if (k == 0) nv0 = nv;
            }
            out_v = (int)Vp; out_x1 = (int)X1; out_x2 = (int)X2;
            if (j <= qa1) row0a += (int)(nv0 & 0xffffu) - cv1;
            if (j <= qb1) row0b += (int)(nv0 >> 16) - cv1;
            // the window's last column: the vertical differences of the lane's rows (pre-scaled, exact multiples of 8)
            // (one lane of a group is there per step, so the wave runs this at most steps of a drain: a lane whose S rows are all inside
            // the window -- every lane but the first and the last -- adds them up as packed halves modulo 2^16, S - 1 instructions)
            auto packed_sum = [&]() -> uint32_t {
                u16x2 acc = __builtin_bit_cast(u16x2, U[0]);
#pragma unroll
                for (int k = 1; k < S; ++k) acc += __builtin_bit_cast(u16x2, U[k]);
                return __builtin_bit_cast(uint32_t, acc);
            };
            if (j == qa1) {
                if (!head && t0 + S <= tlen[0]) tota += (int)(short)(uint16_t)((packed_sum() & 0xffffu) - (uint32_t)(S * cu1));
                else {
#pragma unroll
                    for (int k = 0; k < S; ++k) tota += (t0 + k > 0 && t0 + k < tlen[0]) ? (int)(U[k] & 0xffffu) - cu1 : 0;
                }
            }
            if (j == qb1) {
                if (!head && t0 + S <= tlen[1]) totb += (int)(short)(uint16_t)((packed_sum() >> 16) - (uint32_t)(S * cu1));
                else {
#pragma unroll
                    for (int k = 0; k < S; ++k) totb += (t0 + k > 0 && t0 + k < tlen[1]) ? (int)(U[k] >> 16) - cu1 : 0;
                }
            }
            auto store = [&](uint8_t *dst, const uint32_t *dw) {   // p_off is 16-aligned, W and t0 are multiples of S
                if constexpr (S == 4) *reinterpret_cast<uint32_t *>(dst) = dw[0];
                else if constexpr (S == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(dw[0], dw[1]);
                else if constexpr (S == 16) *reinterpret_cast<uint4 *>(dst) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
                else {
#pragma unroll
                    for (int c = 0; c < S / 4; ++c) __builtin_memcpy(dst + 4 * c, &dw[c], 4);
                    if constexpr ((S & 2) != 0) { const uint16_t h = (uint16_t)dw[S / 4]; __builtin_memcpy(dst + (S & ~3), &h, 2); }
                    if constexpr ((S & 1) != 0) dst[S - 1] = (uint8_t)(dw[S / 4] >> ((S & 2) ? 16 : 0));
                }
            };
            if (j <= qa1 && gl < nl0) store(prow0 + (int64_t)step * W[0], dwa);
            if (j <= qb1 && gl < nl1) store(prow1 + (int64_t)step * W[1], dwb);
        }
    }
#undef MPN_BND
    // H(tlen-1, qlen-1) = H(0,-1) + sum_j v(0,j) + sum_{t>0} u(t, qlen-1)
    int32_t ta = tota + (head ? row0a - 8 * qe : 0), tb = totb + (head ? row0b - 8 * qe : 0);
#pragma unroll
    for (int dlt = GL / 2; dlt; dlt >>= 1) { ta += __shfl_xor(ta, dlt); tb += __shfl_xor(tb, dlt); }
    ta >>= 3; tb >>= 3;
    if (head) {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            if (jid[w] < 0) continue;
            ExtRes out;
            out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = ok[w] ? (w ? tb : ta) : NEG_INF;
            out.reach_end = 0; out.n_cigar = 0; out.r_done = ok[w] ? qlen[w] + tlen[w] - 2 : -1; out.zcode = 0; out.cig_pos = 0;
            out.do_bt = ok[w] ? 1 : 0; out.bt_i = ok[w] ? tlen[w] - 1 : -1; out.bt_j = ok[w] ? qlen[w] - 1 : -1;
            res[jid[w]] = out;
        }
    }
}

// One launch per lane-group width.  GL = 16 / 32 / 64 lanes per window (4 / 2 / 1 windows per wave) for targets up to
// 256 / 512 / 1024 rows: the ramp of the systolic array costs n_lanes - 1 steps per window, so a window should use as few
// lanes -- as tall a strip, S <= 16 -- as it can.  The launch list is grouped by S and every group is padded to whole
// waves, so S is uniform per wave (read from its first window).
template <int GL, bool EXACT, bool RIGHT>
__device__ __forceinline__ void ext_strip_dispatch(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, const int first,
                                                   const int n_list, const ExtParams &prm, const uint8_t *__restrict__ reads,
                                                   const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                   const RefView &rv, uint8_t *__restrict__ P, ExtRes *__restrict__ res, uint8_t *smem,
                                                   const int lds_stride, const int nr_stride) {
    const int j0 = first < n_list ? order[first] : -1;
    if (j0 < 0) return;   // (a wave of padding only: the lists are padded to whole waves at their ends)
    const int S = jobs[j0].strip_s;
#define MPN_CASE(SS) case SS: ext_strip_pack<SS, GL, EXACT, RIGHT>(jobs, order, first, n_list, prm, reads, read_off, read_len, rv, P, res, smem, lds_stride, nr_stride); break
    switch (S) {
        MPN_CASE(1); MPN_CASE(2); MPN_CASE(3); MPN_CASE(4); MPN_CASE(5); MPN_CASE(6); MPN_CASE(7); MPN_CASE(8);
        MPN_CASE(9); MPN_CASE(10); MPN_CASE(11); MPN_CASE(12); MPN_CASE(13); MPN_CASE(14); MPN_CASE(15);
        default: ext_strip_pack<16, GL, EXACT, RIGHT>(jobs, order, first, n_list, prm, reads, read_off, read_len, rv, P, res, smem, lds_stride, nr_stride); break;
    }
#undef MPN_CASE
}

// ONE launch for all the lane-group classes of a variant family (every launch ends in a tail of half-empty CUs: three launches
// per round had three).  Segments are ordered from the widest lane group to the narrowest and, inside a class, from the tallest
// strips and the longest queries down, so the waves with the most cells start first.
struct StripSeg { int32_t first_block, n_list, ord_off, lds_stride, nr_stride, glc, right, pair; };   // pair: two windows per lane group (gap fills)
struct StripSegs { StripSeg s[6]; int32_t n; };

// (the paired gap-fill cell keeps five registers per row -- u, y, y2 and the two windows' target selectors -- ~120 VGPRs at 16 rows:
// four waves per SIMD.  The kernel issues as fast with three as with five: it is bound by issue, not by latency hiding.)
template <bool EXACT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void ext_dp_strip_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, StripSegs segs,
                                                          ExtParams prm, const uint8_t *__restrict__ reads,
                                                          const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                          RefView rv, uint8_t *__restrict__ P, ExtRes *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int k = 0;
    while (k + 1 < segs.n && (int)blockIdx.x >= segs.s[k + 1].first_block) ++k;
    const StripSeg sg = segs.s[k];
    const int32_t *ord = order + sg.ord_off;
    if constexpr (!EXACT) {   // the gap fills: always two windows per lane group
        {
            const int first2 = ((int)blockIdx.x - sg.first_block) * (8 >> sg.glc);
            int S = 16;   // (uniform per wave: the lists are grouped by strip height and padded to whole waves)
            for (int k = 0; k < (8 >> sg.glc); ++k) { const int slot = first2 + k; if (slot < sg.n_list && ord[slot] >= 0) { S = jobs[ord[slot]].strip_s; break; } }
#define MPN_PAIR(SS, GLN) ext_strip_pair<SS, GLN>(jobs, ord, first2, sg.n_list, prm, reads, read_off, read_len, rv, P, res, smem, sg.lds_stride)
#define MPN_PAIR_S(GLN) switch (S) { case 1: MPN_PAIR(1, GLN); break; case 2: MPN_PAIR(2, GLN); break; case 3: MPN_PAIR(3, GLN); break; case 4: MPN_PAIR(4, GLN); break; \
            case 5: MPN_PAIR(5, GLN); break; case 6: MPN_PAIR(6, GLN); break; case 7: MPN_PAIR(7, GLN); break; case 8: MPN_PAIR(8, GLN); break; \
            case 9: MPN_PAIR(9, GLN); break; case 10: MPN_PAIR(10, GLN); break; case 11: MPN_PAIR(11, GLN); break; case 12: MPN_PAIR(12, GLN); break; \
            case 13: MPN_PAIR(13, GLN); break; case 14: MPN_PAIR(14, GLN); break; case 15: MPN_PAIR(15, GLN); break; default: MPN_PAIR(16, GLN); break; }
            if (sg.glc == 0) MPN_PAIR_S(16) else if (sg.glc == 1) MPN_PAIR_S(32) else MPN_PAIR_S(64)
#undef MPN_PAIR_S
#undef MPN_PAIR
        }
    } else {
        const int first = ((int)blockIdx.x - sg.first_block) * (4 >> sg.glc);
#define MPN_GL(GLN, RT) ext_strip_dispatch<GLN, EXACT, RT>(jobs, ord, first, sg.n_list, prm, reads, read_off, read_len, rv, P, res, smem, sg.lds_stride, sg.nr_stride)
        if (sg.right) { if (sg.glc == 0) MPN_GL(16, true); else if (sg.glc == 1) MPN_GL(32, true); else MPN_GL(64, true); }
        else { if (sg.glc == 0) MPN_GL(16, false); else if (sg.glc == 1) MPN_GL(32, false); else MPN_GL(64, false); }
#undef MPN_GL
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tiled, banded strips: the gap fills the strip kernel above cannot take -- targets longer than 1024 rows, or a band that clips
// (the 1.6-kb fills between the sparse anchors of a divergent assembly; w = 750) -- on the SAME cell instead of the band kernel's
// generic 32-bit one (~33 instead of ~120 instructions per cell).  One wave per window.  The target is cut into tiles of 1024 rows
// (64 lanes x 16); a tile is a systolic strip over the query columns its rows have in the band, [T0 - w, T0 + rows - 1 + w]; the
// v / x states and H of a tile's last row go through a 12-byte-per-column boundary in HBM to the next tile's first row.
// ksw2's band in (t, j) terms: a cell is computed iff t - w <= j <= t + w (st = (r - w + 1) >> 1, en = (r + w) >> 1, r = t + j).
// What lies outside is never read: the first in-band cell of a row (j == t - w) takes freshly opened gaps on its left, the last
// one (j == t + w) freshly opened gaps above -- two selects each -- exactly what ksw2's SSE code and the band kernel read there.
// Cells outside the band are computed on whatever flows in and ignored.  H is carried as a 32-bit number (8 H): from the left,
// or from above at a row's entry into the band; the corner's H is the score (ksw2's approximate-maximum walk ends there too).
// Directions: per tile a step-major block like the strip kernel's; layout 3 in the traceback.
constexpr int TILE_ROWS = 1024, TILE_S = 16;
struct TileGeom { int T0, rows, n_lanes, W, jlo, jhi; int64_t base; };
// geometry of tile `tile` of a tlen x qlen window with band w; base = bytes of the tiles before it
__host__ __device__ inline TileGeom tile_geom(int tile, int qlen, int tlen, int w) {
    TileGeom g{};
    int64_t base = 0;
    for (int k = 0;; ++k) {
        const int T0 = k * TILE_ROWS, rows = tlen - T0 < TILE_ROWS ? tlen - T0 : TILE_ROWS, nl = (rows + TILE_S - 1) / TILE_S;
        const int jlo = T0 - w > 0 ? T0 - w : 0, jhi = T0 + rows - 1 + w < qlen - 1 ? T0 + rows - 1 + w : qlen - 1;
        if (k == tile) { g.T0 = T0; g.rows = rows; g.n_lanes = nl; g.W = nl * TILE_S; g.jlo = jlo; g.jhi = jhi; g.base = base; return g; }
        base += (int64_t)((jhi >= jlo ? jhi - jlo + 1 : 0) + nl - 1) * (nl * TILE_S);
    }
}
__host__ __device__ inline int64_t tile_matrix_bytes(int qlen, int tlen, int w) {
    const int nt = (tlen + TILE_ROWS - 1) / TILE_ROWS;
    const TileGeom g = tile_geom(nt - 1, qlen, tlen, w);
    return g.base + (int64_t)((g.jhi >= g.jlo ? g.jhi - g.jlo + 1 : 0) + g.n_lanes - 1) * g.W;
}
// eligible: a global alignment whose corner the band reaches, on scores the packed cell holds
__host__ __device__ inline bool ext_tile_ok(int qlen, int tlen, int w) {
    const int d = qlen > tlen ? qlen - tlen : tlen - qlen;
    return qlen > 0 && tlen > 0 && w >= 1 && d <= w && tlen <= 32768 && qlen <= 60000;
}

// EXACT: an end extension (ksw2's exact maximum per anti-diagonal with its tie order, the z-drop rule, the best cell of the last query
// column): every in-band cell merges (H, tie key) into its anti-diagonal's slot of an LDS table with one ds_max_u32 -- the slot's
// band limits come from a second table -- and the rule is applied over the anti-diagonals in order after the last tile, like the
// exact strip variants do.  RIGHT: right-aligned gaps (the left extension).  LDS: query | score table | BEST[n_r] | (st, en)[n_r].
__host__ __device__ inline bool ext_tile_exact_ok(int qlen, int tlen, int w) {
    return qlen > 0 && tlen > 0 && w >= 1 && tlen <= 8191 && qlen <= 16384 && qlen + tlen - 1 <= 14000;
}
__host__ __device__ inline int ext_tile_lds_bytes(int qlen, int tlen, bool exact) {
    return ((qlen + 3) & ~3) + 64 + (exact ? 8 * ((qlen + tlen + 2) & ~3) : 0);
}

template <bool EXACT, bool RIGHT>
__device__ __forceinline__ void ext_tile_body(const ExtJob &jb, const int jid, const ExtParams &prm, const uint8_t *__restrict__ reads,
                                              const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len, const RefView &rv,
                                              uint8_t *__restrict__ P, int8_t *__restrict__ gstate, ExtRes *__restrict__ res, uint8_t *smem) {
    constexpr int S = TILE_S;
    const int lane = threadIdx.x;
    const int qlen = jb.qlen, tlen = jb.tlen;
    int q = prm.q, e = prm.e, q2 = prm.q2, e2 = prm.e2;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e, qe2 = q2 + e2;
    int w = jb.w;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    ExtRes out;
    out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = -1; out.mqe = NEG_INF; out.score = NEG_INF;
    out.reach_end = 0; out.n_cigar = 0; out.r_done = -1; out.bt_i = out.bt_j = -1; out.do_bt = 0; out.zcode = 0; out.cig_pos = 0;
    if (!(EXACT ? ext_tile_exact_ok(qlen, tlen, w) : ext_tile_ok(qlen, tlen, w)) || -prm.sc_mis > 2 * (q + e)) { if (lane == 0) res[jid] = out; return; }
    // ---- the cell's constants (see ext_strip_pack) ----
    constexpr int RS = RIGHT ? 0 : 4, RA = RIGHT ? 1 : 3, RB = 2, RA2 = RIGHT ? 3 : 1, RB2 = RIGHT ? 4 : 0;
    const uint32_t sb_mch = (uint32_t)(8 * prm.sc_mch + RS + 128) & 0xff, sb_mis = (uint32_t)(8 * prm.sc_mis + RS + 128) & 0xff,
                   sb_n = (uint32_t)(8 * prm.sc_n + RS + 128) & 0xff;
    const int q_pad = (qlen + 3) & ~3;
    uint32_t *tab = reinterpret_cast<uint32_t *>(smem + q_pad);
    if (lane < 5) tab[lane] = lane == 4 ? sb_n * 0x01010101u : (sb_mis * 0x01010101u) ^ ((sb_mch ^ sb_mis) << (8 * lane));
    {
        const int64_t roff = read_off[jb.read];
        const int32_t rlen = read_len[jb.read];
        for (int i = lane; i < qlen; i += 64) smem[i] = (uint8_t)(4 * ext_qbase(reads, roff, rlen, jb.rev, jb.qs + (jb.reversed ? qlen - 1 - i : i)));
    }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
#define MPN_BND(R) (8 * ((R) == 0 ? -qe : (R) < long_thres ? -e : (R) == long_thres ? long_diff : -e2))
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pk = [](int lo, int hi) { return (uint32_t)(hi * 65536 + lo); };
    constexpr int BETA = 0x2000 + 128;
    const int cv_lo = -8 * qe + RA + BETA, cv_hi = -8 * qe2 + RA2 + BETA, cu_lo = -8 * qe + RB + BETA, cu_hi = -8 * qe2 + RB2 + BETA;
    const uint32_t CV = pk(cv_lo, cv_hi), CU = pk(cu_lo, cu_hi);
    const uint32_t FRESH_U = pk(-8 * qe, -8 * qe) + CU, EDGE_V = pk(-8 * qe, -8 * qe) + CV;
    const uint32_t KONST = sb_n | 0x2000u;
    const uint32_t KZZ = pk(-16 * qe + RA + RB + BETA, -16 * qe2 + RA2 + RB2 + BETA);
    constexpr int RU = RIGHT ? 8 : 0;
    const uint32_t KEA = pk(8 * e - 8 * qe + RA - RU, 8 * e2 - 8 * qe2 + RA2 - RU), KEB = pk(8 * e - 8 * qe + RB - RU, 8 * e2 - 8 * qe2 + RB2 - RU);
    const uint32_t EIGHT = 0x00080008u, RANK_CLR = 0xfff8fff8u;
    const uint32_t MCH7 = (uint32_t)(8 * prm.sc_mch + 7 + BETA) * 0x00010001u;
    const int64_t g0 = rv.seq_off[jb.rid] + jb.ts;
    // boundary between tiles: two buffers of qlen x 3 words in the window's global scratch (a tile reads what the one before wrote
    // while it writes for the next)
    uint32_t *bnd_base = reinterpret_cast<uint32_t *>(gstate + jb.state_off);
    const int n_tiles = (tlen + TILE_ROWS - 1) / TILE_ROWS;
    // EXACT: per anti-diagonal the running (H, tie key) maximum and the band limits; per target row H in the last query column
    const int n_r = qlen + tlen - 1;
    uint32_t *BEST = reinterpret_cast<uint32_t *>(smem + q_pad + 64), *SE = BEST + ((n_r + 3) & ~3);
    int32_t *HL = reinterpret_cast<int32_t *>(bnd_base + 6 * (size_t)qlen);
    int r_lim = n_r;   // first anti-diagonal whose band lies outside the matrix (ksw2 stops there as z-dropped)
    if constexpr (EXACT) {
        for (int r = lane; r < n_r; r += 64) {
            int st = 0, en = tlen - 1;
            if (st < r - qlen + 1) st = r - qlen + 1;
            if (en > r) en = r;
            if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
            if (en > (r + w) >> 1) en = (r + w) >> 1;
            if (st > en) { r_lim = r_lim < r ? r_lim : r; st = 1; en = 0; }
            BEST[r] = 0;
            SE[r] = (uint32_t)st | (uint32_t)en << 16;
        }
        for (int t = lane; t < tlen; t += 64) HL[t] = NEG_INF;
        r_lim = wave_reduce_min(r_lim);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    __syncthreads();
    const bool head = lane == 0;
    int32_t score8 = 0;
    int prev_jhi = -1;
    for (int tile = 0; tile < n_tiles; ++tile) {
        const TileGeom G = tile_geom(tile, qlen, tlen, w);
        const int T0 = G.T0, n_lanes = G.n_lanes, W = G.W, jlo = G.jlo, jhi = G.jhi;
        if (jhi < jlo) break;   // (cannot happen for an eligible window: |qlen - tlen| <= w)
        uint32_t *bnd_in = bnd_base + (size_t)(tile & 1) * 3 * (size_t)qlen, *bnd_out = bnd_base + (size_t)((tile + 1) & 1) * 3 * (size_t)qlen;
        const bool write_bnd = tile + 1 < n_tiles;
        const int t0 = T0 + lane * S;
        uint32_t UL[S], YL[S], TSEL[S];
        int32_t H8[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int t = t0 + k;
            const int sq = t < tlen ? ref_code(rv, g0 + (jb.reversed ? tlen - 1 - t : t)) : 4;
            TSEL[k] = 0x0c0c0100u | (sq < 4 ? 4u + (uint32_t)sq : 0u);
            // rows whose band starts at column 0 take the first-column boundary; the others are overridden at their entry
            UL[k] = jlo == 0 ? pk(MPN_BND(t), MPN_BND(t)) + CU : FRESH_U;
            YL[k] = 0;
            H8[k] = 8 * (t < long_thres ? -(q + e * (t + 1)) : -(q2 + e2 * (t + 1)));
        }
        uint8_t *prow = P + jb.p_off + G.base + (int64_t)lane * S;
        const int n_steps = (jhi - jlo + 1) + n_lanes - 1;
        int out_v = 0, out_x = 0, out_h = 0;
        uint32_t qt = 0;
        auto qcode = [&](int col) -> uint32_t { return col < qlen ? (uint32_t)smem[col] : 16u; };
        uint32_t qt_next = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + qcode(jlo));
        uint32_t qc_next = qcode(jlo + 1);
        // the head lane's boundary words, requested two steps ahead
        uint32_t bv_n = 0, bx_n = 0, bh_n = 0, bv_n2 = 0, bx_n2 = 0, bh_n2 = 0;
        auto load_bnd = [&](int col, uint32_t &a, uint32_t &b, uint32_t &c) {
            if (tile > 0 && head && col <= prev_jhi && col < qlen) { a = bnd_in[3 * (size_t)col]; b = bnd_in[3 * (size_t)col + 1]; c = bnd_in[3 * (size_t)col + 2]; }
        };
        load_bnd(jlo, bv_n, bx_n, bh_n);
        load_bnd(jlo + 1, bv_n2, bx_n2, bh_n2);
        for (int step = 0; step < n_steps; ++step) {
            const uint32_t q_in = qt_next;
            qt_next = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(tab) + qc_next);
            qc_next = qcode(jlo + step + 2);
            const uint32_t bv = bv_n, bx = bx_n, bh = bh_n;
            bv_n = bv_n2; bx_n = bx_n2; bh_n = bh_n2;
            load_bnd(jlo + step + 2, bv_n2, bx_n2, bh_n2);
            const uint32_t qt_s = (uint32_t)wave_shr1_zero((int)qt);
            const int v_s = wave_shr1_zero(out_v), x_s = wave_shr1_zero(out_x), h_s = wave_shr1_zero(out_h);
            const int j = jlo + step - lane;
            qt = head ? q_in : qt_s;
            uint32_t Vp, Xp;
            int32_t Hup;
            if (head) {
                if (tile == 0) { const int bj = MPN_BND(j); Vp = pk(bj, bj) + CV; Xp = 0; Hup = 0; }
                else { Vp = bv; Xp = bx; Hup = (int32_t)bh; }
            } else { Vp = (uint32_t)v_s; Xp = (uint32_t)x_s; Hup = h_s; }
            if (j >= jlo && j <= jhi && lane < n_lanes) {
                uint32_t dw[(S + 3) / 4], ecell[4] = {0, 0, 0, 0};
                const int d_in = j - t0 + w;    // row k enters the band at this column iff d_in == k
                const int d_out = j - t0 - w;   // row k is the band's last row of this column iff d_out == k
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const bool enter = d_in == k, edge = d_out == k;
                    const uint32_t Up = enter ? FRESH_U : UL[k];
                    const uint32_t Yl = enter ? 0u : YL[k];
                    Vp = edge ? EDGE_V : Vp;
                    Xp = edge ? 0u : Xp;
                    const uint32_t sc16 = __builtin_amdgcn_perm(qt, KONST, TSEL[k]);
                    const uint32_t A = Xp + Vp, B = Yl + Up;
                    const uint32_t M = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, A), __builtin_bit_cast(s16x2, B)));
                    uint32_t m2, z16;
                    asm("v_pk_max_i16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m2) : "v"(M));
                    asm("v_max_i16 %0, %1, %2" : "=v"(z16) : "v"(m2), "v"(sc16));
                    uint32_t zc;
                    asm("v_pk_min_i16 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(zc) : "v"(z16), "v"(MCH7));
                    const uint32_t Zc = zc & RANK_CLR;
                    const uint32_t ZZ = Zc + KZZ;
                    const uint32_t nu = ZZ - Vp, nv = ZZ - Up;
                    u16x2 An = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, A), __builtin_bit_cast(u16x2, Zc + KEA));
                    u16x2 Bn = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, B), __builtin_bit_cast(u16x2, Zc + KEB));
                    uint32_t HA, HB;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(HA) : "v"(__builtin_bit_cast(uint32_t, An)), "v"(EIGHT));
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(HB) : "v"(__builtin_bit_cast(uint32_t, Bn)), "v"(EIGHT));
                    if constexpr (RIGHT) {   // (right-aligned gaps: the states were one unit high for the >= 0 flags)
                        An = __builtin_elementwise_sub_sat(An, __builtin_bit_cast(u16x2, EIGHT));
                        Bn = __builtin_elementwise_sub_sat(Bn, __builtin_bit_cast(u16x2, EIGHT));
                    }
                    uint32_t Fw;
                    asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(Fw) : "v"(HB), "v"(HA));
                    asm("v_and_or_b32 %0, %1, 7, %2" : "=v"(ecell[k & 3]) : "v"(z16), "v"(Fw));
                    // H of the cell: from the left, or from above where the row enters the band (8 H; u and v in the low halves)
                    const int32_t h_left = H8[k] + (int32_t)(nv & 0xffffu) - cv_lo, h_up = Hup + (int32_t)(nu & 0xffffu) - cu_lo;
                    const int32_t h = enter ? h_up : h_left;
                    H8[k] = h;
                    Hup = h;
                    if constexpr (EXACT) {
                        const int t = t0 + k, r = t + j;
                        const uint32_t se = SE[r < n_r ? r : n_r - 1];
                        const int st_ = (int)(se & 0xffffu), en_ = (int)(se >> 16);
                        const bool inb = t >= st_ && t <= en_;
                        const int en1 = st_ + ((en_ - st_) & ~3);
                        const int cls = t == en_ ? 0 : t < en1 ? 1 + ((t - st_) & 3) : 5;   // ksw2's tie order: the band's last cell, then its 4-lane classes, then the tail
                        const uint32_t word = (uint32_t)((h >> 3) + 32768) << 16 | (0xffffu - ((uint32_t)cls << 13 | (uint32_t)t));
                        atomicMax(&BEST[r < n_r ? r : n_r - 1], inb ? word : 0u);
                        if (inb && j == qlen - 1) HL[t] = h >> 3;
                    }
                    UL[k] = nu; YL[k] = __builtin_bit_cast(uint32_t, Bn);
                    Vp = nv; Xp = __builtin_bit_cast(uint32_t, An);
                    if ((k & 3) == 3) {
                        uint32_t e01, e23;
                        asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(e01) : "v"(ecell[1]), "v"(ecell[0]));
                        asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(e23) : "v"(ecell[3]), "v"(ecell[2]));
                        const uint32_t lo = __builtin_amdgcn_perm(e23, e01, 0x05040100u), hi = __builtin_amdgcn_perm(e23, e01, 0x07060302u);
                        asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(dw[k >> 2]) : "v"(hi), "v"(lo));
                    }
                }
                out_v = (int)Vp; out_x = (int)Xp; out_h = Hup;
                *reinterpret_cast<uint4 *>(prow + (int64_t)step * W) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
                // a full tile's last lane hands its last row to the next tile
                if (write_bnd && lane == 63) { bnd_out[3 * (size_t)j] = (uint32_t)out_v; bnd_out[3 * (size_t)j + 1] = (uint32_t)out_x; bnd_out[3 * (size_t)j + 2] = (uint32_t)out_h; }
            }
        }
        // the corner: H of row tlen - 1 after its last column (qlen - 1 = jhi of the last tile)
        if (tile + 1 == n_tiles) {
            const int lt = tlen - 1 - T0, ll = lt / S, kk = lt % S;
            int32_t hv = 0;
#pragma unroll
            for (int k = 0; k < S; ++k) hv = k == kk ? H8[k] : hv;
            score8 = __shfl(hv, ll);
        }
        prev_jhi = jhi;
        // the boundary the next tile reads was written by lane 63, its first lane reads it: order them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
#undef MPN_BND
    if constexpr (!EXACT) {
        if (lane == 0) {
            out.score = score8 >> 3;
            out.r_done = qlen + tlen - 2;
            out.do_bt = 1; out.bt_i = tlen - 1; out.bt_j = qlen - 1;
            res[jid] = out;
        }
    } else {
        // ---- the z-drop rule over the anti-diagonals in order (as in ext_strip_pack's exact variants, 64 lanes wide) ----
        __syncthreads();
        const int tlm1 = tlen - 1;
        struct Best { int m, t, r; };
        auto scan_first_max = [&](Best x) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const Best o{__shfl_up(x.m, d), __shfl_up(x.t, d), __shfl_up(x.r, d)};
                if (lane >= d && !(x.m > o.m)) x = o;
            }
            return x;
        };
        auto diag = [&](int r, int &H, int &mt) {
            const uint32_t pv = BEST[r];
            H = pv ? (int)(pv >> 16) - 32768 : NEG_INF;
            mt = (int)((0xffffu - (pv & 0xffffu)) & 8191u);
        };
        const int n_eff = r_lim < n_r ? r_lim : n_r;   // anti-diagonals ksw2 computes
        const int C = (n_eff + 63) / 64, rlo = min(n_eff, lane * C), rhi = min(n_eff, rlo + C);
        auto chunk_best = [&](int lim) {
            Best b{NEG_INF, -1, -1};
            for (int r = rlo; r < rhi && r <= lim; ++r) { int H, mt; diag(r, H, mt); if (H > b.m) b = Best{H, mt, r}; }
            return b;
        };
        Best incl = scan_first_max(chunk_best(n_eff));
        Best in{__shfl_up(incl.m, 1), __shfl_up(incl.t, 1), __shfl_up(incl.r, 1)};
        ExtApply ez; ez.max = 0; ez.max_t = ez.max_q = -1; ez.zdropped = 0;
        if (lane > 0 && in.m > 0) { ez.max = in.m; ez.max_t = in.t; ez.max_q = in.r - in.t; }
        int r_break = n_eff;
        for (int r = rlo; r < rhi; ++r) {
            int H, mt; diag(r, H, mt);
            if (ext_apply_zdrop(ez, H, r, mt, jb.zdrop, e2)) { r_break = r; break; }
        }
        r_break = wave_reduce_min(r_break);
        const bool zd = r_break < n_eff;            // the z-drop rule fired
        const bool dropped = zd || n_eff < n_r;      // ... or the band left the matrix (st > en: ksw2 sets zdropped and stops)
        const int r_last = zd ? r_break : n_eff - 1; // last anti-diagonal whose cells count
        Best fin = scan_first_max(chunk_best(r_last));
        fin = Best{__shfl(fin.m, 63), __shfl(fin.t, 63), __shfl(fin.r, 63)};
        // best cell of the last query column among the anti-diagonals qlen - 1 .. r_last (target rows 0 .. r_last - qlen + 1)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int t_lim = min(tlm1, r_last - (qlen - 1));
        const int Ct = (tlen + 63) / 64, tlo = min(tlen, lane * Ct), thi = min(tlen, tlo + Ct);
        Best me{NEG_INF, -1, -1};
        for (int t = tlo; t < thi && t <= t_lim; ++t) { const int h = HL[t]; if (h > me.m) me = Best{h, t, t}; }
        me = scan_first_max(me);
        me = Best{__shfl(me.m, 63), __shfl(me.t, 63), 0};
        if (lane == 0) {
            if (fin.m > 0) { out.max = fin.m; out.max_t = fin.t; out.max_q = fin.r - fin.t; }
            out.zdropped = dropped ? 1 : 0;
            if (me.m > NEG_INF) { out.mqe = me.m; out.mqe_t = me.t; }
            if (!dropped) out.score = HL[tlm1];
            out.r_done = dropped ? (zd ? r_break : n_eff) : n_r - 1;
            if (!dropped && !(jb.flag & EZ_EXTZ_ONLY)) { out.do_bt = 1; out.bt_i = tlen - 1; out.bt_j = qlen - 1; }
            else if (!dropped && (jb.flag & EZ_EXTZ_ONLY) && out.mqe + jb.end_bonus > out.max) { out.reach_end = 1; out.do_bt = 1; out.bt_i = out.mqe_t; out.bt_j = qlen - 1; }
            else if (out.max_t >= 0 && out.max_q >= 0) { out.do_bt = 1; out.bt_i = out.max_t; out.bt_j = out.max_q; }
            res[jid] = out;
        }
    }
}

__global__ __launch_bounds__(64) void ext_dp_tile_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_jobs,
                                                         ExtParams prm, const uint8_t *__restrict__ reads,
                                                         const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                         RefView rv, uint8_t *__restrict__ P, int8_t *__restrict__ gstate, ExtRes *__restrict__ res) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int jid = order[blockIdx.x];
    const ExtJob jb = jobs[jid];
    // (a workgroup is one window: one branch per window into the instantiation that carries only its own bookkeeping)
    if (jb.flag & EZ_APPROX_MAX) ext_tile_body<false, false>(jb, jid, prm, reads, read_off, read_len, rv, P, gstate, res, smem);
    else if (jb.flag & EZ_RIGHT) ext_tile_body<true, true>(jb, jid, prm, reads, read_off, read_len, rv, P, gstate, res, smem);
    else ext_tile_body<true, false>(jb, jid, prm, reads, read_off, read_len, rv, P, gstate, res, smem);
}



// traceback: one lane per job (serial pointer chase; parallelism across jobs hides the latency)
__global__ __launch_bounds__(64) void ext_bt_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_jobs,
                                                    const uint8_t *__restrict__ P, const int32_t *__restrict__ OFF,
                                                    uint32_t *__restrict__ CIG, uint32_t *__restrict__ COMPACT,
                                                    unsigned long long *__restrict__ compact_used, ExtRes *__restrict__ res) {
    // (no early exit: the whole wave meets again at the end to reserve its slice of the compact pool with ONE atomic --
    // one atomic per job on a single counter serialises tens of thousands of lanes at the memory side)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int jid = k < n_jobs ? order[k] : -1;  // -1: also the padding of a strip launch list
    const bool live = jid >= 0;
    ExtJob jb = jobs[live ? jid : 0];
    ExtRes r = res[live ? jid : 0];
    const bool walk = live && r.do_bt;
    if (live && !r.do_bt) res[jid].cig_pos = 0;
    if (!walk) { jb.qlen = jb.tlen = 1; r.bt_i = r.bt_j = -1; }
    const int n_col = jb.n_col, n_r = jb.qlen + jb.tlen - 1;
    const uint8_t *p = P + jb.p_off;
    const int32_t *off = OFF + 2 * jb.row_off, *off_end = off + n_r;
    const bool rowmajor = jb.layout == 1;  // strip kernel: cell (t, j) at [j + t/S][t], band never clips
    const bool rank_is_op = (jb.flag & EZ_RIGHT) != 0;
    const bool byslot = jb.layout == 2;    // band kernel: cell (t, r) at [r][t mod SL], band limits recomputed here
    const bool tiled = jb.layout == 3;     // tiled strips: per 1024-row tile a step-major block over the tile's band columns
    TileGeom tg{};
    int tg_tile = -1;
    const int bw = jb.w < 0 ? (jb.tlen > jb.qlen ? jb.tlen : jb.qlen) : jb.w;
    const bool rev_cigar = (jb.flag & EZ_REV_CIGAR) != 0;
    // ops are generated last-to-first.  REV_CIGAR keeps that order (write forward from the region start),
    // otherwise they are written back to front so that they read forward.
    uint32_t *cend = CIG + jb.cig_off;
    const int cap = jb.qlen + jb.tlen + 2;
    uint32_t *cbeg = cend - cap;
    int n = 0, i = r.bt_i, j = r.bt_j, state = 0;
    uint32_t cur_op = 0xf, cur_len = 0;
#define MPN_FLUSH() do { if (cur_len) { if (rev_cigar) cbeg[n] = cur_len << 4 | cur_op; else cend[-1 - n] = cur_len << 4 | cur_op; ++n; } } while (0)
#define MPN_PUSHOP(OP, LEN) do { if ((uint32_t)(OP) == cur_op) cur_len += (LEN); else { MPN_FLUSH(); cur_op = (OP); cur_len = (LEN); } } while (0)
    constexpr int BT_AHEAD = 8;   // direction bytes fetched at once down the diagonal
    while (i >= 0 && j >= 0) {
        const int rr = i + j;
        int force_state = -1, tmp;
        if (rowmajor) {  // (the strip kernel stores the winner's rank: 4 - operand, or the operand itself for right-aligned gaps)
            if (state == 0) {
                // The walk is a chain of dependent loads, one HBM round trip per cell, and most cells of an alignment continue the
                // diagonal: the bytes of the next BT_AHEAD cells down the diagonal are requested together and consumed while they
                // say "match"; the first one that says otherwise is handled by the general step below (its byte is already here).
                uint8_t ahead[BT_AHEAD];
#pragma unroll
                for (int k = 0; k < BT_AHEAD; ++k) {
                    const int ii = i - k, jj = j - k;
                    ahead[k] = (ii >= 0 && jj >= 0) ? p[(int64_t)(jj + ii / jb.strip_s) * jb.qstride + ii] : (uint8_t)0xff;
                }
                int k = 0;
                tmp = 0;
#pragma unroll
                for (int q = 0; q < BT_AHEAD; ++q) {
                    if (k != q) continue;             // (stopped earlier)
                    if (i - q < 0 || j - q < 0) continue;
                    int b = ahead[q];
                    if (!rank_is_op) b = (b & ~7) | (4 - (b & 7));
                    if ((b & 7) == 0) ++k; else tmp = b;
                }
                if (k > 0) { MPN_PUSHOP(0, (uint32_t)k); i -= k; j -= k; }
                if (k == BT_AHEAD || i < 0 || j < 0) continue;
                // cell (i, j): a gap wins there (tmp holds its decoded byte)
            } else {
                tmp = p[(int64_t)(j + i / jb.strip_s) * jb.qstride + i];
                if (!rank_is_op) tmp = (tmp & ~7) | (4 - (tmp & 7));
            }
        }
        else if (tiled) {
            int st = 0, en = jb.tlen - 1;
            if (st < rr - jb.qlen + 1) st = rr - jb.qlen + 1;
            if (en > rr) en = rr;
            if (st < (rr - bw + 1) >> 1) st = (rr - bw + 1) >> 1;
            if (en > (rr + bw) >> 1) en = (rr + bw) >> 1;
            if (i < st) force_state = 2;
            if (i > en) force_state = 1;
            tmp = 0;
            if (force_state < 0) {
                const int tile = i / TILE_ROWS;
                if (tile != tg_tile) { tg = tile_geom(tile, jb.qlen, jb.tlen, bw); tg_tile = tile; }   // (the walk only descends: a few times per window)
                const int lt = i - tg.T0;
                tmp = p[tg.base + (int64_t)((j - tg.jlo) + lt / TILE_S) * tg.W + lt];
                if (!rank_is_op) tmp = (tmp & ~7) | (4 - (tmp & 7));
            }
        }
        else if (byslot) {
            int st = 0, en = jb.tlen - 1;
            if (st < rr - jb.qlen + 1) st = rr - jb.qlen + 1;
            if (en > rr) en = rr;
            if (st < (rr - bw + 1) >> 1) st = (rr - bw + 1) >> 1;
            if (en > (rr + bw) >> 1) en = (rr + bw) >> 1;
            if (i < st) force_state = 2;
            if (i > en) force_state = 1;
            tmp = force_state < 0 ? p[(int64_t)rr * jb.qstride + (i & (jb.qstride - 1))] : 0;
        } else {
            if (i < off[rr]) force_state = 2;
            if (i > off_end[rr]) force_state = 1;
            tmp = force_state < 0 ? p[(int64_t)rr * n_col + i - off[rr]] : 0;
        }
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) { MPN_PUSHOP(0, 1); --i; --j; }
        else if (state == 1 || state == 3) { MPN_PUSHOP(2, 1); --i; }
        else { MPN_PUSHOP(1, 1); --j; }
    }
    if (walk) {
        if (i >= 0) MPN_PUSHOP(2, (uint32_t)(i + 1));
        if (j >= 0) MPN_PUSHOP(1, (uint32_t)(j + 1));
        MPN_FLUSH();
    }
#undef MPN_PUSHOP
#undef MPN_FLUSH
    // hand the ops over in forward order through a compact pool, so that only used entries travel to the host
    const int lane = threadIdx.x & 63;
    int incl = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    const int total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 0 && total) base = atomicAdd(compact_used, (unsigned long long)total);
    base = (unsigned long long)__shfl((long long)base, 0);
    if (walk) {
        const unsigned long long pos = base + (unsigned long long)(incl - n);
        const uint32_t *src = rev_cigar ? cbeg : cend - n;
        for (int q = 0; q < n; ++q) COMPACT[pos + q] = src[q];
        res[jid].n_cigar = n;
        res[jid].cig_pos = (int64_t)pos;
    }
}

// z-drop test of a finished gap-fill CIGAR (minimap2 mm_test_zdrop without the inversion probe): one lane per job
__global__ __launch_bounds__(64) void ext_ztest_kernel(const ExtJob *__restrict__ jobs, const int32_t *__restrict__ order, int n_jobs,
                                                       ExtParams prm, const uint8_t *__restrict__ reads,
                                                       const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                       RefView rv,
                                                       const uint32_t *__restrict__ CIG, ExtRes *__restrict__ res,
                                                       int32_t *__restrict__ redo_ids, unsigned long long *__restrict__ n_redo,
                                                       InvProbe *__restrict__ probes, unsigned long long *__restrict__ n_probe) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_jobs) return;
    const int jid = order[k];
    if (jid < 0) return;  // padding of a strip launch list
    const ExtJob jb = jobs[jid];
    if (!(jb.flag & EZ_APPROX_MAX)) return;  // only gap fills are tested
    // a window that fails the test is listed for the exact second pass: the host reads the COUNT (8 bytes), and the ids only
    // when there are any -- not a result record per window
    auto report = [&](int z) {
        res[jid].zcode = z;
        if (z) redo_ids[atomicAdd(n_redo, 1ULL)] = jid;
    };
    const ExtRes r = res[jid];
    const int zmin = prm.zdrop_thres < prm.zdrop_inv ? prm.zdrop_thres : prm.zdrop_inv;
    // No walk over the sequences is needed when a drop of more than the smaller threshold is impossible.  A drop is made of the
    // negative contributions between two points of the path, so it is at most all of them: N = (mismatch and ambiguous columns) +
    // (the gaps as THIS test prices them, q + e * len each).  The window's DP score F (exact corner score of the strip kernel) is
    // a * matches - those columns - the gaps as the DP prices them (the cheaper of the two affine costs), hence
    // N <= a * M - F + sum over the gaps of (q + e * len - DP cost), with M = the columns of the CIGAR's match operations.
    // One pass over the CIGAR's few dozen operations, no sequence access.
    const uint32_t *cig = CIG + jb.cig_off - r.n_cigar;  // gap-fill jobs are never REV_CIGAR
    if ((jb.layout == 1 || jb.layout == 3) && r.do_bt && !r.zdropped) {
        int64_t m_cols = 0, extra = 0;
        for (int c = 0; c < r.n_cigar; ++c) {
            const uint32_t op = cig[c] & 0xf;
            const int64_t len = cig[c] >> 4;
            if (op == 0) m_cols += len;
            else if (op == 1 || op == 2) {
                const int64_t w1 = prm.q + prm.e * len, w2 = prm.q2 + prm.e2 * len;
                extra += w1 - (w1 < w2 ? w1 : w2);
            }
        }
        if ((int64_t)prm.sc_mch * m_cols - r.score + extra <= zmin) { report(0); return; }
    }
    const int64_t roff = read_off[jb.read];
    const int32_t rlen = read_len[jb.read];
    const int64_t g0 = rv.seq_off[jb.rid] + jb.ts;
    int32_t score = 0, mx = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
    int32_t p00 = -1, p01 = -1, p10 = -1, p11 = -1;   // where the largest drop starts (the running maximum) and ends
    RefCursor tc(rv);
    ReadCursor qc(reads, roff, rlen, jb.rev);
    for (int c = 0; c < r.n_cigar; ++c) {
        const uint32_t op = cig[c] & 0xf, len = cig[c] >> 4;
        if (op == 0) {
            for (uint32_t l = 0; l < len; ++l) {
                const int ct = tc.at(g0 + i + l), cq = qc.at(jb.qs + j + (int)l);
                score += (ct == 4 || cq == 4) ? prm.sc_n : ct == cq ? prm.sc_mch : prm.sc_mis;
                if (score < mx) {
                    const int li = i + (int)l - max_i, lj = j + (int)l - max_j, diff = li > lj ? li - lj : lj - li;
                    const int z = mx - score - diff * prm.e;
                    if (z > max_zdrop) { max_zdrop = z; p00 = max_i; p01 = max_j; p10 = i + (int)l; p11 = j + (int)l; }
                } else { mx = score; max_i = i + l; max_j = j + l; }
            }
            i += len; j += len;
        } else if (op == 1 || op == 2) {
            score -= prm.q + prm.e * (int)len;
            if (op == 1) j += len; else i += len;
            if (score < mx) {
                const int li = i - max_i, lj = j - max_j, diff = li > lj ? li - lj : lj - li;
                const int z = mx - score - diff * prm.e;
                if (z > max_zdrop) { max_zdrop = z; p00 = max_i; p01 = max_j; p10 = i; p11 = j; }
            } else { mx = score; max_i = i; max_j = j; }
        }
    }
    const int over = max_zdrop > prm.zdrop_thres ? 1 : 0;
    const int q_len = p11 - p01, t_len = p10 - p00;
    if (max_zdrop > prm.zdrop_inv && q_len < prm.max_gap && t_len < prm.max_gap && q_len > 0 && t_len > 0) {
        // inversion candidate: the host decides between "nothing / plain z-drop" and "inversion" (second pass with zdrop_inv)
        res[jid].zcode = over;
        probes[atomicAdd(n_probe, 1ULL)] = InvProbe{jid, p00, p01, p10, p11, over};
        return;
    }
    report(over);
}

}  // namespace mpn
