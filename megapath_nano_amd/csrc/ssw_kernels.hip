// SSW-compatible local alignment on gfx950 (MI355X).  One wavefront (64 lanes) per read/reference pair.
//
// Replaces /root/reference/bin/realignment/realign/ssw.c (ssw_align :762, sw_sse2_byte :123,
// sw_sse2_word :354, banded_sw :532) with identical outputs (see oracle/ssw_oracle.c for the
// cell-level statement of what the SSE2 code computes, incl. the padded rows and the band quirks).
//
// Score passes (kernel ssw_score_kernel<R>): the reference walks reference columns and keeps the read
// striped over SSE lanes; here every lane owns R CONSECUTIVE read rows in VGPRs (H, E), the query
// profile (n x P int8) sits in LDS, and the vertical gap state F -- the only intra-column
// dependency -- is resolved exactly by a max-plus prefix scan: lane-local recurrence + one DPP wave
// scan (row_shr/row_bcast) per column.  Column maxima are reduced with the same DPP ladder.
// No MFMA: integer DP.  HBM traffic is tiny (reference bases streamed once, 64 per load).
//
// Traceback (kernel ssw_banded_kernel): banded DP row by row, band cells across lanes, previous row
// in LDS, horizontal gap state by the same scan; direction codes (1 B/cell) to an HBM scratch;
// lane 0 walks them back.  The host doubles the band until the DP reproduces the score (ssw.c:555-615).
#include "mpn_common.h"
#include "../../include/mpn_ssw.h"

#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

namespace mpn {

struct SswDev {
    const int8_t *reads, *refs;
    const int64_t *read_off, *ref_off;
    const int32_t *read_len, *ref_len, *mask_len;
    int8_t mat[64];
    int32_t nsym, score_size, gapO, gapE, flag, filters, filterd, bias;
    // per-pair results (device)
    int32_t *score1, *score2, *ref_begin1, *ref_end1, *read_begin1, *read_end1, *ref_end2;
    int32_t *status, *need_cigar;
    uint16_t *mcol;           // column maxima scratch
    const int64_t *mcol_off;  // per pair offset into mcol
};

struct PassOut { int maxv, end_ref, end_read, overflow; };

// One score pass over ref[0..refLen) (dir=0 forward, 1 backward).  prof: LDS profile, PL = 64*R row stride.
template <int R>
__device__ __forceinline__ PassOut score_pass(const int8_t *__restrict__ ref, int refLen, int dir, int readLen,
                                              int sse_lanes, int gapO, int gapE, int bias, int terminate,
                                              const int8_t *prof, uint16_t *__restrict__ mcol) {
    constexpr int PL = 64 * R;
    const int lane = threadIdx.x & 63;
    const int P = (readLen + sse_lanes - 1) / sse_lanes * sse_lanes;  // padded rows of the SSE layout
    const int row0 = lane * R;
    const int D = R * gapE;
    int H[R], E[R], Hmax[R];
#pragma unroll
    for (int k = 0; k < R; ++k) H[k] = 0, E[k] = 0, Hmax[k] = 0;
    PassOut o;
    o.maxv = 0;
    o.end_ref = sse_lanes == 16 ? -1 : 0;  // ssw.c:145 vs :371
    o.overflow = 0;
    int rc = 0, mc = 0;
    for (int c = 0; c < refLen; ++c) {
        if ((c & 63) == 0) {
            int idx = c + lane;
            rc = idx < refLen ? ref[dir ? refLen - 1 - idx : idx] : 0;
        }
        const int sym = __builtin_amdgcn_readlane(rc, c & 63);
        const int *pw = reinterpret_cast<const int *>(prof + sym * PL + row0);
        int diag = wave_shr1(H[R - 1], 0);
        int fl = NEG_INF;
#pragma unroll
        for (int q = 0; q < R / 4; ++q) {
            const int w = pw[q];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = q * 4 + b;
                const int s = (w << (24 - 8 * b)) >> 24;
                const int hold = H[k];
                int a = max(diag + s, 0);
                a = max(a, E[k]);
                H[k] = a;  // A(k): H without the vertical gap
                diag = hold;
                fl = max(fl - gapE, a - gapO);
            }
        }
        // fl = F entering the next lane's first row from this lane's rows only; scan with decay D per lane
        int t = wave_scan_max(fl + lane * D);
        t = wave_shr1(t, NEG_INF);
        int f = lane == 0 ? NEG_INF : t - (lane - 1) * D;
        int cm = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int h = max(H[k], f);
            H[k] = h;
            E[k] = max(E[k] - gapE, h - gapO);
            if (row0 + k < P) cm = max(cm, h);
            f = max(f - gapE, h - gapO);
        }
        const int colmax = wave_reduce_max(cm);
        const int i = dir ? refLen - 1 - c : c;
        if (colmax > o.maxv) {
            o.maxv = colmax;
            if (sse_lanes == 16 && colmax + bias >= 255) { o.overflow = 1; break; }  // ssw.c:271
            o.end_ref = i;
#pragma unroll
            for (int k = 0; k < R; ++k) Hmax[k] = H[k];
        }
        if (mcol) {
            if (lane == (c & 63)) mc = colmax;
            if ((c & 63) == 63 || c == refLen - 1) {
                int idx = (c & ~63) + lane;
                if (idx <= c) mcol[idx] = (uint16_t)mc;
            }
        }
        if (colmax == terminate) break;
    }
    int er = 1 << 30;
#pragma unroll
    for (int k = R - 1; k >= 0; --k)
        if (Hmax[k] == o.maxv && row0 + k < P) er = row0 + k;
    er = wave_reduce_min(er);
    o.end_read = min(readLen - 1, er);
    return o;
}

template <int R>
__device__ __forceinline__ void build_profile(int8_t *prof, const int8_t *mat, int nsym, const int8_t *read, int readLen,
                                              bool reversed) {
    constexpr int PL = 64 * R;
    for (int idx = threadIdx.x; idx < nsym * PL; idx += 64) {
        int sym = idx / PL, j = idx - sym * PL;
        int8_t v = 0;
        if (j < readLen) v = mat[sym * nsym + read[reversed ? readLen - 1 - j : j]];
        prof[idx] = v;
    }
    __syncthreads();
}

template <int R>
__global__ __launch_bounds__(64) void ssw_score_kernel(SswDev p, const int32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) int8_t prof[];
    const int job = order[blockIdx.x];
    const int lane = threadIdx.x;
    const int readLen = p.read_len[job], refLen = p.ref_len[job], maskLen = p.mask_len[job];
    const int8_t *read = p.reads + p.read_off[job];
    const int8_t *ref = p.refs + p.ref_off[job];
    uint16_t *mcol = p.mcol + p.mcol_off[job];
    const bool have_byte = p.score_size == 0 || p.score_size == 2, have_word = p.score_size == 1 || p.score_size == 2;

    build_profile<R>(prof, p.mat, p.nsym, read, readLen, false);
    int word = 0;
    PassOut o;
    if (have_byte) {
        o = score_pass<R>(ref, refLen, 0, readLen, 16, p.gapO, p.gapE, p.bias, 255, prof, mcol);
        if (o.overflow) {
            if (!have_word) { if (lane == 0) p.status[job] = MPN_SSW_ENULL; return; }
            o = score_pass<R>(ref, refLen, 0, readLen, 8, p.gapO, p.gapE, 0, 65535, prof, mcol);
            word = 1;
        }
    } else {
        o = score_pass<R>(ref, refLen, 0, readLen, 8, p.gapO, p.gapE, 0, 65535, prof, mcol);
        word = 1;
    }
    // 2nd best outside +-maskLen (ssw.c:310-323 / :512-525): first column (in index order) holding the max
    int s2 = 0, r2 = 0;
    {
        int e1 = max(o.end_ref - maskLen, 0);
        int e2 = min(o.end_ref + maskLen, refLen);
        int start2 = word ? e2 : e2 + 1;
        int bv = 0, bi = 1 << 30;
        __syncthreads();  // mcol written by this wave: make the stores visible to its own later loads
        for (int idx = lane; idx < e1; idx += 64) { int v = mcol[idx]; if (v > bv) bv = v, bi = idx; }
        for (int idx = start2 + lane; idx < refLen; idx += 64) { int v = mcol[idx]; if (v > bv) bv = v, bi = idx; }
        s2 = wave_reduce_max(bv);
        int cand = (bv == s2 && s2 > 0) ? bi : (1 << 30);
        cand = wave_reduce_min(cand);
        r2 = s2 > 0 ? cand : 0;
    }
    int ref_begin = -1, read_begin = -1, need = 0, st = MPN_SSW_OK;
    const int score1 = o.maxv, ref_end1 = o.end_ref, read_end1 = o.end_read;
    const bool stop = p.flag == 0 || (p.flag == 2 && score1 < p.filters);
    if (!stop) {
        const int rl = read_end1 + 1;
        __syncthreads();
        build_profile<R>(prof, p.mat, p.nsym, read, rl, true);
        PassOut rv = score_pass<R>(ref, ref_end1 + 1, 1, rl, word ? 8 : 16, p.gapO, p.gapE, word ? 0 : p.bias,
                                   score1, prof, nullptr);
        ref_begin = rv.end_ref;
        read_begin = read_end1 - rv.end_read;
        const bool nocig = (7 & p.flag) == 0 || ((2 & p.flag) != 0 && score1 < p.filters) ||
                           ((4 & p.flag) != 0 && (ref_end1 - ref_begin > p.filterd || read_end1 - read_begin > p.filterd));
        if (!nocig) {
            if (ref_begin < 0 || read_begin < 0 || ref_end1 - ref_begin + 1 <= 0 || read_end1 - read_begin + 1 <= 0)
                st = MPN_SSW_EUNDEF;
            else
                need = 1;
        }
    }
    if (lane == 0) {
        p.score1[job] = score1;
        p.ref_end1[job] = ref_end1;
        p.read_end1[job] = read_end1;
        p.score2[job] = maskLen >= 15 ? s2 : 0;
        p.ref_end2[job] = maskLen >= 15 ? r2 : -1;
        p.ref_begin1[job] = ref_begin;
        p.read_begin1[job] = read_begin;
        p.need_cigar[job] = need;
        p.status[job] = st;
    }
}

// ---------------------------------------------------------------------------------------------
// Reads longer than 64 x 32 rows: the read is cut into STRIPS of 2048 rows; a strip is one score pass over the whole
// reference with the rows in VGPRs as above, and what crosses a strip boundary travels through HBM scratch, one value
// pair per reference column: H of the strip's last row (the next strip's diagonal / upper neighbour) and the vertical gap
// state leaving it.  ssw.c's outputs are defined on whole columns (first column whose maximum exceeds every earlier
// one; smallest read index holding it; abort of the 8-bit pass at the first saturating column), so the strips merge
// their column maxima (colmax scratch) and the pass's results are derived from the merged columns afterwards: the
// best column is the first one that attains the global maximum, the read end is the smallest row attaining it there.
struct LongScratch {
    int32_t *bh, *bf;      // per column: boundary H and F (int32[refLen] each)
    int32_t *colmax;       // per scan position: maximum over the strips done so far
};

struct StripOut { int maxv, first_col, er; };

// one strip: rows [rbase, rbase + 2048) of the (possibly reversed) read; prof is the strip's profile in LDS
__device__ __forceinline__ StripOut score_strip(const int8_t *__restrict__ ref, int refLen, int dir, int P, int rbase, bool first_strip,
                                                int gapO, int gapE, const int8_t *prof, LongScratch sc) {
    constexpr int R = 32, PL = 64 * R;
    const int lane = threadIdx.x & 63;
    const int row0 = lane * R;
    const int D = R * gapE;
    int H[R], E[R], Hmax[R];
#pragma unroll
    for (int k = 0; k < R; ++k) H[k] = 0, E[k] = 0, Hmax[k] = 0;
    StripOut o;
    o.maxv = 0; o.first_col = -1;
    int rc = 0, t_bh = 0, t_bf = NEG_INF, t_cm = 0, w_bh = 0, w_bf = 0, w_cm = 0, carry_diag = 0;
    for (int c = 0; c < refLen; ++c) {
        if ((c & 63) == 0) {
            const int idx = c + lane;
            rc = idx < refLen ? ref[dir ? refLen - 1 - idx : idx] : 0;
            if (!first_strip && idx < refLen) { t_bh = sc.bh[idx]; t_bf = sc.bf[idx]; }
            t_cm = (!first_strip && idx < refLen) ? sc.colmax[idx] : 0;
        }
        const int sym = __builtin_amdgcn_readlane(rc, c & 63);
        const int in_h = first_strip ? 0 : __builtin_amdgcn_readlane(t_bh, c & 63);        // H(rbase - 1, c)
        const int in_f = first_strip ? NEG_INF : __builtin_amdgcn_readlane(t_bf, c & 63);  // F entering row rbase at c
        const int *pw = reinterpret_cast<const int *>(prof + sym * PL + row0);
        const int sh = wave_shr1(H[R - 1], 0);
        int diag = lane == 0 ? carry_diag : sh;
        carry_diag = in_h;
        int fl = NEG_INF;
#pragma unroll
        for (int q = 0; q < R / 4; ++q) {
            const int w = pw[q];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = q * 4 + b;
                const int s = (w << (24 - 8 * b)) >> 24;
                const int hold = H[k];
                int a = max(diag + s, 0);
                a = max(a, E[k]);
                H[k] = a;
                diag = hold;
                fl = max(fl - gapE, a - gapO);
            }
        }
        int t = wave_scan_max(fl + lane * D);
        t = wave_shr1(t, NEG_INF);
        // F entering this lane's first row: from the lanes before it, and from the previous strip decayed over their rows
        int f = lane == 0 ? in_f : max(t - (lane - 1) * D, in_f - lane * D);
        int cm = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int h = max(H[k], f);
            H[k] = h;
            E[k] = max(E[k] - gapE, h - gapO);
            if (rbase + row0 + k < P) cm = max(cm, h);
            f = max(f - gapE, h - gapO);
        }
        const int colmax = wave_reduce_max(cm);
        if (colmax > o.maxv) {
            o.maxv = colmax;
            o.first_col = c;
#pragma unroll
            for (int k = 0; k < R; ++k) Hmax[k] = H[k];
        }
        // boundary for the next strip (lane 63's last row) and merged column maximum, flushed every 64 columns
        const int out_h = __builtin_amdgcn_readlane(H[R - 1], 63), out_f = __builtin_amdgcn_readlane(f, 63);
        const int merged = max(colmax, __builtin_amdgcn_readlane(t_cm, c & 63));
        if (lane == (c & 63)) { w_bh = out_h; w_bf = out_f; w_cm = merged; }
        if ((c & 63) == 63 || c == refLen - 1) {
            const int idx = (c & ~63) + lane;
            if (idx <= c) { sc.bh[idx] = w_bh; sc.bf[idx] = w_bf; sc.colmax[idx] = w_cm; }
        }
    }
    int er = 1 << 30;
#pragma unroll
    for (int k = R - 1; k >= 0; --k)
        if (Hmax[k] == o.maxv && rbase + row0 + k < P) er = rbase + row0 + k;
    o.er = wave_reduce_min(er);
    return o;
}

__device__ __forceinline__ void build_strip_profile(int8_t *prof, const int8_t *mat, int nsym, const int8_t *read, int readLen,
                                                    bool reversed, int rbase) {
    constexpr int PL = 64 * 32;
    __syncthreads();
    for (int idx = threadIdx.x; idx < nsym * PL; idx += 64) {
        const int sym = idx / PL, j = rbase + idx - sym * PL;
        int8_t v = 0;
        if (j < readLen) v = mat[sym * nsym + read[reversed ? readLen - 1 - j : j]];
        prof[idx] = v;
    }
    __syncthreads();
}

// a whole pass = all strips; results as score_pass defines them
__device__ __forceinline__ PassOut score_pass_long(const int8_t *__restrict__ ref, int refLen, int dir, const int8_t *read, int readLen,
                                                   bool reversed, int sse_lanes, int gapO, int gapE, int bias, const int8_t *mat,
                                                   int nsym, int8_t *prof, LongScratch sc, uint16_t *__restrict__ mcol) {
    const int lane = threadIdx.x & 63;
    const int P = (readLen + sse_lanes - 1) / sse_lanes * sse_lanes;
    const int n_strips = (P + 2047) / 2048;
    int best_er = 1 << 30, gmax = 0;
    // the strips' own results are kept for the merge: (max, first column, row) per strip, lane s holds strip s (<= 64 strips)
    int s_max = 0, s_col = -1, s_er = 1 << 30;
    for (int s = 0; s < n_strips; ++s) {
        build_strip_profile(prof, mat, nsym, read, readLen, reversed, s * 2048);
        const StripOut so = score_strip(ref, refLen, dir, P, s * 2048, s == 0, gapO, gapE, prof, sc);
        if (lane == (s & 63)) { s_max = so.maxv; s_col = so.first_col; s_er = so.er; }
        gmax = max(gmax, so.maxv);
    }
    __syncthreads();  // the column maxima written by this wave are read back below
    PassOut o;
    o.maxv = gmax;
    o.overflow = (sse_lanes == 16 && gmax + bias >= 255) ? 1 : 0;
    o.end_ref = sse_lanes == 16 ? -1 : 0;
    o.end_read = 0;
    if (gmax > 0) {
        int cand = 1 << 30;  // first scan position whose merged column maximum is the global one
        for (int idx = lane; idx < refLen; idx += 64) if (sc.colmax[idx] == gmax) { cand = idx; break; }
        const int cstar = wave_reduce_min(cand);
        o.end_ref = dir ? refLen - 1 - cstar : cstar;
        const int er = (s_max == gmax && s_col == cstar) ? s_er : (1 << 30);
        best_er = wave_reduce_min(er);
        o.end_read = min(readLen - 1, best_er);
    }
    if (mcol) for (int idx = lane; idx < refLen; idx += 64) mcol[idx] = (uint16_t)sc.colmax[idx];
    return o;
}

__global__ __launch_bounds__(64) void ssw_score_long_kernel(SswDev p, const int32_t *__restrict__ order, int32_t *__restrict__ scratch,
                                                            const int64_t *__restrict__ scratch_off) {
    extern __shared__ __attribute__((aligned(16))) int8_t prof[];
    const int job = order[blockIdx.x];
    const int lane = threadIdx.x;
    const int readLen = p.read_len[job], refLen = p.ref_len[job], maskLen = p.mask_len[job];
    const int8_t *read = p.reads + p.read_off[job];
    const int8_t *ref = p.refs + p.ref_off[job];
    uint16_t *mcol = p.mcol + p.mcol_off[job];
    LongScratch sc;
    sc.bh = scratch + scratch_off[job]; sc.bf = sc.bh + refLen; sc.colmax = sc.bf + refLen;
    const bool have_byte = p.score_size == 0 || p.score_size == 2, have_word = p.score_size == 1 || p.score_size == 2;
    int word = 0;
    PassOut o;
    if (have_byte) {
        o = score_pass_long(ref, refLen, 0, read, readLen, false, 16, p.gapO, p.gapE, p.bias, p.mat, p.nsym, prof, sc, mcol);
        if (o.overflow) {
            if (!have_word) { if (lane == 0) p.status[job] = MPN_SSW_ENULL; return; }
            o = score_pass_long(ref, refLen, 0, read, readLen, false, 8, p.gapO, p.gapE, 0, p.mat, p.nsym, prof, sc, mcol);
            word = 1;
        }
    } else {
        o = score_pass_long(ref, refLen, 0, read, readLen, false, 8, p.gapO, p.gapE, 0, p.mat, p.nsym, prof, sc, mcol);
        word = 1;
    }
    int s2 = 0, r2 = 0;
    {
        int e1 = max(o.end_ref - maskLen, 0);
        int e2 = min(o.end_ref + maskLen, refLen);
        int start2 = word ? e2 : e2 + 1;
        int bv = 0, bi = 1 << 30;
        __syncthreads();
        for (int idx = lane; idx < e1; idx += 64) { int v = mcol[idx]; if (v > bv) bv = v, bi = idx; }
        for (int idx = start2 + lane; idx < refLen; idx += 64) { int v = mcol[idx]; if (v > bv) bv = v, bi = idx; }
        s2 = wave_reduce_max(bv);
        int cand = (bv == s2 && s2 > 0) ? bi : (1 << 30);
        cand = wave_reduce_min(cand);
        r2 = s2 > 0 ? cand : 0;
    }
    int ref_begin = -1, read_begin = -1, need = 0, st = MPN_SSW_OK;
    const int score1 = o.maxv, ref_end1 = o.end_ref, read_end1 = o.end_read;
    const bool stop = p.flag == 0 || (p.flag == 2 && score1 < p.filters);
    if (!stop) {
        const int rl = read_end1 + 1;
        // the reverse pass stops at the column whose maximum equals score1: the first column attaining the pass's maximum
        PassOut rv = score_pass_long(ref, ref_end1 + 1, 1, read, rl, true, word ? 8 : 16, p.gapO, p.gapE, word ? 0 : p.bias, p.mat,
                                     p.nsym, prof, sc, nullptr);
        ref_begin = rv.end_ref;
        read_begin = read_end1 - rv.end_read;
        const bool nocig = (7 & p.flag) == 0 || ((2 & p.flag) != 0 && score1 < p.filters) ||
                           ((4 & p.flag) != 0 && (ref_end1 - ref_begin > p.filterd || read_end1 - read_begin > p.filterd));
        if (!nocig) {
            if (ref_begin < 0 || read_begin < 0 || ref_end1 - ref_begin + 1 <= 0 || read_end1 - read_begin + 1 <= 0)
                st = MPN_SSW_EUNDEF;
            else
                need = 1;
        }
    }
    if (lane == 0) {
        p.score1[job] = score1;
        p.ref_end1[job] = ref_end1;
        p.read_end1[job] = read_end1;
        p.score2[job] = maskLen >= 15 ? s2 : 0;
        p.ref_end2[job] = maskLen >= 15 ? r2 : -1;
        p.ref_begin1[job] = ref_begin;
        p.read_begin1[job] = read_begin;
        p.need_cigar[job] = need;
        p.status[job] = st;
    }
}

// ---------------------------------------------------------------------------------------------
// banded traceback
struct BandDev {
    SswDev s;
    const int32_t *jobs;      // pair index per block
    const int32_t *bw;        // band half-width for this round, per block
    const int64_t *dir_off;   // per block offset into dir
    uint8_t *dir;             // packed direction codes, 1 B per band cell
    int32_t *bmax;            // running DP max per pair (persists over rounds, ssw.c:544 `max`)
    int32_t *done;            // per block: 1 = traced, 0 = band too narrow, <0 error
    uint32_t *cig;            // per pair scratch, ops written back to front
    const int64_t *cig_off;   // per pair END offset (exclusive) of its scratch region
    int32_t *cig_len;
};

__global__ __launch_bounds__(64) void ssw_banded_kernel(BandDev b) {
    extern __shared__ __attribute__((aligned(16))) int rows[];  // 4 arrays of W ints
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int job = b.jobs[blk];
    const int bw = b.bw[blk];
    const int width_d = 2 * bw + 1, W = width_d + 2;
    int *hrow[2] = {rows, rows + W};
    int *erow[2] = {rows + 2 * W, rows + 3 * W};
    const SswDev &p = b.s;
    const int rb = p.ref_begin1[job], qb = p.read_begin1[job];
    const int refLen = p.ref_end1[job] - rb + 1, readLen = p.read_end1[job] - qb + 1;
    const int8_t *ref = p.refs + p.ref_off[job] + rb;
    const int8_t *read = p.reads + p.read_off[job] + qb;
    const int gapO = p.gapO, gapE = p.gapE, n = p.nsym, score = p.score1[job];
    uint8_t *dir = b.dir + b.dir_off[blk];
    int mx = 0;
    for (int i = 0; i < readLen; ++i) {
        const int x = max(i - bw, 0), xp = max(i - 1 - bw, 0);
        const int end = min(refLen - 1, i + bw), endp = min(refLen - 1, i - 1 + bw);
        const int ncell = end - x + 1;
        const int *hp = hrow[(i + 1) & 1], *ep = erow[(i + 1) & 1];
        int *hc = hrow[i & 1], *ec = erow[i & 1];
        const int8_t *mrow_base = p.mat;
        const int qc = read[i];
        uint8_t *dl = dir + (int64_t)width_d * i;
        int carry_f = 0, carry_h = 0;  // f and h of the cell left of the chunk (hc[0] = f = 0 at row start, ssw.c:580)
        for (int c0 = 0; c0 < ncell; c0 += 64) {
            const int cj = c0 + lane, j = x + cj;
            const bool act = cj < ncell;
            int up_h = 0, up_e = 0, dg = 0, s = 0;
            if (act) {
                const bool upv = i > 0 && j <= endp && !(j == end && i <= bw + 1);  // ssw.c:579-580 `edge` clearing
                if (upv) up_h = hp[j - xp], up_e = ep[j - xp];
                if (i > 0 && j - 1 >= xp) dg = hp[j - 1 - xp];
                s = mrow_base[(int)ref[j] * n + qc];
            }
            const int t1 = i == 0 ? -gapO : up_h - gapO, t2 = i == 0 ? -gapE : up_e - gapE;
            const int ev = max(t1, t2);
            const int de = t1 > t2 ? 3 : 2;
            const int e1 = max(ev, 0);
            const int temp2 = dg + s;
            const int B = max(e1, temp2);
            // f(j) = max(h(j-1) - gapO, f(j-1) - gapE); exact via scan because gapO > gapE and B >= 0
            int t = wave_scan_max(act ? B - gapO + lane * gapE : NEG_INF);
            t = wave_shr1(t, NEG_INF);
            const int f_first = max(carry_h - gapO, carry_f - gapE);
            int f = lane == 0 ? f_first : max(t - (lane - 1) * gapE, f_first - lane * gapE);
            const int f1 = max(f, 0);
            const int temp1 = max(e1, f1);
            const int h = max(temp1, temp2);
            // DPP reads are hoisted out of the selects: a DPP executed under a partial EXEC mask would see
            // the masked-off source lane as invalid
            const int h_sh = wave_shr1(h, 0), f_sh = wave_shr1(f, 0);
            const int hl = lane == 0 ? carry_h : h_sh;
            const int fleft = lane == 0 ? carry_f : f_sh;
            const int df = (hl - gapO > fleft - gapE) ? 5 : 4;
            const int dh = temp1 <= temp2 ? 1 : (e1 > f1 ? de : df);
            if (act) {
                mx = max(mx, h);
                hc[cj] = h;
                ec[cj] = ev;
                dl[cj] = (uint8_t)((de - 2) | ((df - 4) << 1) | (dh << 2));
            }
            carry_h = __builtin_amdgcn_readlane(h, 63);
            carry_f = __builtin_amdgcn_readlane(f, 63);
        }
        __syncthreads();
    }
    mx = wave_reduce_max(mx);
    const int prev = b.bmax[job];
    mx = max(mx, prev);
    if (lane == 0) b.bmax[job] = mx;
    if (mx < score) { if (lane == 0) b.done[blk] = 0; return; }
    __threadfence_block();
    __syncthreads();
    if (lane != 0) return;
    // traceback, ssw.c:618-697; ops are produced last-to-first, which is the final order reversed
    uint32_t *cend = b.cig + b.cig_off[job];
    int l = 0, e = 0, plane = 2, i = readLen - 1, j = refLen - 1;
    int op = 0, prev_op = 0;  // 0 M, 1 I, 2 D
    while (i > 0) {
        const int x = max(i - bw, 0), cj = j - x, hi = min(refLen - 1, i + bw);
        if (j < 0 || cj < 0 || cj >= width_d || j > hi) { b.done[blk] = -MPN_SSW_EUNDEF; return; }
        const int code = dir[(int64_t)width_d * i + cj];
        const int d = plane == 0 ? 2 + (code & 1) : plane == 1 ? 4 + ((code >> 1) & 1) : (code >> 2);
        switch (d) {
        case 1: --i; --j; plane = 2; op = 0; break;
        case 2: --i; plane = 0; op = 1; break;
        case 3: --i; plane = 2; op = 1; break;
        case 4: --j; plane = 1; op = 2; break;
        case 5: --j; plane = 2; op = 2; break;
        default: b.done[blk] = -MPN_SSW_ENULL; return;
        }
        if (op == prev_op) ++e;
        else {
            ++l; cend[-l] = ((uint32_t)e << 4) | prev_op;
            prev_op = op; e = 1;
        }
    }
    if (op == 0) { ++l; cend[-l] = ((uint32_t)(e + 1) << 4); }
    else {
        ++l; cend[-l] = ((uint32_t)e << 4) | op;
        ++l; cend[-l] = 1u << 4;
    }
    b.cig_len[job] = l;
    b.done[blk] = 1;
}

// ---------------------------------------------------------------------------------------------
// host side
template <int R>
static int launch_score(const SswDev &d, const int32_t *order_dev, int count, hipStream_t st) {
    if (count == 0) return 0;
    size_t lds = (size_t)d.nsym * 64 * R;
    hipLaunchKernelGGL(ssw_score_kernel<R>, dim3(count), dim3(64), lds, st, d, order_dev);
    MPN_HIP_CHECK(hipGetLastError());
    return 0;
}

static int ssw_batch_impl(int32_t n_pairs, const int8_t *reads, const int64_t *read_off, const int32_t *read_len,
                          const int8_t *refs, const int64_t *ref_off, const int32_t *ref_len, const int8_t *mat,
                          int32_t n, int8_t score_size, uint8_t gap_open, uint8_t gap_extend, uint8_t flag,
                          uint16_t filters, int32_t filterd, const int32_t *mask_len, uint16_t *score1,
                          uint16_t *score2, int32_t *ref_begin1, int32_t *ref_end1, int32_t *read_begin1,
                          int32_t *read_end1, int32_t *ref_end2, uint32_t *cigar_pool, int64_t cigar_cap,
                          int64_t *cigar_off, int32_t *cigar_len, int32_t *status) {
    if (n_pairs <= 0) return 0;
    if (n < 1 || n > 8) { set_error("mpn_ssw_align_batch: matrix edge n=%d outside 1..8", n); return -2; }
    hipStream_t st = 0;
    std::vector<int32_t> st_h(n_pairs, MPN_SSW_OK);
    // total extents of the input buffers
    int64_t reads_total = 0, refs_total = 0;
    std::vector<int64_t> mcol_off(n_pairs);
    int64_t mcol_total = 0;
    std::vector<int32_t> cls[5];  // [4]: reads of more than 2048 rows (strip kernel)
    std::vector<int64_t> long_off(n_pairs, 0);
    int64_t long_total = 0;
    for (int i = 0; i < n_pairs; ++i) {
        cigar_len[i] = 0; cigar_off[i] = 0;
        reads_total = std::max<int64_t>(reads_total, read_off[i] + std::max(read_len[i], 0));
        refs_total = std::max<int64_t>(refs_total, ref_off[i] + std::max(ref_len[i], 0));
        mcol_off[i] = mcol_total;
        mcol_total += ((int64_t)std::max(ref_len[i], 0) + 63) / 64 * 64;
        if (gap_open <= gap_extend) st_h[i] = MPN_SSW_EDOMAIN;
        else if (read_len[i] <= 0 || ref_len[i] < 0) st_h[i] = MPN_SSW_EUNDEF;
        else if ((int64_t)read_len[i] > 64 * 2048) st_h[i] = MPN_SSW_ETOOLONG;  // (64 strips of 2048 rows)
        else if (score_size < 0 || score_size > 2) st_h[i] = MPN_SSW_ENULL;  // ssw.c:801-804
        else {
            int P = (read_len[i] + 15) / 16 * 16;
            cls[P <= 256 ? 0 : P <= 512 ? 1 : P <= 1024 ? 2 : P <= 2048 ? 3 : 4].push_back(i);
            if (P > 2048) { long_off[i] = long_total; long_total += 3 * (int64_t)std::max(ref_len[i], 1); }
        }
    }
    SswDev d;
    memset(&d, 0, sizeof(d));
    DevBuf<int8_t> d_reads, d_refs;
    DevBuf<int64_t> d_read_off, d_ref_off, d_mcol_off;
    DevBuf<int32_t> d_read_len, d_ref_len, d_mask, d_res, d_order;
    DevBuf<uint16_t> d_mcol;
    if (d_reads.upload(reads, reads_total, st) || d_refs.upload(refs, refs_total, st) ||
        d_read_off.upload(read_off, n_pairs, st) || d_ref_off.upload(ref_off, n_pairs, st) ||
        d_read_len.upload(read_len, n_pairs, st) || d_ref_len.upload(ref_len, n_pairs, st) ||
        d_mask.upload(mask_len, n_pairs, st) || d_mcol_off.upload(mcol_off.data(), n_pairs, st) ||
        d_mcol.alloc(mcol_total) || d_res.alloc((size_t)n_pairs * 9))
        return -1;
    MPN_HIP_CHECK(hipMemsetAsync(d_res.p, 0, (size_t)n_pairs * 9 * sizeof(int32_t), st));
    d.reads = d_reads.p; d.refs = d_refs.p; d.read_off = d_read_off.p; d.ref_off = d_ref_off.p;
    d.read_len = d_read_len.p; d.ref_len = d_ref_len.p; d.mask_len = d_mask.p;
    memcpy(d.mat, mat, (size_t)n * n);
    d.nsym = n; d.score_size = score_size; d.gapO = gap_open; d.gapE = gap_extend; d.flag = flag;
    d.filters = filters; d.filterd = filterd;
    {
        int bias = 0;
        for (int i = 0; i < n * n; ++i) if (mat[i] < bias) bias = mat[i];
        d.bias = abs(bias) & 0xff;
    }
    int32_t *r = d_res.p;
    d.score1 = r; d.score2 = r + n_pairs; d.ref_begin1 = r + 2 * (size_t)n_pairs; d.ref_end1 = r + 3 * (size_t)n_pairs;
    d.read_begin1 = r + 4 * (size_t)n_pairs; d.read_end1 = r + 5 * (size_t)n_pairs; d.ref_end2 = r + 6 * (size_t)n_pairs;
    d.status = r + 7 * (size_t)n_pairs; d.need_cigar = r + 8 * (size_t)n_pairs;
    d.mcol = d_mcol.p; d.mcol_off = d_mcol_off.p;

    std::vector<int32_t> order;
    int cnt[5], base[5];
    for (int c = 0; c < 5; ++c) { base[c] = (int)order.size(); cnt[c] = (int)cls[c].size(); order.insert(order.end(), cls[c].begin(), cls[c].end()); }
    if (d_order.upload(order.data(), order.size(), st)) return -1;
    if (launch_score<4>(d, d_order.p + base[0], cnt[0], st) || launch_score<8>(d, d_order.p + base[1], cnt[1], st) ||
        launch_score<16>(d, d_order.p + base[2], cnt[2], st) || launch_score<32>(d, d_order.p + base[3], cnt[3], st))
        return -1;
    DevBuf<int32_t> d_long;
    DevBuf<int64_t> d_long_off;
    if (cnt[4] > 0) {
        if (d_long.alloc((size_t)long_total) || d_long_off.upload(long_off.data(), n_pairs, st)) return -1;
        hipLaunchKernelGGL(ssw_score_long_kernel, dim3(cnt[4]), dim3(64), (size_t)d.nsym * 64 * 32, st, d, d_order.p + base[4], d_long.p,
                           (const int64_t *)d_long_off.p);
        MPN_HIP_CHECK(hipGetLastError());
    }
    std::vector<int32_t> res((size_t)n_pairs * 9);
    if (d_res.download(res.data(), res.size(), st)) return -1;
    MPN_HIP_CHECK(hipStreamSynchronize(st));
    const int32_t *h_s1 = &res[0], *h_s2 = &res[n_pairs], *h_rb = &res[2 * (size_t)n_pairs], *h_re = &res[3 * (size_t)n_pairs];
    const int32_t *h_qb = &res[4 * (size_t)n_pairs], *h_qe = &res[5 * (size_t)n_pairs], *h_re2 = &res[6 * (size_t)n_pairs];
    const int32_t *h_st = &res[7 * (size_t)n_pairs], *h_need = &res[8 * (size_t)n_pairs];
    std::vector<int32_t> pending;
    std::vector<int64_t> cig_end(n_pairs, 0);
    int64_t cig_total = 0;
    for (int i = 0; i < n_pairs; ++i) {
        if (st_h[i] != MPN_SSW_OK) {
            score1[i] = score2[i] = 0; ref_begin1[i] = read_begin1[i] = -1; ref_end1[i] = read_end1[i] = ref_end2[i] = 0;
            continue;
        }
        st_h[i] = h_st[i];
        score1[i] = (uint16_t)h_s1[i]; score2[i] = (uint16_t)h_s2[i];
        ref_begin1[i] = h_rb[i]; ref_end1[i] = h_re[i]; read_begin1[i] = h_qb[i]; read_end1[i] = h_qe[i]; ref_end2[i] = h_re2[i];
        if (h_st[i] == MPN_SSW_OK && h_need[i]) {
            pending.push_back(i);
            cig_total += (int64_t)(h_re[i] - h_rb[i] + 1) + (h_qe[i] - h_qb[i] + 1) + 4;
            cig_end[i] = cig_total;
        }
    }
    if (!pending.empty()) {
        DevBuf<int64_t> d_cig_end;
        DevBuf<uint32_t> d_cig;
        DevBuf<int32_t> d_bmax, d_ciglen;
        if (d_cig_end.upload(cig_end.data(), n_pairs, st) || d_cig.alloc(cig_total) || d_bmax.alloc(n_pairs) || d_ciglen.alloc(n_pairs))
            return -1;
        MPN_HIP_CHECK(hipMemsetAsync(d_bmax.p, 0, (size_t)n_pairs * 4, st));
        MPN_HIP_CHECK(hipMemsetAsync(d_ciglen.p, 0, (size_t)n_pairs * 4, st));
        std::vector<int32_t> bw(n_pairs, 0);
        for (int i : pending) bw[i] = abs((h_re[i] - h_rb[i] + 1) - (h_qe[i] - h_qb[i] + 1)) + 1;
        for (int round = 0; !pending.empty() && round < 40; ++round) {
            // chunk the pending list so that the direction scratch stays below ~2 GiB per launch
            size_t pos = 0;
            std::vector<int32_t> next;
            while (pos < pending.size()) {
                std::vector<int32_t> jobs, bws;
                std::vector<int64_t> doff;
                int64_t dtot = 0;
                int maxw = 0;
                while (pos < pending.size()) {
                    int i = pending[pos];
                    int64_t wd = 2 * (int64_t)bw[i] + 1, sz = wd * (h_qe[i] - h_qb[i] + 1);
                    if (sz * 3 > ((int64_t)1 << 31) - 1 || wd + 2 > 9000) { st_h[i] = MPN_SSW_EUNDEF; ++pos; continue; }
                    if (!jobs.empty() && dtot + sz > ((int64_t)1 << 31)) break;
                    jobs.push_back(i); bws.push_back(bw[i]); doff.push_back(dtot);
                    dtot += (sz + 15) / 16 * 16;
                    maxw = std::max<int>(maxw, (int)wd + 2);
                    ++pos;
                }
                if (jobs.empty()) continue;
                DevBuf<int32_t> d_jobs, d_bw, d_done;
                DevBuf<int64_t> d_doff;
                DevBuf<uint8_t> d_dir;
                if (d_jobs.upload(jobs.data(), jobs.size(), st) || d_bw.upload(bws.data(), bws.size(), st) ||
                    d_doff.upload(doff.data(), doff.size(), st) || d_dir.alloc(dtot) || d_done.alloc(jobs.size()))
                    return -1;
                MPN_HIP_CHECK(hipMemsetAsync(d_done.p, 0, jobs.size() * 4, st));
                BandDev b;
                b.s = d; b.jobs = d_jobs.p; b.bw = d_bw.p; b.dir_off = d_doff.p; b.dir = d_dir.p; b.bmax = d_bmax.p;
                b.done = d_done.p; b.cig = d_cig.p; b.cig_off = d_cig_end.p; b.cig_len = d_ciglen.p;
                size_t lds = (size_t)maxw * 4 * sizeof(int);
                if (lds > 64 * 1024)
                    MPN_HIP_CHECK(hipFuncSetAttribute((const void *)ssw_banded_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(ssw_banded_kernel, dim3((unsigned)jobs.size()), dim3(64), lds, st, b);
                MPN_HIP_CHECK(hipGetLastError());
                std::vector<int32_t> done(jobs.size());
                if (d_done.download(done.data(), done.size(), st)) return -1;
                MPN_HIP_CHECK(hipStreamSynchronize(st));
                for (size_t k = 0; k < jobs.size(); ++k) {
                    int i = jobs[k];
                    if (done[k] == 0) { bw[i] *= 2; next.push_back(i); }
                    else if (done[k] < 0) st_h[i] = -done[k];
                }
            }
            pending.swap(next);
        }
        for (int i : pending) st_h[i] = MPN_SSW_EUNDEF;  // band never reproduced the score
        std::vector<uint32_t> cig_h((size_t)cig_total);
        std::vector<int32_t> cl(n_pairs);
        if (d_cig.download(cig_h.data(), cig_h.size(), st) || d_ciglen.download(cl.data(), n_pairs, st)) return -1;
        MPN_HIP_CHECK(hipStreamSynchronize(st));
        int64_t used = 0;
        for (int i = 0; i < n_pairs; ++i) {
            if (st_h[i] != MPN_SSW_OK || !h_need[i] || cl[i] <= 0) continue;
            if (used + cl[i] > cigar_cap) { st_h[i] = MPN_SSW_ECIGAR_CAP; continue; }
            cigar_off[i] = used;
            cigar_len[i] = cl[i];
            memcpy(cigar_pool + used, &cig_h[(size_t)(cig_end[i] - cl[i])], (size_t)cl[i] * 4);
            used += cl[i];
        }
    }
    for (int i = 0; i < n_pairs; ++i) status[i] = st_h[i];
    return 0;
}

}  // namespace mpn

// ---------------------------------------------------------------------------------------------
// C-ABI
struct _profile {  // opaque to callers (pyssw.py only passes the pointer back)
    const int8_t *read;
    const int8_t *mat;
    int32_t readLen, n;
    int8_t score_size;
};

extern "C" {

int mpn_ssw_align_batch(int32_t n_pairs, const int8_t *reads, const int64_t *read_off, const int32_t *read_len,
                        const int8_t *refs, const int64_t *ref_off, const int32_t *ref_len, const int8_t *mat,
                        int32_t n, int8_t score_size, uint8_t gap_open, uint8_t gap_extend, uint8_t flag,
                        uint16_t filters, int32_t filterd, const int32_t *mask_len, uint16_t *score1, uint16_t *score2,
                        int32_t *ref_begin1, int32_t *ref_end1, int32_t *read_begin1, int32_t *read_end1,
                        int32_t *ref_end2, uint32_t *cigar_pool, int64_t cigar_cap, int64_t *cigar_off,
                        int32_t *cigar_len, int32_t *status) {
    return mpn::ssw_batch_impl(n_pairs, reads, read_off, read_len, refs, ref_off, ref_len, mat, n, score_size, gap_open,
                               gap_extend, flag, filters, filterd, mask_len, score1, score2, ref_begin1, ref_end1,
                               read_begin1, read_end1, ref_end2, cigar_pool, cigar_cap, cigar_off, cigar_len, status);
}

s_profile *ssw_init(const int8_t *read, const int32_t readLen, const int8_t *mat, const int32_t n, const int8_t score_size) {
    s_profile *p = (s_profile *)calloc(1, sizeof(s_profile));
    p->read = read; p->mat = mat; p->readLen = readLen; p->n = n; p->score_size = score_size;
    return p;
}

void init_destroy(s_profile *p) { free(p); }

s_align *ssw_align(const s_profile *prof, const int8_t *ref, int32_t refLen, const uint8_t weight_gapO,
                   const uint8_t weight_gapE, const uint8_t flag, const uint16_t filters, const int32_t filterd,
                   const int32_t maskLen) {
    if (maskLen < 15)
        fprintf(stderr, "When maskLen < 15, the function ssw_align doesn't return 2nd best alignment information.\n");
    int64_t zero = 0, coff = 0;
    int32_t rl = prof->readLen, fl = refLen, ml = maskLen, clen = 0, st = 0;
    uint16_t s1 = 0, s2 = 0;
    int32_t rb = -1, re = 0, qb = -1, qe = 0, re2 = 0;
    int64_t cap = (int64_t)std::max(rl, 0) + std::max(fl, 0) + 8;
    uint32_t *cig = (uint32_t *)malloc((size_t)cap * 4);
    int rc = mpn::ssw_batch_impl(1, prof->read, &zero, &rl, ref, &zero, &fl, prof->mat, prof->n, prof->score_size,
                                 weight_gapO, weight_gapE, flag, filters, filterd, &ml, &s1, &s2, &rb, &re, &qb, &qe, &re2,
                                 cig, cap, &coff, &clen, &st);
    if (rc != 0) { fprintf(stderr, "libmpn ssw_align: %s\n", mpn_last_error()); free(cig); return NULL; }
    if (st != MPN_SSW_OK) {
        if (st == MPN_SSW_ENULL && prof->score_size == 0)
            fprintf(stderr, "Please set 2 to the score_size parameter of the function ssw_init, otherwise the alignment results will be incorrect.\n");
        else
            fprintf(stderr, "libmpn ssw_align: pair rejected (status %d, see mpn_ssw.h)\n", st);
        free(cig);
        return NULL;
    }
    s_align *r = (s_align *)calloc(1, sizeof(s_align));
    r->score1 = s1; r->score2 = s2; r->ref_begin1 = rb; r->ref_end1 = re; r->read_begin1 = qb; r->read_end1 = qe; r->ref_end2 = re2;
    if (clen > 0) { r->cigar = cig; r->cigarLen = clen; }
    else { free(cig); r->cigar = 0; r->cigarLen = 0; }
    return r;
}

void align_destroy(s_align *a) {
    if (!a) return;
    free(a->cigar);
    free(a);
}

}  // extern "C"
