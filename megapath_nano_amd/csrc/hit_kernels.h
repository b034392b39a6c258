// Hits from chains on the GPU: minimap2's mm_gen_regs (hash-keyed sort of a read's chains), mm_set_parent, mm_select_sub,
// mm_squeeze_a (as segments for anchor_squeeze_kernel) and mm_join_long, one wave per read, from the chain records
// chain_backtrack_kernel left in HBM.  The host receives the SELECTED hits only (72 bytes each) -- a strain-rich target set has a
// hundred chains per read of which a handful survive -- and never sorts, groups or filters a chain.
//
// The read's chains live in LDS as a struct of arrays.  The two sorts are rank sorts over the 64 lanes (a read has tens of
// chains); the grouping / selection / joining passes are the sequential algorithms of hit.c on lane 0 (their state is a few
// LDS words per step: ~100 us for a read with a hundred chains, thousands of reads in flight).  Reads with more than
// HIT_MAX_CHAINS chains are listed for the host, which runs the same functions (csrc/align.hip) on the downloaded records.
#pragma once
#include "map_types.h"
#include "mpn_common.h"

namespace mpn {

constexpr int HIT_MAX_CHAINS = 384;   // LDS of the large instantiation: 136 bytes per chain
constexpr int HIT_SMALL_CHAINS = 48;  // the small one (most reads of a random target set): 6.5 KB, so that a CU holds many waves

struct HitSelParams {
    float mask_level, pri_ratio, min_join_flank_ratio;
    int32_t min_diff, best_n, max_join_long, max_join_short, min_join_flank_sc, min_cnt, with_cigar;
    int32_t max_chains;  // <= HIT_MAX_CHAINS: reads with more chains are left to the host (MPN_HIT_MAX_CHAINS: tests mix both paths)
    uint32_t seed_mix;   // wang32(opt->seed)
};

// a selected hit as the host takes it over (the rest of its mm_reg1_t follows from these: mm_reg_set_coor)
struct HitRec { uint64_t fx, fy, lx, ly; int32_t score, score0, cnt, as, parent, subsc, n_sub, mlen, blen; uint32_t hash; };
// per read: where its hits are, how many, the anchors of its squeezed list; flags bit 0: ids were re-synchronised (mm_sync_regs
// ran, so mm_set_sam_pri has); n_regs = -1: too many chains, left to the host
struct HitRead { int64_t reg_pos; int32_t n_regs, n_a, flags, pad; };

__device__ __forceinline__ uint64_t hs_hash64(uint64_t key) {
    key = ~key + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}
__device__ __forceinline__ uint32_t hs_wang32(uint32_t key) {
    key += ~(key << 15); key ^= (key >> 10); key += (key << 3);
    key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
    return key;
}

// N: chains a read may have in this instantiation; it takes the reads with lo_excl < chains <= min(N, prm.max_chains), and the
// instantiation with lo_excl == 0 also writes the records of the reads without chains; the one with N == HIT_MAX_CHAINS those
// of the reads it leaves to the host
template <int N>
__global__ __launch_bounds__(64) void hit_select_kernel(HitSelParams prm, int lo_excl, int n_reads, const int32_t *__restrict__ n_chain,
                                                        const int64_t *__restrict__ u_pos, const int64_t *__restrict__ b_pos,
                                                        const uint64_t *__restrict__ Uc, const ChainRec *__restrict__ Rc,
                                                        const int32_t *__restrict__ qlens, const uint32_t *__restrict__ name_hash,
                                                        HitRec *__restrict__ out_regs, SqueezeSeg *__restrict__ out_segs,
                                                        unsigned long long *__restrict__ counters, HitRead *__restrict__ out_reads) {
    // sort keys / scratch
    __shared__ uint64_t k0[N], k1[N];
    __shared__ int32_t ord[N];
    // the hits (index = id)
    __shared__ uint64_t FX[N], FY[N], LX[N], LY[N], cov[N];
    __shared__ int32_t SC[N], SC0[N], CNT[N], AS[N], PAR[N], SUB[N], NSUB[N], ML[N], BL[N], SRC[N], SEG[N], QS[N], QE[N], W[N], TMP[N];
    __shared__ int32_t GSRC[N], GDST[N], GCNT[N];   // the squeeze segments (k1[seg] = long-join mark)
    __shared__ uint32_t HSH[N];
    __shared__ int s_n, s_na, s_nseg, s_flags;
    __shared__ unsigned long long s_rpos, s_spos;
    const int lane = threadIdx.x;
    for (int read = blockIdx.x; read < n_reads; read += gridDim.x) {
        const int nc = n_chain[read];
        if (nc == 0) { if (lane == 0 && lo_excl == 0) out_reads[read] = HitRead{0, 0, 0, 0, 0}; continue; }
        if (nc > prm.max_chains) {
            if (lane == 0 && N == HIT_MAX_CHAINS) { out_reads[read] = HitRead{0, -1, 0, 0, 0}; atomicAdd(&counters[2], 1ULL); }
            continue;
        }
        if (nc <= lo_excl || nc > N) continue;
        const int qlen = qlens[read];
        const uint64_t *u = Uc + u_pos[read];
        const ChainRec *rc = Rc + u_pos[read];
        uint32_t hash = name_hash[read];
        hash ^= hs_wang32((uint32_t)qlen) + prm.seed_mix;
        hash = hs_wang32(hash);
        // ---- chains in the order of their first anchors (ties: pool order), their places `as` in that order ----
        for (int c = lane; c < nc; c += 64) { k0[c] = rc[c].fx; TMP[c] = (int32_t)u[c]; }
        __syncthreads();
        for (int c = lane; c < nc; c += 64) {
            const uint64_t key = k0[c];
            int rank = 0;
            for (int j = 0; j < nc; ++j) { const uint64_t kj = k0[j]; rank += (kj < key || (kj == key && j < c)) ? 1 : 0; }
            ord[rank] = c;
        }
        __syncthreads();
        // prefix sum of the counts in that order (as), and the start of every chain's anchors in the pool (src, pool order)
        {
            int carry = 0, carry_src = 0;
            for (int t0 = 0; t0 < nc; t0 += 64) {
                const int t = t0 + lane;
                const int cnt_sorted = t < nc ? TMP[ord[t]] : 0, cnt_pool = t < nc ? TMP[t] : 0;
                const int inc = wave_scan_add(cnt_sorted), inc2 = wave_scan_add(cnt_pool);
                if (t < nc) { AS[t] = carry + inc - cnt_sorted; W[t] = carry_src + inc2 - cnt_pool; }   // AS by sorted place, W = src by pool index
                carry += __builtin_amdgcn_readlane(inc, 63);
                carry_src += __builtin_amdgcn_readlane(inc2, 63);
            }
        }
        __syncthreads();
        // ---- mm_gen_regs: keys (score ^ hash, as << 32 | cnt), descending ----
        for (int t = lane; t < nc; t += 64) {
            const int c = ord[t];
            const ChainRec r = rc[c];
            const uint32_t h = (uint32_t)hs_hash64((hs_hash64(r.fx) + hs_hash64(r.fy)) ^ hash);
            k0[t] = u[c] ^ (uint64_t)h;
            k1[t] = (uint64_t)AS[t] << 32 | (uint32_t)TMP[c];
        }
        __syncthreads();
        for (int t = lane; t < nc; t += 64) {
            const uint64_t x = k0[t], y = k1[t];
            int rank = 0;
            for (int j = 0; j < nc; ++j) { const uint64_t xj = k0[j], yj = k1[j]; rank += (xj > x || (xj == x && yj > y)) ? 1 : 0; }
            const int c = ord[t];
            const ChainRec r = rc[c];
            const int i = rank;
            FX[i] = r.fx; FY[i] = r.fy; LX[i] = r.lx; LY[i] = r.ly;
            SC[i] = SC0[i] = (int32_t)(x >> 32); HSH[i] = (uint32_t)x; CNT[i] = (int32_t)y; PAR[i] = -1; SUB[i] = 0; NSUB[i] = 0;
            ML[i] = r.mlen; BL[i] = r.blen; SRC[i] = W[c]; SEG[i] = -1;
            TMP[i] = (int32_t)(y >> 32);   // as (moved to AS below: AS is still read by other lanes)
            const int32_t q_span = (int32_t)(r.fy >> 32 & 0xff);
            const bool rev = r.fx >> 63;
            QS[i] = !rev ? (int32_t)r.fy + 1 - q_span : qlen - ((int32_t)r.ly + 1);
            QE[i] = !rev ? (int32_t)r.ly + 1 : qlen - ((int32_t)r.fy + 1 - q_span);
        }
        __syncthreads();
        for (int t = lane; t < nc; t += 64) AS[t] = TMP[t];
        __syncthreads();
        if (lane == 0) {
            int n = nc, flags = 0;
            auto rid_of = [&](int i) { return (int32_t)(FX[i] << 1 >> 33); };
            auto rev_of = [&](int i) { return (int32_t)(FX[i] >> 63); };
            auto rs_of = [&](int i) { const int32_t sp = (int32_t)(FY[i] >> 32 & 0xff); return (int32_t)FX[i] + 1 > sp ? (int32_t)FX[i] + 1 - sp : 0; };
            auto re_of = [&](int i) { return (int32_t)LX[i] + 1; };
            auto set_q = [&](int i) {
                const int32_t sp = (int32_t)(FY[i] >> 32 & 0xff);
                if (!rev_of(i)) { QS[i] = (int32_t)FY[i] + 1 - sp; QE[i] = (int32_t)LY[i] + 1; }
                else { QS[i] = qlen - ((int32_t)LY[i] + 1); QE[i] = qlen - ((int32_t)FY[i] + 1 - sp); }
            };
            auto move = [&](int dst, int src) {
                FX[dst] = FX[src]; FY[dst] = FY[src]; LX[dst] = LX[src]; LY[dst] = LY[src];
                SC[dst] = SC[src]; SC0[dst] = SC0[src]; CNT[dst] = CNT[src]; AS[dst] = AS[src]; PAR[dst] = PAR[src]; SUB[dst] = SUB[src];
                NSUB[dst] = NSUB[src]; ML[dst] = ML[src]; BL[dst] = BL[src]; SRC[dst] = SRC[src]; SEG[dst] = SEG[src]; QS[dst] = QS[src];
                QE[dst] = QE[src]; HSH[dst] = HSH[src]; ord[dst] = ord[src];
            };
            // ids: ord[i] holds hit i's id while hits are dropped and re-numbered (mm_sync_regs)
            for (int i = 0; i < n; ++i) ord[i] = i;
            // mm_sync_regs over hits [0, n) whose ids are in ord[] and whose parents refer to ids
            auto sync = [&]() {
                int max_id = -1;
                for (int i = 0; i < n; ++i) max_id = max_id > ord[i] ? max_id : ord[i];
                for (int i = 0; i <= max_id; ++i) TMP[i] = -1;
                for (int i = 0; i < n; ++i) if (ord[i] >= 0) TMP[ord[i]] = i;
                for (int i = 0; i < n; ++i) {
                    ord[i] = i;
                    const int p = PAR[i];
                    if (p == -2) PAR[i] = i;
                    else if (p >= 0 && p <= max_id && TMP[p] >= 0) PAR[i] = TMP[p];
                    else PAR[i] = -1;
                }
                flags |= 1;
            };
            // ---- mm_set_parent (no hit has a base-level alignment yet) ----
            {
                W[0] = 0; PAR[0] = 0;
                int k = 1;
                for (int i = 1; i < n; ++i) {
                    const int si = QS[i], ei = QE[i];
                    int n_cov = 0, uncov_len = 0, j;
                    for (j = 0; j < k; ++j) {
                        const int p = W[j];
                        int sj = QS[p], ej = QE[p];
                        if (ej <= si || sj >= ei) continue;
                        if (sj < si) sj = si;
                        if (ej > ei) ej = ei;
                        // (insertion into the sorted list: what the std::sort of the host code leaves)
                        const uint64_t v = (uint64_t)sj << 32 | (uint32_t)ej;
                        int q = n_cov++;
                        while (q > 0 && cov[q - 1] > v) { cov[q] = cov[q - 1]; --q; }
                        cov[q] = v;
                    }
                    if (n_cov > 0) {
                        int x = si;
                        for (j = 0; j < n_cov; ++j) {
                            if ((int)(cov[j] >> 32) > x) uncov_len += (int)(cov[j] >> 32) - x;
                            x = (int32_t)cov[j] > x ? (int32_t)cov[j] : x;
                        }
                        if (ei > x) uncov_len += ei - x;
                        for (j = 0; j < k; ++j) {
                            const int p = W[j];
                            const int sj = QS[p], ej = QE[p];
                            if (ej <= si || sj >= ei) continue;
                            const int mn = ej - sj < ei - si ? ej - sj : ei - si, mx = ej - sj > ei - si ? ej - sj : ei - si;
                            const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                            const float f1 = (float)ol / (float)mn, f2 = (float)uncov_len / (float)mx;
                            if (f1 - f2 > prm.mask_level) {
                                PAR[i] = PAR[p];
                                SUB[p] = SUB[p] > SC[i] ? SUB[p] : SC[i];
                                if (CNT[i] >= CNT[p]) ++NSUB[p];
                                break;
                            }
                        }
                    } else j = k;
                    if (j == k) { W[k++] = i; PAR[i] = i; NSUB[i] = 0; }
                }
            }
            // ---- mm_select_sub (in place, like the host code: a parent overwritten by the compaction is read as it is) ----
            if (prm.pri_ratio > 0.0f) {
                int k = 0, n_2nd = 0;
                for (int i = 0; i < n; ++i) {
                    const int p = PAR[i];
                    if (p == i) { if (k != i) move(k, i); ++k; }
                    else {
                        const float thr = (float)SC[p] * prm.pri_ratio;
                        if (((float)SC[i] >= thr || SC[i] + prm.min_diff >= SC[p]) && n_2nd < prm.best_n) {
                            if (!(QS[i] == QS[p] && QE[i] == QE[p] && rid_of(i) == rid_of(p) && rs_of(i) == rs_of(p) && re_of(i) == re_of(p))) {
                                if (k != i) move(k, i);
                                ++k; ++n_2nd;
                            }
                        }
                    }
                }
                if (k != n) { n = k; sync(); }
            }
            // ---- mm_squeeze_a: places in the read's anchor list by ascending `as`; every hit is still one chain: a segment each ----
            int n_a = 0, n_seg = 0;
            {
                for (int i = 0; i < n; ++i) {   // insertion sort of the hit indices by as (W)
                    int q = i;
                    while (q > 0 && AS[W[q - 1]] > AS[i]) { W[q] = W[q - 1]; --q; }
                    W[q] = i;
                }
                for (int t = 0; t < n; ++t) {
                    const int i = W[t];
                    AS[i] = n_a; SEG[i] = n_seg;
                    GSRC[n_seg] = SRC[i]; GDST[n_seg] = n_a; GCNT[n_seg] = CNT[i]; k1[n_seg] = 0;   // (flag: set by a join below)
                    ++n_seg;
                    n_a += CNT[i];
                }
            }
            // ---- mm_join_long ----
            if (n >= 2) {
                int na = 0;   // primaries in the order of their anchors: W holds all hits by as already
                for (int t = 0; t < n; ++t) { const int i = W[t]; if (PAR[i] == i || PAR[i] < 0) TMP[na++] = i; }
                int n_drop = 0;
                for (int t = na - 1; t >= 1; --t) {
                    const int i0 = TMP[t - 1], i1 = TMP[t];
                    if (AS[i0] + CNT[i0] != AS[i1]) continue;
                    if (rid_of(i0) != rid_of(i1) || rev_of(i0) != rev_of(i1)) continue;
                    const uint64_t a0x = LX[i0], a0y = LY[i0], a1x = FX[i1], a1y = FY[i1];
                    if (a1x <= a0x || (int32_t)a1y <= (int32_t)a0y) continue;
                    int max_gap, min_gap;
                    max_gap = min_gap = (int32_t)a1y - (int32_t)a0y;
                    max_gap = max_gap > (int64_t)(a1x - a0x) ? max_gap : (int)(a1x - a0x);
                    min_gap = min_gap < (int64_t)(a1x - a0x) ? min_gap : (int)(a1x - a0x);
                    if (max_gap > prm.max_join_long || min_gap > prm.max_join_short) continue;
                    const float fsc = (float)prm.min_join_flank_sc / (float)prm.max_join_long;
                    const float fsc2 = fsc * (float)max_gap;
                    const int sc_thres = (int)((double)fsc2 + .499);
                    if (SC[i0] < sc_thres || SC[i1] < sc_thres) continue;
                    const float ffl = (float)max_gap * prm.min_join_flank_ratio;
                    const int min_flank_len = (int)ffl;
                    if (re_of(i0) - rs_of(i0) < min_flank_len || QE[i0] - QS[i0] < min_flank_len) continue;
                    if (re_of(i1) - rs_of(i1) < min_flank_len || QE[i1] - QS[i1] < min_flank_len) continue;
                    k1[SEG[i1]] = 1;   // a[r1.as].y |= SEED_LONG_JOIN (applied by the squeeze kernel)
                    CNT[i0] += CNT[i1]; SC[i0] += SC[i1];
                    {
                        const int span = (int)(a1y >> 32 & 0xff);
                        const int tl = (int32_t)a1x - (int32_t)a0x, ql = (int32_t)a1y - (int32_t)a0y;
                        BL[i0] += BL[i1] - span + (tl > ql ? tl : ql);
                        ML[i0] += ML[i1] - span + (tl > span && ql > span ? span : tl < ql ? tl : ql);
                    }
                    LX[i0] = LX[i1]; LY[i0] = LY[i1];
                    set_q(i0);
                    CNT[i1] = 0;
                    PAR[i1] = ord[i0];
                    ++n_drop;
                }
                if (n_drop > 0) {
                    // (ids equal indices here: a sync or the initial numbering left them so)
                    for (int i = 0; i < n; ++i)
                        if (PAR[i] >= 0 && ord[i] != PAR[i])
                            if (PAR[PAR[i]] >= 0 && PAR[PAR[i]] != PAR[i]) PAR[i] = PAR[PAR[i]];
                    // mm_filter_regs: only the count test applies before the base-level alignment
                    int k = 0;
                    for (int i = 0; i < n; ++i)
                        if (!(CNT[i] < prm.min_cnt)) { if (k < i) move(k, i); ++k; }
                    n = k;
                    sync();
                }
            }
            s_n = n; s_na = n_a; s_nseg = prm.with_cigar ? n_seg : 0; s_flags = flags;
            s_rpos = n ? atomicAdd(&counters[0], (unsigned long long)n) : 0ULL;
            s_spos = (prm.with_cigar && n_seg) ? atomicAdd(&counters[1], (unsigned long long)n_seg) : 0ULL;
            out_reads[read] = HitRead{(int64_t)s_rpos, n, n_a, flags, 0};
        }
        __syncthreads();
        const int n = s_n;
        HitRec *dst = out_regs + s_rpos;
        for (int i = lane; i < n; i += 64)
            dst[i] = HitRec{FX[i], FY[i], LX[i], LY[i], SC[i], SC0[i], CNT[i], AS[i], PAR[i], SUB[i], NSUB[i], ML[i], BL[i], HSH[i]};
        const int n_seg = s_nseg;
        SqueezeSeg *sg = out_segs + s_spos;
        const int64_t pool0 = b_pos[read];
        for (int t = lane; t < n_seg; t += 64) sg[t] = SqueezeSeg{pool0 + GSRC[t], GDST[t], GCNT[t], read, (int32_t)k1[t]};
        __syncthreads();
    }
}

}  // namespace mpn
