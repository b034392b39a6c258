// Finishing of stitched alignments on the GPU: minimap2's mm_fix_cigar (left-alignment of gaps, merging of adjacent
// insertion/deletion runs, removal of a leading gap) and mm_update_extra (matching / aligned / ambiguous base counts and the
// maximal clipped running score "dp_max") over the CIGAR the host stitched from its DP windows.  The reads and the 2-bit
// targets are already in HBM, so the host no longer builds per-read code arrays or unpacks target slices: the CIGAR goes
// up, the fixed CIGAR and eight integers per alignment come back.
#pragma once
#include "ext_kernels.h"

namespace mpn {

struct FinJob { int64_t cig_off, code_off; int32_t n_cigar, read, rid, rev, qs1, rs1, qspan, tspan; };  // code_off: scratch of the jobs too large for LDS
struct FinOut { int32_t n_cigar, qshift, tshift, blen, mlen, n_ambi, dp_max, pad; };
struct FinParams { int8_t mat[25]; int8_t q, e; };

// One wave per alignment.  The 64 lanes stage its CIGAR and the codes of both sequences in LDS (coalesced loads, the 2-bit
// target unpacked once) and share the passes; a lane owns a contiguous range of ops.
//  * Gap left-alignment: the position of an op does not depend on earlier shifts (a shift moves length from the match run
//    before a gap to the one after it), and the shift of gap k is min(r_k, length of the run before it INCLUDING what gap
//    k-2 moved into it), r_k = how far the bases before the gap repeat the gap's tail.  Every lane computes r_k up to the
//    run's original length; only gaps that reach it ("saturated", a few per alignment) depend on their predecessor and are
//    finished by lane 0 in order.  The new run lengths then follow per op.
//  * Merging of adjacent insertion/deletion runs, squeezing out empty ops, dropping a leading gap: lane 0, and only when a
//    ballot says there is something to do (sequential scans over LDS).
//  * Statistics: counts add up; the clipped running score s -> max(0, s + x) over a range of columns is the map
//    s -> max(s + D, C) with peak max(s + PM, CM), and these maps compose exactly (max-plus), so every lane folds its range
//    and lane 0 composes the 64 summaries in order.
// Statement for statement these are minimap2's mm_fix_cigar / mm_update_extra; only the order of evaluation differs.
constexpr uint32_t FIN_SAT = 1u << 31;

template <bool IN_LDS>
__global__ __launch_bounds__(64) void aln_finish_wave_kernel(const FinJob *__restrict__ jobs, const int32_t *__restrict__ list, int n_list,
                                                             uint32_t *__restrict__ CIG, uint32_t *__restrict__ AUX, uint8_t *__restrict__ CODES,
                                                             const uint8_t *__restrict__ reads,
                                                             const int64_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
                                                             RefView rv, FinParams prm, FinOut *__restrict__ out,
                                                             const int64_t *__restrict__ code_offs = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fin_lds[];
    __shared__ int32_t sum_l[64][8];
    const int lane = threadIdx.x;
    // (global scratch written by one lane and read by another needs the fence at agent scope)
    auto sync = []() { if constexpr (IN_LDS) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); __builtin_amdgcn_wave_barrier(); };
    auto excl_scan = [&](int v) { int x = v; for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(x, d); if (lane >= d) x += o; } return x - v; };
    for (int li = blockIdx.x; li < n_list; li += gridDim.x) {
        const int jid = list[li];
        const FinJob jb = jobs[jid];
        uint32_t *cg = CIG + jb.cig_off;
        int n = jb.n_cigar;
        // IN_LDS: everything staged in LDS; else (alignments too large for it) the CIGAR is fixed in place and the code arrays
        // and the shift table live in global scratch -- same passes, the 64 lanes still share them
        uint32_t *c_l, *aux;
        uint8_t *q_l;
        if constexpr (IN_LDS) { c_l = reinterpret_cast<uint32_t *>(fin_lds); aux = c_l + n; q_l = reinterpret_cast<uint8_t *>(aux + n); }
        else { c_l = cg; aux = AUX + jb.cig_off; q_l = CODES + (code_offs ? code_offs[li] : jb.code_off); }  // (code_offs: by list position)
        uint8_t *t_l = q_l + ((jb.qspan + 3) & ~3);
        const int64_t roff = read_off[jb.read];
        const int32_t rlen = read_len[jb.read];
        const int64_t g0 = rv.seq_off[jb.rid] + jb.rs1;
        // Staging is a chain of global-load latencies for a single wave, so the loads are batched: eight per lane in flight
        // for the CIGAR and the query bytes, and the target comes as whole 2-bit words (16 bases per load).
        if constexpr (IN_LDS) {
            for (int i0 = 0; i0 < n; i0 += 512) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 + lane; v[u] = i < n ? cg[i] : 0u; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + u * 64 + lane; if (i < n) c_l[i] = v[u]; }
            }
        }
        for (int x0 = 0; x0 < jb.qspan; x0 += 4 * 64 * 4) {   // four loads in flight per lane, four bases (one word) per load
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + 4 * (u * 64 + lane);
                v[u] = 0x4e4e4e4eu;   // "NNNN"
                if (x < jb.qspan) {
                    // the four bases x .. x+3 on the hit's strand: ascending addresses, or (reverse strand) descending from
                    // rlen-1-(qs1+x); the reads buffer is padded, so the word may reach a few bytes past the read
                    const int64_t a = roff + (jb.rev ? rlen - 1 - (jb.qs1 + x) - 3 : jb.qs1 + x);
                    if (a >= 0) __builtin_memcpy(&v[u], reads + a, 4);
                    else for (int j = 0; j < 4; ++j) if (a + j >= 0) reinterpret_cast<uint8_t *>(&v[u])[j] = reads[a + j];
                    if (jb.rev) v[u] = __builtin_bswap32(v[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x = x0 + 4 * (u * 64 + lane);
                if (x < jb.qspan) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = nt4_code((uint8_t)(v[u] >> (8 * j)));
                        packed |= (uint32_t)(jb.rev ? (c < 4 ? 3 - c : 4) : c) << (8 * j);
                    }
                    *reinterpret_cast<uint32_t *>(q_l + x) = packed;   // (q_l and x are 4-aligned; the array is padded to a multiple of 4)
                }
            }
        }
        {
            // lane i of a round unpacks target bases [16 i, 16 i + 16) of the interval from the two words that hold them
            const int n16 = (jb.tspan + 15) >> 4;
            const int sh = 2 * (int)(g0 & 15);
            const int64_t w0 = g0 >> 4, w_last = (g0 + (jb.tspan > 0 ? jb.tspan - 1 : 0)) >> 4;
            for (int i0 = 0; i0 < n16; i0 += 128) {
                unsigned long long w[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * 64 + lane;
                    w[u] = 0;
                    if (i < n16) {  // (the second word only where the interval reaches into it: it may lie past the array)
                        const unsigned long long lo = rv.seq2[w0 + i], hi = sh && w0 + i + 1 <= w_last ? rv.seq2[w0 + i + 1] : 0u;
                        w[u] = sh ? (lo >> sh | hi << (32 - sh)) & 0xffffffffULL : lo;
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = i0 + u * 64 + lane;
                    if (i < n16) {
                        const int x = i << 4, m = min(16, jb.tspan - x);
                        for (int b = 0; b < m; ++b) t_l[x + b] = (uint8_t)(w[u] >> (2 * b) & 3);
                    }
                }
            }
            if (rv.n_runs > 0) {   // ambiguous-base runs that overlap the interval (rare; uniform search)
                sync();
                int lo = 0, hi = rv.n_runs;  // first run that ends after g0
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (rv.nrun_e[mid] <= g0) lo = mid + 1; else hi = mid; }
                for (; lo < rv.n_runs && rv.nrun_s[lo] < g0 + jb.tspan; ++lo) {
                    const int64_t a = rv.nrun_s[lo] > g0 ? rv.nrun_s[lo] - g0 : 0, b = (rv.nrun_e[lo] < g0 + jb.tspan ? rv.nrun_e[lo] : g0 + jb.tspan) - g0;
                    for (int64_t x = a + lane; x < b; x += 64) t_l[x] = 4;
                }
            }
        }
        sync();
        int qshift = 0, tshift = 0;
        if (n > 1) {
            const int per = (n + 63) / 64, k_lo = min(n, lane * per), k_hi = min(n, k_lo + per);
            int qa = 0, ta = 0;
            for (int k = k_lo; k < k_hi; ++k) { const uint32_t op = c_l[k] & 0xf, len = c_l[k] >> 4; qa += op != 2 ? len : 0; ta += op != 1 ? len : 0; }
            int qoff = excl_scan(qa), toff = excl_scan(ta);
            bool shrink = false, any_sat = false;
            for (int k = k_lo; k < k_hi; ++k) {
                const uint32_t op = c_l[k] & 0xf, len = c_l[k] >> 4;
                uint32_t v = 0;
                if (len == 0 && op != 0) shrink = true;   // (a match run is judged below, with what its left gap moved into it)
                if (op == 0) { toff += len; qoff += len; }
                else if (op == 1 || op == 2) {
                    if (k > 0 && k < n - 1 && (c_l[k - 1] & 0xf) == 0 && (c_l[k + 1] & 0xf) == 0) {
                        int l;
                        const int prev_len = c_l[k - 1] >> 4;
                        if (op == 1) { for (l = 0; l < prev_len; ++l) if (q_l[qoff - 1 - l] != q_l[qoff + (int)len - 1 - l]) break; }
                        else { for (l = 0; l < prev_len; ++l) if (t_l[toff - 1 - l] != t_l[toff + (int)len - 1 - l]) break; }
                        v = (uint32_t)l;
                        if (l == prev_len) { v |= FIN_SAT; any_sat = true; }
                    }
                    if (op == 1) qoff += len; else toff += len;
                }
                aux[k] = v;
            }
            sync();
            if (__ballot(any_sat) != 0) {
                if (lane == 0) {  // saturated gaps in order: the run before gap k has grown by what gap k-2 moved into it
                    int qo = 0, to = 0;
                    for (int k = 0; k < n; ++k) {
                        const uint32_t op = c_l[k] & 0xf, len = c_l[k] >> 4, v = aux[k];
                        if (v & FIN_SAT) {
                            const int grown = (int)(c_l[k - 1] >> 4) + (k >= 2 ? (int)(aux[k - 2] & ~FIN_SAT) : 0);
                            int l = (int)(v & ~FIN_SAT);
                            if (op == 1) { for (; l < grown; ++l) if (q_l[qo - 1 - l] != q_l[qo + (int)len - 1 - l]) break; }
                            else { for (; l < grown; ++l) if (t_l[to - 1 - l] != t_l[to + (int)len - 1 - l]) break; }
                            aux[k] = (uint32_t)l | (l == grown ? FIN_SAT : 0u);   // still "the whole run": it becomes empty
                        }
                        qo += op != 2 ? len : 0; to += op != 1 ? len : 0;
                    }
                }
                sync();
            }
            // new lengths of the match runs; empty ops and adjacent insertion/deletion pairs
            bool pair = false;
            for (int k = k_lo; k < k_hi; ++k) {
                const uint32_t op = c_l[k] & 0xf;
                if (op == 0) {
                    const uint32_t from_left = k > 0 ? aux[k - 1] & ~FIN_SAT : 0u, to_right = k < n - 1 ? aux[k + 1] & ~FIN_SAT : 0u;
                    const uint32_t seen = (c_l[k] >> 4) + from_left, len = seen - to_right;   // seen: its length when the scan reaches it
                    c_l[k] = len << 4;
                    if (seen == 0 || len == 0) shrink = true;
                } else if (k < n - 2 && op + (c_l[k + 1] & 0xf) == 3) pair = true;   // (types do not change; a neighbour's length may)
            }
            sync();
            if (__ballot(pair) != 0) {
                int sh2 = 0;
                if (lane == 0) {
                    for (int k = 0; k < n - 2; ++k) {
                        if ((c_l[k] & 0xf) > 0 && (c_l[k] & 0xf) + (c_l[k + 1] & 0xf) == 3) {
                            uint32_t l, sacc[3] = {0, 0, 0};
                            for (l = k; l < (uint32_t)n; ++l) {
                                const uint32_t op = c_l[l] & 0xf;
                                if (op == 1 || op == 2 || c_l[l] >> 4 == 0) sacc[op] += c_l[l] >> 4;
                                else break;
                            }
                            if (sacc[1] > 0 && sacc[2] > 0 && l - k > 2) {
                                c_l[k] = sacc[1] << 4 | 1;
                                c_l[k + 1] = sacc[2] << 4 | 2;
                                for (k += 2; k < (int)l; ++k) c_l[k] &= 0xf;
                                sh2 = 1;
                            }
                            k = l;
                        }
                    }
                }
                if (__builtin_amdgcn_readfirstlane(sh2)) shrink = true;
                sync();
            }
            if (__ballot(shrink) != 0) {
                int n2 = n;
                if (lane == 0) {
                    int32_t l = 0;
                    for (int k = 0; k < n; ++k) if (c_l[k] >> 4 != 0) c_l[l++] = c_l[k];
                    n2 = l;
                    l = 0;
                    for (int k = 0; k < n2; ++k)
                        if (k == n2 - 1 || (c_l[k] & 0xf) != (c_l[k + 1] & 0xf)) c_l[l++] = c_l[k];
                        else c_l[k + 1] += c_l[k] >> 4 << 4;
                    n2 = l;
                }
                n = __builtin_amdgcn_readfirstlane(n2);
                sync();
            }
            if (n > 0 && ((c_l[0] & 0xf) == 1 || (c_l[0] & 0xf) == 2)) {   // (uniform: every lane reads the same word)
                const int32_t l = c_l[0] >> 4;
                if ((c_l[0] & 0xf) == 1) qshift = l; else tshift = l;
                sync();
                --n;
                for (int k0 = 0; k0 < n; k0 += 64) {
                    const int k = k0 + lane;
                    const uint32_t v = k < n ? c_l[k + 1] : 0u;
                    sync();
                    if (k < n) c_l[k] = v;
                }
                sync();
            }
        }
        // ---- statistics ----
        {
            const int per = (n + 63) / 64, k_lo = min(n, lane * per), k_hi = min(n, k_lo + per);
            int qa = 0, ta = 0;
            for (int k = k_lo; k < k_hi; ++k) { const uint32_t op = c_l[k] & 0xf, len = c_l[k] >> 4; qa += op != 2 ? len : 0; ta += op != 1 ? len : 0; }
            int qoff = qshift + excl_scan(qa), toff = tshift + excl_scan(ta);
            int32_t blen = 0, mlen = 0, n_ambi_all = 0;
            const int sc_mch = prm.mat[0], sc_mis = prm.mat[1], sc_amb = prm.mat[4];
            // the lane's columns as one map: D total, P running prefix, mn its minimum, PM its maximum, CM the largest P - min so far
            int32_t P = 0, mn = 0, PM = NEG_INF, CM = 0;
            bool first = true;
            auto step = [&](int x) {
                P += x;
                if (first) { mn = P; first = false; } else mn = mn < P ? mn : P;
                PM = PM > P ? PM : P;
                const int c = P - mn;
                CM = CM > c ? CM : c;
            };
            for (int k = k_lo; k < k_hi; ++k) {
                const uint32_t op = c_l[k] & 0xf, len = c_l[k] >> 4;
                if (op == 0) {
                    int n_ambi = 0, n_diff = 0;
                    for (uint32_t l = 0; l < len; ++l) {
                        const int cq = q_l[qoff + (int)l], ct = t_l[toff + (int)l];
                        if (ct > 3 || cq > 3) ++n_ambi;
                        else if (ct != cq) ++n_diff;
                        step(ct > 3 || cq > 3 ? sc_amb : ct == cq ? sc_mch : sc_mis);   // (not a table in the kernel arguments: that is a global load per column)
                    }
                    blen += len - n_ambi; mlen += len - (n_ambi + n_diff); n_ambi_all += n_ambi;
                    toff += len; qoff += len;
                } else if (op == 1) {
                    int n_ambi = 0;
                    for (uint32_t l = 0; l < len; ++l) if (q_l[qoff + (int)l] > 3) ++n_ambi;
                    blen += len - n_ambi; n_ambi_all += n_ambi;
                    step(-(prm.q + prm.e * (int)len));
                    qoff += len;
                } else if (op == 2) {
                    int n_ambi = 0;
                    for (uint32_t l = 0; l < len; ++l) if (t_l[toff + (int)l] > 3) ++n_ambi;
                    blen += len - n_ambi; n_ambi_all += n_ambi;
                    step(-(prm.q + prm.e * (int)len));
                    toff += len;
                }
            }
            sum_l[lane][0] = first ? 1 : 0; sum_l[lane][1] = P; sum_l[lane][2] = first ? 0 : P - mn; sum_l[lane][3] = PM; sum_l[lane][4] = CM;
            for (int d = 32; d; d >>= 1) { blen += __shfl_xor(blen, d); mlen += __shfl_xor(mlen, d); n_ambi_all += __shfl_xor(n_ambi_all, d); }
            sync();
            if (lane == 0) {
                int32_t s = 0, mx = 0;
                for (int l = 0; l < 64; ++l) {
                    if (sum_l[l][0]) continue;               // empty range
                    const int32_t D = sum_l[l][1], C = sum_l[l][2], PMl = sum_l[l][3], CMl = sum_l[l][4];
                    const int32_t pk = s + PMl > CMl ? s + PMl : CMl;
                    mx = mx > pk ? mx : pk;
                    s = s + D > C ? s + D : C;
                }
                FinOut o;
                o.n_cigar = n; o.qshift = qshift; o.tshift = tshift; o.blen = blen; o.mlen = mlen; o.n_ambi = n_ambi_all; o.dp_max = mx; o.pad = 0;
                out[jid] = o;
            }
        }
        sync();
        if constexpr (IN_LDS) for (int i = lane; i < n; i += 64) cg[i] = c_l[i];
        sync();
    }
}

}  // namespace mpn
