// Device kernels of the mapper's seed + chain stages (gfx950).  Included by mapper.hip only.
//
// Data layout in HBM (all struct-of-arrays, CSR over the reads of a batch):
//   reads      ASCII bytes, concatenated; seq_off[i], seq_len[i]
//   mz         read minimizers as 16-byte pairs (x = hash<<8|span, y = read<<32|last_pos<<1|strand); mz_off[n+1]
//   index      keys[n_keys] (sorted 2k-bit hashes), key_off[n_keys+1], pos[] (rid<<32|last_pos<<1|strand)
//   anchors    16-byte pairs (x = strand<<63|rid<<32|ref_pos, y = flags|span<<32|query_pos); anchor_off[n+1]
//   chaining   f,p,t,v int32 per anchor; chain ends u (8 B each)
// Every stage is integer / byte work bound by HBM latency and bandwidth, not by arithmetic: no MFMA.
#pragma once
#include "mpn_common.h"
#include "map_types.h"

#include <type_traits>

namespace mpn {

__device__ __forceinline__ uint64_t hash64m(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// ---------------------------------------------------------------------------------------------------------
// (w,k)-minimizers.  A sequence is cut into CHUNKS of C positions; a chunk OWNS the minimizers whose position (the last
// base of the k-mer) lies in it, and the output of a sequence is the concatenation of its chunks' outputs (minimap2
// reports minimizers in position order).  Two kernels produce a chunk:
//
//  * sketch_fast_kernel -- regular chunks: no ambiguous base within reach, not at the ends of the sequence, odd k (no
//    k-mer equals its reverse complement).  There minimap2's window automaton reports exactly the positions p whose
//    hash is the minimum (ties included) of at least one window of w consecutive k-mers containing p: with L / R the
//    numbers of consecutive neighbours to the left / right whose hash is not smaller, that is L + R >= w - 1.  One
//    wave per chunk, a lane per position: bases are loaded coalesced, packed 2 bits per base in LDS, the k-mer of a
//    position is two LDS words and a shift, the hash runs in 32-bit registers when 2k <= 32, and the window test is
//    2(w-1) LDS reads of 4 bytes.  No divergence, ~100 lane-instructions per base (the automaton below: ~5000).
//  * sketch_chunk_kernel -- every other chunk (listed by the fast kernel): exact replay of minimap2's sequential
//    automaton by one lane per chunk.  The ring of the last w window elements lives in LDS as [slot][lane].  The state
//    at p0 depends only on the last w + k - 1 window elements and on min(l, w + k), so the lane reconstructs it by
//    replaying a short warm-up [p0 - D, p0) from the reset state: if the warm-up holds an ambiguous base the automaton
//    was reset there anyway; otherwise l saturated inside it (for even k, where symmetric k-mers do not advance l, D
//    is doubled until that holds).  It then runs w window steps past the chunk, because a minimizer is reported when
//    it is replaced or leaves the window, and keeps the reports whose position the chunk owns.
// Two passes: the first counts (chunk_cnt) and remembers what it found -- the fast kernel the emit mask of every 64 positions,
// the automaton kernel its reports in a staging area; after the prefix sums the second writes at mz_off[seq] + chunk_rel[chunk]:
// sketch_fill_kernel hashes the emitted positions of the regular chunks again (a fifth of the k-mers), sketch_stage_copy_kernel
// moves the staged reports (the automaton runs a second time, FILL = true, only for chunks the staging area had no room for).
constexpr int SKETCH_FAST_MAX_W = 32;

template <bool HASH64>
__device__ __forceinline__ uint64_t sketch_hash(uint64_t key, uint64_t mask) {
    if (HASH64) return hash64m(key, mask);
    // the same mixing on 32-bit registers: every step of hash64m is reduced modulo 2^(2k) <= 2^32 before a right shift
    uint32_t h = (uint32_t)key;
    const uint32_t m = (uint32_t)mask;
    h = (~h + (h << 21)) & m;
    h = h ^ h >> 24;
    h = ((h + (h << 3)) + (h << 8)) & m;
    h = h ^ h >> 14;
    h = ((h + (h << 2)) + (h << 4)) & m;
    h = h ^ h >> 28;
    h = (h + (h << 31)) & m;
    return h;
}

// Is the chunk [p0, p1) of a sequence of length len regular, as far as its place in the sequence goes?  [A0, A1) is the range
// of bases that must also be free of ambiguous codes: an ambiguous base resets the automaton's run length l, and while
// l < w + k (the w + k - 1 steps after it) nothing is reported, not even a minimum whose turn ends there; a minimizer at p
// is reported at step p + w at the latest.  So the chunk's reports are the regular ones iff no such base lies in
// [p0 - 2w - k, p1 + w], the sequence starts before that range and ends after it (the end of a sequence reports the
// last minimum unconditionally).
__device__ __forceinline__ bool sketch_chunk_in_range(int p0, int p1, int len, int w, int k, int *A0, int *A1) {
    *A0 = p0 - 2 * w - k;
    *A1 = p1 + w + 1;
    return (k & 1) && w <= SKETCH_FAST_MAX_W && *A0 >= 0 && *A1 <= len;
}

// chunk -> sequence table: a wave per sequence writes its index over the sequence's chunks
__global__ __launch_bounds__(256) void sketch_chunk_read_kernel(const int64_t *__restrict__ chunk_off, int n, int32_t *__restrict__ chunk_read) {
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4)
        for (int64_t c = chunk_off[i] + lane; c < chunk_off[i + 1]; c += 64) chunk_read[c] = i;
}

// (W_CT: the window length at compile time, 0 = the run-time argument; with it the 2 (w - 1) LDS reads of the window test are
// unrolled and issued together instead of one round trip each)
template <bool HASH64, int W_CT = 0>
__global__ __launch_bounds__(256) void sketch_fast_kernel(const uint8_t *__restrict__ seqs, const int64_t *__restrict__ seq_off,
                                                          const int32_t *__restrict__ seq_len, int n,
                                                          const int64_t *__restrict__ chunk_off, const int32_t *__restrict__ chunk_read, int64_t n_chunks,
                                                          int C, int w, int k, int32_t *__restrict__ chunk_cnt,
                                                          int64_t *__restrict__ slow_list, unsigned long long *__restrict__ n_slow,
                                                          unsigned long long *__restrict__ emask, uint8_t *__restrict__ cfast) {
    // per wave: packed bases (2 bits each, first base in the top bits of a word), hashes of the k-mer end positions
    constexpr int MAX_EXT = 256 + 3 * SKETCH_FAST_MAX_W + 28 + 1 + 16;
    typedef typename std::conditional<HASH64, uint64_t, uint32_t>::type hash_t;
    __shared__ uint32_t s_words[4][MAX_EXT / 16 + 2];
    __shared__ hash_t s_hash[4][256 + 2 * (SKETCH_FAST_MAX_W - 1) + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *words = s_words[wv];
    hash_t *hs = s_hash[wv];
    const uint64_t mask = (1ULL << 2 * k) - 1;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wv; c < n_chunks; c += (int64_t)gridDim.x * 4) {
        const int i = chunk_read[c];   // (the sequence of the chunk: a table, not a search over chunk_off per chunk)
        const uint8_t *sq = seqs + seq_off[i];
        const int len = seq_len[i];
        const int p0 = (int)(c - chunk_off[i]) * C, p1 = min(len, p0 + C);
        int a0, a1;  // bases loaded: [a0, a1)
        const bool in_range = sketch_chunk_in_range(p0, p1, len, w, k, &a0, &a1);
        bool regular = in_range;
        const int ext = a1 - a0;
        const int qbase = w + 2;  // loaded index of the first base of the first k-mer needed: (p0 - (w-1) - (k-1)) - a0
        if (regular) {
            // bases -> 2-bit words: 16 consecutive lanes OR their codes together
            for (int t0 = 0; t0 < ext; t0 += 64) {
                const int t = t0 + lane;
                const int code = t < ext ? nt4_code(sq[a0 + t]) : 0;
                if (__ballot(code > 3)) regular = false;
                uint32_t v = (uint32_t)(code & 3) << (2 * (15 - (lane & 15)));
                v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4); v |= __shfl_xor(v, 8);
                if ((lane & 15) == 0) words[(t0 >> 4) + (lane >> 4)] = v;
            }
        }
        if (!regular) {
            // chunks that are irregular by their place in the sequence are on the host-built part of the list already;
            // one that holds an ambiguous base is appended here (rare, so the atomic on one counter does not matter)
            if (in_range && lane == 0) slow_list[atomicAdd(n_slow, 1ULL)] = c;
            if (lane == 0) cfast[c] = 0;
            continue;
        }
        if (lane == 0) cfast[c] = 1;
        // hash and strand of the k-mers ending at p0 - (w-1) + q, q in [0, n_q)
        const int n_q = (p1 - p0) + 2 * (w - 1);
        for (int q0 = 0; q0 < n_q; q0 += 64) {
            const int q = q0 + lane;
            uint64_t h = 0;
            bool z = false;
            if (q < n_q) {
                // bases qbase + q .. qbase + q + k - 1 of the loaded range; the k-mer's first base is the most significant
                const int wi = (qbase + q) >> 4, sh = 2 * ((qbase + q) & 15);
                const uint64_t hi64 = (uint64_t)words[wi] << 32 | words[wi + 1];
                // 64 bits hold bases 16 wi .. 16 wi + 31; k <= 28 and (q & 15) + k <= 43 may exceed them: take a third word
                uint64_t f;
                if (sh + 2 * k <= 64) f = hi64 << sh >> (64 - 2 * k);
                else f = ((hi64 << sh) | ((uint64_t)words[wi + 2] >> (32 - sh))) >> (64 - 2 * k);  // (sh >= 10 here)
                // reverse complement: complement, then reverse the 2-bit groups
                uint64_t r = ~f & mask;
                r = __builtin_bitreverse64(r);
                r = ((r & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((r & 0x5555555555555555ULL) << 1);
                r >>= 64 - 2 * k;
                z = !(f < r);
                h = sketch_hash<HASH64>(z ? r : f, mask);
            }
            if (q < n_q) hs[q] = (hash_t)h;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // window test of the owned positions p = p0 + t: q index of p is t + (w - 1)
        int run = 0;
        for (int t0 = 0; t0 < p1 - p0; t0 += 64) {
            const int t = t0 + lane;
            bool emit = false;
            if (t < p1 - p0) {
                const int q = t + (w - 1);
                const hash_t hp = hs[q];
                int L = 0, R = 0;
                bool go = true;
                if constexpr (W_CT > 0) {
                    hash_t hl[W_CT - 1], hr[W_CT - 1];
#pragma unroll
                    for (int d = 1; d < W_CT; ++d) { hl[d - 1] = hs[q - d]; hr[d - 1] = hs[q + d]; }
#pragma unroll
                    for (int d = 1; d < W_CT; ++d) { go = go && hl[d - 1] >= hp; L += go; }
                    go = true;
#pragma unroll
                    for (int d = 1; d < W_CT; ++d) { go = go && hr[d - 1] >= hp; R += go; }
                } else {
                    for (int d = 1; d < w; ++d) { go = go && hs[q - d] >= hp; L += go; }
                    go = true;
                    for (int d = 1; d < w; ++d) { go = go && hs[q + d] >= hp; R += go; }
                }
                emit = L + R >= w - 1;
            }
            const unsigned long long em = __ballot(emit);
            if (lane == 0) emask[c * 4 + (t0 >> 6)] = em;   // what sketch_fill_kernel emits
            run += __popcll(em);
        }
        if (lane == 0) chunk_cnt[c] = run;
        __builtin_amdgcn_wave_barrier();
    }
}

// The second pass of the regular chunks without the window test: the count pass left the emit mask of every 64 positions
// (emask) and whether it handled the chunk (cfast).  The wave packs the chunk's bases again, compacts the emitted positions
// (~47 of 256) into a list and hashes only those -- one round of 64 lanes instead of five over all k-mers -- and stores the
// records in order, 16 bytes per lane.
template <bool HASH64>
__global__ __launch_bounds__(256) void sketch_fill_kernel(const uint8_t *__restrict__ seqs, const int64_t *__restrict__ seq_off,
                                                          const int32_t *__restrict__ seq_len, int n,
                                                          const int64_t *__restrict__ chunk_off, const int32_t *__restrict__ chunk_read, int64_t n_chunks,
                                                          int C, int w, int k,
                                                          const int64_t *__restrict__ mz_off, const int32_t *__restrict__ chunk_rel,
                                                          u128 *__restrict__ mz, uint32_t rid_base,
                                                          const unsigned long long *__restrict__ emask, const uint8_t *__restrict__ cfast) {
    constexpr int MAX_EXT = 256 + 3 * SKETCH_FAST_MAX_W + 28 + 1 + 16;
    __shared__ uint32_t s_words[4][MAX_EXT / 16 + 2];
    __shared__ uint16_t s_list[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *words = s_words[wv];
    uint16_t *list = s_list[wv];
    const uint64_t mask = (1ULL << 2 * k) - 1;
    const unsigned long long lane_lt = (1ULL << lane) - 1;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wv; c < n_chunks; c += (int64_t)gridDim.x * 4) {
        if (!cfast[c]) continue;   // the automaton kernel's chunk
        const int i = chunk_read[c];   // (the sequence of the chunk: a table, not a search over chunk_off per chunk)
        const uint8_t *sq = seqs + seq_off[i];
        const int len = seq_len[i];
        const int p0 = (int)(c - chunk_off[i]) * C, p1 = min(len, p0 + C);
        int a0, a1;
        (void)sketch_chunk_in_range(p0, p1, len, w, k, &a0, &a1);
        const int ext = a1 - a0;
        const int qbase = w + 2;
        for (int t0 = 0; t0 < ext; t0 += 64) {
            const int t = t0 + lane;
            const int code = t < ext ? nt4_code(sq[a0 + t]) : 0;
            uint32_t v = (uint32_t)(code & 3) << (2 * (15 - (lane & 15)));
            v |= __shfl_xor(v, 1); v |= __shfl_xor(v, 2); v |= __shfl_xor(v, 4); v |= __shfl_xor(v, 8);
            if ((lane & 15) == 0) words[(t0 >> 4) + (lane >> 4)] = v;
        }
        int run = 0;
        for (int t0 = 0; t0 < p1 - p0; t0 += 64) {
            const unsigned long long em = emask[c * 4 + (t0 >> 6)];
            if (em >> lane & 1) list[run + __popcll(em & lane_lt)] = (uint16_t)(t0 + lane);
            run += __popcll(em);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        u128 *out = mz + mz_off[i] + chunk_rel[c];
        for (int r0 = 0; r0 < run; r0 += 64) {
            if (r0 + lane < run) {
                const int t = list[r0 + lane];
                const int q = t + (w - 1);   // index among the k-mers ending at p0 - (w-1) + q
                const int wi = (qbase + q) >> 4, sh = 2 * ((qbase + q) & 15);
                const uint64_t hi64 = (uint64_t)words[wi] << 32 | words[wi + 1];
                uint64_t f;
                if (sh + 2 * k <= 64) f = hi64 << sh >> (64 - 2 * k);
                else f = ((hi64 << sh) | ((uint64_t)words[wi + 2] >> (32 - sh))) >> (64 - 2 * k);
                uint64_t r = ~f & mask;
                r = __builtin_bitreverse64(r);
                r = ((r & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((r & 0x5555555555555555ULL) << 1);
                r >>= 64 - 2 * k;
                const bool z = !(f < r);
                const uint64_t h = sketch_hash<HASH64>(z ? r : f, mask);
                u128 rec;
                rec.x = h << 8 | (uint64_t)k;
                rec.y = (uint64_t)(rid_base + (uint32_t)i) << 32 | (uint32_t)(p0 + t) << 1 | (uint32_t)z;
                out[r0 + lane] = rec;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

constexpr int SKETCH_STAGE_CAP = 256;   // reports per staged chunk (= C: one per owned position at most)

template <bool FILL>
__global__ __launch_bounds__(64) void sketch_chunk_kernel(const uint8_t *__restrict__ seqs, const int64_t *__restrict__ seq_off,
                                                          const int32_t *__restrict__ seq_len, int n,
                                                          const int64_t *__restrict__ chunk_off, const int32_t *__restrict__ chunk_read,
                                                          const int64_t *__restrict__ slow_list,
                                                          const unsigned long long *__restrict__ n_slow, int C, int w, int k,
                                                          const int64_t *__restrict__ mz_off, const int32_t *__restrict__ chunk_rel,
                                                          int32_t *__restrict__ chunk_cnt, u128 *__restrict__ mz, uint32_t rid_base,
                                                          u128 *__restrict__ stage, int64_t stage_chunks, int64_t li_begin) {
    // stage (count pass): the reports of the first stage_chunks listed chunks are kept, SKETCH_STAGE_CAP slots per chunk, and a
    // copy kernel puts them in place -- the automaton runs once for them; li_begin: the fill pass of the chunks beyond that
    extern __shared__ __attribute__((aligned(16))) uint64_t ring[];  // bx[w][64] then by[w][64]
    const int tid = threadIdx.x;
    uint64_t *bx = ring, *by = ring + (size_t)w * blockDim.x;
    const int64_t n_list = (int64_t)*n_slow;
    for (int64_t li = li_begin + (int64_t)blockIdx.x * blockDim.x + tid; li < n_list; li += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = slow_list[li];
    const int i = chunk_read[c];
    const uint8_t *s = seqs + seq_off[i];
    const int len = seq_len[i];
    const int p0 = (int)(c - chunk_off[i]) * C, p1 = min(len, p0 + C);
    const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
    const int lsat = w + k;
    const uint32_t rid = rid_base + (uint32_t)i;
    int64_t cnt = 0;
    u128 *out = FILL ? mz + mz_off[i] + chunk_rel[c] : (stage && li < stage_chunks) ? stage + li * SKETCH_STAGE_CAP : nullptr;
    // a report belongs to this chunk iff the reported position does (a position is reported at most once: <= C reports)
#define MPN_PUSH(X, Y) do { const int pp_ = (int)((uint32_t)(Y) >> 1); if (pp_ >= p0 && pp_ < p1) { if (out) { out[cnt].x = (X); out[cnt].y = (Y); } ++cnt; } } while (0)
    int D = w + k + 8;
    for (;;) {
        const int ps = max(0, p0 - D);
        uint64_t kmer0 = 0, kmer1 = 0, minx = ~0ULL, miny = ~0ULL;
        int l = 0, buf_pos = 0, min_pos = 0, extra = 0;
        bool saw_n = false, redo = false;
        int p;
        for (int j = 0; j < w; ++j) bx[j * blockDim.x + tid] = ~0ULL, by[j * blockDim.x + tid] = ~0ULL;
        for (p = ps; p < len; ++p) {
            if (p == p0 && ps > 0 && !saw_n && l < lsat) { redo = true; break; }  // warm-up too short (even k only)
            if (p >= p1 && ++extra > w) break;  // w window steps past the chunk: everything it owns has been reported
            const bool warm = p < p0;
            const int cc = nt4_code(s[p]);
            uint64_t ix = ~0ULL, iy = ~0ULL;
            if (cc < 4) {
                const int kmer_span = l + 1 < k ? l + 1 : k;
                kmer0 = (kmer0 << 2 | (uint64_t)cc) & mask;
                kmer1 = (kmer1 >> 2) | (3ULL ^ (uint64_t)cc) << shift1;
                if (kmer0 == kmer1) { if (p >= p1) --extra; continue; }  // (the window does not move)
                const int z = kmer0 < kmer1 ? 0 : 1;
                if (l < lsat) ++l;
                if (l >= k) {
                    ix = hash64m(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)kmer_span;
                    iy = (uint64_t)rid << 32 | (uint32_t)p << 1 | (uint32_t)z;
                }
            } else { l = 0; if (warm) saw_n = true; }
            bx[buf_pos * blockDim.x + tid] = ix, by[buf_pos * blockDim.x + tid] = iy;
            if (l == w + k - 1 && minx != ~0ULL) {
                for (int j = buf_pos + 1; j < w; ++j) {
                    uint64_t x = bx[j * blockDim.x + tid], y = by[j * blockDim.x + tid];
                    if (minx == x && y != miny) MPN_PUSH(x, y);
                }
                for (int j = 0; j < buf_pos; ++j) {
                    uint64_t x = bx[j * blockDim.x + tid], y = by[j * blockDim.x + tid];
                    if (minx == x && y != miny) MPN_PUSH(x, y);
                }
            }
            if (ix <= minx) {
                if (l >= w + k && minx != ~0ULL) MPN_PUSH(minx, miny);
                minx = ix, miny = iy, min_pos = buf_pos;
            } else if (buf_pos == min_pos) {
                if (l >= w + k - 1 && minx != ~0ULL) MPN_PUSH(minx, miny);
                minx = ~0ULL;
                for (int j = buf_pos + 1; j < w; ++j) {
                    uint64_t x = bx[j * blockDim.x + tid];
                    if (minx >= x) minx = x, miny = by[j * blockDim.x + tid], min_pos = j;
                }
                for (int j = 0; j <= buf_pos; ++j) {
                    uint64_t x = bx[j * blockDim.x + tid];
                    if (minx >= x) minx = x, miny = by[j * blockDim.x + tid], min_pos = j;
                }
                if (l >= w + k - 1 && minx != ~0ULL) {
                    for (int j = buf_pos + 1; j < w; ++j) {
                        uint64_t x = bx[j * blockDim.x + tid], y = by[j * blockDim.x + tid];
                        if (minx == x && miny != y) MPN_PUSH(x, y);
                    }
                    for (int j = 0; j <= buf_pos; ++j) {
                        uint64_t x = bx[j * blockDim.x + tid], y = by[j * blockDim.x + tid];
                        if (minx == x && miny != y) MPN_PUSH(x, y);
                    }
                }
            }
            if (++buf_pos == w) buf_pos = 0;
        }
        if (redo) { D *= 2; cnt = 0; continue; }
        if (p == len && minx != ~0ULL) MPN_PUSH(minx, miny);  // the end of the sequence reports the last minimum
        break;
    }
#undef MPN_PUSH
    if (!FILL) chunk_cnt[c] = (int32_t)cnt;
    }
}

// the staged reports of the automaton kernel's count pass go to their places: a wave per listed chunk
__global__ __launch_bounds__(256) void sketch_stage_copy_kernel(const int32_t *__restrict__ chunk_read, const int64_t *__restrict__ slow_list,
                                                                const unsigned long long *__restrict__ n_slow, int64_t stage_chunks,
                                                                const int64_t *__restrict__ mz_off, const int32_t *__restrict__ chunk_rel,
                                                                const int32_t *__restrict__ chunk_cnt, const u128 *__restrict__ stage,
                                                                u128 *__restrict__ mz) {
    const int lane = threadIdx.x & 63;
    const int64_t n_list = min((int64_t)*n_slow, stage_chunks);
    for (int64_t li = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); li < n_list; li += (int64_t)gridDim.x * 4) {
        const int64_t c = slow_list[li];
        const int cnt = chunk_cnt[c];
        u128 *out = mz + mz_off[chunk_read[c]] + chunk_rel[c];
        const u128 *in = stage + li * SKETCH_STAGE_CAP;
        for (int q = lane; q < cnt; q += 64) out[q] = in[q];
    }
}

// per sequence (one lane): offsets of its chunks' minimizers relative to the sequence start, and the total
__global__ __launch_bounds__(256) void sketch_chunk_prefix_kernel(const int64_t *__restrict__ chunk_off, int n,
                                                                  const int32_t *__restrict__ chunk_cnt,
                                                                  int32_t *__restrict__ chunk_rel, int64_t *__restrict__ mz_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t run = 0;
    for (int64_t c = chunk_off[i]; c < chunk_off[i + 1]; ++c) { chunk_rel[c] = run; run += chunk_cnt[c]; }
    mz_cnt[i] = run;
}

// exclusive scan of n int64 by ONE block (n is the number of reads of a batch: small); out[n] = total
// two independent scans in one launch (a stage often needs two of them at the same point: a launch each was a third of the
// seed stage's small commands): block 0 scans the first array, block 1 the second
__global__ __launch_bounds__(1024) void scan_i64_kernel(const int64_t *__restrict__ in, int64_t *__restrict__ out, int n);
__global__ __launch_bounds__(1024) void scan_i64x2_kernel(const int64_t *__restrict__ in0, int64_t *__restrict__ out0, int n0,
                                                          const int64_t *__restrict__ in1, int64_t *__restrict__ out1, int n1) {
    __shared__ int64_t part[1024];
    const int64_t *in = blockIdx.x ? in1 : in0;
    int64_t *out = blockIdx.x ? out1 : out0;
    const int n = blockIdx.x ? n1 : n0;
    const int tid = threadIdx.x, per = (n + 1023) / 1024;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int64_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) { int64_t v = in[i]; out[i] = run; run += v; }
    if (tid == 1023) out[n] = part[1023];
}

__global__ __launch_bounds__(1024) void scan_i64_kernel(const int64_t *__restrict__ in, int64_t *__restrict__ out, int n) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x, per = (n + 1023) / 1024;
    const int lo = min(n, tid * per), hi = min(n, lo + per);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int64_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) { int64_t v = in[i]; out[i] = run; run += v; }
    if (tid == 1023) out[n] = part[1023];
}

// ---------------------------------------------------------------------------------------------------------
// seeds: one lane per read minimizer.  A bucket table over the top bits of the hash (bucket_start[b] = first key of bucket
// b, 2^bucket_bits + 1 entries, ~4 keys per bucket) narrows the binary search to one cache line of (key, first position)
// pairs, which also holds the answer: two or three lines per lookup instead of ~18 uncached probes over the whole array.
__global__ __launch_bounds__(256) void idx_bucket_table_kernel(const uint64_t *__restrict__ keys, int64_t n_keys, int shift, int64_t n_buckets,
                                                               int64_t *__restrict__ bucket_start) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_keys; i += (int64_t)gridDim.x * blockDim.x) {
        // entry i closes the buckets (b_prev, b_i]: they all start at key i
        const int64_t b_prev = i > 0 ? (int64_t)(keys[i - 1] >> shift) : -1;
        const int64_t b_i = i < n_keys ? (int64_t)(keys[i] >> shift) : n_buckets;
        for (int64_t b = b_prev + 1; b <= b_i; ++b) bucket_start[b] = i;
    }
}

__global__ __launch_bounds__(256) void idx_kv_kernel(const uint64_t *__restrict__ keys, const int64_t *__restrict__ key_off, int64_t n_keys,
                                                     u128 *__restrict__ kv) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_keys; i += (int64_t)gridDim.x * blockDim.x)
        kv[i] = u128{i < n_keys ? keys[i] : ~0ULL, (uint64_t)key_off[i]};
}

__global__ __launch_bounds__(256) void seed_lookup_kernel(const u128 *__restrict__ kv,
                                                          int64_t n_keys, const int64_t *__restrict__ bucket_start, int bucket_shift,
                                                          const u128 *__restrict__ mz, int64_t n_mz,
                                                          int32_t max_occ, int32_t *__restrict__ occ,
                                                          int64_t *__restrict__ pos_start) {
    for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < n_mz; m += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = mz[m].x >> 8;
        const int64_t b = (int64_t)(key >> bucket_shift);
        int64_t lo = bucket_start[b], hi = bucket_start[b + 1];
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (kv[mid].x < key) lo = mid + 1; else hi = mid;
        }
        int32_t t = 0;
        int64_t st = 0;
        if (lo < n_keys) {
            const u128 e = kv[lo];
            if (e.x == key) { st = (int64_t)e.y; t = (int32_t)min<int64_t>((int64_t)kv[lo + 1].y - st, 0x7fffffff); }
        }
        occ[m] = t >= max_occ ? -1 : t;  // -1: repetitive, skipped (counts into rep_len)
        pos_start[m] = st;
    }
}

// per read (one wave): anchor offsets of its minimizers, anchor total and rep_len (minimap2 collect_matches).
// rep_len is the length of the union of the query intervals [st, en) of the too-frequent minimizers; their `en` grows
// with the minimizer index, so an interval adds en - max(st, en of the previous such interval): a max-scan, no loop.
// span_sum = sum of the seed lengths of ALL the read's anchors (mm_chain_dp's average seed length is taken over every
// anchor, also over those the stray-hit filter below never materialises); n_blk = 64-minimizer blocks of the read.
__global__ __launch_bounds__(64) void seed_prefix_kernel(const u128 *__restrict__ mz, const int64_t *__restrict__ mz_off, int n,
                                                         const int32_t *__restrict__ occ, int64_t *__restrict__ rel_off,
                                                         int64_t *__restrict__ n_anchor, int32_t *__restrict__ rep_len,
                                                         unsigned long long *__restrict__ span_sum, int64_t *__restrict__ n_blk) {
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t m0 = mz_off[i], m1 = mz_off[i + 1];
        int64_t run = 0;
        int32_t rl = 0, prev_en = 0;
        unsigned long long ss = 0;
        for (int64_t c = m0; c < m1; c += 64) {
            const int64_t m = c + lane;
            int32_t t = 0, st = 0, en = 0;
            uint32_t q_span = 0;
            if (m < m1) {
                t = occ[m];
                q_span = mz[m].x & 0xff;
                if (t < 0) {
                    const uint32_t q_pos = (uint32_t)mz[m].y;
                    en = (int)(q_pos >> 1) + 1; st = en - (int)q_span;
                }
            }
            const bool rep = t < 0;
            // exclusive prefix sum of the hit counts of this tile
            int32_t incl = t > 0 ? t : 0;
            const int32_t own = incl;
            ss += (unsigned long long)own * q_span;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            if (m < m1) rel_off[m] = run + (incl - own);
            run += __shfl(incl, 63);
            // union length of the repetitive intervals
            const int32_t e_incl = wave_scan_max(rep ? en : 0);
            const int32_t before = max(wave_shr1(e_incl, 0), prev_en);
            int32_t add = rep ? en - max(st, before) : 0;
            for (int d = 32; d; d >>= 1) add += __shfl_xor(add, d);
            rl += add;
            prev_en = max(prev_en, __builtin_amdgcn_readlane(e_incl, 63));
        }
        for (int d = 32; d; d >>= 1) ss += __shfl_xor(ss, d);
        if (lane == 0) { n_anchor[i] = run; rep_len[i] = rl; span_sum[i] = ss; n_blk[i] = (m1 - m0 + 63) / 64; }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stray-hit filter.  Against a large target set a read minimizer finds tens of index positions, of which one or two
// belong to the read's true locus; the rest are isolated hits that cannot chain.  The anchor compaction below drops
// every SEGMENT (maximal run of a read's sorted anchors on one strand of one target whose consecutive reference gaps
// are <= max_dist_x) of fewer than min_cnt anchors.  A member of a segment of >= min_cnt anchors has T = min(min_cnt, 3)
// consecutive members around it that span <= (T-1) * max_dist_x reference bases, so with bins of width W >= 2 (T-1)
// max_dist_x it shares an "A" bin (pos / W) or a "B" bin ((pos + W/2) / W) with T-1 other anchors of its read.  The filter
// counts a read's hits per bin WITHOUT materialising the anchors and emits only those whose A or B bin holds >= T hits:
// a superset of what the compaction keeps, so the chains are identical, while the anchor array, its partition, sort and
// compaction shrink by an order of magnitude.
// Counting: one workgroup per read, counters in LDS as count-min sketches (two hash functions per bin family) of
// saturating 3-level counters -- three bitmaps "seen once / twice / three times", raised by atomicOr (the returned old
// word tells whether to climb a level).  Collisions only make the filter keep more.
// The hits are walked like a load-balanced expand: a wave takes 64 consecutive minimizers of its read, whose hits are one
// contiguous range of virtual slots, FLT_UNROLL 64-slot windows per step (as many independent gathers in flight per lane).
// Pass 1 counts, pass 2 (after a barrier) gathers again, tests, and leaves one keep word per window + the block's total.
// A read with many hits overfills its tables (24 kb: ~120 k hits on 82 k slots, a third of the strays pass): it gets a SECOND
// ROUND over the survivors only -- fresh tables, count, test, the keep words rewritten.  Every member of a segment the
// compaction keeps survived the first round together with the T-1 neighbours that share its bin, so it passes again: still a
// superset, from tables that hold a third of the hits.
constexpr int FLT_THREADS = 1024, FLT_WAVES = FLT_THREADS / 64, FLT_UNROLL = 4;  // windows of 64 slots per wave step
constexpr int FLT_SLOTS = 81920, FLT_WORDS = FLT_SLOTS / 32;      // bits / words per bitmap; 4 tables x 3 levels = 120 KB
constexpr int FLT_CHUNK = 4096;                                   // slots per start-bit chunk of a wave (64 words)
constexpr size_t FLT_LDS_BYTES = (size_t)12 * FLT_WORDS * 4 + (size_t)FLT_WAVES * (64 * 8 + FLT_CHUNK / 8 + 64 * 4) + 16;
struct FilterParams { int shift; uint32_t half; int level; int64_t second_round; };   // bin = pos >> shift; level = T - 1 (bitmap tested);
                                                                  // second_round: reads with at least this many hits are counted twice

__device__ __forceinline__ void flt_slots(uint32_t hi, uint32_t bin, uint32_t &s1, uint32_t &s2) {
    uint32_t k = hi * 0x9E3779B1u ^ (bin + 0x7F4A7C15u) * 0x85EBCA77u;
    k ^= k >> 15; k *= 0x2C1B3C6Du; k ^= k >> 12;
    s1 = __umulhi(k, (uint32_t)FLT_SLOTS);
    uint32_t k2 = k * 0x297A2D39u; k2 ^= k2 >> 15;
    s2 = __umulhi(k2 * 0x85EBCA6Bu, (uint32_t)FLT_SLOTS);
}
__device__ __forceinline__ void flt_add(uint32_t *tab, uint32_t slot) {  // tab: the 3 level bitmaps of one table
    const uint32_t w = slot >> 5, bit = 1u << (slot & 31);
    if (atomicOr(&tab[w], bit) & bit)
        if (atomicOr(&tab[FLT_WORDS + w], bit) & bit) atomicOr(&tab[2 * FLT_WORDS + w], bit);
}
__device__ __forceinline__ bool flt_test(const uint32_t *tab, uint32_t slot, int level) {
    return tab[level * FLT_WORDS + (slot >> 5)] >> (slot & 31) & 1;
}

// word of the keep bitmap that holds window `it` of block gblk whose first virtual slot is vfirst: blocks are unaligned
// ranges of the virtual slot space, one extra word per block keeps them disjoint
__device__ __forceinline__ int64_t flt_word(int64_t vfirst, int64_t gblk, int64_t it) { return (vfirst >> 6) + gblk + it; }

__global__ __launch_bounds__(FLT_THREADS) void seed_filter_kernel(const u128 *__restrict__ mz, const int64_t *__restrict__ mz_off, int n_reads,
                                                                  const int32_t *__restrict__ occ, const int64_t *__restrict__ pos_start,
                                                                  const int64_t *__restrict__ rel_off, const uint64_t *__restrict__ pos,
                                                                  const int64_t *__restrict__ full_off, const int64_t *__restrict__ blk_base,
                                                                  const int32_t *__restrict__ order, FilterParams fp,
                                                                  unsigned long long *__restrict__ keep,
                                                                  int64_t *__restrict__ blk_kept, int32_t *__restrict__ blk_read,
                                                                  unsigned int *__restrict__ next_read) {
    extern __shared__ uint32_t flt_lds[];
    uint32_t *bm = flt_lds;                                            // [table A1, A2, B1, B2][level][FLT_WORDS]
    // per wave: the block's minimizers that have hits, compacted -- index position of the first hit | parity << 62,
    // block-relative first slot -- and the start-of-minimizer bits of the chunk of FLT_CHUNK slots being walked
    unsigned long long *ps_all = (unsigned long long *)(flt_lds + 12 * FLT_WORDS);
    unsigned long long *sb_all = ps_all + FLT_WAVES * 64;
    uint32_t *g_all = (uint32_t *)(sb_all + FLT_WAVES * (FLT_CHUNK / 64));
    uint32_t *next_blk = g_all + FLT_WAVES * 64;                       // [2]: the block queue of each pass; [2]: the read taken
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long *ps = ps_all + wv * 64, *sb = sb_all + wv * (FLT_CHUNK / 64);
    uint32_t *g = g_all + wv * 64;
    const unsigned long long lane_le = lane == 63 ? ~0ULL : (2ULL << lane) - 1;
    // Persistent workgroups (one per CU) take reads from a counter: a grid of one workgroup per read would keep its hardware
    // queue's dispatcher busy for the whole kernel (each workgroup needs a whole CU), and every kernel behind it on that
    // pipe -- of any worker -- waits to be dispatched.
    for (;;) {
        if (tid == 0) next_blk[2] = atomicAdd(next_read, 1u);
        __syncthreads();
        const int ri = (int)next_blk[2];
        __syncthreads();
        if (ri >= n_reads) break;
        const int read = order[ri];                                    // reads with many hits first (the grid's long pole)
        const int64_t m0 = mz_off[read], m1 = mz_off[read + 1];
        const int64_t v0 = full_off[read];
        if (full_off[read + 1] == v0) continue;                        // no hits: its blocks keep blk_kept = 0
        const int64_t nblk = (m1 - m0 + 63) / 64, gb0 = blk_base[read];
        const int n_pass = full_off[read + 1] - v0 >= fp.second_round ? 4 : 2;
        for (int pass = 0; pass < n_pass; ++pass) {
            const bool counting = !(pass & 1), again = pass >= 2;   // again: only the hits the first round kept take part
            if (counting) for (int k = tid; k < 12 * FLT_WORDS; k += FLT_THREADS) bm[k] = 0;
            if (tid == 0) next_blk[0] = 0;
            __syncthreads();
            for (;;) {  // the waves take the read's blocks from a queue: the blocks differ in their hit counts
                uint32_t bq = 0;
                if (lane == 0) bq = atomicAdd(&next_blk[0], 1u);
                const int64_t b = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)bq);
                if (b >= nblk) break;
                const int64_t m = m0 + b * 64 + lane;
                int32_t t = 0;
                int64_t rel = 0;
                unsigned long long psv = 0;
                if (m < m1) {
                    t = occ[m];
                    if (t < 0) t = 0;
                    rel = rel_off[m];
                    psv = (unsigned long long)pos_start[m] | (unsigned long long)((uint32_t)mz[m].y & 1) << 62;
                }
                const int64_t first = __shfl(rel, 0);
                const uint32_t start = m < m1 ? (uint32_t)(rel - first) : 0u;   // (64 minimizers x mid_occ hits: far below 2^32)
                uint32_t end_all = start + (uint32_t)t;
#pragma unroll
                for (int d = 32; d; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)end_all, d); end_all = o > end_all ? o : end_all; }
                // owners of the slots: the minimizers with hits, compacted, so that their starts are distinct
                const unsigned long long nz = __ballot(t > 0);
                if (t > 0) { const int c = __popcll(nz & (lane_le >> 1)); g[c] = start; ps[c] = psv; }
                const int64_t vfirst = v0 + first;
                int64_t kept = 0;
                for (uint32_t c0 = 0; c0 < end_all; c0 += FLT_CHUNK) {
                    sb[lane] = 0;                                       // (FLT_CHUNK / 64 == 64 words)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    if (t > 0 && start >= c0 && start - c0 < (uint32_t)FLT_CHUNK) atomicOr(&sb[(start - c0) >> 6], 1ULL << ((start - c0) & 63));
                    int before = __popcll(__ballot(t > 0 && start < c0));  // owners that start before the chunk
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t c_end = end_all - c0 < (uint32_t)FLT_CHUNK ? end_all - c0 : (uint32_t)FLT_CHUNK;
                    for (uint32_t q0 = 0; q0 < c_end; q0 += 64 * FLT_UNROLL) {
                        uint64_t r[FLT_UNROLL];
                        uint32_t par[FLT_UNROLL];
                        bool live[FLT_UNROLL];
#pragma unroll
                        for (int u = 0; u < FLT_UNROLL; ++u) {
                            r[u] = 0; par[u] = 0;
                            const uint32_t w0 = q0 + u * 64;
                            live[u] = false;
                            if (w0 < c_end) {                           // (uniform)
                                const unsigned long long word = sb[w0 >> 6];
                                const uint32_t p = c0 + w0 + lane;
                                live[u] = p < end_all;
                                if (again) live[u] = live[u] && (keep[flt_word(vfirst, gb0 + b, (int64_t)((c0 + w0) >> 6))] >> lane & 1);
                                if (live[u]) {
                                    const int own = before + __popcll(word & lane_le) - 1;   // last owner that starts at or before p
                                    const unsigned long long pv = ps[own];
                                    par[u] = (uint32_t)(pv >> 62) & 1;
                                    r[u] = pos[(pv & 0x3fffffffffffffffULL) + (p - g[own])];
                                }
                                before += __popcll(word);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < FLT_UNROLL; ++u) {
                            const uint32_t w0 = q0 + u * 64;
                            if (w0 >= c_end) break;
                            const uint32_t p = c0 + w0 + lane;
                            bool keep_it = false;
                            if (live[u]) {
                                const uint32_t hi = (uint32_t)(r[u] >> 32) | (((uint32_t)r[u] & 1) ^ par[u]) << 31;   // strand | target
                                const uint32_t rpos = (uint32_t)r[u] >> 1;
                                const uint32_t ba = (uint32_t)((uint64_t)rpos >> fp.shift), bb = (uint32_t)(((uint64_t)rpos + fp.half) >> fp.shift);
                                uint32_t a1, a2, b1, b2;
                                flt_slots(hi, ba, a1, a2);
                                flt_slots(hi ^ 0x5bd1e995u, bb, b1, b2);
                                if (counting) {
                                    flt_add(bm, a1); flt_add(bm + 3 * FLT_WORDS, a2);
                                    flt_add(bm + 6 * FLT_WORDS, b1); flt_add(bm + 9 * FLT_WORDS, b2);
                                } else {
                                    keep_it = (flt_test(bm, a1, fp.level) && flt_test(bm + 3 * FLT_WORDS, a2, fp.level)) ||
                                              (flt_test(bm + 6 * FLT_WORDS, b1, fp.level) && flt_test(bm + 9 * FLT_WORDS, b2, fp.level));
                                }
                            }
                            if (!counting) {
                                const unsigned long long km = __ballot(keep_it);
                                if (lane == 0) keep[flt_word(vfirst, gb0 + b, (int64_t)((c0 + w0) >> 6))] = km;
                                kept += __popcll(km);
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (!counting && lane == 0) { blk_kept[gb0 + b] = kept; blk_read[gb0 + b] = read; }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
        }
    }
}

// processing order of the filter's workgroups: reads by decreasing hit count, in classes of a factor of two (one workgroup)
__global__ __launch_bounds__(1024) void seed_order_kernel(const int64_t *__restrict__ n_anchor, int n, int32_t *__restrict__ order) {
    __shared__ uint32_t cnt[64], cur[64];
    const int tid = threadIdx.x;
    if (tid < 64) cnt[tid] = 0;
    __syncthreads();
    for (int r = tid; r < n; r += 1024) atomicAdd(&cnt[__clzll((unsigned long long)n_anchor[r] + 1)], 1u);  // class 0 = the largest
    __syncthreads();
    if (tid == 0) { uint32_t run = 0; for (int c = 0; c < 64; ++c) { cur[c] = run; run += cnt[c]; } }
    __syncthreads();
    for (int r = tid; r < n; r += 1024) order[atomicAdd(&cur[__clzll((unsigned long long)n_anchor[r] + 1)], 1u)] = r;
}

// anchor offsets per read from the block offsets: the blocks of a read are consecutive
__global__ __launch_bounds__(256) void seed_read_off_kernel(const int64_t *__restrict__ blk_off, const int64_t *__restrict__ blk_base, int n,
                                                            int64_t *__restrict__ anchor_off) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n) anchor_off[r] = blk_off[blk_base[r]];
}

// Emission of the kept hits: one wave per 64-minimizer block; it walks the block's keep words, and a lane whose bit is set
// finds the owner of its slot, gathers the index position and stores the 16-byte anchor at the block's offset + its rank.
__global__ __launch_bounds__(256) void seed_emit_kernel(const u128 *__restrict__ mz, const int64_t *__restrict__ mz_off,
                                                        const int32_t *__restrict__ occ, const int64_t *__restrict__ pos_start,
                                                        const int64_t *__restrict__ rel_off, const uint64_t *__restrict__ pos,
                                                        const int64_t *__restrict__ full_off, const int64_t *__restrict__ blk_base, int n_reads,
                                                        const unsigned long long *__restrict__ keep, const int64_t *__restrict__ blk_kept,
                                                        const int64_t *__restrict__ blk_off, const int32_t *__restrict__ blk_read,
                                                        const int32_t *__restrict__ seq_len, u128 *__restrict__ anchors) {
    __shared__ int64_t s_g[4][64], s_ps[4][64];
    __shared__ uint64_t s_y[4][64];   // flags | span << 32 | query position (forward strand)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t *g = s_g[wv], *ps = s_ps[wv];
    uint64_t *yy = s_y[wv];
    const int64_t n_blocks = blk_base[n_reads];
    for (int64_t gblk = (int64_t)blockIdx.x * 4 + wv; gblk < n_blocks; gblk += (int64_t)gridDim.x * 4) {
        if (blk_kept[gblk] == 0) continue;
        const int read = blk_read[gblk];
        const int64_t m0 = mz_off[read], m1 = mz_off[read + 1];
        const int64_t m = m0 + (gblk - blk_base[read]) * 64 + lane;
        const int32_t qlen = seq_len[read];
        int32_t t = 0;
        int64_t rel = 0;
        if (m < m1) {
            t = occ[m];
            if (t < 0) t = 0;
            const uint64_t mx = mz[m].x, my = mz[m].y;
            bool tandem = false;
            if (m > m0 && mz[m - 1].x >> 8 == mx >> 8) tandem = true;
            if (m + 1 < m1 && mz[m + 1].x >> 8 == mx >> 8) tandem = true;
            rel = rel_off[m];
            ps[lane] = pos_start[m];
            yy[lane] = (tandem ? 1ULL << 42 : 0ULL) | (uint64_t)(mx & 0xff) << 32 | (uint32_t)my;
        }
        const int64_t first = __shfl(rel, 0);
        int64_t end_all = rel - first + t;
#pragma unroll
        for (int d = 32; d; d >>= 1) { const int64_t o = __shfl_xor(end_all, d); end_all = o > end_all ? o : end_all; }
        g[lane] = m < m1 ? rel - first : end_all;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int64_t vfirst = full_off[read] + first;
        int64_t out = blk_off[gblk];
        for (int64_t p0 = 0, it = 0; p0 < end_all; p0 += 64, ++it) {
            const unsigned long long km = keep[flt_word(vfirst, gblk, it)];
            if (km >> lane & 1) {
                const int64_t p = p0 + lane;
                int lo = 0, hi = 63;
#pragma unroll
                for (int step = 0; step < 6; ++step) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (g[mid] <= p) lo = mid; else hi = mid - 1;
                }
                const uint64_t r = pos[ps[lo] + (p - g[lo])];
                const uint64_t y = yy[lo];
                const uint32_t q_pos = (uint32_t)y, q_span = (uint32_t)(y >> 32) & 0xff;
                const uint32_t rpos = (uint32_t)r >> 1;
                u128 a;
                if ((r & 1) == (q_pos & 1)) {
                    a.x = (r & 0xffffffff00000000ULL) | rpos;
                    a.y = (uint64_t)q_span << 32 | q_pos >> 1;
                } else {
                    a.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | rpos;
                    a.y = (uint64_t)q_span << 32 | (uint32_t)(qlen - ((int32_t)(q_pos >> 1) + 1 - (int32_t)q_span) - 1);
                }
                a.y |= y & (1ULL << 42);
                anchors[out + __popcll(km & ((1ULL << lane) - 1))] = a;
            }
            out += __popcll(km);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Segmented LSD radix sort, one workgroup (256 threads) per segment, 8-bit digits, stable.
// Keys are the 96 bits (x, low 32 bits of y) of 16-byte records; a pass whose digit is constant over the
// segment costs only its histogram.  Ranks inside a 256-element tile come from ballot-based multisplit.
template <int NY>
__device__ __forceinline__ uint32_t sort_digit(const u128 &r, int pass) {
    return pass < NY ? (uint32_t)(r.y >> (8 * pass)) & 0xff : (uint32_t)(r.x >> (8 * (pass - NY))) & 0xff;
}

// NY = 4: anchors, key (x, low 32 bits of y).  NY = 8: index records, key (x, y).
// Statistics counter "records moved" (bench.py's algorithmic bytes of the sort).  It is spread over WORK_SLOTS addresses:
// atomics of thousands of workgroups on ONE address serialise at the memory side and each workgroup then waits for its
// own (measured: they were the largest cost of the window-sort and list kernels).
constexpr int WORK_SLOTS = 256;
__device__ __forceinline__ void add_work(unsigned long long *work, unsigned long long n) {
    if (work) atomicAdd(&work[blockIdx.x & (WORK_SLOTS - 1)], n);
}

struct SegSortLds {
    uint32_t hist[256], bins[256], wcnt[4][256];
    int skip;
};

// one segment [data, data + n) sorted by the calling workgroup (256 threads); tmp: bounce buffer of the same extent
template <int NY>
__device__ __forceinline__ void wg_radix_sort(SegSortLds &L, u128 *__restrict__ data, u128 *__restrict__ tmp, int64_t n,
                                              unsigned long long *__restrict__ work) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u128 *src = data, *dst = tmp;
    for (int pass = 0; pass < NY + 8; ++pass) {
        L.hist[tid] = 0;
        __syncthreads();
        for (int64_t i = tid; i < n; i += 256) atomicAdd(&L.hist[sort_digit<NY>(src[i], pass)], 1u);
        __syncthreads();
        if (tid == 0) L.skip = 0;
        __syncthreads();
        if (L.hist[tid] == (uint32_t)n) L.skip = 1;
        __syncthreads();
        if (L.skip) { __syncthreads(); continue; }
        if (tid == 0) add_work(work, (unsigned long long)n);  // records moved by this pass
        // exclusive scan of the histogram (256 entries, thread per bin)
        {
            uint32_t v = L.hist[tid];
            L.bins[tid] = v;
            __syncthreads();
            for (int d = 1; d < 256; d <<= 1) {
                uint32_t a = tid >= d ? L.bins[tid - d] : 0;
                __syncthreads();
                L.bins[tid] += a;
                __syncthreads();
            }
            uint32_t excl = L.bins[tid] - v;
            __syncthreads();
            L.bins[tid] = excl;
            __syncthreads();
        }
        for (int64_t t0 = 0; t0 < n; t0 += 256) {
            const int64_t i = t0 + tid;
            const bool act = i < n;
            u128 r;
            uint32_t dg = 0;
            if (act) { r = src[i]; dg = sort_digit<NY>(r, pass); }
            L.wcnt[0][tid] = 0; L.wcnt[1][tid] = 0; L.wcnt[2][tid] = 0; L.wcnt[3][tid] = 0;
            __syncthreads();
            // lanes of this wave holding the same digit
            unsigned long long same = __ballot(act);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                unsigned long long m = __ballot((dg >> b) & 1);
                same &= ((dg >> b) & 1) ? m : ~m;
            }
            const unsigned long long lt = (1ULL << lane) - 1;
            const uint32_t rank = __popcll(same & lt);
            if (act && rank == 0) L.wcnt[wv][dg] = __popcll(same);
            __syncthreads();
            if (act) {
                uint32_t o = L.bins[dg] + rank;
                for (int w2 = 0; w2 < wv; ++w2) o += L.wcnt[w2][dg];
                dst[o] = r;
            }
            __syncthreads();
            L.bins[tid] += L.wcnt[0][tid] + L.wcnt[1][tid] + L.wcnt[2][tid] + L.wcnt[3][tid];
            __syncthreads();
        }
        u128 *sw = src; src = dst; dst = sw;
        __threadfence_block();
        __syncthreads();
    }
    if (src != data) {
        for (int64_t i = tid; i < n; i += 256) data[i] = src[i];
    }
    __syncthreads();
}

// segments = consecutive CSR rows (seg_off[n_seg + 1])
template <int NY>
__global__ __launch_bounds__(256) void seg_sort_kernel(u128 *__restrict__ data, u128 *__restrict__ tmp,
                                                       const int64_t *__restrict__ seg_off, int n_seg,
                                                       unsigned long long *__restrict__ work) {
    __shared__ SegSortLds L;
    for (int seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const int64_t base = seg_off[seg];
        const int64_t n = seg_off[seg + 1] - base;
        if (n < 2) continue;
        wg_radix_sort<NY>(L, data + base, tmp + base, n, work);
    }
}

// segments = an explicit work list (offset, length) whose length is read from the device
struct SortSeg { int64_t off; int64_t len; };
template <int NY>
__global__ __launch_bounds__(256) void seg_sort_list_kernel(u128 *__restrict__ data, u128 *__restrict__ tmp,
                                                            const SortSeg *__restrict__ list, const unsigned int *__restrict__ n_list,
                                                            unsigned int cap, unsigned long long *__restrict__ work) {
    __shared__ SegSortLds L;
    const unsigned int n_seg = *n_list < cap ? *n_list : cap;
    for (unsigned int seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const SortSeg sg = list[seg];
        if (sg.len < 2) continue;
        wg_radix_sort<NY>(L, data + sg.off, tmp + sg.off, sg.len, work);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Anchor sort for large target sets.  Against tens of Gbp of targets a read collects tens of thousands of stray seed
// hits spread evenly over (strand, target): an LSD radix sort moves every one of them through HBM eight times.  Instead:
//   1. anchor_msd_kernel: ONE partition pass per read (workgroup per read) on the top bits of the sort key -- strand,
//      then (target, position) squeezed to what the target set needs -- with 2^14 counters in LDS.  A bucket then holds
//      a couple of strays, or the read's true locus.
//      Buckets of more than 16 anchors (the true loci: a few per cent of the anchors) go to work lists by size.
//   2. anchor_window_sort_kernel: the partitioned array is streamed through LDS in windows; a lane sorts a small bucket
//      by insertion (a few LDS moves).
//   3. anchor_bitonic_list_kernel: listed buckets of up to 4096 anchors, one load + a bitonic network in LDS + one store;
//      seg_sort_list_kernel: the rest by radix passes through HBM.
// Keys are unique (a read position and a target position identify an anchor), so the result does not depend on the
// order in which the atomics of pass 1 placed the records.
constexpr int MSD_BITS = 14, MSD_NB = 1 << MSD_BITS, MSD_THREADS = 1024, SMALL_BUCKET = 16;
constexpr int BITONIC_SMALL = 1024, BITONIC_MID = 4096;  // work lists: buckets of 17..1024, 1025..4096, more (radix passes)
struct BinParams { int pos_bits, shift; };

__device__ __forceinline__ uint32_t anchor_bin(uint64_t x, const BinParams bp) {
    const uint64_t comp = ((x >> 32 & 0x7fffffffULL) << bp.pos_bits) | (uint64_t)(uint32_t)x;
    return (uint32_t)(x >> 63) << (MSD_BITS - 1) | (uint32_t)(comp >> bp.shift);
}

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}

__device__ __forceinline__ bool anchor_less(uint64_t ax, uint64_t ay, uint64_t bx, uint64_t by) {
    return ax < bx || (ax == bx && (uint32_t)ay < (uint32_t)by);
}

struct SortLists {           // three work lists in one allocation: [0] <= BITONIC_SMALL, [1] <= BITONIC_MID, [2] larger
    SortSeg *seg[3];
    unsigned int *count;     // count[3]
    unsigned int cap[3];
};

// 16 waves per workgroup, two workgroups per CU (64 KB of counters each): the loops are bound by the latency of their
// global loads, so what counts is loads in flight -- resident waves x the 4 independent loads each lane issues per step.
__global__ __launch_bounds__(MSD_THREADS) void anchor_msd_kernel(const u128 *__restrict__ src, u128 *__restrict__ dst,
                                                                 const int64_t *__restrict__ anchor_off, int n_reads, BinParams bp,
                                                                 SortLists lists, unsigned long long *__restrict__ work) {
    extern __shared__ uint32_t msd_lds[];  // MSD_NB counters + one total per wave + 6 list words (launch with (MSD_NB + 32) * 4 bytes)
    uint32_t *cnt = msd_lds, *wtot = msd_lds + MSD_NB;
    constexpr int NT = MSD_THREADS, NW = NT / 64, PER_WAVE = MSD_NB / NW;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int read = blockIdx.x; read < n_reads; read += gridDim.x) {
        const int64_t base = anchor_off[read];
        const int64_t n = anchor_off[read + 1] - base;
        if (n == 0) continue;
        const u128 *in = src + base;
        u128 *out = dst + base;
        for (int k = tid; k < MSD_NB; k += NT) cnt[k] = 0;
        __syncthreads();
        for (int64_t i0 = tid; i0 < n; i0 += 4 * NT) {
            uint64_t x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int64_t i = i0 + u * NT; x[u] = i < n ? in[i].x : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + u * NT < n) atomicAdd(&cnt[anchor_bin(x[u], bp)], 1u);
        }
        __syncthreads();
        // exclusive scan of the counters: every wave owns a slice, 64 consecutive counters per step (conflict-free)
        uint32_t s = 0;
        for (int c = 0; c < PER_WAVE; c += 64) s += cnt[wv * PER_WAVE + c + lane];
        for (int d = 32; d; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) wtot[wv] = s;
        __syncthreads();
        uint32_t carry = 0;
        for (int w2 = 0; w2 < wv; ++w2) carry += wtot[w2];
        for (int c = 0; c < PER_WAVE; c += 64) {
            const uint32_t v = cnt[wv * PER_WAVE + c + lane];
            const uint32_t incl = wave_incl_add(v, lane);
            cnt[wv * PER_WAVE + c + lane] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
        __syncthreads();
        // buckets the window kernel will not sort in LDS go to the work lists (their extents are known here).  The workgroup
        // reserves its list entries with ONE atomic per list (thousands of workgroups pushing one by one would queue on it)
        uint32_t *lcnt = wtot + NW, *lbase = lcnt + 3;  // [3] each
        if (tid < 3) lcnt[tid] = 0;
        __syncthreads();
        for (int b = tid; b < MSD_NB; b += NT) {
            const uint32_t st = cnt[b], en = b + 1 < MSD_NB ? cnt[b + 1] : (uint32_t)n;
            const uint32_t len = en - st;
            if (len > SMALL_BUCKET) atomicAdd(&lcnt[len <= BITONIC_SMALL ? 0 : len <= BITONIC_MID ? 1 : 2], 1u);
        }
        __syncthreads();
        if (tid < 3) { lbase[tid] = lcnt[tid] ? atomicAdd(&lists.count[tid], lcnt[tid]) : 0u; lcnt[tid] = 0; }
        __syncthreads();
        for (int b = tid; b < MSD_NB; b += NT) {
            const uint32_t st = cnt[b], en = b + 1 < MSD_NB ? cnt[b + 1] : (uint32_t)n;
            const uint32_t len = en - st;
            if (len > SMALL_BUCKET) {
                const int c = len <= BITONIC_SMALL ? 0 : len <= BITONIC_MID ? 1 : 2;
                const uint32_t w = lbase[c] + atomicAdd(&lcnt[c], 1u);
                if (w < lists.cap[c]) lists.seg[c][w] = SortSeg{base + st, (int64_t)len};
            }
        }
        __syncthreads();
        for (int64_t i0 = tid; i0 < n; i0 += 4 * NT) {
            u128 r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int64_t i = i0 + u * NT; if (i < n) r[u] = in[i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * NT < n) { const uint32_t p = atomicAdd(&cnt[anchor_bin(r[u].x, bp)], 1u); out[p] = r[u]; }
        }
        if (tid == 0) add_work(work, (unsigned long long)n);
        __syncthreads();
    }
}

// Small buckets, sorted in LDS.  The batch's partitioned anchor array is cut into windows of SORT_WIN anchors regardless of
// read boundaries; a workgroup owns the buckets that START in its window (a bucket start = the bin changes, or a read's
// first anchor) and loads SMALL_BUCKET more records so that an owned small bucket is complete.  Owners are unique, so
// the windows are independent: the grid is the number of windows, not the number of reads.  Buckets of more than
// SMALL_BUCKET anchors are left as they are (the partition kernel has put them on the work lists).
constexpr int SORT_WIN = 1024, SORT_LOAD = SORT_WIN + SMALL_BUCKET;

__global__ __launch_bounds__(256) void anchor_window_sort_kernel(u128 *__restrict__ data, const int64_t *__restrict__ anchor_off,
                                                                 int n_reads, int64_t n_a, BinParams bp,
                                                                 unsigned long long *__restrict__ work) {
    __shared__ uint64_t kx[SORT_LOAD], ky[SORT_LOAD];
    __shared__ unsigned long long flags[SORT_LOAD / 64 + 2];  // bit i: a bucket starts at window position i (i <= cnt: the end of the data counts)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t n_win = (n_a + SORT_WIN - 1) / SORT_WIN;
    for (int64_t win = blockIdx.x; win < n_win; win += gridDim.x) {
        const int64_t g0 = win * SORT_WIN;
        const int cnt = (int)(n_a - g0 < SORT_LOAD ? n_a - g0 : SORT_LOAD);       // records loaded
        const int own = (int)(n_a - g0 < SORT_WIN ? n_a - g0 : SORT_WIN);         // positions whose buckets are ours
        {   // all loads of the window are issued before the first one is consumed (the kernel is bound by their latency)
            constexpr int NL = (SORT_LOAD + 255) / 256;
            u128 r[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) { const int i = tid + u * 256; if (i < cnt) r[u] = data[g0 + i]; }
#pragma unroll
            for (int u = 0; u < NL; ++u) { const int i = tid + u * 256; if (i < cnt) { kx[i] = r[u].x; ky[i] = r[u].y; } }
        }
        for (int k = tid; k < SORT_LOAD / 64 + 2; k += 256) flags[k] = 0;
        __syncthreads();
        // the read that holds g0, then every read boundary inside the loaded range forces a bucket start
        int r0;
        {
            int lo = 0, hi = n_reads;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (anchor_off[mid] <= g0) lo = mid; else hi = mid; }
            r0 = lo;
        }
        for (int r = r0 + tid; r <= n_reads; r += 256) {
            const int64_t d = anchor_off[r] - g0;
            if (d > cnt) break;
            if (d >= 0) atomicOr(&flags[d >> 6], 1ULL << (d & 63));
        }
        if (tid == 0 && g0 + cnt == n_a) atomicOr(&flags[cnt >> 6], 1ULL << (cnt & 63));
        // bin changes
        for (int i0 = wv * 64; i0 < cnt; i0 += 256) {
            const int i = i0 + lane;
            bool st = false;
            if (i < cnt) {
                const uint64_t prev = i > 0 ? kx[i - 1] : (g0 > 0 ? data[g0 - 1].x : ~kx[0]);
                st = anchor_bin(kx[i], bp) != anchor_bin(prev, bp);
            }
            const unsigned long long m = __ballot(st);
            if (lane == 0 && m) atomicOr(&flags[i0 >> 6], m);
        }
        __syncthreads();
        // every bucket start in the owned range is taken by the thread that owns its position
        for (int i = tid; i < own; i += 256) {
            if (!(flags[i >> 6] >> (i & 63) & 1)) continue;
            // end of the bucket: the next start within SMALL_BUCKET positions, else the bucket is a large one (not ours to sort)
            int e = -1;
            for (int q = i + 1; q <= i + SMALL_BUCKET && q <= cnt; ++q)
                if (flags[q >> 6] >> (q & 63) & 1) { e = q; break; }
            if (e < 0 || e - i <= 1) continue;
            for (int u = i + 1; u < e; ++u) {  // insertion sort in LDS
                const uint64_t vx = kx[u], vy = ky[u];
                int v = u - 1;
                while (v >= i && anchor_less(vx, vy, kx[v], ky[v])) { kx[v + 1] = kx[v]; ky[v + 1] = ky[v]; --v; }
                kx[v + 1] = vx; ky[v + 1] = vy;
            }
        }
        __syncthreads();
        // write back what this window owns: from its first bucket start to the first bucket start at or after SORT_WIN
        int lo = cnt, hi = cnt;
        {
            const int w_end = (cnt >> 6) + 1;
            for (int w = 0; w < w_end; ++w) if (flags[w]) { lo = (w << 6) + __builtin_ctzll(flags[w]); break; }
            if (own < cnt) {
                int w = own >> 6;
                unsigned long long m = flags[w] >> (own & 63) << (own & 63);
                while (!m && ++w < w_end) m = flags[w];
                if (m) hi = (w << 6) + __builtin_ctzll(m);
                if (hi > cnt) hi = cnt;
            }
            if (lo > own) lo = hi;  // no bucket starts in the owned range: nothing to write
        }
        for (int i = lo + tid; i < hi; i += 256) { u128 r; r.x = kx[i]; r.y = ky[i]; data[g0 + i] = r; }
        if (tid == 0 && hi > lo) add_work(work, (unsigned long long)(hi - lo));
        __syncthreads();
    }
}

// work-list segments of at most CAP anchors: one load, a bitonic network in LDS, one store
template <int CAP>
__global__ __launch_bounds__(256) void anchor_bitonic_list_kernel(u128 *__restrict__ data, const SortSeg *__restrict__ list,
                                                                  const unsigned int *__restrict__ n_list, unsigned int cap,
                                                                  unsigned long long *__restrict__ work) {
    __shared__ uint64_t kx[CAP], ky[CAP];
    const int tid = threadIdx.x;
    const unsigned int n_seg = *n_list < cap ? *n_list : cap;
    for (unsigned int seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const SortSeg sg = list[seg];
        const int len = (int)sg.len;
        u128 *a = data + sg.off;
        int N = 2;
        while (N < len) N <<= 1;
        for (int i = tid; i < N; i += 256) {
            if (i < len) { const u128 r = a[i]; kx[i] = r.x; ky[i] = r.y; }
            else { kx[i] = ~0ULL; ky[i] = ~0ULL; }
        }
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (N >> 1); t += 256) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const bool up = (i & k) == 0;
                    const uint64_t ax = kx[i], ay = ky[i], bx = kx[l], by = ky[l];
                    if (anchor_less(bx, by, ax, ay) == up) { kx[i] = bx; ky[i] = by; kx[l] = ax; ky[l] = ay; }
                }
                __syncthreads();
            }
        for (int i = tid; i < len; i += 256) { u128 r; r.x = kx[i]; r.y = ky[i]; a[i] = r; }
        if (tid == 0) add_work(work, (unsigned long long)len);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Chaining DP (minimap2 mm_chain_dp), one wavefront per read.  Anchor i is processed sequentially; its
// predecessors j = i-1, i-2, ... are evaluated 64 at a time across the lanes.  The sequential semantics of
// the max_skip heuristic (a counter that depends on the order in which predecessors are visited) are kept
// exactly: the running maximum comes from a DPP prefix-max, the "already on a better chain" marks of the
// current tile are exchanged through LDS, and the skip counter is replayed over the (few) event lanes.
struct ChainParams {
    int max_dist_x, max_dist_y, bw, max_skip, max_iter, min_cnt, min_sc;
};

__device__ __forceinline__ int ilog2_32(uint32_t v) { return 31 - __clz((int)v); }

// The last CW anchors (coordinates, f, p, v and the per-iteration mark t) are mirrored in an LDS ring and the anchors
// themselves are fetched / their results written back 64 at a time (one per lane, coalesced, next chunk prefetched),
// so the sequential loop touches global memory only for predecessors further back than CW.  The lower end of the
// predecessor range needs no state: a predecessor is in range iff its reference coordinate is within max_dist_x and it
// is at most max_iter anchors back, and the anchors are sorted.  Single-wave workgroup: LDS accesses of one wave are
// ordered, so only a compiler/LDS fence separates the phases (a full __syncthreads would also drain the global stores).
// CW = 128 (4 KB of LDS per wave): the kernel is latency-bound, its throughput is the number of resident waves; a
// 512-entry ring limited a CU to 9 waves and cost 25 % of the kernel's time, 64 entries gain nothing more.
constexpr int CHAIN_CW = 128;
#define MPN_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return (uint64_t)hi << 32 | lo;
}

// Chaining work items.  The predecessor scan of anchor i stops at the first j with x_i > x_j + max_dist_x, and the
// anchors of a read are sorted by x = (strand, target, position), so wherever two consecutive anchors are further
// apart than max_dist_x no chain can cross: the read's anchor list splits into independent SEGMENTS (one per target
// locus, typically; a stray seed hit is a segment of its own).  One wave per read (this kernel) cuts the list at segment
// boundaries into work items of at least CHAIN_ITEM anchors (whole segments only, so an item is chained exactly like a
// short read) and computes the read-wide average seed length the gap cost uses; the DP kernel takes items, not reads,
// from a shared queue -- the long pole of a batch becomes its longest locus instead of its longest read.  Items of at
// least CHAIN_BIG anchors are queued first.
constexpr int CHAIN_ITEM = 256, CHAIN_BIG = 4096;
struct ChainSeg { int32_t read, start, end; };

// ---------------------------------------------------------------------------------------------------------
// Anchor compaction.  A chain never crosses a gap of more than max_dist_x between consecutive sorted anchors (see above),
// and a chain of fewer than min_cnt anchors is discarded by the backtrack, so a segment with fewer than min_cnt anchors
// cannot contribute a chain: its anchors are dropped BEFORE the chaining DP.  Against a large target set most anchors
// are such strays (random k-mer hits, one per locus), so the DP, its f/p/t/v state, the chain-end scan and the backtrack
// shrink by an order of magnitude, with identical chains.
// The batch's anchor array is cut into pieces of COMPACT_PIECE anchors regardless of read boundaries; one wave per piece.
// Start-of-segment flags of the current and the next 64-anchor tile (a read's first anchor always starts a segment) give
// every anchor the saturated distances to its segment's start and end.
// WRITE = false: piece_kept[piece]; per read (atomics): anchors kept.   WRITE = true: kept anchors -> out, in order.
constexpr int COMPACT_PIECE = 4096;

template <bool WRITE>
__global__ __launch_bounds__(64) void anchor_compact_kernel(const u128 *__restrict__ anchors, const int64_t *__restrict__ anchor_off,
                                                            int n_reads, int64_t n_a, int max_dist_x, int min_cnt,
                                                            int64_t *__restrict__ piece_kept, unsigned long long *__restrict__ read_kept,
                                                            const int64_t *__restrict__ piece_off, u128 *__restrict__ out) {
    const int lane = threadIdx.x;
    const int K = min_cnt < 64 ? (min_cnt > 1 ? min_cnt : 1) : 64;
    const int64_t n_pieces = (n_a + COMPACT_PIECE - 1) / COMPACT_PIECE;
    for (int64_t piece = blockIdx.x; piece < n_pieces; piece += gridDim.x) {
        const int64_t g0 = piece * COMPACT_PIECE, g1 = g0 + COMPACT_PIECE < n_a ? g0 + COMPACT_PIECE : n_a;
        int r0;  // the read that holds anchor g0: last r with anchor_off[r] <= g0
        {
            int lo = 0, hi = n_reads;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (anchor_off[mid] <= g0) lo = mid; else hi = mid; }
            r0 = lo;
        }
        // distance from the start of g0's segment to g0, looking back at most K anchors inside its read
        int64_t since;
        {
            const int64_t rb = anchor_off[r0], j = g0 - 1 - lane;
            bool st = false;
            if (lane < K && j >= rb) st = j == rb || anchors[j].x > anchors[j - 1].x + (uint64_t)max_dist_x;
            const unsigned long long m = __ballot(st);
            since = m ? __builtin_ctzll(m) + 1 : K;
        }
        u128 cur{0, 0}, nxt{0, 0};
        bool s_cur = false, s_nxt = false;
        int r_cur = r0, r_nxt = r0;                             // per lane: the read of its anchor, with the read's bounds cached
        int64_t rb_cur = anchor_off[r0], re_cur = anchor_off[r0 + 1], rb_nxt, re_nxt;
        auto load = [&](int64_t t0, u128 &rec, bool &st, int &r, int64_t &rb, int64_t &re) {
            const int64_t i = t0 + lane;
            st = false;
            if (i < n_a) {
                if (i >= re) { do { ++r; re = anchor_off[r + 1]; } while (i >= re); rb = anchor_off[r]; }
                rec = anchors[i];
                st = i == rb || rec.x > anchors[i - 1].x + (uint64_t)max_dist_x;
            }
        };
        load(g0, cur, s_cur, r_cur, rb_cur, re_cur);
        int64_t run = 0;
        int acc_r = -1;                                         // count pass: totals of the read the wave is inside (uniform)
        unsigned long long acc_k = 0;
        for (int64_t t0 = g0; t0 < g1; t0 += 64) {
            const bool has_next = t0 + 64 < n_a;
            r_nxt = r_cur; rb_nxt = rb_cur; re_nxt = re_cur;
            if (has_next) load(t0 + 64, nxt, s_nxt, r_nxt, rb_nxt, re_nxt); else s_nxt = false;
            const unsigned long long m_cur = __ballot(s_cur), m_nxt = __ballot(s_nxt);
            const int64_t i = t0 + lane;
            const bool act = i < g1;
            bool keep = false;
            if (act) {
                const unsigned long long le = lane == 63 ? ~0ULL : (2ULL << lane) - 1;
                const unsigned long long below = m_cur & le, above = m_cur & ~le;
                const int64_t d_back = below ? lane - (63 - __builtin_clzll(below)) : since + lane;
                int64_t e;  // next segment start, saturated
                if (above) e = t0 + __builtin_ctzll(above);
                else if (!has_next) e = n_a;
                else if (m_nxt) e = t0 + 64 + __builtin_ctzll(m_nxt);
                else e = n_a < t0 + 128 ? n_a : t0 + 128;
                keep = d_back + (e - i) >= K;
            }
            const unsigned long long km = __ballot(keep);
            if (WRITE) {
                if (keep) out[piece_off[piece] + run + __popcll(km & ((1ULL << lane) - 1))] = cur;
            } else {
                const int r_first = __builtin_amdgcn_readfirstlane(r_cur);  // (lane 0 is active: t0 < g1)
                const bool uniform = __ballot(act && r_cur != r_first) == 0;
                if (!uniform || r_first != acc_r) {  // leaving the read the total belongs to: flush it (one atomic per run)
                    if (lane == 0 && acc_r >= 0 && acc_k) atomicAdd(&read_kept[acc_r], acc_k);
                    acc_r = -1; acc_k = 0;
                }
                if (uniform) {
                    acc_r = r_first; acc_k += (unsigned long long)__popcll(km);
                } else if (act && keep) {
                    atomicAdd(&read_kept[r_cur], 1ULL);
                }
            }
            run += __popcll(km);
            since = m_cur ? 64 - (63 - __builtin_clzll(m_cur)) : since + 64;
            cur = nxt; s_cur = s_nxt; r_cur = r_nxt; rb_cur = rb_nxt; re_cur = re_nxt;
        }
        if (!WRITE && lane == 0 && acc_r >= 0 && acc_k) atomicAdd(&read_kept[acc_r], acc_k);
        if (!WRITE && lane == 0) piece_kept[piece] = run;
    }
}

// per read: kept anchors as int64 (input of the offset scan) and the average seed length over ALL its anchors (n_full
// hits, seed lengths summed by seed_prefix_kernel: mm_chain_dp's average uses every anchor)
__global__ __launch_bounds__(256) void anchor_compact_finish_kernel(const int64_t *__restrict__ n_full, int n_reads,
                                                                    const unsigned long long *__restrict__ read_kept,
                                                                    const unsigned long long *__restrict__ span_sum,
                                                                    int64_t *__restrict__ kept, float *__restrict__ avg_qspan) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const int64_t n = n_full[r];
    kept[r] = (int64_t)read_kept[r];
    avg_qspan[r] = n > 0 ? (float)span_sum[r] / (float)n : 0.f;
}

__global__ __launch_bounds__(64) void chain_segments_kernel(const u128 *__restrict__ anchors, const int64_t *__restrict__ anchor_off,
                                                            int n_reads, ChainParams cp, float *__restrict__ avg_qspan, int have_avg,
                                                            ChainSeg *__restrict__ seg_big, ChainSeg *__restrict__ seg_small,
                                                            unsigned int *__restrict__ counters, int item_min) {
    const int lane = threadIdx.x;
    for (int read = blockIdx.x; read < n_reads; read += gridDim.x) {
        const int64_t base = anchor_off[read];
        const int64_t n = anchor_off[read + 1] - base;
        if (n == 0) continue;
        const u128 *a = anchors + base;
        unsigned long long sum = 0;
        int32_t open = 0;  // start of the segment being scanned (uniform)
        for (int64_t i0 = 0; i0 < n; i0 += 64) {
            const int64_t i = i0 + lane;
            bool start = false;
            if (i < n) {
                sum += a[i].y >> 32 & 0xff;
                start = i > 0 && a[i].x > a[i - 1].x + (uint64_t)cp.max_dist_x;
            }
            unsigned long long m = __ballot(start);
            if (lane == 0) {
                while (m) {
                    const int32_t s_new = (int32_t)i0 + __builtin_ctzll(m);
                    if (s_new - open >= item_min) {
                        const bool big = s_new - open >= CHAIN_BIG;
                        const unsigned int pos = atomicAdd(&counters[big ? 0 : 1], 1u);
                        (big ? seg_big : seg_small)[pos] = ChainSeg{read, open, s_new};
                        open = s_new;
                    }
                    m &= m - 1;
                }
            }
            open = __builtin_amdgcn_readfirstlane(open);
        }
        for (int d = 32; d; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) {
            if (!have_avg) avg_qspan[read] = (float)sum / (float)n;  // (given: the mean over the read's anchors before compaction)
            const bool big = (int32_t)n - open >= CHAIN_BIG;
            const unsigned int pos = atomicAdd(&counters[big ? 0 : 1], 1u);
            (big ? seg_big : seg_small)[pos] = ChainSeg{read, open, (int32_t)n};
        }
    }
}

// An anchor in the LDS ring: one ds_read_b128.  x is NOT the reference coordinate but its running sum with every step between
// consecutive anchors clamped to max_dist_x + 1 (mod 2^32): inside a segment -- where every step is at most max_dist_x -- differences
// are the true ones, across a segment boundary (another target, another strand) they exceed max_dist_x like the true ones; the
// predecessors an anchor can see are at most max_iter anchors back, so a difference never wraps.  The DP then runs on 32-bit numbers.
struct ChainSlot { uint32_t x; int32_t y, f, p; };

__global__ __launch_bounds__(64) void chain_dp_kernel(const u128 *__restrict__ anchors, const int64_t *__restrict__ anchor_off,
                                                      const float *__restrict__ avg_qspan_r, const ChainSeg *__restrict__ seg_big,
                                                      const ChainSeg *__restrict__ seg_small, unsigned int *__restrict__ counters,
                                                      ChainParams cp, int32_t *__restrict__ F, int32_t *__restrict__ P,
                                                      int32_t *__restrict__ T, int32_t *__restrict__ V) {
    __shared__ int mark[64];
    __shared__ __attribute__((aligned(16))) ChainSlot ring[CHAIN_CW];
    __shared__ int32_t wt[CHAIN_CW], wv[CHAIN_CW];
    constexpr int M = CHAIN_CW - 1;
    const int lane = threadIdx.x;
    const unsigned int n_big = counters[0], n_seg = n_big + counters[1];
    const uint32_t clamp_x = (uint32_t)cp.max_dist_x + 1u;
    const int32_t max_dq = cp.max_dist_y < cp.max_dist_x ? cp.max_dist_y : cp.max_dist_x;
    for (;;) {
        unsigned int sidx = 0;
        if (lane == 0) sidx = atomicAdd(&counters[2], 1u);  // shared queue: big segments first
        sidx = (unsigned int)__builtin_amdgcn_readfirstlane((int)sidx);
        if (sidx >= n_seg) break;
        const ChainSeg sg = sidx < n_big ? seg_big[sidx] : seg_small[sidx - n_big];
        // the segment is chained like a read of its own; predecessor indices are stored relative to the read
        const int64_t base = anchor_off[sg.read] + sg.start;
        const int32_t n = sg.end - sg.start;
        const int32_t rel = sg.start;
        const u128 *a = anchors + base;
        int32_t *f = F + base, *p = P + base, *t = T + base, *v = V + base;
        for (int32_t i = lane; i < n; i += 64) t[i] = 0;
        const float avg_qspan = avg_qspan_r[sg.read];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // chunk registers: lane l holds anchor c0 + l of the current chunk and of the next one; results of the chunk
        // (cx: the clamped running coordinate, cx64 the true one for predecessors that have left the ring)
        auto step_of = [&](int32_t k, uint64_t x) -> uint32_t {   // clamped step from anchor k - 1 to anchor k
            if (k == 0) return 0u;
            const uint64_t d = x - a[k - 1].x;
            return d > (uint64_t)clamp_x ? clamp_x : (uint32_t)d;
        };
        uint64_t cx64 = 0, nx64 = 0;
        uint32_t cx = 0, cyl = 0, cyh = 0, nstep = 0, nyl = 0, nyh = 0;
        if (lane < n) { const u128 r = a[lane]; cx64 = r.x; cyl = (uint32_t)r.y; cyh = (uint32_t)(r.y >> 32); cx = step_of(lane, r.x); }
        cx = (uint32_t)wave_scan_add((int)cx);
        int32_t rf = 0, rp = -1, rv = 0;
        for (int32_t i = 0; i < n; ++i) {
            const int li = i & 63;
            if (li == 0) {  // prefetch the next chunk
                const int32_t k = i + 64 + lane;
                nstep = 0;
                if (k < n) { const u128 r = a[k]; nx64 = r.x; nyl = (uint32_t)r.y; nyh = (uint32_t)(r.y >> 32); nstep = step_of(k, r.x); }
            }
            const uint32_t ri = (uint32_t)__builtin_amdgcn_readlane((int)cx, li);
            const int32_t qi = __builtin_amdgcn_readlane((int)cyl, li), q_span = __builtin_amdgcn_readlane((int)cyh, li) & 0xff;
            int32_t max_f = q_span, n_skip = 0, max_j = -1;
            const int32_t win_lo = i - CHAIN_CW;  // anchors j >= win_lo (and < i) are in the LDS ring
            bool done = false;
            for (int32_t j0 = i - 1; j0 >= 0 && !done; j0 -= 64) {
                const int32_t j = j0 - lane;
                // A tile that lies in the LDS ring as a whole takes the short way with the marks below: every predecessor that
                // passes the filters marks its own predecessor in the ring BEFORE the marks are read.  A mark goes to a LATER
                // lane (an earlier anchor) only, so a lane still sees exactly the marks of the lanes before it; the marks of
                // lanes at or past the break land on lanes past the break, which are not evaluated, and a stale mark `i` means
                // nothing to the anchors after i.  Two LDS round trips per tile instead of four.
                const bool tile_in_ring = j0 - 63 >= win_lo;   // (uniform)
                bool in = j >= 0 && i - j <= cp.max_iter;
                int32_t pj, tj = 0, fj, dq;
                uint32_t dr;
                if (tile_in_ring) {
                    // (a lane with j < 0 reads a slot of the ring too: whatever it holds is discarded with `in`)
                    const uint4 sl = *reinterpret_cast<const uint4 *>(&ring[j & M]);
                    dr = ri - sl.x; dq = qi - (int32_t)sl.y; fj = (int32_t)sl.z; pj = (int32_t)sl.w;
                } else {
                    dr = ~0u; dq = 0; fj = 0; pj = -1;
                    const uint64_t ri64 = readlane_u64(cx64, li);
                    if (in) {
                        if (j >= win_lo) {
                            const uint4 sl = *reinterpret_cast<const uint4 *>(&ring[j & M]);
                            dr = ri - sl.x; dq = qi - (int32_t)sl.y; fj = (int32_t)sl.z; pj = (int32_t)sl.w; tj = wt[j & M];
                        } else {
                            const u128 r = a[j];
                            const uint64_t d64 = ri64 - r.x;   // (sorted: never negative)
                            dr = d64 > (uint64_t)clamp_x ? clamp_x : (uint32_t)d64;
                            dq = qi - (int32_t)r.y; pj = p[j]; pj = pj >= 0 ? pj - rel : pj; tj = t[j]; fj = f[j];
                        }
                    }
                }
                in = in && dr <= (uint32_t)cp.max_dist_x;   // out of range: so is everything before it
                const int32_t dd = (int32_t)dr > dq ? (int32_t)dr - dq : dq - (int32_t)dr;
                const bool cont = !(in && dr != 0u && dq > 0 && dq <= max_dq && dd <= cp.bw);
                int32_t sc;
                {
                    const int32_t min_d = dq < (int32_t)dr ? dq : (int32_t)dr;
                    const int32_t s = min_d > q_span ? q_span : min_d;
                    const int32_t log_dd = dd ? ilog2_32((uint32_t)dd) : 0;
                    const int32_t gap_cost = (int)((double)dd * .01 * (double)avg_qspan) + (log_dd >> 1);
                    sc = cont ? NEG_INF : s - gap_cost + fj;
                }
                if (!in) pj = -1;
                // the range ends inside (or right after) this tile if any lane fell out of it
                if (__ballot(!in)) done = true;
                // marks made by earlier lanes of this tile (p[j'] == my j)
                bool tmark, gmark = false;
                if (tile_in_ring) {
                    if (!cont && pj >= 0) {
                        if (pj >= win_lo) wt[pj & M] = i;
                        else { t[pj] = i; gmark = true; }
                    }
                    MPN_LDS_FENCE();
                    tmark = wt[j & M] == i && j >= 0;
                } else {
                    mark[lane] = 0;
                    MPN_LDS_FENCE();
                    if (!cont && pj >= 0 && j0 - pj < 64 && j0 - pj >= 0) mark[j0 - pj] = 1;
                    MPN_LDS_FENCE();
                    tmark = (tj == i) || mark[lane];
                }
                // running maximum before each lane (sequential order = lane order)
                const int incl = wave_scan_max(sc);
                const int excl_raw = wave_shr1(incl, NEG_INF);
                const int before = max(max_f, excl_raw);
                const bool newmax = !cont && sc > before;
                const bool skipev = !cont && !newmax && tmark;
                // minimap2's skip counter over the lanes in order -- a new maximum takes one off (not below 0), a marked predecessor
                // that does not improve adds one, and the walk breaks when it exceeds max_skip -- is the clamped running sum
                // s_l = max(s_{l-1} + d_l, 0) = Q_l + max(s_before, -min_{t<=l} Q_t) with Q the prefix sum of d: two wave scans
                // (a serial loop over the set lanes was half of this kernel's time)
                const int Qs = wave_scan_add(newmax ? -1 : skipev ? 1 : 0);
                const int minQ = -wave_scan_max(-Qs);
                const int skips = Qs + max(n_skip, -minQ);
                const unsigned long long over = __ballot(skipev && skips > cp.max_skip);
                const int brk = over ? __builtin_ctzll(over) : 64;
                if (!over) n_skip = __builtin_amdgcn_readlane(skips, 63);
                const bool elig = !cont && lane < brk;
                // the best eligible score is the running maximum at the last eligible lane: no second scan
                const int best = brk > 0 ? __builtin_amdgcn_readlane(incl, brk - 1) : NEG_INF;
                if (best > max_f) {
                    const unsigned long long who = __ballot(elig && sc == best);
                    const int wl = __builtin_ctzll(who);
                    max_f = best;
                    max_j = j0 - wl;
                }
                if (!tile_in_ring && elig && pj >= 0) {
                    if (pj >= win_lo) wt[pj & M] = i;
                    else { t[pj] = i; gmark = true; }
                }
                if (brk < 64) done = true;
                // a mark that went to global memory must be visible to the later tiles of this anchor
                if (__ballot(gmark)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                MPN_LDS_FENCE();
            }
            int32_t vprev = 0;
            if (max_j >= 0) vprev = max_j >= win_lo ? wv[max_j & M] : v[max_j];
            const int32_t vi = (max_j >= 0 && vprev > max_f) ? vprev : max_f;
            if (lane == li) { rf = max_f; rp = max_j >= 0 ? max_j + rel : -1; rv = vi; }
            if (lane == 0) {
                const int sl = i & M;
                *reinterpret_cast<uint4 *>(&ring[sl]) = make_uint4(ri, (uint32_t)qi, (uint32_t)max_f, (uint32_t)max_j);
                wv[sl] = vi; wt[sl] = 0;
            }
            MPN_LDS_FENCE();
            if (li == 63 || i == n - 1) {  // write the chunk back, coalesced; move to the prefetched chunk
                const int32_t k = (i & ~63) + lane;
                if (k <= i) { f[k] = rf; p[k] = rp; v[k] = rv; }
                const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)cx, 63);
                cx = (uint32_t)wave_scan_add((int)nstep) + carry; cx64 = nx64; cyl = nyl; cyh = nyh;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
    }
}

// chain ends: anchors nobody points to with peak score >= min_sc; each is walked back to its peak
__global__ __launch_bounds__(64) void chain_ends_kernel(const int64_t *__restrict__ anchor_off, int n_reads, ChainParams cp,
                                                        const int32_t *__restrict__ F, const int32_t *__restrict__ P,
                                                        int32_t *__restrict__ T, const int32_t *__restrict__ V,
                                                        uint64_t *__restrict__ U, int32_t *__restrict__ n_ends) {
    const int lane = threadIdx.x;
    for (int read = blockIdx.x; read < n_reads; read += gridDim.x) {
        const int64_t base = anchor_off[read];
        const int64_t n = anchor_off[read + 1] - base;
        const int32_t *f = F + base, *p = P + base, *v = V + base;
        int32_t *t = T + base;
        uint64_t *u = U + base;
        for (int64_t i = lane; i < n; i += 64) t[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        for (int64_t i = lane; i < n; i += 64) if (p[i] >= 0) t[p[i]] = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        int cnt = 0;
        for (int64_t i0 = 0; i0 < n; i0 += 64) {
            const int64_t i = i0 + lane;
            bool is_end = i < n && t[i] == 0 && v[i] >= cp.min_sc;
            uint64_t val = 0;
            if (is_end) {
                int64_t j = i;
                while (j >= 0 && f[j] < v[j]) j = p[j];
                if (j < 0) j = i;
                val = (uint64_t)(uint32_t)f[j] << 32 | (uint64_t)j;
            }
            const unsigned long long m = __ballot(is_end);
            if (is_end) u[cnt + __popcll(m & ((1ULL << lane) - 1))] = val;
            cnt += __popcll(m);
        }
        if (lane == 0) n_ends[read] = cnt;
        __syncthreads();
    }
}

// sort the chain ends of every read in DESCENDING (score, index) order: 64-bit keys, reuse of the radix
// machinery with a complemented key; one workgroup per read
__global__ __launch_bounds__(256) void chain_sort_ends_kernel(uint64_t *__restrict__ U, uint64_t *__restrict__ Utmp,
                                                              const int64_t *__restrict__ anchor_off,
                                                              const int32_t *__restrict__ n_ends, int n_reads) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t bins[256];
    __shared__ uint32_t wcnt[4][256];
    __shared__ int s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int read = blockIdx.x; read < n_reads; read += gridDim.x) {
        const int64_t base = anchor_off[read];
        const int n = n_ends[read];
        if (n < 2) continue;
        uint64_t *src = U + base, *dst = Utmp + base;
        for (int pass = 0; pass < 8; ++pass) {
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 256) atomicAdd(&hist[(uint32_t)(~src[i] >> (8 * pass)) & 0xff], 1u);
            __syncthreads();
            if (tid == 0) s_skip = 0;
            __syncthreads();
            if (hist[tid] == (uint32_t)n) s_skip = 1;
            __syncthreads();
            if (s_skip) { __syncthreads(); continue; }
            {
                uint32_t v = hist[tid];
                bins[tid] = v;
                __syncthreads();
                for (int d = 1; d < 256; d <<= 1) {
                    uint32_t a = tid >= d ? bins[tid - d] : 0;
                    __syncthreads();
                    bins[tid] += a;
                    __syncthreads();
                }
                uint32_t excl = bins[tid] - v;
                __syncthreads();
                bins[tid] = excl;
                __syncthreads();
            }
            for (int t0 = 0; t0 < n; t0 += 256) {
                const int i = t0 + tid;
                const bool act = i < n;
                uint64_t r = 0;
                uint32_t dg = 0;
                if (act) { r = src[i]; dg = (uint32_t)(~r >> (8 * pass)) & 0xff; }
                wcnt[0][tid] = 0; wcnt[1][tid] = 0; wcnt[2][tid] = 0; wcnt[3][tid] = 0;
                __syncthreads();
                unsigned long long same = __ballot(act);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    unsigned long long m = __ballot((dg >> b) & 1);
                    same &= ((dg >> b) & 1) ? m : ~m;
                }
                const uint32_t rank = __popcll(same & ((1ULL << lane) - 1));
                if (act && rank == 0) wcnt[wv][dg] = __popcll(same);
                __syncthreads();
                if (act) {
                    uint32_t o = bins[dg] + rank;
                    for (int w2 = 0; w2 < wv; ++w2) o += wcnt[w2][dg];
                    dst[o] = r;
                }
                __syncthreads();
                bins[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
                __syncthreads();
            }
            uint64_t *sw = src; src = dst; dst = sw;
            __threadfence_block();
            __syncthreads();
        }
        if (src != U + base) for (int i = tid; i < n; i += 256) U[base + i] = src[i];
        __syncthreads();
    }
}

constexpr int BT_PAR_MIN = 16;   // chain ends of a read from which the backtrack runs a lane per end

// backtrack from the best chain end down: an anchor belongs to one chain only.  One wave per read; with few chain ends the
// walk is sequential (lane 0) and the copy of the chained anchors parallel, with many a lane owns an end.
//   out: u[k] = score<<32|cnt (in U, compacted), chained anchors in B (chain by chain, forward order), n_chain, n_chained
__global__ __launch_bounds__(64) void chain_backtrack_kernel(const u128 *__restrict__ anchors, const int64_t *__restrict__ anchor_off,
                                                             int n_reads, ChainParams cp, const int32_t *__restrict__ F,
                                                             const int32_t *__restrict__ P, int32_t *__restrict__ T,
                                                             int32_t *__restrict__ V, uint64_t *__restrict__ U,
                                                             const int32_t *__restrict__ n_ends, u128 *__restrict__ B,
                                                             uint64_t *__restrict__ Uc, unsigned long long *__restrict__ used,
                                                             int64_t *__restrict__ u_pos, int64_t *__restrict__ b_pos,
                                                             int32_t *__restrict__ n_chain, int64_t *__restrict__ n_chained,
                                                             ChainRec *__restrict__ Rc, int par_min) {
    __shared__ int s_k;
    __shared__ unsigned long long s_up, s_bp;
    const int lane = threadIdx.x;
    for (int read = blockIdx.x; read < n_reads; read += gridDim.x) {
        const int64_t base = anchor_off[read];
        const int64_t n = anchor_off[read + 1] - base;
        const u128 *a = anchors + base;
        const int32_t *f = F + base, *p = P + base;
        int32_t *t = T + base, *v = V + base;
        uint64_t *u = U + base;
        const int n_u = n_ends[read];
        if (n_u >= par_min) {
            // Many chain ends (a strain-rich target set: a hundred loci per read): a lane per end instead of lane 0 for all.
            // An anchor belongs to the best-ranked end whose way back passes through it, i.e. to the smallest rank in the subtree
            // above it: every end walks back and leaves its rank with atomicMin, stopping where a better end has already been
            // (that end goes on, or stopped at a still better one: nothing further back can be this end's).  The chain of an
            // end is then the stretch of its way back that carries its own rank -- exactly what the ordered walk with "taken"
            // marks yields, including the anchors that ends discarded later keep taken.
            for (int64_t i = lane; i < n; i += 64) t[i] = 0x7fffffff;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            for (int e0 = 0; e0 < n_u; e0 += 64) {   // (ranks in order, 64 at a time: a better end is rarely overwritten)
                const int e = e0 + lane;
                if (e < n_u) {
                    int32_t j = (int32_t)u[e];
                    // (the predecessor is fetched beside the atomic, not after it: one memory round trip per step of the walk)
                    while (j >= 0) { const int32_t nj = p[j]; if (atomicMin(&t[j], e) < e) break; j = nj; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __syncthreads();
            // count and decide (v[e] = anchors of end e's chain, 0 if it does not survive) ...
            int k_run = 0, nv_run = 0;
            for (int e0 = 0; e0 < n_u; e0 += 64) {
                const int e = e0 + lane;
                int cnt = 0;
                if (e < n_u) {
                    const uint64_t ui = u[e];
                    int32_t j = p[(int32_t)ui];
                    cnt = 1;   // (the ordered walk visits its start even when a better end has taken it)
                    while (j >= 0) { const int32_t tj = t[j], nj = p[j]; if (tj != e) break; ++cnt; j = nj; }
                    const bool keep = cnt >= cp.min_cnt && (j < 0 || (int32_t)(ui >> 32) - f[j] >= cp.min_sc);
                    if (!keep) cnt = 0;
                    v[e] = cnt;
                }
                k_run += __builtin_amdgcn_readlane(wave_scan_add(cnt > 0 ? 1 : 0), 63);
                nv_run += __builtin_amdgcn_readlane(wave_scan_add(cnt), 63);
            }
            if (lane == 0) {
                s_k = k_run;
                n_chain[read] = k_run;
                n_chained[read] = nv_run;
                s_up = k_run ? atomicAdd(&used[0], (unsigned long long)k_run) : 0ULL;
                s_bp = nv_run ? atomicAdd(&used[1], (unsigned long long)nv_run) : 0ULL;
                u_pos[read] = (int64_t)s_up;
                b_pos[read] = (int64_t)s_bp;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
            __syncthreads();
            // ... then place: (score, count) words, records and anchors of the surviving chains in rank order, as the ordered
            // walk leaves them; the second walk copies the chain back to front and measures it on the way (mm_cal_fuzzy_len)
            u128 *b = B + s_bp;
            uint64_t *uc = Uc + s_up;
            ChainRec *rc = Rc + s_up;
            k_run = 0; nv_run = 0;
            for (int e0 = 0; e0 < n_u; e0 += 64) {
                const int e = e0 + lane;
                const int cnt = e < n_u ? v[e] : 0;
                const int incl = wave_scan_add(cnt), kin = wave_scan_add(cnt > 0 ? 1 : 0);
                if (cnt > 0) {
                    const int kidx = k_run + kin - 1, off = nv_run + incl - cnt;
                    const uint64_t ui = u[e];
                    int32_t j = (int32_t)ui;
                    u128 nxt = a[j];
                    const u128 last = nxt;
                    int ml = 0, bl = 0;
                    b[off + cnt - 1] = nxt;
                    for (int q = cnt - 2; q >= 0; --q) {
                        j = p[j];
                        const u128 cur = a[j];
                        b[off + q] = cur;
                        const int span = (int)(nxt.y >> 32 & 0xff);
                        const int tl = (int32_t)nxt.x - (int32_t)cur.x, ql = (int32_t)nxt.y - (int32_t)cur.y;
                        bl += tl > ql ? tl : ql;
                        ml += tl > span && ql > span ? span : tl < ql ? tl : ql;
                        nxt = cur;
                    }
                    const int span0 = (int)(nxt.y >> 32 & 0xff);
                    ml += span0; bl += span0;
                    j = p[j];   // where the chain stops: nothing, or an anchor a better end owns (its score is taken off)
                    const uint64_t sc = j < 0 ? ui >> 32 : (ui >> 32) - (uint64_t)f[j];
                    uc[kidx] = sc << 32 | (uint32_t)cnt;
                    rc[kidx] = ChainRec{nxt.x, nxt.y, last.x, last.y, ml, bl};
                }
                k_run += __builtin_amdgcn_readlane(kin, 63);
                nv_run += __builtin_amdgcn_readlane(incl, 63);
            }
            __syncthreads();
            continue;
        }
        for (int64_t i = lane; i < n; i += 64) t[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (lane == 0) {
            int n_v = 0, k = 0;
            for (int i = 0; i < n_u; ++i) {
                const int n_v0 = n_v, k0 = k;
                const uint64_t ui = u[i];
                int64_t j = (int32_t)ui;
                do { v[n_v++] = (int32_t)j; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
                if (j < 0) {
                    if (n_v - n_v0 >= cp.min_cnt) u[k++] = ui >> 32 << 32 | (uint32_t)(n_v - n_v0);
                } else if ((int32_t)(ui >> 32) - f[j] >= cp.min_sc) {
                    if (n_v - n_v0 >= cp.min_cnt) u[k++] = ((ui >> 32) - (uint64_t)f[j]) << 32 | (uint32_t)(n_v - n_v0);
                }
                if (k0 == k) n_v = n_v0;
            }
            s_k = k;
            n_chain[read] = k;
            n_chained[read] = n_v;
            // compact pools: only chains that survived travel to the host
            s_up = k ? atomicAdd(&used[0], (unsigned long long)k) : 0ULL;
            s_bp = n_v ? atomicAdd(&used[1], (unsigned long long)n_v) : 0ULL;
            u_pos[read] = (int64_t)s_up;
            b_pos[read] = (int64_t)s_bp;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        const int k = s_k;
        u128 *b = B + s_bp;
        uint64_t *uc = Uc + s_up;
        ChainRec *rc = Rc + s_up;
        int off = 0;
        for (int c = 0; c < k; ++c) {
            const int ni = (int32_t)u[c];
            // the copy also measures the chain (mm_cal_fuzzy_len): the host makes its hits from these records and never sees an anchor
            int ml = 0, bl = 0;
            for (int j = lane; j < ni; j += 64) {
                const u128 cur = a[v[off + (ni - j - 1)]];
                b[off + j] = cur;
                const int span = (int)(cur.y >> 32 & 0xff);
                if (j == 0) { ml += span; bl += span; }
                else {
                    const u128 prev = a[v[off + (ni - j)]];
                    const int tl = (int32_t)cur.x - (int32_t)prev.x, ql = (int32_t)cur.y - (int32_t)prev.y;
                    bl += tl > ql ? tl : ql;
                    ml += tl > span && ql > span ? span : tl < ql ? tl : ql;
                }
            }
            ml = __builtin_amdgcn_readlane(wave_scan_add(ml), 63);
            bl = __builtin_amdgcn_readlane(wave_scan_add(bl), 63);
            if (lane == 0) {
                const u128 first = a[v[off + ni - 1]], last = a[v[off]];
                rc[c] = ChainRec{first.x, first.y, last.x, last.y, ml, bl};
            }
            off += ni;
        }
        for (int c = lane; c < k; c += 64) uc[c] = u[c];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// GPU index build: minimizers (hash<<8|span, position) -> records (hash, position), MSD partition by the top
// bits of the hash into buckets, segmented radix sort of every bucket by (hash, position), then key/offset arrays.
__global__ __launch_bounds__(256) void idx_bucket_hist_kernel(const u128 *__restrict__ mz, int64_t n, int shift, unsigned long long *__restrict__ hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&hist[(mz[i].x >> 8) >> shift], 1ULL);
}

__global__ __launch_bounds__(256) void idx_bucket_scatter_kernel(const u128 *__restrict__ mz, int64_t n, int shift,
                                                                 unsigned long long *__restrict__ cursor, u128 *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        u128 r = mz[i];
        r.x >>= 8;  // drop the span: the key is the hash
        const unsigned long long p = atomicAdd(&cursor[r.x >> shift], 1ULL);
        out[p] = r;
    }
}

// boundaries of equal-key runs: per block of 2048 records the number of run starts
__global__ __launch_bounds__(256) void idx_flag_count_kernel(const u128 *__restrict__ rec, int64_t n, int64_t *__restrict__ block_cnt) {
    __shared__ int s[256];
    const int64_t base = (int64_t)blockIdx.x * 2048;
    int c = 0;
    for (int k = 0; k < 8; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n && (i == 0 || rec[i].x != rec[i - 1].x)) ++c;
    }
    s[threadIdx.x] = c;
    __syncthreads();
    for (int d = 128; d; d >>= 1) { if (threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(256) void idx_emit_kernel(const u128 *__restrict__ rec, int64_t n, const int64_t *__restrict__ block_off,
                                                       uint64_t *__restrict__ keys, int64_t *__restrict__ key_off,
                                                       uint64_t *__restrict__ pos) {
    __shared__ int s[256];
    const int64_t base = (int64_t)blockIdx.x * 2048;
    // thread t owns records base + t*8 .. +8 (contiguous, so that local order = global order)
    int flags = 0, c = 0;
    for (int k = 0; k < 8; ++k) {
        const int64_t i = base + (int64_t)threadIdx.x * 8 + k;
        if (i < n) {
            pos[i] = rec[i].y;
            if (i == 0 || rec[i].x != rec[i - 1].x) { flags |= 1 << k; ++c; }
        }
    }
    s[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int v = threadIdx.x >= d ? s[threadIdx.x - d] : 0;
        __syncthreads();
        s[threadIdx.x] += v;
        __syncthreads();
    }
    int64_t o = block_off[blockIdx.x] + s[threadIdx.x] - c;
    for (int k = 0; k < 8; ++k)
        if (flags >> k & 1) {
            const int64_t i = base + (int64_t)threadIdx.x * 8 + k;
            keys[o] = rec[i].x;
            key_off[o] = i;
            ++o;
        }
}

// ---------------------------------------------------------------------------------------------------------
// Targets already resident in HBM as ASCII -> 2 bits per base (16 bases per word) + the list of ambiguous-base runs.
// One lane per output word; a run start / end is a base whose N-ness differs from its predecessor's, so the lanes need
// one byte of context each and no scan.  Starts and ends are appended unordered (the host sorts the two short lists:
// the i-th smallest start pairs with the i-th smallest end).  counters[0|1] = number of starts | ends found (may exceed cap).
__global__ __launch_bounds__(256) void idx_pack2_kernel(const uint8_t *__restrict__ seqs, int64_t total, uint32_t *__restrict__ words,
                                                        int64_t n_words, unsigned long long *__restrict__ counters,
                                                        int64_t *__restrict__ starts, int64_t *__restrict__ ends, int64_t cap) {
    const bool aligned = (reinterpret_cast<uintptr_t>(seqs) & 15) == 0;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g0 = wi * 16;
        uint8_t b[16];
        if (aligned && g0 + 16 <= total) {
            const uint4 v = *reinterpret_cast<const uint4 *>(seqs + g0);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = (uint8_t)(q[k >> 2] >> (8 * (k & 3)));
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) b[k] = g0 + k < total ? seqs[g0 + k] : (uint8_t)'A';
        }
        bool prev_n = g0 > 0 && g0 - 1 < total && nt4_code(seqs[g0 - 1]) > 3;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t g = g0 + k;
            const int c = nt4_code(b[k]);
            const bool isn = g < total && c > 3;
            if (g < total && !isn) word |= (uint32_t)c << (2 * k);
            if (isn != prev_n && g <= total) {
                const unsigned long long p = atomicAdd(&counters[isn ? 0 : 1], 1ULL);
                if ((int64_t)p < cap) (isn ? starts : ends)[p] = g;
            }
            prev_n = isn;
        }
        words[wi] = word;
    }
}

// occurrence histogram of the index keys (mm_idx_cal_max_occ needs a quantile of it): bins 0..nbins-1, the last one open-ended
// (the key offsets are read from the lookup's (key, first position) pairs: the separate key / offset arrays of the build are
// released once that table exists)
__global__ __launch_bounds__(256) void idx_occ_hist_kernel(const u128 *__restrict__ kv, int64_t n_keys, int nbins,
                                                           unsigned long long *__restrict__ hist) {
    __shared__ unsigned int local[1024];
    for (int k = threadIdx.x; k < 1024; k += blockDim.x) local[k] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t occ = (int64_t)(kv[i + 1].y - kv[i].y);
        if (occ < 1024) atomicAdd(&local[(int)occ], 1u);
        else atomicAdd(&hist[occ < nbins - 1 ? occ : nbins - 1], 1ULL);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 1024; k += blockDim.x) if (local[k]) atomicAdd(&hist[k], (unsigned long long)local[k]);
}

}  // namespace mpn
