// Small shared device/host types of the mapper.
#pragma once
#include "mpn_common.h"

namespace mpn {

struct u128 { uint64_t x, y; };

// What the host needs of a chain to make a hit of it (minimap2's mm_reg1_t before the base-level extension): its first and
// last anchor and the approximate match / block lengths (mm_reg_set_coor + mm_cal_fuzzy_len).  The chained anchors themselves
// stay in HBM; chain_backtrack_kernel writes one record per surviving chain beside the (score, count) word.
struct ChainRec { uint64_t fx, fy, lx, ly; int32_t mlen, blen; };

// One surviving chain of a read on its way into the read's squeezed anchor list (minimap2's mm_squeeze_a after hit selection):
// cnt anchors from src (index into the chain stage's pool) to sq_off[read] + dst; flag: the first anchor gets SEED_LONG_JOIN.
struct SqueezeSeg { int64_t src; int32_t dst, cnt, read, flag; };

// The split of a z-dropped hit (mm_split_reg) as the host needs it: fuzzy lengths of both halves and the first anchor of the
// remainder.  Written by stitch_kernel for the hits it cuts.
struct SplitRec { uint64_t fx, fy, lx_left, ly_left; int32_t mlen_l, blen_l, mlen_r, blen_r; };

__device__ __forceinline__ int nt4_code(uint8_t c) {
    c |= 0x20;
    return c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : (c == 't' || c == 'u') ? 3 : 4;
}

// Target sequences in HBM: 2 bits per base (16 bases per 32-bit word, all targets concatenated) plus the sorted list
// of ambiguous-base runs [start, end) in concatenated coordinates (RefSeq assemblies hold few, long N runs).
struct RefView {
    const uint32_t *seq2;
    const int64_t *seq_off;   // per target: offset in concatenated coordinates
    const int64_t *nrun_s, *nrun_e;
    int32_t n_runs;
};

__device__ __forceinline__ int ref_code(const RefView &rv, int64_t g) {
    int c = (int)((rv.seq2[g >> 4] >> (2 * (int)(g & 15))) & 3u);
    if (rv.n_runs > 0) {  // last run with start <= g
        int lo = 0, hi = rv.n_runs;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (rv.nrun_s[mid] <= g) lo = mid + 1; else hi = mid; }
        if (lo > 0 && g < rv.nrun_e[lo - 1]) c = 4;
    }
    return c;
}

// Sequential reader of target codes for a lane that walks along a target (CIGAR walks): the current 16-base word and the
// ambiguous-run / run-free interval around the position stay in registers, so a step costs a shift instead of a word
// load plus a binary search.
struct RefCursor {
    const RefView &rv;
    int64_t widx = -1, run_s = 1, run_e = 0, clear_s = 1, clear_e = 0;
    uint32_t word = 0;
    __device__ __forceinline__ explicit RefCursor(const RefView &r) : rv(r) {}
    __device__ __forceinline__ int at(int64_t g) {
        const int64_t wi = g >> 4;
        if (wi != widx) { word = rv.seq2[wi]; widx = wi; }
        int c = (int)((word >> (2 * (int)(g & 15))) & 3u);
        if (rv.n_runs > 0) {
            if (g >= run_s && g < run_e) c = 4;
            else if (!(g >= clear_s && g < clear_e)) {
                int lo = 0, hi = rv.n_runs;  // last run with start <= g
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (rv.nrun_s[mid] <= g) lo = mid + 1; else hi = mid; }
                if (lo > 0 && g < rv.nrun_e[lo - 1]) { run_s = rv.nrun_s[lo - 1]; run_e = rv.nrun_e[lo - 1]; c = 4; }
                else { clear_s = lo > 0 ? rv.nrun_e[lo - 1] : INT64_MIN; clear_e = lo < rv.n_runs ? rv.nrun_s[lo] : INT64_MAX; }
            }
        }
        return c;
    }
};

// Sequential reader of read bases on the hit's strand (0..4 codes): four ASCII bytes per load.  The reads buffer is 4-byte
// aligned and padded (mpn_map.h), so the aligned word around any base may be read.
struct ReadCursor {
    const uint32_t *words;
    int64_t roff, widx = -1;
    int32_t rlen, rev;
    uint32_t word = 0;
    __device__ __forceinline__ ReadCursor(const uint8_t *reads, int64_t roff_, int32_t rlen_, int rev_)
        : words(reinterpret_cast<const uint32_t *>(reads)), roff(roff_), rlen(rlen_), rev(rev_) {}
    __device__ __forceinline__ int at(int x) {
        const int64_t a = roff + (rev ? rlen - 1 - x : x), wi = a >> 2;
        if (wi != widx) { word = words[wi]; widx = wi; }
        const int c = nt4_code((uint8_t)(word >> (8 * (int)(a & 3))));
        return rev ? (c < 4 ? 3 - c : 4) : c;
    }
};

// Several small regions zeroed by ONE launch (counters, cursors and flag arrays of a stage: a hipMemsetAsync each is a launch and a
// completion signal each).  Sizes in 4-byte words; block b works on region b % n.
struct ZeroList { uint32_t *p[8]; uint32_t words[8]; int n; };
static __global__ __launch_bounds__(256) void zero_regions_kernel(ZeroList z) {
    const int r = blockIdx.x % z.n, part = blockIdx.x / z.n, parts = gridDim.x / z.n;
    for (uint32_t i = part * 256 + threadIdx.x; i < z.words[r]; i += parts * 256) z.p[r][i] = 0;
}
inline void zero_list_push(ZeroList &z, void *p, size_t bytes) {
    if (!p || !bytes) return;
    z.p[z.n] = (uint32_t *)p; z.words[z.n] = (uint32_t)((bytes + 3) / 4); ++z.n;
}
inline hipError_t zero_regions(ZeroList &z, hipStream_t st) {
    if (z.n == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_regions_kernel, dim3((unsigned)z.n * 4), dim3(256), 0, st, z);
    return hipGetLastError();
}


}  // namespace mpn
