// Small shared device/host types of the mapper.
#pragma once
#include "mpn_common.h"

namespace mpn {

struct u128 { uint64_t x, y; };

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

}  // namespace mpn
