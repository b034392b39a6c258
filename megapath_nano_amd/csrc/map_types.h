// Small shared device/host types of the mapper.
#pragma once
#include "mpn_common.h"

namespace mpn {

struct u128 { uint64_t x, y; };

__device__ __forceinline__ int nt4_code(uint8_t c) {
    c |= 0x20;
    return c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : (c == 't' || c == 'u') ? 3 : 4;
}

}  // namespace mpn
