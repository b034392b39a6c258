// Shared helpers for libmpn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace mpn {

void set_error(const char *fmt, ...);

#define MPN_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            mpn::set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__, hipGetErrorString(_e)); \
            return -1;                                                                           \
        }                                                                                        \
    } while (0)

// ---- device buffer with RAII (host-side plumbing) --------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        MPN_HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
        return 0;
    }
    int zero(hipStream_t st) {
        if (n) MPN_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), st));
        return 0;
    }
    int upload(const T *h, size_t count, hipStream_t st) {
        if (alloc(count)) return -1;
        if (count) MPN_HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
        return 0;
    }
    int download(T *h, size_t count, hipStream_t st) const {
        if (count) MPN_HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, st));
        return 0;
    }
};

// ---- wave64 cross-lane primitives (DPP; no LDS traffic) ---------------------------------
constexpr int NEG_INF = -(1 << 28);

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}

// lane i receives lane i-1's value; lane 0 receives `fill`  (wave_shr:1)
__device__ __forceinline__ int wave_shr1(int v, int fill) { return dpp_mov<0x138, 0xf>(fill, v); }

// inclusive prefix max over the 64 lanes: row_shr 1,2,4,8 then row_bcast15 / row_bcast31
__device__ __forceinline__ int wave_scan_max(int v) {
    v = max(v, dpp_mov<0x111, 0xf>(NEG_INF, v));
    v = max(v, dpp_mov<0x112, 0xf>(NEG_INF, v));
    v = max(v, dpp_mov<0x114, 0xf>(NEG_INF, v));
    v = max(v, dpp_mov<0x118, 0xf>(NEG_INF, v));
    v = max(v, dpp_mov<0x142, 0xa>(NEG_INF, v));
    v = max(v, dpp_mov<0x143, 0xc>(NEG_INF, v));
    return v;
}
__device__ __forceinline__ int wave_reduce_max(int v) { return __builtin_amdgcn_readlane(wave_scan_max(v), 63); }
__device__ __forceinline__ int wave_reduce_min(int v) { return -wave_reduce_max(-v); }

}  // namespace mpn
