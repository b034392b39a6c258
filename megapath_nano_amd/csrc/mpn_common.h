// Shared helpers for libmpn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <algorithm>
#include <atomic>
#include <vector>

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

// ---- grow-only device arena ---------------------------------------------------------------
// hipMalloc/hipFree per batch cost more than the kernels they serve (and hipFree synchronises the device, which would
// serialise the pipelined workers).  A worker thread points tl_arena at its slot's arena; every DevBuf it creates then
// bump-allocates from chunks that persist across calls.  reset() rewinds; nothing is freed until process exit.
// Set by the calling thread's failed device allocation (arena, pool or plain buffer) when the device is out of memory: the
// pipelined mapper sheds a worker on it (align.hip) instead of failing the call.  A distinct flag, not a match on error text.
extern thread_local bool tl_oom;
// test hook: MPN_TEST_ARENA_BUDGET=<bytes> makes the arenas of all workers share that budget; an arena that would grow past it
// fails like a device that is out of memory (tests/test_map_e2e_gpu.py sheds workers with it)
extern std::atomic<long long> g_arena_bytes;
long long arena_test_budget();

// hipMalloc that leaves the device a reserve: the workers of the pipelined mapper grow their scratch until an allocation fails and
// then shed (align.hip) -- but a device that is filled to the last byte cannot serve the runtime's own needs either (kernel scratch,
// signals), and THAT failure aborts the process (HSA_STATUS_ERROR_OUT_OF_RESOURCES).  Large requests are therefore refused, as
// out of memory, while they would leave less than the reserve free.
inline hipError_t guarded_malloc(void **p, size_t bytes) {
    if (bytes >= ((size_t)32 << 20)) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t reserve = std::max<size_t>((size_t)4 << 30, total_b / 40);
            if (free_b < bytes + reserve) return hipErrorOutOfMemory;
        }
    }
    return hipMalloc(p, bytes);
}

struct Arena {
    struct Chunk { void *p; size_t cap; };
    std::vector<Chunk> chunks;
    size_t cur = 0, off = 0, used = 0;  // used: bytes handed out since the last reset (incl. alignment)
    // Rewind.  A pass that spilled over several chunks leaves holes the next pass may not be able to use (requests come in
    // a different size mix every sub-batch), so the chunks are then merged into ONE slab with head-room: after a few
    // sub-batches a worker owns a single slab and bump-allocates from it, and the footprint stops growing.  (hipFree
    // synchronises the device, which is why this happens here, between sub-batches, and only while the slab still grows.)
    void reset() {
        if (chunks.size() > 1) {
            size_t total = 0;
            for (const Chunk &c : chunks) { total += c.cap; (void)hipFree(c.p); }
            chunks.clear();
            g_arena_bytes -= (long long)total;
            const size_t want = std::max(used + used / 4, total / 2) + ((size_t)64 << 20);
            Chunk c{nullptr, want};
            const long long budget = arena_test_budget();
            if ((budget <= 0 || g_arena_bytes + (long long)want <= budget) && guarded_malloc(&c.p, c.cap) == hipSuccess) { chunks.push_back(c); g_arena_bytes += (long long)want; }
            else (void)hipGetLastError();  // take() will report the failure if the memory is really gone
        }
        cur = 0; off = 0; used = 0;
    }
    // what has been handed out up to here / back to there: the allocations made after mark() are dead (a worker maps one sub-batch
    // against several index parts and keeps the sub-batch's sketch, taken before the mark, across them)
    struct Mark { size_t cur, off, used; };
    Mark mark() const { return Mark{cur, off, used}; }
    void rewind(const Mark &m) { cur = m.cur; off = m.off; used = m.used; }
    // give everything back (a worker that leaves, or a slot that idles while the others need the memory)
    void release_all() {
        for (const Chunk &c : chunks) { (void)hipFree(c.p); g_arena_bytes -= (long long)c.cap; }
        chunks.clear();
        cur = off = used = 0;
    }
    void *take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        used += bytes;
        for (; cur < chunks.size(); ++cur, off = 0)
            if (off + bytes <= chunks[cur].cap) { void *r = (char *)chunks[cur].p + off; off += bytes; return r; }
        Chunk c;
        c.cap = bytes > ((size_t)1 << 30) ? bytes : ((size_t)1 << 30);
        const long long budget = arena_test_budget();
        if (budget > 0) c.cap = bytes > ((size_t)16 << 20) ? bytes : ((size_t)16 << 20);   // (small chunks, so that a small budget binds)
        if ((budget > 0 && g_arena_bytes + (long long)c.cap > budget) || guarded_malloc(&c.p, c.cap) != hipSuccess) {
            (void)hipGetLastError();
            tl_oom = true;
            mpn::set_error("arena: out of memory (hipMalloc of %zu bytes failed)", c.cap);
            return nullptr;
        }
        g_arena_bytes += (long long)c.cap;
        chunks.push_back(c);
        cur = chunks.size() - 1;
        off = bytes;
        return c.p;
    }
};
extern thread_local Arena *tl_arena;

// ---- device buffer with RAII (host-side plumbing) --------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        if (tl_arena) {
            p = (T *)tl_arena->take(count * sizeof(T));
            owned = false;
            return p ? 0 : -1;
        }
        const hipError_t e_ = hipMalloc((void **)&p, count * sizeof(T));
        if (e_ != hipSuccess) {
            if (e_ == hipErrorOutOfMemory) tl_oom = true;
            p = nullptr;
            mpn::set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e_));
            (void)hipGetLastError();
            return -1;
        }
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
// the same with lane 0 receiving 0: bound_ctrl supplies the zero, so no register has to be set up with the fill value
__device__ __forceinline__ int wave_shr1_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }

// inclusive prefix max over the 64 lanes: row_shr 1,2,4,8 then row_bcast15 / row_bcast31.  The lanes a shift does not reach
// take INT_MIN, the identity of the signed max: the compiler then folds each move into the max (one v_max_i32_dpp per step
// instead of a constant, a v_mov_dpp and a max).
__device__ __forceinline__ int wave_scan_max(int v) {
    constexpr int ID = -2147483647 - 1;
    v = max(v, dpp_mov<0x111, 0xf>(ID, v));
    v = max(v, dpp_mov<0x112, 0xf>(ID, v));
    v = max(v, dpp_mov<0x114, 0xf>(ID, v));
    v = max(v, dpp_mov<0x118, 0xf>(ID, v));
    v = max(v, dpp_mov<0x142, 0xa>(ID, v));
    v = max(v, dpp_mov<0x143, 0xc>(ID, v));
    return v;
}
// inclusive prefix sum over the 64 lanes (the same DPP ladder; lanes a shift does not reach add 0)
__device__ __forceinline__ int wave_scan_add(int v) {
    v += dpp_mov<0x111, 0xf>(0, v);
    v += dpp_mov<0x112, 0xf>(0, v);
    v += dpp_mov<0x114, 0xf>(0, v);
    v += dpp_mov<0x118, 0xf>(0, v);
    v += dpp_mov<0x142, 0xa>(0, v);
    v += dpp_mov<0x143, 0xc>(0, v);
    return v;
}
__device__ __forceinline__ int wave_reduce_max(int v) { return __builtin_amdgcn_readlane(wave_scan_max(v), 63); }
__device__ __forceinline__ int wave_reduce_min(int v) { return -wave_reduce_max(-v); }

}  // namespace mpn
