// Sorting and interval-union kernels behind include/mpn_abundance.h: the coordinate sort of BAM records
// (`samtools sort`, /root/reference/bin/lib/aligner.py:246-252) and the covered base pairs per assembly
// (`bedtools sort | merge` + sum, /root/reference/bin/megapath_nano.py:313-347).  Byte/integer work bound by HBM traffic:
// a record is 24 bytes and a radix pass reads and writes it once (48 B per record and pass; constant digits are skipped).
#include "mpn_common.h"
#include "../../include/mpn_abundance.h"

#include <vector>

namespace mpn {

struct Rec3 { uint64_t hi, lo; int64_t idx; };

__device__ __forceinline__ uint32_t rec_digit(const Rec3 &r, int pass) {   // pass 0..7: bytes of lo, 8..15: bytes of hi
    return (uint32_t)((pass < 8 ? r.lo >> (8 * pass) : r.hi >> (8 * (pass - 8))) & 0xff);
}

constexpr int RS_THREADS = 256;

// per block: histogram of its chunk's digits -> hist[digit * n_blocks + block]; and which digits vary at all (any_diff)
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const Rec3 *__restrict__ src, int64_t n, int64_t chunk, int pass,
                                                             uint32_t *__restrict__ hist, int n_blocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += RS_THREADS) atomicAdd(&h[rec_digit(src[i], pass)], 1u);
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// one block: exclusive scan of the digit-major table (256 * n_blocks entries); flag[0] = 1 if one digit holds every record
__global__ __launch_bounds__(1024) void rs_scan_kernel(uint32_t *__restrict__ hist, int n_blocks, int64_t n, int *__restrict__ flag) {
    __shared__ unsigned long long part[1024];
    __shared__ int single;
    const int total = 256 * n_blocks, t = threadIdx.x, per = (total + 1023) / 1024, lo = min(total, t * per), hi = min(total, lo + per);
    if (t == 0) single = 0;
    __syncthreads();
    unsigned long long s = 0;
    for (int k = lo; k < hi; ++k) s += hist[k];
    part[t] = s;
    // a digit that holds all n records: its row of the table sums to n
    if (t < 256) { unsigned long long d = 0; for (int b = 0; b < n_blocks; ++b) d += hist[(size_t)t * n_blocks + b]; if ((int64_t)d == n) single = 1; }
    __syncthreads();
    if (t == 0) { unsigned long long acc = 0; for (int k = 0; k < 1024; ++k) { const unsigned long long v = part[k]; part[k] = acc; acc += v; } flag[0] = single; }
    __syncthreads();
    if (single) return;
    unsigned long long o = part[t];
    for (int k = lo; k < hi; ++k) { const uint32_t v = hist[k]; hist[k] = (uint32_t)o; o += v; }
}

// stable scatter: the block walks its chunk in order, 256 records at a time; a record's slot = running offset of its digit
// + its rank among the earlier records of the tile with the same digit (ballots inside a wave, per-wave counts across waves)
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const Rec3 *__restrict__ src, Rec3 *__restrict__ dst, int64_t n, int64_t chunk,
                                                                int pass, const uint32_t *__restrict__ offs, int n_blocks, const int *__restrict__ flag) {
    if (flag[0]) return;   // constant digit: the pass is skipped (the host keeps src as the current buffer)
    __shared__ uint32_t bins[256], wcnt[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    bins[tid] = offs[(size_t)tid * n_blocks + blockIdx.x];
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (int64_t t0 = lo; t0 < hi; t0 += RS_THREADS) {
        const int64_t i = t0 + tid;
        const bool act = i < hi;
        Rec3 r{0, 0, 0};
        uint32_t dg = 0;
        if (act) { r = src[i]; dg = rec_digit(r, pass); }
        wcnt[0][tid] = 0; wcnt[1][tid] = 0; wcnt[2][tid] = 0; wcnt[3][tid] = 0;
        __syncthreads();
        unsigned long long same = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((dg >> b) & 1);
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
}

__global__ __launch_bounds__(256) void rs_fill_kernel(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, int64_t n, Rec3 *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = Rec3{hi[i], lo[i], i};
}
__global__ __launch_bounds__(256) void rs_order_kernel(const Rec3 *__restrict__ rec, int64_t n, int64_t *__restrict__ order) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) order[i] = rec[i].idx;
}

// sorts d_a in place by (hi, lo), stable; d_b: bounce buffer of the same size.  Returns the buffer that holds the result.
static int radix_sort_rec3(Rec3 *d_a, Rec3 *d_b, int64_t n, Rec3 **result, hipStream_t st) {
    *result = d_a;
    if (n < 2) return 0;
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 4095) / 4096));
    const int64_t chunk = (n + n_blocks - 1) / n_blocks;
    DevBuf<uint32_t> hist;
    DevBuf<int> flag;
    if (hist.alloc((size_t)256 * n_blocks) || flag.alloc(1)) return -1;
    int *h_flag = nullptr;
    MPN_HIP_CHECK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocDefault));
    Rec3 *src = d_a, *dst = d_b;
    int rc = 0;
    for (int pass = 0; pass < 16 && !rc; ++pass) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3(n_blocks), dim3(RS_THREADS), 0, st, (const Rec3 *)src, n, chunk, pass, hist.p, n_blocks);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, hist.p, n_blocks, n, flag.p);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(n_blocks), dim3(RS_THREADS), 0, st, (const Rec3 *)src, dst, n, chunk, pass, (const uint32_t *)hist.p, n_blocks,
                           (const int *)flag.p);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h_flag, flag.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { set_error("radix sort pass %d failed", pass); rc = -1; break; }
        if (!*h_flag) { Rec3 *t = src; src = dst; dst = t; }
    }
    (void)hipHostFree(h_flag);
    *result = src;
    return rc;
}

// ---- union length of intervals per (group, sequence), summed per group ----------------------------------------------------
// records sorted by (hi = group << 32 | seq, lo = start << 32 | end).  A thread sweeps one chunk of the sorted list; the running
// interval that enters a chunk comes from a serial pass over the chunk summaries (phase 2), like a three-phase scan.
struct SweepSum { uint64_t first_key, last_key; uint32_t run_end; int32_t one_key; };

__global__ __launch_bounds__(256) void cov_phase1_kernel(const Rec3 *__restrict__ rec, int64_t n, int64_t chunk, int n_chunks, SweepSum *__restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int64_t lo = (int64_t)c * chunk, hi = lo + chunk < n ? lo + chunk : n;
    SweepSum s{0, 0, 0, 1};
    if (lo < hi) {
        s.first_key = rec[lo].hi;
        uint64_t key = s.first_key;
        uint32_t run_end = 0;
        for (int64_t i = lo; i < hi; ++i) {
            const Rec3 r = rec[i];
            if (r.hi != key) { key = r.hi; run_end = 0; s.one_key = 0; }
            const uint32_t e = (uint32_t)r.lo;
            run_end = e > run_end ? e : run_end;
        }
        s.last_key = key; s.run_end = run_end;
    }
    sums[c] = s;
}

// carry[c] = (key, running end) of the interval group that is open when chunk c starts (key = ~0: none)
__global__ void cov_phase2_kernel(const SweepSum *__restrict__ sums, int n_chunks, uint64_t *__restrict__ carry_key, uint32_t *__restrict__ carry_end) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t key = ~0ULL;
    uint32_t end = 0;
    for (int c = 0; c < n_chunks; ++c) {
        carry_key[c] = key; carry_end[c] = end;
        const SweepSum s = sums[c];
        if (s.one_key && s.first_key == key) end = s.run_end > end ? s.run_end : end;   // the open group runs through the whole chunk
        else { key = s.last_key; end = s.run_end; }
        // (one_key with another key, or several keys: the chunk's last group is the open one, with the chunk's own running end --
        //  unless that last group started before the chunk, which only happens in the one_key case handled above)
    }
}

__global__ __launch_bounds__(256) void cov_phase3_kernel(const Rec3 *__restrict__ rec, int64_t n, int64_t chunk, int n_chunks,
                                                         const uint64_t *__restrict__ carry_key, const uint32_t *__restrict__ carry_end,
                                                         unsigned long long *__restrict__ covered, int32_t n_groups) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int64_t lo = (int64_t)c * chunk, hi = lo + chunk < n ? lo + chunk : n;
    uint64_t key = carry_key[c];
    uint32_t run_end = carry_end[c];
    unsigned long long acc = 0;
    uint32_t acc_group = 0xffffffffu;
    for (int64_t i = lo; i < hi; ++i) {
        const Rec3 r = rec[i];
        const uint32_t s = (uint32_t)(r.lo >> 32), e = (uint32_t)r.lo, g = (uint32_t)(r.hi >> 32);
        if (g != acc_group) {
            if (acc && acc_group < (uint32_t)n_groups) atomicAdd(&covered[acc_group], acc);
            acc = 0; acc_group = g;
        }
        if (r.hi != key) { key = r.hi; run_end = 0; acc += e - s; run_end = e; continue; }
        // same (group, sequence): overlapping and book-ended intervals merge (start <= running end)
        if (s > run_end) acc += e - s;
        else if (e > run_end) acc += e - run_end;
        run_end = e > run_end ? e : run_end;
    }
    if (acc && acc_group < (uint32_t)n_groups) atomicAdd(&covered[acc_group], acc);
}

}  // namespace mpn

using namespace mpn;

extern "C" int mpn_sort_order(int64_t n, const uint64_t *hi, const uint64_t *lo, int64_t *order) {
    if (n < 0 || (n > 0 && (!hi || !lo || !order))) { set_error("mpn_sort_order: bad arguments"); return -2; }
    if (n == 0) return 0;
    hipStream_t st = 0;
    DevBuf<uint64_t> d_hi, d_lo;
    DevBuf<Rec3> a, b;
    DevBuf<int64_t> d_order;
    if (d_hi.upload(hi, (size_t)n, st) || d_lo.upload(lo, (size_t)n, st) || a.alloc((size_t)n) || b.alloc((size_t)n) || d_order.alloc((size_t)n)) return -1;
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16));
    hipLaunchKernelGGL(rs_fill_kernel, dim3(g), dim3(256), 0, st, (const uint64_t *)d_hi.p, (const uint64_t *)d_lo.p, n, a.p);
    Rec3 *res = nullptr;
    if (radix_sort_rec3(a.p, b.p, n, &res, st)) return -1;
    hipLaunchKernelGGL(rs_order_kernel, dim3(g), dim3(256), 0, st, (const Rec3 *)res, n, d_order.p);
    MPN_HIP_CHECK(hipGetLastError());
    if (d_order.download(order, (size_t)n, st)) return -1;
    MPN_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int mpn_cover_by_group(int64_t n, const int32_t *group, const int32_t *seq, const int64_t *start, const int64_t *end,
                                  int32_t n_groups, int64_t *covered) {
    if (n < 0 || n_groups < 0 || (n_groups > 0 && !covered) || (n > 0 && (!group || !seq || !start || !end))) { set_error("mpn_cover_by_group: bad arguments"); return -2; }
    for (int32_t g = 0; g < n_groups; ++g) covered[g] = 0;
    if (n == 0 || n_groups == 0) return 0;
    std::vector<uint64_t> hi((size_t)n), lo((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (group[i] < 0 || group[i] >= n_groups || seq[i] < 0 || start[i] < 0 || end[i] < start[i] || end[i] > 0xffffffffLL) {
            set_error("mpn_cover_by_group: record %lld outside the domain (group in [0, n_groups), seq >= 0, 0 <= start <= end < 2^32)", (long long)i);
            return -2;
        }
        hi[(size_t)i] = (uint64_t)(uint32_t)group[i] << 32 | (uint32_t)seq[i];
        lo[(size_t)i] = (uint64_t)start[i] << 32 | (uint64_t)end[i];
    }
    hipStream_t st = 0;
    DevBuf<uint64_t> d_hi, d_lo;
    DevBuf<Rec3> a, b;
    if (d_hi.upload(hi.data(), (size_t)n, st) || d_lo.upload(lo.data(), (size_t)n, st) || a.alloc((size_t)n) || b.alloc((size_t)n)) return -1;
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 256 * 16));
    hipLaunchKernelGGL(rs_fill_kernel, dim3(g), dim3(256), 0, st, (const uint64_t *)d_hi.p, (const uint64_t *)d_lo.p, n, a.p);
    Rec3 *res = nullptr;
    if (radix_sort_rec3(a.p, b.p, n, &res, st)) return -1;
    const int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(16384, (n + 255) / 256));
    const int64_t chunk = (n + n_chunks - 1) / n_chunks;
    DevBuf<SweepSum> sums;
    DevBuf<uint64_t> ck;
    DevBuf<uint32_t> ce;
    DevBuf<unsigned long long> d_cov;
    if (sums.alloc((size_t)n_chunks) || ck.alloc((size_t)n_chunks) || ce.alloc((size_t)n_chunks) || d_cov.alloc((size_t)n_groups) || d_cov.zero(st)) return -1;
    const int cg = (n_chunks + 255) / 256;
    hipLaunchKernelGGL(cov_phase1_kernel, dim3(cg), dim3(256), 0, st, (const Rec3 *)res, n, chunk, n_chunks, sums.p);
    hipLaunchKernelGGL(cov_phase2_kernel, dim3(1), dim3(64), 0, st, (const SweepSum *)sums.p, n_chunks, ck.p, ce.p);
    hipLaunchKernelGGL(cov_phase3_kernel, dim3(cg), dim3(256), 0, st, (const Rec3 *)res, n, chunk, n_chunks, (const uint64_t *)ck.p, (const uint32_t *)ce.p,
                       d_cov.p, n_groups);
    MPN_HIP_CHECK(hipGetLastError());
    if (d_cov.download((unsigned long long *)covered, (size_t)n_groups, st)) return -1;
    MPN_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}
