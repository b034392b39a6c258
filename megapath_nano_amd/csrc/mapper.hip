// Host orchestration of the mapper (index build, seed + chain stages) behind include/mpn_map.h.
#include "map_kernels.h"
#include "mapper_internal.h"
#include "../../include/mpn_map.h"

#include <algorithm>
#include <mutex>
#include <string.h>
#include <time.h>
#include <sys/stat.h>
#include <string>
#include <vector>

namespace mpn {

hipError_t stream_sync(hipStream_t st) {
    struct Ev {  // one event per host thread
        hipEvent_t e = nullptr;
        ~Ev() { if (e) (void)hipEventDestroy(e); }
    };
    static thread_local Ev ev;
    hipError_t rc;
    if (!ev.e && (rc = hipEventCreateWithFlags(&ev.e, hipEventDisableTiming)) != hipSuccess) return rc;
    if ((rc = hipEventRecord(ev.e, st)) != hipSuccess) return rc;
    // Poll with sleeps instead of hipEventSynchronize: on ROCm 7.2 even a blocking-sync event wait keeps the calling thread
    // on a core for most of the wait (measured: the 8 workers burnt ~4 CPU-seconds per 0.9 s step waiting for the GPU,
    // a quarter of the container's CPU quota, which the host phases of the other workers need).
    unsigned int us = 20;
    for (;;) {
        rc = hipEventQuery(ev.e);
        if (rc != hipErrorNotReady) return rc;
        timespec ts{0, (long)us * 1000};
        nanosleep(&ts, nullptr);
        if (us < 200) us += 20;
    }
}


// Small device -> host reads go through pinned memory of the calling thread: a copy into pageable memory is synchronous
// inside the runtime, which waits for the stream with the thread spinning on a core.
struct PinScratch {
    void *p = nullptr;
    size_t cap = 0;
    void *get(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return nullptr; }
            cap = want;
        }
        return p;
    }
    ~PinScratch() { if (p) (void)hipHostFree(p); }
};
static thread_local PinScratch tl_pin;

static int read_i64(const int64_t *d, int64_t *out, hipStream_t st) {
    int64_t *s = (int64_t *)tl_pin.get(64);
    if (!s) { set_error("pinned scratch allocation failed"); return -1; }
    MPN_HIP_CHECK(hipMemcpyAsync(s, d, 8, hipMemcpyDeviceToHost, st));
    MPN_HIP_CHECK(stream_sync(st));
    *out = *s;
    return 0;
}

thread_local int64_t g_stats[MPN_NSTATS] = {0};
PhaseLog g_phase_log;
bool g_worker_cpu_on = false;
std::atomic<long long> g_worker_cpu_ns[64];
thread_local int tl_worker_id = -1;

void pack_2bit(const uint8_t *codes, int64_t n, std::vector<uint32_t> &words, std::vector<int64_t> &ns, std::vector<int64_t> &ne) {
    words.assign((size_t)(n + 15) / 16 + 1, 0u);
    ns.clear(); ne.clear();
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t c = codes[i];
        if (c > 3) {
            if (!ne.empty() && ne.back() == i) ne.back() = i + 1;
            else { ns.push_back(i); ne.push_back(i + 1); }
        } else words[(size_t)(i >> 4)] |= (uint32_t)c << (2 * (int)(i & 15));
    }
}

static int grid_1d(int64_t n, int block, int cap = 256 * 16) {
    int64_t g = (n + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// sketch a batch that is already on the device; fills mz_off (device, n+1) and allocates mz.
// h_len: host copy of the sequence lengths (chunk table).
int sketch_device(const uint8_t *d_seqs, const int64_t *d_off, const int32_t *d_len, const int32_t *h_len, int n, int k, int w,
                  uint32_t rid_base, DevBuf<int64_t> &mz_off, DevBuf<u128> &mz, int64_t *n_mz, hipStream_t st, EvTimer *ev = nullptr) {
    const int C = 256;
    std::vector<int64_t> chunk_off((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) chunk_off[i + 1] = chunk_off[i] + (h_len[i] + C - 1) / C;
    const int64_t n_chunks = chunk_off[n];
    DevBuf<int64_t> cnt, d_chunk_off, slow_list;
    DevBuf<int32_t> chunk_cnt, chunk_rel;
    DevBuf<unsigned long long> n_slow, emask;
    DevBuf<uint8_t> cfast;
    DevBuf<int32_t> chunk_read;
    DevBuf<u128> stage;
    int64_t stage_chunks = 0;
    if (cnt.alloc((size_t)n + 1) || mz_off.alloc((size_t)n + 1) || d_chunk_off.upload(chunk_off.data(), (size_t)n + 1, st) ||
        chunk_cnt.alloc((size_t)n_chunks) || chunk_rel.alloc((size_t)n_chunks) || slow_list.alloc((size_t)n_chunks) || n_slow.alloc(1) ||
        emask.alloc((size_t)n_chunks * 4 + 4) || cfast.alloc((size_t)n_chunks + 1) || chunk_read.alloc((size_t)n_chunks + 1))
        return -1;
    hipLaunchKernelGGL(sketch_chunk_read_kernel, dim3(std::max(1, std::min((n + 3) / 4, 4096))), dim3(256), 0, st, (const int64_t *)d_chunk_off.p, n, chunk_read.p);
    {
        // the chunks that are irregular by their place in the sequence (its first chunk(s), its last one(s); all of them for
        // even k or a very wide window: the same rule as sketch_chunk_in_range) start the automaton kernel's list
        std::vector<int64_t> slow;
        const bool none_fast = !(k & 1) || w > SKETCH_FAST_MAX_W;
        for (int i = 0; i < n; ++i) {
            const int64_t nc = chunk_off[i + 1] - chunk_off[i];
            for (int64_t c = 0; c < nc; ++c) {
                const int64_t p0 = c * C, p1 = std::min<int64_t>(h_len[i], p0 + C);
                if (none_fast || p0 - 2 * w - k < 0 || p1 + w + 1 > h_len[i]) slow.push_back(chunk_off[i] + c);
                else if (p1 + C + w + 1 <= h_len[i]) c = std::max<int64_t>(c, (h_len[i] - w - 1) / C - 2);  // skip the regular middle
            }
        }
        const unsigned long long ns = slow.size();
        // staging for the automaton kernel's reports: the chunks listed here + some of those the fast kernel appends (ambiguous bases)
        stage_chunks = (int64_t)std::min<unsigned long long>((unsigned long long)n_chunks, ns + 2048);
        if (stage_chunks * SKETCH_STAGE_CAP * sizeof(u128) > ((size_t)1 << 30)) stage_chunks = ((size_t)1 << 30) / (SKETCH_STAGE_CAP * sizeof(u128));
        if (stage.alloc((size_t)stage_chunks * SKETCH_STAGE_CAP + 1)) return -1;
        if (ns) MPN_HIP_CHECK(hipMemcpyAsync(slow_list.p, slow.data(), ns * 8, hipMemcpyHostToDevice, st));
        MPN_HIP_CHECK(hipMemcpyAsync(n_slow.p, &ns, 8, hipMemcpyHostToDevice, st));
        MPN_HIP_CHECK(stream_sync(st));  // (the host vectors leave scope)
    }
    // regular chunks by the position-parallel kernel (a wave per chunk); it lists the others for the automaton kernel
    const size_t lds = (size_t)2 * w * 64 * sizeof(uint64_t);
    const unsigned fast_grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 3) / 4, 256 * 64));
    const unsigned slow_grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 63) / 64, 256 * 16));
    const bool hash64 = 2 * k > 32;
    auto pass = [&](bool fill, u128 *out) {
        if (n_chunks <= 0) return;
        const int64_t *moff = fill ? (const int64_t *)mz_off.p : nullptr;
        const int32_t *crel = fill ? (const int32_t *)chunk_rel.p : nullptr;
        int32_t *ccnt = fill ? nullptr : chunk_cnt.p;
        // (the count pass leaves the emit masks; the second pass of the regular chunks hashes the emitted positions only)
        if (fill) {
            if (hash64) hipLaunchKernelGGL(sketch_fill_kernel<true>, dim3(fast_grid), dim3(256), 0, st, d_seqs, d_off, d_len, n, (const int64_t *)d_chunk_off.p, (const int32_t *)chunk_read.p, n_chunks, C,
                                           w, k, moff, crel, out, rid_base, (const unsigned long long *)emask.p, (const uint8_t *)cfast.p);
            else hipLaunchKernelGGL(sketch_fill_kernel<false>, dim3(fast_grid), dim3(256), 0, st, d_seqs, d_off, d_len, n, (const int64_t *)d_chunk_off.p, (const int32_t *)chunk_read.p, n_chunks, C,
                                    w, k, moff, crel, out, rid_base, (const unsigned long long *)emask.p, (const uint8_t *)cfast.p);
        } else {
#define MPN_SKETCH_FAST(H, WCT) hipLaunchKernelGGL((sketch_fast_kernel<H, WCT>), dim3(fast_grid), dim3(256), 0, st, d_seqs, d_off, d_len, n, \
                                                  (const int64_t *)d_chunk_off.p, (const int32_t *)chunk_read.p, n_chunks, C, w, k, ccnt, slow_list.p, n_slow.p, emask.p, cfast.p)
            if (hash64) MPN_SKETCH_FAST(true, 0); else if (w == 10) MPN_SKETCH_FAST(false, 10); else MPN_SKETCH_FAST(false, 0);   // (-x map-ont: w = 10)
#undef MPN_SKETCH_FAST
        }
        if (fill) {
            // the staged chunks are copied into place; the automaton runs a second time only for the listed chunks beyond the staging
            hipLaunchKernelGGL(sketch_stage_copy_kernel, dim3(slow_grid), dim3(256), 0, st, (const int32_t *)chunk_read.p, (const int64_t *)slow_list.p,
                               (const unsigned long long *)n_slow.p, stage_chunks, moff, crel, (const int32_t *)chunk_cnt.p, (const u128 *)stage.p, out);
            if (stage_chunks < n_chunks)
                hipLaunchKernelGGL(sketch_chunk_kernel<true>, dim3(slow_grid), dim3(64), lds, st, d_seqs, d_off, d_len, n, (const int64_t *)d_chunk_off.p, (const int32_t *)chunk_read.p,
                                   (const int64_t *)slow_list.p, (const unsigned long long *)n_slow.p, C, w, k, moff, crel, ccnt, out, rid_base,
                                   (u128 *)nullptr, (int64_t)0, stage_chunks);
        } else hipLaunchKernelGGL(sketch_chunk_kernel<false>, dim3(slow_grid), dim3(64), lds, st, d_seqs, d_off, d_len, n, (const int64_t *)d_chunk_off.p, (const int32_t *)chunk_read.p,
                                  (const int64_t *)slow_list.p, (const unsigned long long *)n_slow.p, C, w, k, moff, crel, ccnt, out, rid_base,
                                  stage.p, stage_chunks, (int64_t)0);
    };
    if (ev) ev->skip();  // (host work above: the span of the count pass starts at its launch)
    pass(false, nullptr);
    MPN_HIP_CHECK(hipGetLastError());
    if (ev) ev->mark(10, 33);
    hipLaunchKernelGGL(sketch_chunk_prefix_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const int64_t *)d_chunk_off.p, n,
                       (const int32_t *)chunk_cnt.p, chunk_rel.p, cnt.p);
    MPN_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(1024), 0, st, cnt.p, mz_off.p, n);
    MPN_HIP_CHECK(hipGetLastError());
    int64_t total = 0;
    if (read_i64(mz_off.p + n, &total, st)) return -1;
    if (mz.alloc((size_t)total)) return -1;
    if (ev) ev->mark(10);
    if (total > 0) {
        pass(true, mz.p);
        MPN_HIP_CHECK(hipGetLastError());
        if (ev) ev->mark(10, 34);
    }
    *n_mz = total;
    return 0;
}

int upload_seqs(int32_t n, const char *seqs, const int64_t *seq_off, const int32_t *seq_len, DevBuf<uint8_t> &d_seqs,
                DevBuf<int64_t> &d_off, DevBuf<int32_t> &d_len, int64_t *total_bases, hipStream_t st) {
    int64_t extent = 0, bases = 0;
    for (int i = 0; i < n; ++i) {
        if (seq_len[i] < 0 || seq_off[i] < 0) { set_error("negative sequence length/offset at %d", i); return -2; }
        extent = std::max<int64_t>(extent, seq_off[i] + seq_len[i]);
        bases += seq_len[i];
    }
    if (d_seqs.alloc((size_t)extent + 16)) return -1;
    if (extent) MPN_HIP_CHECK(hipMemcpyAsync(d_seqs.p, seqs, (size_t)extent, hipMemcpyHostToDevice, st));
    if (d_off.upload(seq_off, n, st) || d_len.upload(seq_len, n, st)) return -1;
    *total_bases = bases;
    return 0;
}

// the sketch of a batch of reads on its own (shared by the index parts a sub-batch is mapped against)
int sketch_reads(int k, int w, int n, const uint8_t *d_seqs, const int64_t *d_off, const int32_t *d_len, const int32_t *h_len, ReadSketch &sk,
                 hipStream_t st) {
    EvTimer ev(st);
    sk.valid = false;
    if (sketch_device(d_seqs, d_off, d_len, h_len, n, k, w, 0, sk.mz_off, sk.mz, &sk.n_mz, st, &ev)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    ev.resolve();
    sk.k = k; sk.w = w; sk.valid = true;
    return 0;
}

// seeds -> sorted anchors -> chains for a batch resident on the device; pre: the batch's sketch if the caller has it (same k, w)
int seed_chain_device(const mpn_index *idx, const mpn_map_opt *opt, int n, const uint8_t *d_seqs, const int64_t *d_off,
                      const int32_t *d_len, const int32_t *h_len, SeedChainOut &o, hipStream_t st, const ReadSketch *pre) {
    int64_t n_mz = 0;
    DevBuf<int64_t> mz_off;
    DevBuf<u128> mz;
    EvTimer ev(st);
    if (pre && pre->valid && pre->k == idx->k && pre->w == idx->w) {
        // (views of the caller's buffers: nothing is copied, nothing is released here)
        mz_off.p = pre->mz_off.p; mz_off.n = pre->mz_off.n; mz_off.owned = false;
        mz.p = pre->mz.p; mz.n = pre->mz.n; mz.owned = false;
        n_mz = pre->n_mz;
    } else if (sketch_device(d_seqs, d_off, d_len, h_len, n, idx->k, idx->w, 0, mz_off, mz, &n_mz, st, &ev)) return -1;
    g_stats[1] += n_mz;
    int32_t mid_occ = opt->mid_occ > 0 ? opt->mid_occ : mpn_index_mid_occ(idx, opt->mid_occ_frac);
    if (mid_occ > (1 << 25)) {
        // `-f 0` (no cut-off, valid in minimap2) arrives as INT32_MAX: nothing occurs more often than the index's most frequent
        // minimizer, so that count + 1 is the same cut-off
        const int32_t top = mpn_index_mid_occ(idx, 1e-30f);
        if (top > 0) mid_occ = std::min(mid_occ, top);
    }
    DevBuf<int32_t> occ;
    DevBuf<int64_t> pos_start, rel_off, full_off, n_blk, blk_base;
    DevBuf<unsigned long long> span_sum;
    {   // layout of the read-back block (download_chains copies it in one piece): five int64[N] tables, two int32[N], the counters
        const size_t N = (size_t)n, o1 = (N + 1) * 8, o2 = o1 + N * 8, o3 = o2 + N * 8, o4 = o3 + N * 8, o5 = o4 + N * 4, o6 = (o5 + N * 4 + 7) & ~(size_t)7;
        o.tables_bytes = o6 + (size_t)(2 + WORK_SLOTS) * 8;
        if (o.tables.alloc(o.tables_bytes)) return -1;
        auto view = [&](auto &buf, size_t off, size_t count) { buf.release(); buf.p = reinterpret_cast<decltype(buf.p)>(o.tables.p + off); buf.n = count; buf.owned = false; };
        view(o.n_anchor, 0, N + 1); view(o.n_chained, o1, N); view(o.u_pos, o2, N); view(o.b_pos, o3, N);
        view(o.n_chain, o4, N); view(o.rep_len, o5, N); view(o.used, o6, 2 + WORK_SLOTS);
    }
    if (occ.alloc(n_mz) || pos_start.alloc(n_mz) || rel_off.alloc(n_mz) || full_off.alloc((size_t)n + 1) ||
        n_blk.alloc((size_t)n + 1) || blk_base.alloc((size_t)n + 1) || span_sum.alloc(n) || o.anchor_off.alloc((size_t)n + 1))
        return -1;
    if (n_mz > 0) {
        hipLaunchKernelGGL(seed_lookup_kernel, dim3(grid_1d(n_mz, 256)), dim3(256), 0, st, (const u128 *)idx->kv.p,
                           idx->n_keys, (const int64_t *)idx->bucket_start.p, idx->bucket_shift, mz.p, n_mz, mid_occ, occ.p, pos_start.p);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(11, 35);
    }
    hipLaunchKernelGGL(seed_prefix_kernel, dim3(std::max(1, std::min(n, 256 * 32))), dim3(64), 0, st, mz.p, mz_off.p, n, occ.p, rel_off.p,
                       o.n_anchor.p, o.rep_len.p, span_sum.p, n_blk.p);
    MPN_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(scan_i64x2_kernel, dim3(2), dim3(1024), 0, st, (const int64_t *)o.n_anchor.p, full_off.p, n, (const int64_t *)n_blk.p, blk_base.p, n);
    MPN_HIP_CHECK(hipGetLastError());
    int64_t n_full = 0;  // hits of the batch's minimizers; only those that pass the stray-hit filter become anchors
    if (read_i64(full_off.p + n, &n_full, st)) return -1;
    g_stats[2] += n_full;
    // stray-hit filter: keep words over the virtual hit slots + per-block totals, then the kept hits' offsets
    const int64_t nb_cap = n_mz / 64 + n + 1;
    DevBuf<unsigned long long> keep;
    DevBuf<int64_t> blk_kept, blk_off;
    DevBuf<int32_t> blk_read, order;
    DevBuf<unsigned int> next_read;
    // the counters, cursors and flag arrays of the whole stage are allocated here and zeroed by ONE launch (regions must be sized
    // in whole 4-byte words: all of them are arrays of 4- or 8-byte items)
    DevBuf<unsigned int> n_list, seg_counters;
    DevBuf<unsigned long long> read_kept;
    if (next_read.alloc(4) || order.alloc(n) || keep.alloc((size_t)(n_full / 64 + nb_cap + 8)) || blk_kept.alloc((size_t)nb_cap + 1) || blk_off.alloc((size_t)nb_cap + 2) ||
        blk_read.alloc((size_t)nb_cap + 1) || n_list.alloc(4) || seg_counters.alloc(4) || read_kept.alloc(n))
        return -1;
    {
        ZeroList z{};
        zero_list_push(z, next_read.p, 4 * sizeof(unsigned int));
        zero_list_push(z, blk_kept.p, ((size_t)nb_cap + 1) * 8);
        zero_list_push(z, o.used.p, o.used.n * sizeof(unsigned long long));
        zero_list_push(z, n_list.p, 4 * sizeof(unsigned int));
        zero_list_push(z, seg_counters.p, 4 * sizeof(unsigned int));
        zero_list_push(z, read_kept.p, (size_t)n * 8);
        MPN_HIP_CHECK(zero_regions(z, st));
    }
    if (nb_cap > 0x7fffffff) { set_error("sub-batch too large for the seed filter"); return -1; }
    if (mid_occ > (1 << 25)) { set_error("occurrence cut-off %d too large (a block of 64 minimizers must hold fewer than 2^32 hits)", mid_occ); return -1; }
    if (n_full > 0) {
        FilterParams fp;
        {
            const int T = std::max(1, std::min(opt->min_cnt, 3));
            const int64_t need = 2 * (int64_t)(T - 1) * std::max(opt->max_gap, 1);
            int sh = 1;
            while (sh < 32 && ((int64_t)1 << sh) < need) ++sh;
            fp.shift = sh; fp.half = (uint32_t)((uint64_t)1 << (sh - 1)); fp.level = T - 1;
            // second round for the reads with at least MPN_FLT_ROUND2 hits.  Off by default: at 20000 it emits 660 M anchors per
            // bench step instead of 1609 M, but the filter takes 30 % longer and the sort stage only 16 % less: 127 -> 121 Gbp/min
            // (profiles/r03/r03_filter_round2.txt)
            static const int64_t round2 = []() { const char *e = getenv("MPN_FLT_ROUND2"); return e ? atoll(e) : INT64_MAX; }();
            fp.second_round = round2;
        }
        // (set on every call and checked: the call is a table update, and a failure must not be hidden behind a later "invalid value")
        MPN_HIP_CHECK(hipFuncSetAttribute((const void *)seed_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLT_LDS_BYTES));
        hipLaunchKernelGGL(seed_order_kernel, dim3(1), dim3(1024), 0, st, (const int64_t *)o.n_anchor.p, n, order.p);
        static const int flt_wgs = []() { const char *e = getenv("MPN_FLT_WGS"); return e ? std::max(1, atoi(e)) : 256; }();
        hipLaunchKernelGGL(seed_filter_kernel, dim3(std::max(1, std::min(n, flt_wgs))), dim3(FLT_THREADS), FLT_LDS_BYTES, st, mz.p, mz_off.p, n, occ.p,
                           pos_start.p, rel_off.p, idx->pos.p, (const int64_t *)full_off.p, (const int64_t *)blk_base.p, (const int32_t *)order.p,
                           fp, keep.p, blk_kept.p, blk_read.p, next_read.p);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(11, 50);
    }
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(1024), 0, st, blk_kept.p, blk_off.p, (int)nb_cap);
    hipLaunchKernelGGL(seed_read_off_kernel, dim3((n + 256) / 256), dim3(256), 0, st, (const int64_t *)blk_off.p, (const int64_t *)blk_base.p, n,
                       o.anchor_off.p);
    MPN_HIP_CHECK(hipGetLastError());
    int64_t n_a = 0;
    if (read_i64(o.anchor_off.p + n, &n_a, st)) return -1;
    o.n_anchors = n_a;
    g_stats[51] += n_a;
    DevBuf<u128> tmp;
    if (o.anchors.alloc(n_a) || tmp.alloc(n_a) || o.n_ends.alloc(n)) return -1;
    ev.mark(11);
    if (n_a > 0) {
        // kept hits in (minimizer, hit) order -> tmp; partition per read on the top key bits -> o.anchors; small buckets are
        // sorted in LDS, the large ones (true loci) by radix passes with tmp as the bounce buffer
        hipLaunchKernelGGL(seed_emit_kernel, dim3(grid_1d(nb_cap, 4, 256 * 64)), dim3(256), 0, st, mz.p, mz_off.p, occ.p, pos_start.p, rel_off.p,
                           idx->pos.p, (const int64_t *)full_off.p, (const int64_t *)blk_base.p, n, (const unsigned long long *)keep.p,
                           (const int64_t *)blk_kept.p, (const int64_t *)blk_off.p, (const int32_t *)blk_read.p, d_len, tmp.p);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(11, 36);
        BinParams bp;
        {
            int32_t max_len = 1;
            for (int32_t l : idx->lens) max_len = std::max(max_len, l);
            int rid_bits = 0, pos_bits = 1;
            while (((int64_t)1 << rid_bits) < (int64_t)idx->n_seq) ++rid_bits;
            while (((int64_t)1 << pos_bits) < (int64_t)max_len) ++pos_bits;
            bp.pos_bits = pos_bits;
            bp.shift = std::max(0, rid_bits + pos_bits - (MSD_BITS - 1));
        }
        // work lists of the buckets that are not sorted in place by the chunk kernel (more than SMALL_BUCKET anchors each)
        SortLists lists;
        DevBuf<SortSeg> list_mem;
        {
            const int64_t c0 = n_a / (SMALL_BUCKET + 1) + (int64_t)n + 16, c1 = n_a / (BITONIC_SMALL + 1) + 16, c2 = n_a / (BITONIC_MID + 1) + 16;
            if (c0 > 0x7fffffff) { set_error("sub-batch too large for the anchor sort"); return -1; }
            if (list_mem.alloc((size_t)(c0 + c1 + c2))) return -1;
            lists.seg[0] = list_mem.p; lists.seg[1] = list_mem.p + c0; lists.seg[2] = list_mem.p + c0 + c1;
            lists.cap[0] = (unsigned int)c0; lists.cap[1] = (unsigned int)c1; lists.cap[2] = (unsigned int)c2;
            lists.count = n_list.p;
        }
        const size_t msd_lds = (size_t)(MSD_NB + 32) * sizeof(uint32_t);
        MPN_HIP_CHECK(hipFuncSetAttribute((const void *)anchor_msd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)msd_lds));
        const int gr = std::max(1, std::min(n, 256 * 4));
        hipLaunchKernelGGL(anchor_msd_kernel, dim3(gr), dim3(MSD_THREADS), msd_lds, st, (const u128 *)tmp.p, o.anchors.p, (const int64_t *)o.anchor_off.p, n, bp,
                           lists, o.used.p + 2);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(12, 47);
        const int64_t n_win = (n_a + SORT_WIN - 1) / SORT_WIN;
        hipLaunchKernelGGL(anchor_window_sort_kernel, dim3((unsigned)std::min<int64_t>(n_win, 256 * 64)), dim3(256), 0, st, o.anchors.p,
                           (const int64_t *)o.anchor_off.p, n, n_a, bp, o.used.p + 2);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(12, 48);
        hipLaunchKernelGGL(anchor_bitonic_list_kernel<BITONIC_SMALL>, dim3(256 * 16), dim3(256), 0, st, o.anchors.p, (const SortSeg *)lists.seg[0],
                           (const unsigned int *)n_list.p, lists.cap[0], o.used.p + 2);
        hipLaunchKernelGGL(anchor_bitonic_list_kernel<BITONIC_MID>, dim3(256 * 4), dim3(256), 0, st, o.anchors.p, (const SortSeg *)lists.seg[1],
                           (const unsigned int *)n_list.p + 1, lists.cap[1], o.used.p + 2);
        hipLaunchKernelGGL(seg_sort_list_kernel<4>, dim3(256 * 4), dim3(256), 0, st, o.anchors.p, tmp.p, (const SortSeg *)lists.seg[2],
                           (const unsigned int *)n_list.p + 2, lists.cap[2], o.used.p + 2);
        MPN_HIP_CHECK(hipGetLastError());
        ev.mark(12, 49);
    }
    ChainParams cp;
    cp.max_dist_x = opt->max_gap; cp.max_dist_y = opt->max_gap; cp.bw = opt->bw; cp.max_skip = opt->max_chain_skip;
    cp.max_iter = opt->max_chain_iter; cp.min_cnt = opt->min_cnt; cp.min_sc = opt->min_chain_score;
    const int g = std::max(1, std::min(n, 256 * 32));
    // anchors of segments too short to chain are dropped; everything below runs on the compact list (c_off, tmp)
    const int64_t n_pieces = (n_a + COMPACT_PIECE - 1) / COMPACT_PIECE;
    DevBuf<int64_t> kept, piece_kept, piece_off;
    DevBuf<float> avg_qspan;
    if (kept.alloc((size_t)n + 1) || o.c_off.alloc((size_t)n + 1) || avg_qspan.alloc(n) || piece_kept.alloc((size_t)n_pieces + 1) ||
        piece_off.alloc((size_t)n_pieces + 1))
        return -1;
    const int gp = (int)std::max<int64_t>(1, std::min<int64_t>(n_pieces, 256 * 64));
    if (n_a > 0) {
        hipLaunchKernelGGL(anchor_compact_kernel<false>, dim3(gp), dim3(64), 0, st, (const u128 *)o.anchors.p, (const int64_t *)o.anchor_off.p, n, n_a,
                           cp.max_dist_x, cp.min_cnt, piece_kept.p, read_kept.p, (const int64_t *)nullptr, (u128 *)nullptr);
        MPN_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(anchor_compact_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const int64_t *)o.n_anchor.p, n,
                       (const unsigned long long *)read_kept.p, (const unsigned long long *)span_sum.p, kept.p, avg_qspan.p);
    hipLaunchKernelGGL(scan_i64x2_kernel, dim3(2), dim3(1024), 0, st, (const int64_t *)kept.p, o.c_off.p, n, (const int64_t *)piece_kept.p, piece_off.p, (int)n_pieces);
    MPN_HIP_CHECK(hipGetLastError());
    int64_t n_c = 0;
    if (read_i64(o.c_off.p + n, &n_c, st)) return -1;
    g_stats[45] += n_c;
    DevBuf<int32_t> F, P, T, V;
    DevBuf<uint64_t> Utmp;
    // (a surviving chain holds at least min_cnt anchors)
    if (F.alloc(n_c) || P.alloc(n_c) || T.alloc(n_c) || V.alloc(n_c) || o.u.alloc(n_c) || Utmp.alloc(n_c) || o.chained.alloc(n_c) ||
        o.recs.alloc((size_t)(n_c / std::max(1, opt->min_cnt)) + 1))
        return -1;
    u128 *ca = tmp.p;  // the sort's bounce buffer is free now: it receives the compact anchors
    if (n_a > 0) {
        hipLaunchKernelGGL(anchor_compact_kernel<true>, dim3(gp), dim3(64), 0, st, (const u128 *)o.anchors.p, (const int64_t *)o.anchor_off.p, n, n_a,
                           cp.max_dist_x, cp.min_cnt, (int64_t *)nullptr, (unsigned long long *)nullptr, (const int64_t *)piece_off.p, ca);
        MPN_HIP_CHECK(hipGetLastError());
    }
    ev.mark(46);
    // work items of the chain DP: runs of whole independent segments of each read's anchor list, cut on the device
    DevBuf<ChainSeg> seg_big, seg_small;
    static const int chain_item = []() { const char *e = getenv("MPN_CHAIN_ITEM"); return e ? std::min(CHAIN_BIG, std::max(16, atoi(e))) : CHAIN_ITEM; }();
    if (seg_big.alloc((size_t)n_c / CHAIN_BIG + (size_t)n + 1) || seg_small.alloc((size_t)n_c / chain_item + (size_t)n + 1)) return -1;
    hipLaunchKernelGGL(chain_segments_kernel, dim3(g), dim3(64), 0, st, (const u128 *)ca, (const int64_t *)o.c_off.p, n, cp, avg_qspan.p, 1, seg_big.p,
                       seg_small.p, seg_counters.p, chain_item);
    MPN_HIP_CHECK(hipGetLastError());
    ev.mark(13);
    hipLaunchKernelGGL(chain_dp_kernel, dim3(256 * 32), dim3(64), 0, st, (const u128 *)ca, (const int64_t *)o.c_off.p, (const float *)avg_qspan.p,
                       (const ChainSeg *)seg_big.p, (const ChainSeg *)seg_small.p, seg_counters.p, cp, F.p, P.p, T.p, V.p);
    MPN_HIP_CHECK(hipGetLastError());
    ev.mark(13, 37);
    hipLaunchKernelGGL(chain_ends_kernel, dim3(g), dim3(64), 0, st, (const int64_t *)o.c_off.p, n, cp, F.p, P.p, T.p, V.p, o.u.p, o.n_ends.p);
    MPN_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(chain_sort_ends_kernel, dim3(std::max(1, std::min(n, 256 * 8))), dim3(256), 0, st, o.u.p, Utmp.p,
                       (const int64_t *)o.c_off.p, o.n_ends.p, n);
    MPN_HIP_CHECK(hipGetLastError());
    // chained anchors and surviving chains go to compact pools
    o.u_compact.p = Utmp.p; o.u_compact.n = Utmp.n; o.u_compact.owned = Utmp.owned; Utmp.p = nullptr; Utmp.n = 0;
    static const int bt_par_min = []() { const char *e = getenv("MPN_BT_PAR_MIN"); return e ? std::max(1, atoi(e)) : BT_PAR_MIN; }();   // (tests force either path)
    hipLaunchKernelGGL(chain_backtrack_kernel, dim3(g), dim3(64), 0, st, (const u128 *)ca, (const int64_t *)o.c_off.p, n, cp, F.p, P.p, T.p, V.p,
                       o.u.p, o.n_ends.p, o.chained.p, o.u_compact.p, o.used.p, o.u_pos.p, o.b_pos.p, o.n_chain.p, o.n_chained.p, o.recs.p, bt_par_min);
    MPN_HIP_CHECK(hipGetLastError());
    ev.mark(14);
    MPN_HIP_CHECK(stream_sync(st));
    ev.resolve();
    return 0;
}

// download the per-read tables and the compact chain pools of a batch (pinned staging owned by the caller): the (score, count)
// words and the chain records; the chained anchors only for the stage test (with_anchors) -- the mapper leaves them in HBM
// mode 0: the per-read tables only (the hits are made on the GPU: hit_kernels.h); 1: + the (score, count) words and the chain
// records; 2: + the words and the chained anchors (stage test)
int download_chains(int n, SeedChainOut &o, HostChains &h, PoolBuf &pin_u, PoolBuf &pin_b, hipStream_t st, int mode) {
    const bool with_anchors = mode == 2;
    h.n_anchor.resize((size_t)n);
    h.n_chain.resize(n); h.n_chained.resize(n); h.rep_len.resize(n); h.u_pos.resize(n); h.b_pos.resize(n);
    unsigned long long used[2 + WORK_SLOTS] = {0};
    {   // the per-read tables: one device block, one pinned staging block, one copy, one wait
        const size_t N = (size_t)n;
        const size_t o1 = (N + 1) * 8, o2 = o1 + N * 8, o3 = o2 + N * 8, o4 = o3 + N * 8, o5 = o4 + N * 4, o6 = (o5 + N * 4 + 7) & ~(size_t)7;
        char *pn = (char *)tl_pin.get(o.tables_bytes + 64);
        if (!pn) { set_error("pinned scratch allocation failed"); return -1; }
        MPN_HIP_CHECK(hipMemcpyAsync(pn, o.tables.p, o.tables_bytes, hipMemcpyDeviceToHost, st));
        MPN_HIP_CHECK(stream_sync(st));
        memcpy(h.n_anchor.data(), pn, N * 8); memcpy(h.n_chained.data(), pn + o1, N * 8);
        memcpy(h.u_pos.data(), pn + o2, N * 8); memcpy(h.b_pos.data(), pn + o3, N * 8);
        memcpy(h.n_chain.data(), pn + o4, N * 4); memcpy(h.rep_len.data(), pn + o5, N * 4);
        memcpy(used, pn + o6, sizeof(used));
    }
    for (int k = 0; k < WORK_SLOTS; ++k) g_stats[44] += (int64_t)used[2 + k];
    ++g_stats[32];
    h.n_pool_chains = (int64_t)used[0];
    if (mode == 0) { h.u_all = nullptr; h.rec_all = nullptr; h.b_all = nullptr; }
    else if (with_anchors) {
        if (pin_u.ensure((size_t)used[0] * 8 + 16) || pin_b.ensure((size_t)used[1] * 16 + 16)) return -1;
        if (o.u_compact.download(pin_u.as<uint64_t>(), (size_t)used[0], st) || o.chained.download(pin_b.as<u128>(), (size_t)used[1], st)) return -1;
        h.u_all = pin_u.as<uint64_t>(); h.b_all = pin_b.as<u128>();
    } else {
        // [records | (score, count) words] in one staging block
        const size_t rec_bytes = ((size_t)used[0] * sizeof(ChainRec) + 15) & ~(size_t)15;
        if (pin_u.ensure(rec_bytes + (size_t)used[0] * 8 + 16)) return -1;
        h.rec_all = pin_u.as<ChainRec>();
        h.u_all = reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(pin_u.p) + rec_bytes);
        h.b_all = nullptr;
        if (o.recs.download(const_cast<ChainRec *>(h.rec_all), (size_t)used[0], st) || o.u_compact.download(const_cast<uint64_t *>(h.u_all), (size_t)used[0], st)) return -1;
    }
    h.chain_off.assign((size_t)n + 1, 0);
    h.b_off.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) { h.chain_off[i + 1] = h.chain_off[i] + h.n_chain[i]; h.b_off[i + 1] = h.b_off[i] + h.n_chained[i]; }
    MPN_HIP_CHECK(stream_sync(st));
    return 0;
}

void HostChains::read_chains(int i, uint64_t *uo, u128 *bo) const {
    const int nc = n_chain[i];
    if (nc == 0) return;
    const uint64_t *u = u_all + u_pos[i];
    const u128 *b = b_all + b_pos[i];
    std::vector<std::pair<std::pair<uint64_t, uint64_t>, int>> w(nc);
    int64_t k = 0;
    for (int c = 0; c < nc; ++c) { w[c] = {{b[k].x, (uint64_t)k << 32 | (uint32_t)c}, c}; k += (int32_t)u[c]; }
    std::sort(w.begin(), w.end());
    int64_t kk = 0;
    for (int c = 0; c < nc; ++c) {
        const int j = w[c].second;
        const int64_t start = (int64_t)(w[c].first.second >> 32);
        const int32_t cnt = (int32_t)u[j];
        uo[c] = u[j];
        memcpy(bo + kk, b + start, (size_t)cnt * sizeof(u128));
        kk += cnt;
    }
}

// the records and (score, count) words of a batch whose tables are already down (the reads the hit kernel left to the host)
int download_chain_records(SeedChainOut &o, HostChains &h, PoolBuf &pin_u, hipStream_t st) {
    const size_t nc = (size_t)h.n_pool_chains;
    const size_t rec_bytes = (nc * sizeof(ChainRec) + 15) & ~(size_t)15;
    if (pin_u.ensure(rec_bytes + nc * 8 + 16)) return -1;
    h.rec_all = pin_u.as<ChainRec>();
    h.u_all = reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(pin_u.p) + rec_bytes);
    if (o.recs.download(const_cast<ChainRec *>(h.rec_all), nc, st) || o.u_compact.download(const_cast<uint64_t *>(h.u_all), nc, st)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    return 0;
}

void HostChains::chain_order(int i, int32_t *order, int64_t *src) const {
    const int nc = n_chain[i];
    if (nc == 0) return;
    const uint64_t *u = u_all + u_pos[i];
    const ChainRec *r = rec_all + u_pos[i];
    // (the same key as read_chains: first anchor, then the chain's position in the pool)
    static thread_local std::vector<std::pair<std::pair<uint64_t, uint64_t>, int>> w;
    w.resize((size_t)nc);
    int64_t k = 0;
    for (int c = 0; c < nc; ++c) { w[c] = {{r[c].fx, (uint64_t)k << 32 | (uint32_t)c}, c}; k += (int32_t)u[c]; }
    std::sort(w.begin(), w.end());
    for (int c = 0; c < nc; ++c) { order[c] = w[c].second; src[c] = (int64_t)(w[c].first.second >> 32); }
}

}  // namespace mpn

using namespace mpn;

extern "C" {

void mpn_map_opt_init(mpn_map_opt *o) {
    memset(o, 0, sizeof(*o));
    o->k = 15; o->w = 10;
    o->mid_occ_frac = 2e-4f;
    o->min_cnt = 3; o->min_chain_score = 40; o->bw = 500; o->max_gap = 5000;
    o->max_chain_skip = 25; o->max_chain_iter = 5000;
    o->mask_level = 0.5f; o->pri_ratio = 0.8f; o->best_n = 5;
    o->max_join_long = 20000; o->max_join_short = 2000; o->min_join_flank_sc = 1000; o->min_join_flank_ratio = 0.5f;
    o->a = 2; o->b = 4; o->q = 4; o->e = 2; o->q2 = 24; o->e2 = 1;
    o->sc_ambi = 1; o->zdrop = 400; o->zdrop_inv = 200; o->end_bonus = -1;
    o->min_dp_max = o->min_chain_score * o->a;
    o->min_ksw_len = 200;
    o->max_clip_ratio = 1.0f;
    o->max_sw_mat = 100000000;
    o->with_cigar = 1;
    o->seed = 11;
    o->host_threads = 0;
}

static int build_bucket_table(mpn_index *idx, hipStream_t st);

// 2-bit packed targets + ambiguous-base runs from the ASCII targets resident in HBM; keeps the host copy too
static int pack_targets_device(mpn_index *idx, const uint8_t *d_seqs, int64_t total, hipStream_t st) {
    const int64_t n_words = (total + 15) / 16 + 1;
    DevBuf<unsigned long long> cnt;
    DevBuf<int64_t> starts, ends;
    if (idx->d_seq2.alloc((size_t)n_words) || cnt.alloc(2)) return -1;
    std::vector<int64_t> ns, ne;
    for (int64_t cap = 1 << 16;;) {
        if (cnt.zero(st) || starts.alloc((size_t)cap) || ends.alloc((size_t)cap)) return -1;
        hipLaunchKernelGGL(idx_pack2_kernel, dim3(grid_1d(n_words, 256, 256 * 64)), dim3(256), 0, st, d_seqs, total, idx->d_seq2.p, n_words,
                           cnt.p, starts.p, ends.p, cap);
        MPN_HIP_CHECK(hipGetLastError());
        unsigned long long h_cnt[2] = {0, 0};
        if (cnt.download(h_cnt, 2, st)) return -1;
        MPN_HIP_CHECK(stream_sync(st));
        if (h_cnt[0] != h_cnt[1]) { set_error("index build: unbalanced ambiguous-base runs"); return -1; }
        if ((int64_t)h_cnt[0] > cap) { cap = (int64_t)h_cnt[0] + 16; continue; }
        ns.resize((size_t)h_cnt[0]); ne.resize((size_t)h_cnt[0]);
        if (starts.download(ns.data(), ns.size(), st) || ends.download(ne.data(), ne.size(), st)) return -1;
        MPN_HIP_CHECK(stream_sync(st));
        break;
    }
    std::sort(ns.begin(), ns.end());
    std::sort(ne.begin(), ne.end());
    idx->n_nruns = (int32_t)ns.size();
    // (no host copy of the packed targets: since the CIGAR fix-up moved to the GPU nothing on the host reads target bases;
    // mpn_index_save fetches the words when it needs them)
    if (idx->d_nrun_s.upload(ns.data(), ns.size(), st) || idx->d_nrun_e.upload(ne.data(), ne.size(), st)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    idx->h_nrun_s.swap(ns); idx->h_nrun_e.swap(ne);
    return 0;
}

// The whole build runs on the GPU from targets that are resident in HBM as concatenated ASCII (d_seqs, off[n_seq+1] on the
// host): 2-bit packing, sketch, MSD bucket partition on the top bits of the hash, one workgroup per bucket for the
// (hash, position) radix sort, key/offset/position arrays.  Peak HBM ~ 6 bytes per target base beside the input
// (minimizer records twice while they are partitioned); the resident index is ~ 2.2 bytes per base.
static int build_index_device(mpn_index *idx, const uint8_t *d_seqs, const std::vector<int64_t> &off, const int32_t *lens, hipStream_t st) {
    const int n_seq = idx->n_seq, k = idx->k, w = idx->w;
    const int64_t total = off[(size_t)n_seq];
    if (pack_targets_device(idx, d_seqs, total, st)) return -1;
    DevBuf<int64_t> d_off;
    DevBuf<int32_t> d_len;
    if (d_off.upload(off.data(), (size_t)n_seq, st) || d_len.upload(lens, (size_t)n_seq, st)) return -1;
    DevBuf<u128> rec;
    int64_t n_mz = 0;
    const int hbits = 2 * k;
    int bbits = 8, shift = 0, nb = 0;
    DevBuf<int64_t> bucket_off;
    std::vector<int64_t> h_bucket_off;
    {
        DevBuf<int64_t> mz_off;
        DevBuf<u128> mz;
        if (sketch_device(d_seqs, d_off.p, d_len.p, lens, n_seq, k, w, 0, mz_off, mz, &n_mz, st)) return -1;
        idx->n_mz = n_mz;
        // buckets of ~64k records: a bucket is sorted by one workgroup, and there should be many more buckets than CUs
        while (bbits < 16 && ((int64_t)1 << (bbits + 16)) < n_mz) ++bbits;
        bbits = std::min(bbits, hbits);
        shift = hbits - bbits; nb = 1 << bbits;
        DevBuf<unsigned long long> hist, cursor;
        if (hist.alloc((size_t)nb) || hist.zero(st) || cursor.alloc((size_t)nb + 1) || bucket_off.alloc((size_t)nb + 1) || rec.alloc((size_t)n_mz))
            return -1;
        const int g = grid_1d(n_mz, 256);
        if (n_mz > 0) hipLaunchKernelGGL(idx_bucket_hist_kernel, dim3(g), dim3(256), 0, st, mz.p, n_mz, shift, hist.p);
        hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(1024), 0, st, (const int64_t *)hist.p, bucket_off.p, nb);
        MPN_HIP_CHECK(hipMemcpyAsync(cursor.p, bucket_off.p, ((size_t)nb + 1) * 8, hipMemcpyDeviceToDevice, st));
        if (n_mz > 0) hipLaunchKernelGGL(idx_bucket_scatter_kernel, dim3(g), dim3(256), 0, st, mz.p, n_mz, shift, cursor.p, rec.p);
        MPN_HIP_CHECK(hipGetLastError());
        h_bucket_off.resize((size_t)nb + 1);
        if (bucket_off.download(h_bucket_off.data(), (size_t)nb + 1, st)) return -1;
        MPN_HIP_CHECK(stream_sync(st));
    }   // the unsorted minimizers are released here
    if (n_mz > 0) {
        // sort the buckets in groups that share one bounce buffer of ~1/8 of the records
        int64_t big = 0;
        for (int b = 0; b < nb; ++b) big = std::max(big, h_bucket_off[(size_t)b + 1] - h_bucket_off[(size_t)b]);
        const int64_t tmp_cap = std::max<int64_t>(big, (n_mz + 7) / 8);
        DevBuf<u128> tmp;
        if (tmp.alloc((size_t)tmp_cap)) return -1;
        for (int b0 = 0; b0 < nb;) {
            int b1 = b0 + 1;
            while (b1 < nb && h_bucket_off[(size_t)b1 + 1] - h_bucket_off[(size_t)b0] <= tmp_cap) ++b1;
            if (h_bucket_off[(size_t)b1] > h_bucket_off[(size_t)b0])
                hipLaunchKernelGGL(seg_sort_kernel<8>, dim3(std::min(b1 - b0, 256 * 32)), dim3(256), 0, st, rec.p, tmp.p - h_bucket_off[(size_t)b0],
                                   (const int64_t *)bucket_off.p + b0, b1 - b0, (unsigned long long *)nullptr);
            b0 = b1;
        }
        MPN_HIP_CHECK(hipGetLastError());
        MPN_HIP_CHECK(stream_sync(st));
    }
    const int64_t n_blocks = (n_mz + 2047) / 2048;
    DevBuf<int64_t> block_cnt, block_off;
    if (block_cnt.alloc((size_t)n_blocks + 1) || block_off.alloc((size_t)n_blocks + 1)) return -1;
    if (n_blocks > 0) hipLaunchKernelGGL(idx_flag_count_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, rec.p, n_mz, block_cnt.p);
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(1024), 0, st, block_cnt.p, block_off.p, (int)n_blocks);
    int64_t n_keys = 0;
    MPN_HIP_CHECK(hipMemcpyAsync(&n_keys, block_off.p + n_blocks, 8, hipMemcpyDeviceToHost, st));
    MPN_HIP_CHECK(stream_sync(st));
    idx->n_keys = n_keys;
    if (idx->keys.alloc((size_t)n_keys) || idx->key_off.alloc((size_t)n_keys + 1) || idx->pos.alloc((size_t)n_mz)) return -1;
    if (n_blocks > 0) hipLaunchKernelGGL(idx_emit_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, rec.p, n_mz, block_off.p, idx->keys.p,
                                         idx->key_off.p, idx->pos.p);
    MPN_HIP_CHECK(hipMemcpyAsync(idx->key_off.p + n_keys, &n_mz, 8, hipMemcpyHostToDevice, st));
    MPN_HIP_CHECK(stream_sync(st));
    MPN_HIP_CHECK(hipGetLastError());
    if (build_bucket_table(idx, st) || idx->d_seq_off.upload(off.data(), off.size(), st) || idx->d_lens.upload(idx->lens.data(), idx->lens.size(), st)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    return 0;
}

static mpn_index *index_shell(int32_t n_seq, const char *const *names, const int32_t *lens, int32_t k, int32_t w) {
    if (n_seq <= 0 || !names || !lens || k < 1 || k > 28 || w < 1 || w > 255) { set_error("mpn_index_build: bad arguments"); return nullptr; }
    mpn_index *idx = new mpn_index();
    idx->k = k; idx->w = w; idx->n_seq = n_seq;
    idx->seq_off.assign((size_t)n_seq + 1, 0);
    for (int i = 0; i < n_seq; ++i) {
        if (lens[i] < 0) { set_error("mpn_index_build: negative length of target %d", i); delete idx; return nullptr; }
        idx->names.push_back(names[i] ? names[i] : "*");
        idx->lens.push_back(lens[i]);
        idx->seq_off[(size_t)i + 1] = idx->seq_off[(size_t)i] + lens[i];
    }
    return idx;
}

mpn_index *mpn_index_build(int32_t n_seq, const char *const *names, const char *const *seqs, const int32_t *lens,
                           int32_t k, int32_t w) {
    if (!seqs) { set_error("mpn_index_build: bad arguments"); return nullptr; }
    mpn_index *idx = index_shell(n_seq, names, lens, k, w);
    if (!idx) return nullptr;
    hipStream_t st = 0;
    const int64_t total = idx->seq_off.back();
    // stage the targets through a bounded pinned buffer: sequence by sequence, no second host copy of the whole set
    DevBuf<uint8_t> d_seqs;
    if (d_seqs.alloc((size_t)total + 16)) { delete idx; return nullptr; }
    for (int i = 0; i < n_seq; ++i)
        if (lens[i] > 0 && hipMemcpyAsync(d_seqs.p + idx->seq_off[(size_t)i], seqs[i], (size_t)lens[i], hipMemcpyHostToDevice, st) != hipSuccess) {
            set_error("mpn_index_build: upload of target %d failed", i);
            delete idx;
            return nullptr;
        }
    if (stream_sync(st) != hipSuccess || build_index_device(idx, d_seqs.p, idx->seq_off, lens, st)) { delete idx; return nullptr; }
    return idx;
}

mpn_index *mpn_index_build_device(int32_t n_seq, const char *const *names, const void *d_seqs, const int64_t *seq_off, const int32_t *lens,
                                  int32_t k, int32_t w) {
    if (!d_seqs || !seq_off) { set_error("mpn_index_build_device: bad arguments"); return nullptr; }
    mpn_index *idx = index_shell(n_seq, names, lens, k, w);
    if (!idx) return nullptr;
    for (int i = 0; i < n_seq; ++i)
        if (seq_off[i] != idx->seq_off[(size_t)i]) { set_error("mpn_index_build_device: targets must be concatenated without gaps (target %d)", i); delete idx; return nullptr; }
    if (build_index_device(idx, (const uint8_t *)d_seqs, idx->seq_off, lens, 0)) { delete idx; return nullptr; }
    return idx;
}

// bucket table over the sorted keys (see seed_lookup_kernel); rebuilt from the keys, so it is not part of the file format
// and the lookup's own copy of the keys: (key, first position) pairs in one array, so that the binary search inside a bucket
// and the hit range it ends on share a cache line.  Buckets are sized for ~4 keys (one 64-byte line): a lookup touches the
// bucket table and one or two lines of pairs instead of three dependent arrays.
static int build_bucket_table(mpn_index *idx, hipStream_t st) {
    int want = 8;
    while (want < 26 && ((int64_t)1 << (want + 2)) < idx->n_keys) ++want;
    const int hbits = 2 * idx->k, bbits = std::min(want, hbits);
    idx->bucket_shift = hbits - bbits;
    const int64_t nb = (int64_t)1 << bbits;
    if (idx->bucket_start.alloc((size_t)nb + 1) || idx->kv.alloc((size_t)idx->n_keys + 1)) return -1;
    hipLaunchKernelGGL(idx_bucket_table_kernel, dim3(grid_1d(idx->n_keys + 1, 256)), dim3(256), 0, st, idx->keys.p, idx->n_keys,
                       idx->bucket_shift, nb, idx->bucket_start.p);
    hipLaunchKernelGGL(idx_kv_kernel, dim3(grid_1d(idx->n_keys + 1, 256)), dim3(256), 0, st, idx->keys.p, idx->key_off.p, idx->n_keys, idx->kv.p);
    MPN_HIP_CHECK(hipGetLastError());
    // the pairs now hold both arrays (16 B per key; 4 GB at the 256 M keys of a 20 Gbp target set): save, export and the
    // occurrence histogram read them from there
    MPN_HIP_CHECK(stream_sync(st));
    idx->keys.release(); idx->key_off.release();
    return 0;
}

// host copies of the sorted keys and their offsets (n_keys + 1 entries), split out of the lookup's pairs
static int fetch_keys(const mpn_index *idx, uint64_t *keys, int64_t *key_off, hipStream_t st) {
    const size_t n = (size_t)idx->n_keys + 1, piece = (size_t)1 << 22;
    std::vector<u128> buf(std::min(n, piece));
    for (size_t lo = 0; lo < n; lo += piece) {
        const size_t m = std::min(piece, n - lo);
        MPN_HIP_CHECK(hipMemcpyAsync(buf.data(), idx->kv.p + lo, m * sizeof(u128), hipMemcpyDeviceToHost, st));
        MPN_HIP_CHECK(stream_sync(st));
        for (size_t i = 0; i < m; ++i) {
            if (keys && lo + i < (size_t)idx->n_keys) keys[lo + i] = buf[i].x;
            if (key_off) key_off[lo + i] = (int64_t)buf[i].y;
        }
    }
    return 0;
}

// ---- persistent form of the index (minimap2 -d) ----------------------------------------------------
// Little-endian file: magic "MPNIDX01", then k, w, n_seq, n_nruns (int32), n_keys, n_mz, total bases, n 2-bit words (int64),
// then lens[n_seq] int32, names (u32 length + bytes each), N runs (start[], end[] int64), 2-bit words, keys, key_off, pos.
static const char MPN_IDX_MAGIC[8] = {'M', 'P', 'N', 'I', 'D', 'X', '0', '1'};

static int index_save(const mpn_index *idx, const char *path, bool append);
int mpn_index_save(const mpn_index *idx, const char *path) { return index_save(idx, path, false); }
// minimap2 -d with a target set of several parts: every part is dumped into the one file, one after another
int mpn_index_save_append(const mpn_index *idx, const char *path) { return index_save(idx, path, true); }

static int index_save(const mpn_index *idx, const char *path, bool append) {
    if (!idx || !path) { set_error("mpn_index_save: null argument"); return -1; }
    hipStream_t st = 0;
    const int64_t total = idx->seq_off.empty() ? 0 : idx->seq_off.back();
    std::vector<uint32_t> words(idx->d_seq2.n);
    const std::vector<int64_t> &ns = idx->h_nrun_s, &ne = idx->h_nrun_e;
    std::vector<uint64_t> keys((size_t)idx->n_keys), pos((size_t)idx->n_mz);
    std::vector<int64_t> h_key_off((size_t)idx->n_keys + 1);
    if (fetch_keys(idx, keys.data(), h_key_off.data(), st) || idx->pos.download(pos.data(), pos.size(), st) ||
        idx->d_seq2.download(words.data(), words.size(), st))
        return -1;
    MPN_HIP_CHECK(stream_sync(st));
    FILE *f = fopen(path, append ? "ab" : "wb");
    if (!f) { set_error("mpn_index_save: cannot open %s", path); return -1; }
    bool ok = true;
    auto put = [&](const void *p, size_t bytes) { if (bytes && fwrite(p, 1, bytes, f) != bytes) ok = false; };
    const int32_t h32[4] = {idx->k, idx->w, idx->n_seq, (int32_t)ns.size()};
    const int64_t h64[4] = {idx->n_keys, idx->n_mz, total, (int64_t)words.size()};
    put(MPN_IDX_MAGIC, 8); put(h32, sizeof(h32)); put(h64, sizeof(h64));
    put(idx->lens.data(), idx->lens.size() * 4);
    for (const std::string &nm : idx->names) { const uint32_t l = (uint32_t)nm.size(); put(&l, 4); put(nm.data(), l); }
    put(ns.data(), ns.size() * 8); put(ne.data(), ne.size() * 8);
    put(words.data(), words.size() * 4);
    put(keys.data(), keys.size() * 8);
    put(h_key_off.data(), h_key_off.size() * 8);
    put(pos.data(), pos.size() * 8);
    if (fclose(f) != 0) ok = false;
    if (!ok) { set_error("mpn_index_save: short write to %s", path); return -1; }
    return 0;
}

mpn_index *mpn_index_load(const char *path) { return mpn_index_load_at(path, 0, nullptr); }

// One part of a saved index: the part that starts at byte `offset` of the file; *next_offset = where the next part starts, or
// -1 after the last one (a file written by mpn_index_save holds one part, one extended by mpn_index_save_append several).
mpn_index *mpn_index_load_at(const char *path, int64_t offset, int64_t *next_offset) {
    if (next_offset) *next_offset = -1;
    if (!path || offset < 0) { set_error("mpn_index_load: null path or negative offset"); return nullptr; }
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("mpn_index_load: cannot open %s", path); return nullptr; }
    if (offset && fseeko(f, (off_t)offset, SEEK_SET) != 0) { fclose(f); set_error("mpn_index_load: cannot seek in %s", path); return nullptr; }
    mpn_index *idx = new mpn_index();
    bool ok = true;
    auto get = [&](void *p, size_t bytes) { if (bytes && fread(p, 1, bytes, f) != bytes) ok = false; };
    auto fail = [&](const char *why) { set_error("mpn_index_load: %s (%s)", why, path); if (f) fclose(f); delete idx; return (mpn_index *)nullptr; };
    char magic[8];
    int32_t h32[4];
    int64_t h64[4];
    int64_t file_size = 0;
    get(magic, 8); get(h32, sizeof(h32)); get(h64, sizeof(h64));
    if (!ok || memcmp(magic, MPN_IDX_MAGIC, 8) != 0) return fail("not an mpn index file");
    idx->k = h32[0]; idx->w = h32[1]; idx->n_seq = h32[2]; idx->n_nruns = h32[3];
    idx->n_keys = h64[0]; idx->n_mz = h64[1];
    const int64_t total = h64[2], n_words = h64[3];
    if (idx->k < 1 || idx->k > 28 || idx->w < 1 || idx->w > 255 || idx->n_seq <= 0 || idx->n_keys < 0 || idx->n_mz < idx->n_keys || total < 0 ||
        n_words < (total + 15) / 16 || n_words > (total + 15) / 16 + 1 || idx->n_nruns < 0)
        return fail("corrupt header");
    {   // the counts of the header must agree with the size of the file before anything is allocated from them
        struct stat sb;
        if (fstat(fileno(f), &sb) != 0) return fail("cannot stat");
        const int64_t fixed = 8 + 16 + 32 + (int64_t)idx->n_seq * 8 /* lens + name length words */ + (int64_t)idx->n_nruns * 16 + n_words * 4 +
                              idx->n_keys * 8 + (idx->n_keys + 1) * 8 + idx->n_mz * 8;
        if ((int64_t)sb.st_size < offset + fixed) return fail("truncated file");
        file_size = (int64_t)sb.st_size;
    }
    idx->lens.resize((size_t)idx->n_seq);
    get(idx->lens.data(), idx->lens.size() * 4);
    idx->seq_off.assign((size_t)idx->n_seq + 1, 0);
    for (int i = 0; i < idx->n_seq && ok; ++i) {
        uint32_t l = 0;
        get(&l, 4);
        if (!ok || l > (1u << 20) || idx->lens[(size_t)i] < 0) return fail("corrupt name table");
        std::string nm(l, '\0');
        get(&nm[0], l);
        idx->names.push_back(nm);
        idx->seq_off[(size_t)i + 1] = idx->seq_off[(size_t)i] + idx->lens[(size_t)i];
    }
    if (!ok || idx->seq_off.back() != total) return fail("corrupt sequence table");
    std::vector<int64_t> ns((size_t)idx->n_nruns), ne((size_t)idx->n_nruns);
    idx->h_seq2.resize((size_t)n_words);
    std::vector<uint64_t> keys((size_t)idx->n_keys), pos((size_t)idx->n_mz);
    std::vector<int64_t> h_key_off((size_t)idx->n_keys + 1);
    get(ns.data(), ns.size() * 8); get(ne.data(), ne.size() * 8);
    get(idx->h_seq2.data(), idx->h_seq2.size() * 4);
    get(keys.data(), keys.size() * 8);
    get(h_key_off.data(), h_key_off.size() * 8);
    get(pos.data(), pos.size() * 8);
    if (!ok || h_key_off.back() != idx->n_mz) return fail("truncated file");
    // the kernels index with these arrays: reject anything that would send them out of bounds
    for (size_t r = 0; r < ns.size(); ++r)
        if (ns[r] < 0 || ne[r] > total || ns[r] >= ne[r] || (r > 0 && ns[r] < ne[r - 1])) return fail("corrupt N runs");
    if (h_key_off[0] != 0) return fail("corrupt key offsets");
    const uint64_t hmask = idx->k < 32 ? ((uint64_t)1 << 2 * idx->k) - 1 : ~(uint64_t)0;
    for (int64_t i = 0; i < idx->n_keys; ++i) {
        if (h_key_off[(size_t)i + 1] <= h_key_off[(size_t)i]) return fail("corrupt key offsets");
        if (keys[(size_t)i] > hmask || (i > 0 && keys[(size_t)i] <= keys[(size_t)i - 1])) return fail("keys are not sorted");
    }
    for (int64_t i = 0; i < idx->n_mz; ++i) {
        const uint64_t rid = pos[(size_t)i] >> 32, p = (uint32_t)pos[(size_t)i] >> 1;
        if (rid >= (uint64_t)idx->n_seq || p >= (uint64_t)idx->lens[(size_t)rid]) return fail("position outside its target");
    }
    {
        const int64_t end = (int64_t)ftello(f);
        if (next_offset && end >= 0 && end < file_size) *next_offset = end;
    }
    fclose(f);
    f = nullptr;
    hipStream_t st = 0;
    if (idx->keys.upload(keys.data(), keys.size(), st) || idx->key_off.upload(h_key_off.data(), h_key_off.size(), st) ||
        idx->pos.upload(pos.data(), pos.size(), st) || idx->d_seq_off.upload(idx->seq_off.data(), idx->seq_off.size(), st) ||
        idx->d_lens.upload(idx->lens.data(), idx->lens.size(), st) ||
        idx->d_seq2.upload(idx->h_seq2.data(), idx->h_seq2.size(), st) || idx->d_nrun_s.upload(ns.data(), ns.size(), st) ||
        idx->d_nrun_e.upload(ne.data(), ne.size(), st) || build_bucket_table(idx, st) || stream_sync(st) != hipSuccess) {
        delete idx;  // (the failing HIP call has set the error text)
        return nullptr;
    }
    idx->h_nrun_s.swap(ns); idx->h_nrun_e.swap(ne);
    std::vector<uint32_t>().swap(idx->h_seq2);   // (only the staging of the load)
    return idx;
}

void mpn_index_destroy(mpn_index *idx) { delete idx; }
int64_t mpn_index_n_minimizers(const mpn_index *idx) { return idx->n_mz; }
int64_t mpn_index_n_keys(const mpn_index *idx) { return idx->n_keys; }
int32_t mpn_index_n_seq(const mpn_index *idx) { return idx->n_seq; }
int32_t mpn_index_k(const mpn_index *idx) { return idx->k; }
int32_t mpn_index_w(const mpn_index *idx) { return idx->w; }
int32_t mpn_index_seq_len(const mpn_index *idx, int32_t i) { return i >= 0 && i < idx->n_seq ? idx->lens[(size_t)i] : -1; }
int32_t mpn_index_seq_name(const mpn_index *idx, int32_t i, char *buf, int32_t cap) {
    if (i < 0 || i >= idx->n_seq || !buf || cap <= 0) return -1;
    const std::string &nm = idx->names[(size_t)i];
    const int32_t l = (int32_t)std::min<size_t>(nm.size(), (size_t)cap - 1);
    memcpy(buf, nm.data(), (size_t)l);
    buf[l] = 0;
    return (int32_t)nm.size();
}

// minimap2's mm_idx_cal_max_occ: the (1 - f) quantile of the occurrence counts of the keys, + 1.  The counts are
// histogrammed on the device (bins 0..65534 exact, the last one open-ended); the exact host path is kept for the case
// that the quantile falls into the open bin.
int32_t mpn_index_mid_occ(const mpn_index *idx, float f) {
    if (f <= 0.f) return INT32_MAX;
    const int64_t n = idx->n_keys;
    if (n == 0) return 1;
    std::lock_guard<std::mutex> g(idx->mu);
    for (auto &kv : idx->mid_occ_cache) if (kv.first == f) return kv.second;
    int64_t kk = (int64_t)(uint32_t)((1. - (double)f) * (double)n);
    if (kk >= n) kk = n - 1;
    hipStream_t st = 0;
    const int nbins = 1 << 16;
    if (idx->occ_hist.empty()) {
        DevBuf<unsigned long long> hist;
        idx->occ_hist.assign((size_t)nbins, 0);
        if (hist.alloc((size_t)nbins) || hist.zero(st)) return -1;
        hipLaunchKernelGGL(idx_occ_hist_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, st, (const u128 *)idx->kv.p, n, nbins, hist.p);
        if (hist.download((unsigned long long *)idx->occ_hist.data(), (size_t)nbins, st) || stream_sync(st) != hipSuccess) {
            idx->occ_hist.clear();
            set_error("mpn_index_mid_occ: histogram failed");
            return -1;
        }
    }
    int64_t cum = 0, v = -1;
    for (int b = 0; b < nbins; ++b) { cum += (int64_t)idx->occ_hist[(size_t)b]; if (cum > kk) { v = b; break; } }
    if (v < 0 || v == nbins - 1) {  // open bin: exact selection on the host
        std::vector<int64_t> h_key_off((size_t)n + 1);
        if (fetch_keys(idx, nullptr, h_key_off.data(), st)) return -1;
        std::vector<uint32_t> a((size_t)n);
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = (uint32_t)(h_key_off[(size_t)i + 1] - h_key_off[(size_t)i]);
        std::nth_element(a.begin(), a.begin() + kk, a.end());
        v = a[(size_t)kk];
    }
    idx->mid_occ_cache.push_back({f, (int32_t)(v + 1)});
    return (int32_t)(v + 1);
}

__global__ void idx_decode_kernel(RefView rv, int64_t g0, int64_t len, char *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < len) out[i] = "ACGTN"[ref_code(rv, g0 + i)];
}

int64_t mpn_index_fetch_seq(const mpn_index *idx, int32_t i, int64_t start, int64_t len, char *out) {
    if (!idx || !out || i < 0 || i >= idx->n_seq || start < 0 || len < 0 || start + len > idx->lens[(size_t)i]) {
        set_error("mpn_index_fetch_seq: range outside the target");
        return -2;
    }
    if (len == 0) return 0;
    hipStream_t st = 0;
    DevBuf<char> d;
    if (d.alloc((size_t)len)) return -1;
    const RefView rv{idx->d_seq2.p, idx->d_seq_off.p, idx->d_nrun_s.p, idx->d_nrun_e.p, idx->n_nruns};
    hipLaunchKernelGGL(idx_decode_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, rv, idx->seq_off[(size_t)i] + start, len, d.p);
    MPN_HIP_CHECK(hipGetLastError());
    if (d.download(out, (size_t)len, st)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    return len;
}

int mpn_index_export(const mpn_index *idx, uint64_t *keys, int64_t *key_off, uint64_t *pos) {
    hipStream_t st = 0;
    if (fetch_keys(idx, keys, key_off, st) || idx->pos.download(pos, (size_t)idx->n_mz, st)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    return 0;
}

int64_t mpn_sketch_batch(int32_t n, const char *seqs, const int64_t *seq_off, const int32_t *seq_len, int32_t k, int32_t w,
                         int64_t *mz_off, uint64_t *mz, int64_t cap) {
    hipStream_t st = 0;
    DevBuf<uint8_t> d_seqs;
    DevBuf<int64_t> d_off, d_mz_off;
    DevBuf<int32_t> d_len;
    DevBuf<u128> d_mz;
    int64_t bases = 0, n_mz = 0;
    if (upload_seqs(n, seqs, seq_off, seq_len, d_seqs, d_off, d_len, &bases, st)) return -1;
    if (sketch_device(d_seqs.p, d_off.p, d_len.p, seq_len, n, k, w, 0, d_mz_off, d_mz, &n_mz, st)) return -1;
    if (d_mz_off.download(mz_off, (size_t)n + 1, st)) return -1;
    if (n_mz <= cap && d_mz.download((u128 *)mz, (size_t)n_mz, st)) return -1;
    MPN_HIP_CHECK(stream_sync(st));
    return n_mz <= cap ? n_mz : -3;
}

int mpn_seed_chain_batch(const mpn_index *idx, const mpn_map_opt *opt, int32_t n, const char *seqs, const int64_t *seq_off,
                         const int32_t *seq_len, int64_t *n_anchor, int32_t *rep_len, int64_t *chain_off, uint64_t *u,
                         int64_t u_cap, int64_t *anchor_off, uint64_t *b, int64_t b_cap) {
    hipStream_t st = 0;
    memset(g_stats, 0, sizeof(g_stats));
    DevBuf<uint8_t> d_seqs;
    DevBuf<int64_t> d_off;
    DevBuf<int32_t> d_len;
    int64_t bases = 0;
    if (upload_seqs(n, seqs, seq_off, seq_len, d_seqs, d_off, d_len, &bases, st)) return -1;
    g_stats[0] = bases;
    SeedChainOut o;
    if (seed_chain_device(idx, opt, n, d_seqs.p, d_off.p, d_len.p, seq_len, o, st, nullptr)) return -1;
    HostChains h;
    PoolBuf pin_u{nullptr, 0, true}, pin_b{nullptr, 0, true};
    struct Free { PoolBuf &a, &b; ~Free() { a.release(); b.release(); } } free_pins{pin_u, pin_b};
    if (download_chains(n, o, h, pin_u, pin_b, st, 2)) return -1;
    for (int i = 0; i < n; ++i) { n_anchor[i] = h.n_anchor[i]; rep_len[i] = h.rep_len[i]; }
    memcpy(chain_off, h.chain_off.data(), ((size_t)n + 1) * 8);
    memcpy(anchor_off, h.b_off.data(), ((size_t)n + 1) * 8);
    if (h.chain_off[n] > u_cap || h.b_off[n] > b_cap) return -3;
    for (int i = 0; i < n; ++i) h.read_chains(i, u + h.chain_off[i], (u128 *)b + h.b_off[i]);
    return 0;
}

void mpn_map_last_stats(int64_t stats[32]) { memcpy(stats, g_stats, 32 * sizeof(int64_t)); }
int32_t mpn_map_last_stats_ex(int64_t *stats, int32_t n) {
    const int32_t m = n < MPN_NSTATS ? n : MPN_NSTATS;
    if (stats && m > 0) memcpy(stats, g_stats, (size_t)m * sizeof(int64_t));
    return MPN_NSTATS;
}

}  // extern "C"
