// Internal types shared by mapper.hip (seed + chain stages) and align.hip (base-level extension, hit bookkeeping).
#pragma once
#include "mpn_common.h"
#include "map_types.h"

#include <atomic>
#include <chrono>
#include <time.h>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

struct mpn_map_opt_s;


struct mpn_index {
    int k = 15, w = 10;
    int32_t n_seq = 0;
    std::vector<std::string> names;
    std::vector<int32_t> lens;
    std::vector<int64_t> seq_off;     // host: offset of each target in concatenated coordinates
    std::vector<uint32_t> h_seq2;               // staging of mpn_index_load only (the packed targets live in d_seq2)
    std::vector<int64_t> h_nrun_s, h_nrun_e;    // ambiguous-base runs (concatenated coordinates), kept for mpn_index_save
    int64_t n_keys = 0, n_mz = 0;
    mpn::DevBuf<uint64_t> keys, pos;
    mpn::DevBuf<int64_t> key_off;
    mpn::DevBuf<int64_t> bucket_start;  // first key of every hash bucket (top bucket_bits of the 2k-bit hash), + end sentinel
    mpn::DevBuf<mpn::u128> kv;          // (key, key_off) pairs + end sentinel: what seed_lookup_kernel reads
    int bucket_shift = 0;
    mpn::DevBuf<uint32_t> d_seq2;     // device: targets packed 2 bits per base
    mpn::DevBuf<int64_t> d_seq_off, d_nrun_s, d_nrun_e;  // + ambiguous-base runs (concatenated coordinates)
    mpn::DevBuf<int32_t> d_lens;      // target lengths (the planning kernel clips the extension windows with them)
    int32_t n_nruns = 0;
    mutable std::mutex mu;
    mutable std::vector<std::pair<float, int32_t>> mid_occ_cache;
    mutable std::vector<uint64_t> occ_hist;   // occurrence histogram of the keys (filled on first use)
};

namespace mpn {

struct SeedChainOut {
    int64_t n_anchors = 0;
    // the per-read tables the host reads back (n_anchor, n_chained, u_pos, b_pos, n_chain, rep_len, used) are slices of ONE
    // block: a single device-to-host copy instead of seven (a copy is a queue round trip, and queues are shared)
    DevBuf<unsigned char> tables;
    size_t tables_bytes = 0;
    DevBuf<int64_t> n_anchor;                                    // per read: index hits of its minimizers (what minimap2 calls n_a)
    DevBuf<int64_t> anchor_off, c_off, n_chained, u_pos, b_pos;  // anchor_off: CSR of the hits that passed the stray-hit filter; c_off: of those kept for chaining
    DevBuf<int32_t> rep_len, n_ends, n_chain;
    DevBuf<u128> anchors, chained;
    DevBuf<uint64_t> u, u_compact;
    DevBuf<ChainRec> recs;                                       // one record per surviving chain, parallel to u_compact
    DevBuf<unsigned long long> used;
};

// minimizers of a batch of reads that is resident on the device (mm_sketch of every read): index parts built with the same k, w
// share it -- a sub-batch is sketched once, not once per part
struct ReadSketch {
    DevBuf<int64_t> mz_off;
    DevBuf<u128> mz;
    int64_t n_mz = 0;
    int k = 0, w = 0;
    bool valid = false;
};

struct HostChains {
    std::vector<int64_t> n_anchor, chain_off, b_off;    // per read: anchors found; prefix sums of chains and of chained anchors
    std::vector<int64_t> n_chained, u_pos, b_pos;       // per read: chained anchors; start in the compact pools
    std::vector<int32_t> n_chain, rep_len;
    const uint64_t *u_all = nullptr;                    // compact pools as downloaded (caller-owned pinned memory)
    const u128 *b_all = nullptr;                        // (only when the anchors were asked for: the stage test)
    const ChainRec *rec_all = nullptr;
    int64_t n_pool_chains = 0;                          // chains in the compact pools of the batch
    // chains of read i in ascending order of their first anchor (minimap2 re-sorts them like this so that neighbouring
    // chains can be joined): u_out[n_chain[i]], b_out[n_chained[i]]
    void read_chains(int i, uint64_t *u_out, u128 *b_out) const;
    // the same order without the anchors: chain c of the sorted order is chain order[c] of the pool (u_all / rec_all at
    // u_pos[i] + order[c]) and its anchors start at src[c] relative to b_pos[i] in the chain stage's device pool
    void chain_order(int i, int32_t *order, int64_t *src) const;
};

constexpr int MPN_NSTATS = 64;
extern thread_local int64_t g_stats[MPN_NSTATS];

// 2-bit packing of 0..4 codes (N -> 0 + run list)
void pack_2bit(const uint8_t *codes, int64_t n, std::vector<uint32_t> &words, std::vector<int64_t> &ns, std::vector<int64_t> &ne);

// Waits for a stream without spinning: the worker threads of the pipelined mapper share the host cores with the
// host phases of the other workers, so a waiting thread must sleep (blocking-sync event), not poll.
hipError_t stream_sync(hipStream_t st);

// HIP-event timer for groups of launches on one stream.  mark() only RECORDS an event and remembers which counter the
// span since the previous mark belongs to; the elapsed times are read back in resolve(), which the caller invokes
// after a synchronisation it needs anyway -- timing never adds a host/device round trip of its own.
struct EvTimer {
    hipStream_t st;
    std::vector<hipEvent_t> ev;
    std::vector<int> slot, slot2;  // slot[i] (and slot2[i]): g_stats indices charged with ev[i] -> ev[i+1]; -1 = not charged
    static bool enabled() { static const bool on = []() { const char *e = getenv("MPN_KERNEL_EVENTS"); return !e || atoi(e) != 0; }(); return on; }
    explicit EvTimer(hipStream_t s) : st(s) { push(); }
    void push() {
        if (!enabled()) return;   // MPN_KERNEL_EVENTS=0: no per-kernel timing (an event per kernel group is a completion signal the runtime's thread handles)
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, st);
        ev.push_back(e);
        slot.push_back(-1); slot2.push_back(-1);
    }
    // the span that ends here goes to stat_index (a kernel family) and, if given, to `single` (one kernel's own slot)
    void mark(int stat_index, int single = -1) { if (!enabled()) return; slot.back() = stat_index; slot2.back() = single; push(); }
    void skip() { push(); }                                            // the span that ends here is not charged
    void resolve() {  // call after the stream has been synchronised
        if (!enabled()) return;
        for (size_t i = 0; i + 1 < ev.size(); ++i) {
            if (slot[i] < 0 && slot2[i] < 0) continue;
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) continue;
            if (slot[i] >= 0) g_stats[slot[i]] += (int64_t)(ms * 1e6);
            if (slot2[i] >= 0) g_stats[slot2[i]] += (int64_t)(ms * 1e6);
        }
        for (size_t i = 0; i + 1 < ev.size(); ++i) (void)hipEventDestroy(ev[i]);
        hipEvent_t last = ev.back();
        ev.assign(1, last); slot.assign(1, -1); slot2.assign(1, -1);
    }
    ~EvTimer() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};

// Grow-only device / pinned-host scratch that survives across calls (hipMalloc of multi-GB scratch per batch costs
// more than the kernels that use it).  One pool per process; the mapper entry points are not re-entrant.
struct PoolBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool pinned_host = false;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { if (pinned_host) (void)hipHostFree(p); else (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 4096;
        const hipError_t e_ = pinned_host ? hipHostMalloc(&p, want, hipHostMallocDefault) : guarded_malloc(&p, want);
        if (e_ != hipSuccess) {
            if (e_ == hipErrorOutOfMemory && !pinned_host) tl_oom = true;
            p = nullptr;
            set_error("%s of %zu bytes failed: %s", pinned_host ? "hipHostMalloc" : "hipMalloc", want, hipGetErrorString(e_));
            (void)hipGetLastError();
            return -1;
        }
        cap = want;
        return 0;
    }
    void release() { if (p) { if (pinned_host) (void)hipHostFree(p); else (void)hipFree(p); } p = nullptr; cap = 0; }
    template <typename T> T *as() const { return (T *)p; }
};

// phase log of the pipeline workers (MPN_DEBUG_PHASES=1): every stop_into() of a worker thread appends
// (worker, slot of the phase that ended, start ns, end ns); dumped to stderr at the end of the call
struct PhaseLog {
    struct Rec { int worker, slot; int64_t t0, t1; };
    std::mutex mu;
    std::vector<Rec> recs;
    bool on = false;
    std::chrono::steady_clock::time_point origin;
    void add(int worker, int slot, std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        std::lock_guard<std::mutex> g(mu);
        recs.push_back({worker, slot, std::chrono::duration_cast<std::chrono::nanoseconds>(a - origin).count(),
                        std::chrono::duration_cast<std::chrono::nanoseconds>(b - origin).count()});
    }
};
extern PhaseLog g_phase_log;
extern thread_local int tl_worker_id;

// MPN_DEBUG_CPU=1: thread CPU time of the calling (worker) thread per phase slot, next to the wall time
extern bool g_worker_cpu_on;
extern std::atomic<long long> g_worker_cpu_ns[64];
static inline long long thread_cpu_ns() {
    timespec t;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
    return t.tv_sec * 1000000000LL + t.tv_nsec;
}

struct WallTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    long long c0 = g_worker_cpu_on ? thread_cpu_ns() : 0;
    void stop_into(int64_t &acc) {
        auto t1 = std::chrono::steady_clock::now();
        acc += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
        if (g_phase_log.on) g_phase_log.add(tl_worker_id, (int)(&acc - g_stats), t0, t1);
        if (g_worker_cpu_on) { const long long c1 = thread_cpu_ns(); g_worker_cpu_ns[(&acc - g_stats) & 63] += c1 - c0; c0 = c1; }
        t0 = t1;
    }
};

}  // namespace mpn
