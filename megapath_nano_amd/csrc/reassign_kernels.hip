// Read reassignment on gfx950: replaces the pandas body of /root/reference/bin/lib/reassignment.py
// (Reassign :66-108) and the reductions at /root/reference/bin/megapath_nano.py:1287-1289,:3664-3667.
//
// Data layout in HBM: struct-of-arrays over alignment rows grouped by read (CSR read_ptr), all int32/int64/f64:
// name code, score, tiebreak, aligned_bp, species code; per-name counters are int64[n_names].
// Reads are independent after the counters are known, so both kernels give one read to one lane
// (a read has at most best_n+1 rows; -N 50 => <= 51) and stream the rows with coalesced-enough loads.
// Both kernels are HBM/atomic bound integer work (no MFMA): algorithmic traffic = 36 B/row in, 5 B/row out.
#include "mpn_common.h"
#include "../../include/mpn_reassign.h"

#include <vector>
#include <algorithm>

namespace mpn {

struct ReDev {
    int64_t n_rows;
    int32_t n_reads, n_names, n_species;
    const int64_t *read_ptr;
    const int32_t *name, *score, *species;
    const double *tiebreak;
    const int64_t *aligned_bp;
    uint8_t *keep;
    unsigned long long *all_count, *u_count, *n_multi;
};

// reassignment.py:73 : among rows of one read with the same name keep the best score (last row on ties).
__global__ __launch_bounds__(256) void reassign_counts_kernel(ReDev d) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < d.n_reads; r += gridDim.x * blockDim.x) {
        const int64_t p0 = d.read_ptr[r], p1 = d.read_ptr[r + 1];
        int nk = 0;
        for (int64_t a = p0; a < p1; ++a) {
            const int na = d.name[a], sa = d.score[a];
            bool k = true;
            for (int64_t b = p0; b < p1; ++b) {
                if (b != a && d.name[b] == na) {
                    const int sb = d.score[b];
                    if (sb > sa || (sb == sa && b > a)) { k = false; break; }
                }
            }
            d.keep[a] = k;
            nk += k;
        }
        for (int64_t a = p0; a < p1; ++a) {
            if (!d.keep[a]) continue;
            atomicAdd(&d.all_count[d.name[a]], 1ULL);                 // :77
            if (nk == 1) atomicAdd(&d.u_count[d.name[a]], 1ULL);      // :80-81
        }
        if (nk > 1) atomicAdd(d.n_multi, 1ULL);                        // :84
    }
}

struct ApplyDev {
    ReDev d;
    const double *thr;        // error_rate * u_count[i]
    const uint8_t *cond1;     // all_count[i] - MCount(=0) >= ratio * all_count[i]
    const uint8_t *explainer; // i explains at least one j
    const int32_t *rank;      // processing order of explainer names
    const long long *u_count_g;
    double as_threshold;
    int32_t *new_name;
    unsigned long long *read_count, *bp;
    int relabel;              // 0 = relation empty: skip :38-64
};

__device__ __forceinline__ bool explains(const ApplyDev &a, int i, int j) {
    // reassignment.py:32 with MCount == 0
    return a.cond1[i] && (double)a.u_count_g[j] < a.thr[i];
}

__global__ __launch_bounds__(256) void reassign_apply_kernel(ApplyDev a) {
    const ReDev &d = a.d;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < d.n_reads; r += gridDim.x * blockDim.x) {
        const int64_t p0 = d.read_ptr[r], p1 = d.read_ptr[r + 1];
        for (int64_t x = p0; x < p1; ++x) a.new_name[x] = d.name[x];
        if (a.relabel) {
            // explainer rows of this read in ascending rank of their ORIGINAL name (:59-63); names are unique
            // inside a read after the dedupe, so ranks are distinct
            int last_rank = -1;
            for (;;) {
                int64_t pick = -1;
                int best = 0x7fffffff;
                for (int64_t x = p0; x < p1; ++x) {
                    if (!d.keep[x]) continue;
                    const int nm = d.name[x];
                    if (!a.explainer[nm]) continue;
                    const int rk = a.rank[nm];
                    if (rk > last_rank && rk < best) best = rk, pick = x;
                }
                if (pick < 0) break;
                last_rank = best;
                const int nm = d.name[pick];
                const double first = (double)d.score[pick];
                for (int64_t y = p0; y < p1; ++y) {
                    if (!d.keep[y]) continue;
                    const int cur = a.new_name[y];
                    if (cur != nm && explains(a, nm, cur) && (double)d.score[y] * a.as_threshold <= first)  // :48,:53
                        a.new_name[y] = nm;                                                                // :56
                }
            }
        }
        // megapath_nano.py:1287: best row = max (score, tiebreak), last on full ties
        int64_t bx = -1;
        for (int64_t x = p0; x < p1; ++x) {
            if (!d.keep[x]) continue;
            if (bx < 0 || d.score[x] > d.score[bx] ||
                (d.score[x] == d.score[bx] && d.tiebreak[x] >= d.tiebreak[bx]))
                bx = x;
        }
        if (bx >= 0) {
            atomicAdd(&a.read_count[a.new_name[bx]], 1ULL);                                 // :3666
            atomicAdd(&a.bp[d.species[bx]], (unsigned long long)d.aligned_bp[bx]);         // :1289
        }
    }
}

}  // namespace mpn

struct mpn_reassign_plan {
    int64_t n_rows;
    int32_t n_reads, n_names, n_species;
    mpn::DevBuf<int64_t> read_ptr, aligned_bp;
    mpn::DevBuf<int32_t> name, score, species, new_name, rank;
    mpn::DevBuf<double> tiebreak, thr;
    mpn::DevBuf<uint8_t> keep, cond1, explainer;
    mpn::DevBuf<unsigned long long> counters;  // all_count | u_count | n_multi
    mpn::DevBuf<unsigned long long> out;       // read_count | bp
    mpn::DevBuf<long long> u_count_g;
    bool counted = false;
};

static mpn::ReDev make_dev(mpn_reassign_plan *p) {
    mpn::ReDev d;
    d.n_rows = p->n_rows; d.n_reads = p->n_reads; d.n_names = p->n_names; d.n_species = p->n_species;
    d.read_ptr = p->read_ptr.p; d.name = p->name.p; d.score = p->score.p; d.species = p->species.p;
    d.tiebreak = p->tiebreak.p; d.aligned_bp = p->aligned_bp.p; d.keep = p->keep.p;
    d.all_count = p->counters.p; d.u_count = p->counters.p + p->n_names; d.n_multi = p->counters.p + 2 * (size_t)p->n_names;
    return d;
}

static int grid_for(int n_reads) { return std::max(1, std::min((n_reads + 255) / 256, 256 * 8)); }

extern "C" {

int mpn_reassign_create(int64_t n_rows, int32_t n_reads, int32_t n_names, int32_t n_species,
                        const int64_t *read_ptr, const int32_t *name_idx, const int32_t *score,
                        const double *tiebreak, const int64_t *aligned_bp, const int32_t *species_idx,
                        mpn_reassign_plan **plan) {
    if (n_rows < 0 || n_reads < 0 || n_names <= 0 || n_species <= 0) { mpn::set_error("mpn_reassign_create: bad sizes"); return -2; }
    if (read_ptr[0] != 0 || read_ptr[n_reads] != n_rows) { mpn::set_error("mpn_reassign_create: read_ptr does not cover the rows"); return -2; }
    for (int64_t i = 0; i < n_rows; ++i)
        if (name_idx[i] < 0 || name_idx[i] >= n_names || species_idx[i] < 0 || species_idx[i] >= n_species) {
            mpn::set_error("mpn_reassign_create: code out of range at row %lld", (long long)i);
            return -2;
        }
    mpn_reassign_plan *p = new mpn_reassign_plan();
    p->n_rows = n_rows; p->n_reads = n_reads; p->n_names = n_names; p->n_species = n_species;
    hipStream_t st = 0;
    if (p->read_ptr.upload(read_ptr, (size_t)n_reads + 1, st) || p->name.upload(name_idx, n_rows, st) ||
        p->score.upload(score, n_rows, st) || p->tiebreak.upload(tiebreak, n_rows, st) ||
        p->aligned_bp.upload(aligned_bp, n_rows, st) || p->species.upload(species_idx, n_rows, st) ||
        p->keep.alloc(n_rows) || p->new_name.alloc(n_rows) || p->counters.alloc(2 * (size_t)n_names + 1) ||
        p->out.alloc((size_t)n_names + n_species) || p->thr.alloc(n_names) || p->cond1.alloc(n_names) ||
        p->explainer.alloc(n_names) || p->rank.alloc(n_names) || p->u_count_g.alloc(n_names)) {
        delete p;
        return -1;
    }
    if (hipStreamSynchronize(st) != hipSuccess) { mpn::set_error("mpn_reassign_create: upload failed"); delete p; return -1; }
    *plan = p;
    return 0;
}

int mpn_reassign_counts(mpn_reassign_plan *p, int64_t *all_count, int64_t *u_count, int64_t *n_multi_reads) {
    hipStream_t st = 0;
    if (p->counters.zero(st)) return -1;
    mpn::ReDev d = make_dev(p);
    if (p->n_reads > 0) {
        hipLaunchKernelGGL(mpn::reassign_counts_kernel, dim3(grid_for(p->n_reads)), dim3(256), 0, st, d);
        MPN_HIP_CHECK(hipGetLastError());
    }
    std::vector<unsigned long long> h(2 * (size_t)p->n_names + 1);
    if (p->counters.download(h.data(), h.size(), st)) return -1;
    MPN_HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < p->n_names; ++i) { all_count[i] = (int64_t)h[i]; u_count[i] = (int64_t)h[p->n_names + i]; }
    *n_multi_reads = (int64_t)h[2 * (size_t)p->n_names];
    p->counted = true;
    return 0;
}

int mpn_reassign_apply(mpn_reassign_plan *p, const int64_t *all_count, const int64_t *u_count, const int32_t *name_rank,
                       double error_rate, double ratio, double as_threshold, uint8_t *keep, int32_t *new_name,
                       uint8_t *explainer, int64_t *read_count_by_name, int64_t *aligned_bp_by_species,
                       int64_t *n_relations) {
    if (!p->counted) { mpn::set_error("mpn_reassign_apply: call mpn_reassign_counts first (it also performs the dedupe)"); return -2; }
    const int S = p->n_names;
    hipStream_t st = 0;
    // relation (reassignment.py:27-36): i explains j  <=>  i != j, both present, cond1(i) and u_count[j] < error_rate*u_count[i]
    std::vector<double> thr(S);
    std::vector<uint8_t> cond1(S), expl(S);
    std::vector<long long> ucg(S);
    std::vector<int64_t> present_u;  // u_count of present names, sorted
    for (int i = 0; i < S; ++i) if (all_count[i] > 0) present_u.push_back(u_count[i]);
    std::sort(present_u.begin(), present_u.end());
    int64_t nrel = 0;
    for (int i = 0; i < S; ++i) {
        thr[i] = error_rate * (double)u_count[i];
        cond1[i] = all_count[i] > 0 && (double)all_count[i] >= ratio * (double)all_count[i];
        ucg[i] = u_count[i];
        expl[i] = 0;
        if (!cond1[i]) continue;
        // number of present j with u_count[j] < thr[i], minus i itself if it qualifies
        int64_t cnt = std::lower_bound(present_u.begin(), present_u.end(), thr[i],
                                       [](int64_t v, double t) { return (double)v < t; }) - present_u.begin();
        if ((double)u_count[i] < thr[i]) --cnt;
        if (cnt > 0) { expl[i] = 1; nrel += cnt; }
    }
    *n_relations = nrel;
    if (p->thr.upload(thr.data(), S, st) || p->cond1.upload(cond1.data(), S, st) || p->explainer.upload(expl.data(), S, st) ||
        p->rank.upload(name_rank, S, st) || p->u_count_g.upload(ucg.data(), S, st) || p->out.zero(st))
        return -1;
    mpn::ApplyDev a;
    a.d = make_dev(p);
    a.thr = p->thr.p; a.cond1 = p->cond1.p; a.explainer = p->explainer.p; a.rank = p->rank.p; a.u_count_g = p->u_count_g.p;
    a.as_threshold = as_threshold; a.new_name = p->new_name.p; a.read_count = p->out.p; a.bp = p->out.p + S;
    a.relabel = nrel > 0;
    if (p->n_reads > 0) {
        hipLaunchKernelGGL(mpn::reassign_apply_kernel, dim3(grid_for(p->n_reads)), dim3(256), 0, st, a);
        MPN_HIP_CHECK(hipGetLastError());
    }
    std::vector<unsigned long long> h((size_t)S + p->n_species);
    if (p->out.download(h.data(), h.size(), st) || p->keep.download(keep, p->n_rows, st) ||
        p->new_name.download(new_name, p->n_rows, st))
        return -1;
    MPN_HIP_CHECK(hipStreamSynchronize(st));
    for (int i = 0; i < S; ++i) { read_count_by_name[i] = (int64_t)h[i]; explainer[i] = expl[i]; }
    for (int i = 0; i < p->n_species; ++i) aligned_bp_by_species[i] = (int64_t)h[S + i];
    return 0;
}

void mpn_reassign_destroy(mpn_reassign_plan *p) { delete p; }

}  // extern "C"
