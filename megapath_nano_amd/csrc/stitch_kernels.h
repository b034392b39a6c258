// Stitching of a hit's DP windows on the GPU (the second half of minimap2's mm_align1, after the ksw calls): the CIGARs of
// the left extension, the gap fills and the right extension are concatenated in order (adjacent operations of the same kind
// merged), the DP score and the end coordinates are accumulated, and a gap fill that z-dropped cuts the hit there.  The
// CIGARs never leave HBM between the traceback and the finishing kernel (fin_kernels.h); the host receives 56 bytes per hit.
#pragma once
#include "fin_kernels.h"

namespace mpn {

// placeholder job of a window that gets no DP (refused by max_sw_mat, or empty): it keeps the job list of a hit complete
// (left?, fills..., right?) and counts as z-dropped at its start, exactly like the `ez.zdropped = 1` of the host code it replaces
enum { EZ_REFUSED = 0x100, EZ_INV = 0x200 };   // EZ_INV: the z-drop test found an inversion in this fill (second pass with zdrop_inv)

struct StitchReg {
    int32_t first_job, n_jobs;        // jobs [first_job, first_job + n_jobs): left extension (if any), gap fills in order, right extension (if any)
    int32_t qs, rs, qe, re;           // first anchor's start / last anchor's end (mm_align1's rs, qs, re, qe before the extensions)
    int32_t qs0, qe0;                 // query limits of the two extensions
    int32_t read, rid, rev, pad;
};

struct StitchOut {
    int64_t cig_off;                  // start of the stitched CIGAR in the round's pool
    int32_t n_ops, dp_score, rs1, re1, qs1, qe1;
    int32_t has_p, dropped, drop_fill, drop_max_t, drop_max_q;   // drop_fill: index of the z-dropped gap fill among the hit's fills
    int32_t split_n;                  // > 0: a z-drop cuts the hit after its first split_n anchors (mm_align1's mm_split_reg call)
    int32_t split_inv, split_rec;     // the cut is at an inversion: the remainder is marked (mm_align1: r2->split_inv = 1); index of the hit's SplitRec
};



// One hit to align in this round, as the host hands it to the planning kernel (plan_kernels.h), and what that kernel leaves
// for the split of a z-dropped hit
struct PlanReg {
    int64_t a_off;                   // the read's (squeezed) chained anchors in the device copy
    int32_t n_a, as, cnt, mlen;      // anchors of the read; the hit's anchors [as, as + cnt); its approximate match length
    int32_t read, qlen, split_inv, pad;
};
struct PlanSum { int32_t as1, cnt1; };   // the hit's anchors after mm_fix_bad_ends

constexpr uint32_t OP_NONE = 0xf;

// One wave per hit.  Lanes own jobs (64 per pass over the hit's list); the running "last operation kind" and the output
// offset are carried from pass to pass, so the merge rule of mm_append_cigar (first op of a window joins the last op before
// it when they are of the same kind) is a scan over windows.
__global__ __launch_bounds__(64) void stitch_kernel(const StitchReg *__restrict__ regs, int n_regs, const ExtJob *__restrict__ jobs,
                                                    const ExtRes *__restrict__ res, const uint32_t *__restrict__ COMPACT,
                                                    uint32_t *__restrict__ OUT, unsigned long long *__restrict__ out_used,
                                                    StitchOut *__restrict__ outs, FinJob *__restrict__ fin_jobs,
                                                    const PlanReg *__restrict__ pregs, const PlanSum *__restrict__ psum,
                                                    const int32_t *__restrict__ job_anchor, const u128 *__restrict__ A, int min_cnt,
                                                    SplitRec *__restrict__ splits, unsigned long long *__restrict__ n_splits) {
    const int lane = threadIdx.x;
    for (int ri = blockIdx.x; ri < n_regs; ri += gridDim.x) {
        const StitchReg sr = regs[ri];
        const ExtJob *jb0 = jobs + sr.first_job;
        const ExtRes *rs0 = res + sr.first_job;
        const int n = sr.n_jobs;
        // ---- where does the hit stop: the first gap fill that z-dropped (or got no DP) ----
        int k_stop = -1;
        for (int c0 = 0; c0 < n && k_stop < 0; c0 += 64) {
            const int k = c0 + lane;
            bool zd = false;
            if (k < n) {
                const int fl = jb0[k].flag;
                if (!(fl & EZ_EXTZ_ONLY)) zd = (fl & EZ_REFUSED) || rs0[k].zdropped;
            }
            const unsigned long long m = __ballot(zd);
            if (m) k_stop = c0 + __builtin_ctzll(m);
        }
        const int dropped = k_stop >= 0;
        const int n_inc = dropped ? k_stop + 1 : n;
        const bool has_left = n > 0 && (jb0[0].flag & EZ_EXTZ_ONLY) && jb0[0].reversed;
        // ---- sizes, score, flags ----
        int total = 0, dp = 0, any_ops = 0;
        uint32_t carry_op = OP_NONE;
        auto chunk = [&](int c0, uint32_t &carry_last, int &carry_off, int &nc, int64_t &pos, int &off, int &merged, int &score_part) {
            const int k = c0 + lane;
            nc = 0; pos = 0; merged = 0; score_part = 0;
            uint32_t first = OP_NONE, last = OP_NONE;
            if (k < n_inc) {
                const ExtJob &jb = jb0[k];
                if (!(jb.flag & EZ_REFUSED)) {
                    const ExtRes &r = rs0[k];
                    nc = r.n_cigar; pos = r.cig_pos;
                    if (nc > 0) { first = COMPACT[pos] & 0xf; last = COMPACT[pos + nc - 1] & 0xf; }
                    if (jb.flag & EZ_EXTZ_ONLY) score_part = nc > 0 ? r.max : 0;
                    else score_part = r.zdropped ? r.max : r.score;
                }
            }
            uint32_t incl = last;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d && incl == OP_NONE) incl = o; }
            uint32_t prev = __shfl_up(incl, 1);
            if (lane == 0 || prev == OP_NONE) prev = carry_last;
            merged = nc > 0 && prev != OP_NONE && prev == first;
            const int contrib = nc - merged;
            int x = contrib;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(x, d); if (lane >= d) x += o; }
            off = carry_off + x - contrib;
            carry_off += __shfl(x, 63);
            const uint32_t tail = __shfl(incl, 63);
            if (tail != OP_NONE) carry_last = tail;
        };
        for (int c0 = 0; c0 < n_inc; c0 += 64) {
            int nc, off, merged, sp; int64_t pos;
            chunk(c0, carry_op, total, nc, pos, off, merged, sp);
            for (int d = 32; d; d >>= 1) sp += __shfl_xor(sp, d);
            dp += sp;
            any_ops |= __ballot(nc > 0) != 0;
        }
        unsigned long long base = 0;
        if (lane == 0 && total > 0) base = atomicAdd(out_used, (unsigned long long)total);
        base = (unsigned long long)__shfl((long long)base, 0);
        // ---- copy (window by window, a lane per window), then the merged first operations ----
        if (total > 0) {
            uint32_t carry2 = OP_NONE;
            int off_c = 0;
            for (int c0 = 0; c0 < n_inc; c0 += 64) {
                int nc, off, merged, sp; int64_t pos;
                chunk(c0, carry2, off_c, nc, pos, off, merged, sp);
                uint32_t *dst = OUT + base + off;
                for (int q = merged; q < nc; ++q) dst[q - merged] = COMPACT[pos + q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            carry2 = OP_NONE; off_c = 0;
            for (int c0 = 0; c0 < n_inc; c0 += 64) {
                int nc, off, merged, sp; int64_t pos;
                chunk(c0, carry2, off_c, nc, pos, off, merged, sp);
                if (merged) atomicAdd(OUT + base + off - 1, (COMPACT[pos] >> 4) << 4);
            }
        }
        // ---- coordinates ----
        int split_n = 0, split_slot = -1;
        if (lane == 0) {
            int rs1 = sr.rs, qs1 = sr.qs, re1 = sr.re, qe1 = sr.qe;
            if (has_left) {
                int reach = 0, max_t = -1, max_q = -1, mqe_t = -1;
                if (!(jb0[0].flag & EZ_REFUSED)) { const ExtRes &e = rs0[0]; reach = e.reach_end; max_t = e.max_t; max_q = e.max_q; mqe_t = e.mqe_t; }
                rs1 = sr.rs - (reach ? mqe_t + 1 : max_t + 1);
                qs1 = sr.qs - (reach ? sr.qs - sr.qs0 : max_q + 1);
            }
            StitchOut o;
            o.drop_fill = -1; o.drop_max_t = o.drop_max_q = -1; o.split_n = 0; o.split_inv = 0; o.split_rec = -1;
            if (dropped) {
                const ExtJob &jb = jb0[k_stop];
                int max_t = -1, max_q = -1;
                if (!(jb.flag & EZ_REFUSED)) { max_t = rs0[k_stop].max_t; max_q = rs0[k_stop].max_q; }
                re1 = jb.ts + (max_t + 1); qe1 = jb.qs + (max_q + 1);
                o.drop_fill = k_stop - (has_left ? 1 : 0); o.drop_max_t = max_t; o.drop_max_q = max_q;
                // the hit is cut after the last anchor that lies before the drop, if enough anchors remain behind it
                const u128 *a = A + pregs[ri].a_off;
                const int as1 = psum[ri].as1, cnt1 = psum[ri].cnt1;
                int j;
                for (j = job_anchor[sr.first_job + k_stop] - 1; j >= 0; --j) if ((int32_t)a[as1 + j].x <= jb.ts + max_t) break;
                if (j < 0) j = 0;
                if (cnt1 - (j + 1) >= min_cnt) { o.split_n = as1 + j + 1 - pregs[ri].as; o.split_inv = (jb.flag & EZ_INV) ? 1 : 0; }
                // (mm_split_reg does nothing for a cut outside the hit)
                if (o.split_n > 0 && o.split_n < pregs[ri].cnt) { split_n = o.split_n; split_slot = (int)atomicAdd(n_splits, 1ULL); o.split_rec = split_slot; }
            } else if (n > 0 && (jb0[n - 1].flag & EZ_EXTZ_ONLY) && !jb0[n - 1].reversed) {
                int reach = 0, max_t = -1, max_q = -1, mqe_t = -1;
                if (!(jb0[n - 1].flag & EZ_REFUSED)) { const ExtRes &e = rs0[n - 1]; reach = e.reach_end; max_t = e.max_t; max_q = e.max_q; mqe_t = e.mqe_t; }
                re1 = sr.re + (reach ? mqe_t + 1 : max_t + 1);
                qe1 = sr.qe + (reach ? sr.qe0 - sr.qe : max_q + 1);
            }
            o.cig_off = (int64_t)base; o.n_ops = total; o.dp_score = dp; o.rs1 = rs1; o.re1 = re1; o.qs1 = qs1; o.qe1 = qe1;
            o.has_p = any_ops || dropped; o.dropped = dropped;
            outs[ri] = o;
            FinJob f;
            f.cig_off = (int64_t)base; f.code_off = 0; f.n_cigar = total; f.read = sr.read; f.rid = sr.rid; f.rev = sr.rev;
            f.qs1 = qs1; f.rs1 = rs1; f.qspan = qe1 > qs1 ? qe1 - qs1 : 0; f.tspan = re1 > rs1 ? re1 - rs1 : 0;
            fin_jobs[ri] = f;
        }
        // ---- the two halves of a cut hit, measured for the host (mm_split_reg's mm_reg_set_coor calls): the anchors stay here ----
        split_n = __shfl(split_n, 0); split_slot = __shfl(split_slot, 0);
        if (split_n > 0) {
            const PlanReg pr = pregs[ri];
            const u128 *a = A + pr.a_off + pr.as;
            int ml[2] = {0, 0}, bl[2] = {0, 0};
            for (int j = lane; j < pr.cnt; j += 64) {
                const int h = j >= split_n;
                const u128 cur = a[j];
                const int span = (int)(cur.y >> 32 & 0xff);
                if (j == 0 || j == split_n) { ml[h] += span; bl[h] += span; }
                else {
                    const u128 prev = a[j - 1];
                    const int tl = (int32_t)cur.x - (int32_t)prev.x, ql = (int32_t)cur.y - (int32_t)prev.y;
                    bl[h] += tl > ql ? tl : ql;
                    ml[h] += tl > span && ql > span ? span : tl < ql ? tl : ql;
                }
            }
            for (int h = 0; h < 2; ++h)
                for (int d = 32; d; d >>= 1) { ml[h] += __shfl_xor(ml[h], d); bl[h] += __shfl_xor(bl[h], d); }
            if (lane == 0) {
                const u128 f = a[split_n], ll = a[split_n - 1];
                splits[split_slot] = SplitRec{f.x, f.y, ll.x, ll.y, ml[0], bl[0], ml[1], bl[1]};
            }
        }
    }
}

}  // namespace mpn
