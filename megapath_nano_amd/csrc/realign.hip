// Amplicon realigner -- MI355X-native counterpart of ReAligner::AlignReads
// (/root/reference/bin/realignment/realign/realigner.cpp:88-117) behind the reference's own C entry points
// (realigner.cpp:854-869) and a batched form (include/mpn_realign.h).
//
//   GPU   realign_fast_kernel : the k-mer seeded Hamming placement of every read on every haplotype
//                               (BuildIndex + FastAlignReadsToHaplotype, realigner.cpp:147-230, 429-451)
//         ssw_* kernels       : haplotype -> reference and read -> haplotype Smith-Waterman (ssw_cpp.cpp:268-300)
//   host  the bookkeeping between them: per-haplotype coverage test and scores, CIGAR composition
//         (CalculateReadToRefAlignment, realigner.cpp:653-777), output strings.
//
// The reference finds placements through a hash of the reads' 32-mers: for haplotype position i and every read k-mer
// (read r, offset p) equal to the haplotype's k-mer at i it evaluates the placement start = max(0, i - p).  Here one
// wavefront takes a (haplotype, read) pair and walks its DIAGONALS d = i - p, a lane per diagonal: the run length of equal
// characters along the diagonal gives every exact 32-mer match (run >= 32), the first of them is the time i at which the
// reference discovers the placement, and the same walk counts the mismatches of the placement.  Negative diagonals all
// evaluate start 0 (the reference clamps), so their earliest match competes with diagonal 0's own.  Per haplotype the
// kernel also marks which positions hold a k-mer of ANY read (the reference's coverage test only runs at those).
#include "mpn_common.h"
#include "../../include/mpn_realign.h"
#include "../../include/mpn_ssw.h"

#include <algorithm>
#include <sstream>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <unordered_map>
#include <vector>

namespace mpn {

constexpr int RA_KMER = 32, RA_MAX_MM = 2;                      // realigner.cpp:66,68
constexpr int RA_MATCH = 4, RA_MISMATCH = 6, RA_GAP_O = 8, RA_GAP_E = 2;   // :70-73

struct FastHit { int32_t pair, start, mm, t, p; };

__global__ __launch_bounds__(256) void realign_fast_kernel(const uint8_t *__restrict__ text, const int64_t *__restrict__ read_off,
                                                           const int32_t *__restrict__ read_len, const int64_t *__restrict__ hap_off,
                                                           const int32_t *__restrict__ hap_len, const int32_t *__restrict__ pair_hap,
                                                           const int32_t *__restrict__ pair_read, int n_pairs,
                                                           const int64_t *__restrict__ hk_off, uint32_t *__restrict__ has_kmer,
                                                           FastHit *__restrict__ hits, unsigned int *__restrict__ n_hits, unsigned int cap) {
    __shared__ unsigned long long s_clamp[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int pair = blockIdx.x * 4 + wv; pair < n_pairs; pair += gridDim.x * 4) {
        const int hap = pair_hap[pair], rd = pair_read[pair];
        const int L = read_len[rd], HL = hap_len[hap];
        if (L <= RA_KMER || HL < RA_KMER) continue;           // reads of <= 32 bases are not indexed (realigner.cpp:436)
        const uint8_t *q = text + read_off[rd], *h = text + hap_off[hap];
        uint32_t *hk = has_kmer + hk_off[hap];
        if (lane == 0) s_clamp[wv] = ~0ULL;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int d_lo = -(L - RA_KMER), d_hi = HL - RA_KMER;  // diagonals that can hold a k-mer match
        for (int d0 = d_lo; d0 <= d_hi; d0 += 64) {
            const int d = d0 + lane;
            bool found = false;
            int t = 0, pf = 0, mm = 0;
            if (d <= d_hi) {
                const int p_lo = d < 0 ? -d : 0, p_hi = min(L, HL - d);
                int run = 0;
                for (int p = p_lo; p < p_hi; ++p) {
                    const uint8_t c1 = h[d + p], c2 = q[p];
                    if (c1 == c2) {
                        if (++run >= RA_KMER) {
                            const int p0 = p - (RA_KMER - 1), i0 = d + p0;
                            atomicOr(&hk[i0 >> 5], 1u << (i0 & 31));
                            if (!found) { found = true; t = i0; pf = p0; }
                        }
                    } else {
                        run = 0;
                        if (c1 != 'N' && c2 != 'N') ++mm;           // FastAlignStrings, realigner.cpp:241
                    }
                }
                if (d < 0 && found) atomicMin(&s_clamp[wv], (unsigned long long)t << 32 | (unsigned)pf);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (d >= 0 && d <= d_hi && d + L <= HL && mm <= RA_MAX_MM) {
                if (d == 0) {  // the clamped evaluations of the negative diagonals are evaluations of start 0 too
                    const unsigned long long c = s_clamp[wv];
                    if (c != ~0ULL) {
                        const int tc = (int)(c >> 32), pc = (int)(uint32_t)c;
                        if (!found || tc < t || (tc == t && pc < pf)) { found = true; t = tc; pf = pc; }
                    }
                }
                if (found) {
                    const unsigned int w = atomicAdd(n_hits, 1u);
                    if (w < cap) hits[w] = FastHit{pair, d, mm, t, pf};
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
enum { OP_UNSPEC = 0, OP_MATCH = 1, OP_INS = 2, OP_DEL = 3, OP_SKIP = 4, OP_SOFT = 5, OP_HARD = 6 };   // realigner.h:47-55
struct CigOp { int op, len; };

struct Translation {   // ssw_cpp.cpp:8-25 kBaseTranslation and :27-47 BuildSwScoreMatrix
    int8_t tr[128], mat[25];
    Translation() {
        memset(tr, 4, sizeof(tr));
        tr['A'] = tr['a'] = 0; tr['C'] = tr['c'] = 1; tr['G'] = tr['g'] = 2; tr['T'] = tr['t'] = 3; tr['U'] = tr['u'] = 0;
        for (int i = 0; i < 25; ++i) mat[i] = -RA_MISMATCH;
        for (int i = 0; i < 4; ++i) mat[i * 5 + i] = RA_MATCH;
    }
};
static const Translation g_tr;

static void parse_cigar(const std::string &c, std::vector<CigOp> &ops) {  // CigarStringToVector: (\d+)([XIDS=]), icase
    ops.clear();
    size_t i = 0;
    const size_t n = c.size();
    while (i < n) {
        if (c[i] < '0' || c[i] > '9') { ++i; continue; }
        size_t j = i;
        while (j < n && c[j] >= '0' && c[j] <= '9') ++j;
        if (j >= n) break;
        const char ch = c[j];
        const char up = (char)(ch >= 'a' && ch <= 'z' ? ch - 32 : ch);
        if (up == 'X' || up == 'I' || up == 'D' || up == 'S' || ch == '=') {
            // the regex takes the longest digit run that ends at the operator; atoi of a run that long saturates nowhere
            // near real CIGARs
            const int len = atoi(c.substr(i, j - i).c_str());
            int op = OP_UNSPEC;                                   // CigarOperationFromChar is case-sensitive
            if (ch == '=' || ch == 'X') op = OP_MATCH; else if (ch == 'S') op = OP_SOFT; else if (ch == 'D') op = OP_DEL; else if (ch == 'I') op = OP_INS;
            ops.push_back(CigOp{op, len});
            i = j + 1;
        } else i = j;  // digits followed by something else: the regex search moves on
    }
}

static std::string cigar_string(const std::vector<CigOp> &ops) {  // CigarVectorToString: a match prints 'X'
    std::string s;
    for (const CigOp &o : ops) {
        s += std::to_string(o.len);
        if (o.op == OP_MATCH) s += 'X'; else if (o.op == OP_INS) s += 'I'; else if (o.op == OP_DEL) s += 'D'; else if (o.op == OP_SOFT) s += 'S';
    }
    return s;
}

static int aligned_length(const std::vector<CigOp> &c) {
    int n = 0;
    for (const CigOp &o : c) if (o.op != OP_DEL) n += o.len;
    return n;
}

static void merge_op(int op, int len, int read_len, std::vector<CigOp> &c) {  // MergeCigarOp, realigner.cpp:551-574
    const int last = c.empty() ? OP_UNSPEC : c.back().op;
    const int before = aligned_length(c);
    const int n = op != OP_DEL ? std::min(len, read_len - before) : len;
    if (n <= 0 || before == read_len) return;
    if (op == last) c.back().len += n; else c.push_back(CigOp{op, n});
}

// A list consumed from the front where an element is only ever pushed back right after one was taken.
struct OpQueue {
    std::vector<CigOp> v;
    size_t head = 0;
    bool empty() const { return head >= v.size(); }
    CigOp &front() { return v[head]; }
    CigOp pop() { return v[head++]; }
    void push_front(const CigOp &o) { if (head > 0) v[--head] = o; else v.insert(v.begin(), o); }
};

// CalculateReadToRefAlignment (realigner.cpp:653-777) with LeftTrimHaplotypeToRefAlignment (:578-607)
static bool read_to_ref(int read_len, int position, const std::string &read_cigar, const std::vector<CigOp> &hap_ops, std::vector<CigOp> &out) {
    out.clear();
    OpQueue r2h, h2r;
    parse_cigar(read_cigar, r2h.v);
    h2r.v = hap_ops;
    {
        int cur = 0;
        while (cur != position) {
            if (h2r.empty()) return false;   // (the reference reads the front of an empty list here)
            const CigOp o = h2r.pop();
            if (o.op == OP_MATCH || o.op == OP_HARD || o.op == OP_SOFT || o.op == OP_INS) {
                if (o.len + cur > position) h2r.push_front(CigOp{o.op, o.len - (position - cur)});
                cur = std::min(o.len + cur, position);
            }
        }
        if (h2r.empty()) return false;
        if (h2r.front().op == OP_DEL) h2r.pop();
    }
    auto is_m = [](int op) { return op == OP_MATCH || op == OP_SOFT; };
    if (!r2h.empty() && r2h.front().op == OP_SOFT) { merge_op(OP_SOFT, r2h.front().len, read_len, out); r2h.pop(); }
    while ((!r2h.empty() || !h2r.empty()) && aligned_length(out) < read_len) {
        if (!r2h.empty() && h2r.empty()) { const CigOp o = r2h.pop(); merge_op(o.op, o.len, read_len, out); continue; }
        if (r2h.empty() && !h2r.empty()) break;
        CigOp a = r2h.pop(), b = h2r.pop();
        if (is_m(a.op) && is_m(b.op)) {
            const int n = std::min(a.len, b.len);
            merge_op(a.op == OP_SOFT || b.op == OP_SOFT ? OP_SOFT : OP_MATCH, n, read_len, out);
            a.len -= n; if (a.len > 0) r2h.push_front(a);
            b.len -= n; if (b.len > 0) h2r.push_front(b);
        } else if (a.op == OP_DEL && is_m(b.op)) {
            merge_op(OP_DEL, a.len, read_len, out);
            b.len -= a.len; if (b.len > 0) h2r.push_front(b);
        } else if (b.op == OP_DEL && is_m(a.op)) {
            merge_op(OP_DEL, b.len, read_len, out);
            if (a.len > 0) r2h.push_front(a);
        } else if (a.op == OP_DEL && b.op == OP_DEL) {
            merge_op(OP_DEL, a.len + b.len, read_len, out);
        } else if (a.op == OP_INS && is_m(b.op)) {
            a.len = std::min(read_len - aligned_length(out), a.len);
            merge_op(OP_INS, a.len, read_len, out);
            if (b.len > 0) h2r.push_front(b);
        } else if (b.op == OP_INS && is_m(a.op)) {
            b.len = std::min(read_len - aligned_length(out), b.len);
            merge_op(OP_INS, b.len, read_len, out);
            a.len = std::max(0, a.len - b.len);
            if (a.len > 0) r2h.push_front(a);
        } else if (a.op == OP_INS && b.op == OP_INS) {
            merge_op(OP_INS, a.len + b.len, read_len, out);
        } else { out.clear(); return true; }
    }
    return true;
}

static void positions_map(int hap_len, const std::string &cigar, std::vector<int> &pm) {  // SetPositionsMap, realigner.cpp:453-507
    pm.assign((size_t)hap_len, 0);
    std::vector<CigOp> raw;
    int shift = 0;
    size_t pos = 0;
    // the same tokens as parse_cigar, but by character ('=' and 'X' both advance with the current shift)
    size_t i = 0;
    const size_t n = cigar.size();
    while (i < n) {
        if (cigar[i] < '0' || cigar[i] > '9') { ++i; continue; }
        size_t j = i;
        while (j < n && cigar[j] >= '0' && cigar[j] <= '9') ++j;
        if (j >= n) break;
        const char op = cigar[j];
        const int len = atoi(cigar.substr(i, j - i).c_str());
        i = j + 1;
        if (op == '=' || op == 'X') { for (int k = 0; k < len && pos < pm.size(); ++k) pm[pos++] = shift; }
        else if (op == 'S') { shift -= len; for (int k = 0; k < len && pos < pm.size(); ++k) pm[pos++] = shift; }
        else if (op == 'D') shift += len;
        else if (op == 'I') { for (int k = 0; k < len && pos < pm.size(); ++k) { pm[pos++] = shift; --shift; } }
    }
}

struct SswOut { int score = 0, ref_begin = 0; std::string cigar; };

// ssw_cpp.cpp:50-203: ConvertAlignment + CalculateNumberMismatch on one result of the batched kernel
static void ssw_cpp_convert(const int8_t *q, int qlen, const int8_t *ref, uint16_t score1, int32_t ref_begin, int32_t q_begin, int32_t q_end,
                            const uint32_t *cig, int n_cig, SswOut &o) {
    o.score = score1; o.ref_begin = ref_begin; o.cigar.clear();
    if (n_cig <= 0) return;
    std::string &s = o.cigar;
    if (q_begin > 0) { s += std::to_string(q_begin); s += 'S'; }
    const int8_t *r = ref + ref_begin, *p = q + q_begin;
    bool in_m = false, in_x = false;
    uint32_t len_m = 0, len_x = 0;
    auto flush = [&]() {
        if (in_m) { s += std::to_string(len_m); s += '='; } else if (in_x) { s += std::to_string(len_x); s += 'X'; }
        in_m = in_x = false; len_m = len_x = 0;
    };
    for (int k = 0; k < n_cig; ++k) {
        const uint32_t n = cig[k] >> 4, op = cig[k] & 15;
        if (op == 0) {
            for (uint32_t j = 0; j < n; ++j) {
                if (*r != *p) {
                    if (in_m) { s += std::to_string(len_m); s += '='; }
                    len_m = 0; ++len_x; in_m = false; in_x = true;
                } else {
                    if (in_x) { s += std::to_string(len_x); s += 'X'; }
                    ++len_m; len_x = 0; in_m = true; in_x = false;
                }
                ++r; ++p;
            }
        } else if (op == 1) { p += n; flush(); s += std::to_string(n); s += 'I'; }
        else if (op == 2) { r += n; flush(); s += std::to_string(n); s += 'D'; }
    }
    flush();
    const int end = qlen - q_end - 1;
    if (end > 0) { s += std::to_string(end); s += 'S'; }
}

// one batched SSW call with the reference's fixed parameters (ssw_cpp.cpp: default Aligner 4/6/8/2, default Filter:
// flag 0x0f, score filter 0, distance filter 32767, maskLen = query length, score_size 2)
static int ssw_many(const std::vector<const std::vector<int8_t> *> &queries, const std::vector<const std::vector<int8_t> *> &refs, std::vector<SswOut> &out) {
    const int n = (int)queries.size();
    out.assign((size_t)n, SswOut());
    if (n == 0) return 0;
    std::vector<int8_t> qbuf, rbuf;
    std::vector<int64_t> qoff(n), roff(n), coff(n);
    std::vector<int32_t> qlen(n), rlen(n), mask(n), rb(n), re(n), qb(n), qe(n), re2(n), clen(n), status(n);
    std::vector<uint16_t> s1(n), s2(n);
    int64_t cap = 0;
    {
        // identical sequences (one reference per window, one haplotype for many reads) are stored once
        std::unordered_map<const void *, int64_t> seen_q, seen_r;
        auto place = [](std::unordered_map<const void *, int64_t> &seen, std::vector<int8_t> &buf, const std::vector<int8_t> *v) {
            auto it = seen.find((const void *)v);
            if (it != seen.end()) return it->second;
            const int64_t off = (int64_t)buf.size();
            buf.insert(buf.end(), v->begin(), v->end());
            seen.emplace((const void *)v, off);
            return off;
        };
        for (int i = 0; i < n; ++i) {
            qoff[i] = place(seen_q, qbuf, queries[i]); qlen[i] = (int32_t)queries[i]->size(); mask[i] = qlen[i];
            roff[i] = place(seen_r, rbuf, refs[i]); rlen[i] = (int32_t)refs[i]->size();
            // CIGAR entries <= alignment columns; a positive score (match 4, gap extension 2) keeps the deleted bases
            // below twice the query length
            cap += std::min<int64_t>((int64_t)qlen[i] + rlen[i], 4 * (int64_t)qlen[i]) + 8;
        }
    }
    std::vector<uint32_t> pool((size_t)cap);
    const int rc = mpn_ssw_align_batch(n, qbuf.data(), qoff.data(), qlen.data(), rbuf.data(), roff.data(), rlen.data(), g_tr.mat, 5, 2, RA_GAP_O,
                                       RA_GAP_E, 0x0f, 0, 32767, mask.data(), s1.data(), s2.data(), rb.data(), re.data(), qb.data(), qe.data(),
                                       re2.data(), pool.data(), cap, coff.data(), clen.data(), status.data());
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        if (status[i] != MPN_SSW_OK) { set_error("realign: SSW pair %d has status %d (the reference has no defined result here)", i, status[i]); return -4; }
        ssw_cpp_convert(queries[i]->data(), qlen[i], refs[i]->data(), s1[i], rb[i], qb[i], qe[i], pool.data() + coff[i], clen[i], out[(size_t)i]);
    }
    return 0;
}

struct ReadAln { int position = -1, score = 0; std::string cigar; };
struct HapAln {
    int index = 0, score = 0;
    std::vector<ReadAln> reads;
    std::string cigar;
    std::vector<CigOp> ops;
    int ref_pos = 0;
    bool is_ref = false;
    std::vector<int> pmap;
    bool operator<(const HapAln &o) const { return score < o.score; }   // realigner.h HaplotypeReadsAlignment::operator<
};

static int realign_windows(int n_win, const mpn_realign_window *W, int32_t *out_pos, std::vector<std::string> &out_cig) {
    // ---- flatten ----
    std::vector<uint8_t> text;
    std::vector<int64_t> read_off, hap_off, hk_off;
    std::vector<int32_t> read_len, hap_len, pair_hap, pair_read;
    std::vector<int> win_read0(n_win + 1, 0), win_hap0(n_win + 1, 0);
    int64_t hk_words = 0;
    for (int w = 0; w < n_win; ++w) {
        const mpn_realign_window &x = W[w];
        if (x.n_reads < 0 || x.n_haps < 0 || !x.reference) { set_error("realign: window %d is malformed", w); return -2; }
        win_read0[w + 1] = win_read0[w] + x.n_reads; win_hap0[w + 1] = win_hap0[w] + x.n_haps;
        for (int r = 0; r < x.n_reads; ++r) {
            const size_t L = strlen(x.seqs[r]);
            read_off.push_back((int64_t)text.size()); read_len.push_back((int32_t)L);
            text.insert(text.end(), x.seqs[r], x.seqs[r] + L);
        }
        for (int h = 0; h < x.n_haps; ++h) {
            const size_t L = strlen(x.haplotypes[h]);
            if (L < (size_t)RA_KMER) { set_error("realign: window %d haplotype %d has %zu bases (< 32: undefined in the reference)", w, h, L); return -2; }
            hap_off.push_back((int64_t)text.size()); hap_len.push_back((int32_t)L);
            text.insert(text.end(), x.haplotypes[h], x.haplotypes[h] + L);
            hk_off.push_back(hk_words); hk_words += ((int64_t)L + 31) / 32;
            for (int r = 0; r < x.n_reads; ++r) { pair_hap.push_back(win_hap0[w] + h); pair_read.push_back(win_read0[w] + r); }
        }
    }
    const int n_reads = win_read0[n_win], n_haps = win_hap0[n_win], n_pairs = (int)pair_hap.size();
    // ---- GPU: placements of every read on every haplotype ----
    std::vector<FastHit> hits;
    std::vector<uint32_t> has_kmer((size_t)hk_words, 0);
    if (n_pairs > 0) {
        hipStream_t st = 0;
        DevBuf<uint8_t> d_text;
        DevBuf<int64_t> d_roff, d_hoff, d_hk_off;
        DevBuf<int32_t> d_rlen, d_hlen, d_ph, d_pr;
        DevBuf<uint32_t> d_hk;
        DevBuf<unsigned int> d_n;
        DevBuf<FastHit> d_hits;
        text.push_back(0);
        // every (pair, start) is reported at most once, and a read fits a haplotype with <= 2 mismatches at a handful of starts:
        // the list is sized for 16 per pair and the kernel is run again in the (pathological) case that it overflows
        int64_t cap_max = 0;
        for (int p = 0; p < n_pairs; ++p) cap_max += std::max(0, hap_len[pair_hap[p]] - read_len[pair_read[p]] + 1);
        int64_t cap = std::min<int64_t>(std::max<int64_t>(16 * (int64_t)n_pairs, 1 << 16), std::max<int64_t>(cap_max, 1));
        if (cap_max > 0xfffffff0LL) { set_error("realign: batch too large (split it)"); return -2; }
        if (d_text.upload(text.data(), text.size(), st) || d_roff.upload(read_off.data(), read_off.size(), st) ||
            d_rlen.upload(read_len.data(), read_len.size(), st) || d_hoff.upload(hap_off.data(), hap_off.size(), st) ||
            d_hlen.upload(hap_len.data(), hap_len.size(), st) || d_ph.upload(pair_hap.data(), pair_hap.size(), st) ||
            d_pr.upload(pair_read.data(), pair_read.size(), st) || d_hk_off.upload(hk_off.data(), hk_off.size(), st) ||
            d_hk.alloc((size_t)hk_words) || d_hk.zero(st) || d_n.alloc(1))
            return -1;
        const int grid = std::max(1, std::min((n_pairs + 3) / 4, 256 * 32));
        unsigned int nh = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if (d_n.zero(st) || d_hits.alloc((size_t)cap)) return -1;
            hipLaunchKernelGGL(realign_fast_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t *)d_text.p, (const int64_t *)d_roff.p,
                               (const int32_t *)d_rlen.p, (const int64_t *)d_hoff.p, (const int32_t *)d_hlen.p, (const int32_t *)d_ph.p,
                               (const int32_t *)d_pr.p, n_pairs, (const int64_t *)d_hk_off.p, d_hk.p, d_hits.p, d_n.p, (unsigned int)cap);
            MPN_HIP_CHECK(hipGetLastError());
            if (d_n.download(&nh, 1, st)) return -1;
            MPN_HIP_CHECK(hipStreamSynchronize(st));
            if ((int64_t)nh <= cap) break;
            cap = nh;   // (the k-mer marks are idempotent; only the list is rebuilt)
        }
        if ((int64_t)nh > cap) { set_error("realign: placement list overflow (%u > %lld)", nh, (long long)cap); return -1; }
        hits.resize(nh);
        if (d_hits.download(hits.data(), nh, st) || d_hk.download(has_kmer.data(), (size_t)hk_words, st)) return -1;
        MPN_HIP_CHECK(hipStreamSynchronize(st));
    }
    // order of evaluation in the reference: haplotype position i, then the k-mer's occurrences by (read, offset)
    std::sort(hits.begin(), hits.end(), [&](const FastHit &a, const FastHit &b) {
        const int ha = pair_hap[a.pair], hb = pair_hap[b.pair];
        if (ha != hb) return ha < hb;
        if (a.t != b.t) return a.t < b.t;
        const int ra = pair_read[a.pair], rb = pair_read[b.pair];
        if (ra != rb) return ra < rb;
        return a.p < b.p;
    });
    // ---- per haplotype: coverage test, best placement per read, haplotype score (realigner.cpp:147-230) ----
    std::vector<HapAln> haps((size_t)n_haps);
    std::vector<std::vector<int8_t>> read_codes((size_t)n_reads), hap_codes((size_t)n_haps), ref_codes((size_t)n_win);
    auto translate = [](const char *s, size_t n, std::vector<int8_t> &out) {
        out.resize(n);
        for (size_t i = 0; i < n; ++i) out[i] = g_tr.tr[(unsigned char)s[i] & 127];
    };
    size_t hp = 0;
    for (int w = 0; w < n_win; ++w) {
        const mpn_realign_window &x = W[w];
        translate(x.reference, strlen(x.reference), ref_codes[(size_t)w]);
        for (int h = 0; h < x.n_haps; ++h) {
            const int gh = win_hap0[w] + h, HL = hap_len[gh];
            HapAln &A = haps[(size_t)gh];
            A.index = h;
            A.reads.assign((size_t)x.n_reads, ReadAln());
            const bool is_ref = strcmp(x.haplotypes[h], x.reference) == 0;
            const size_t h0 = hp;
            while (hp < hits.size() && pair_hap[hits[hp].pair] == gh) ++hp;
            bool dropped = false;
            {
                const uint32_t *hk = has_kmer.data() + hk_off[gh];
                size_t k = h0;
                int max_end = -1;
                for (int i = 0; i + RA_KMER <= HL && !is_ref; ++i) {
                    while (k < hp && hits[k].t <= i) { max_end = std::max(max_end, hits[k].start + read_len[pair_read[hits[k].pair]]); ++k; }
                    if (!(hk[i >> 5] >> (i & 31) & 1)) continue;      // no read holds this k-mer: the test is skipped (:181-183)
                    const bool before_suffix = x.ref_suffix > HL || i < HL - x.ref_suffix;   // (size_t arithmetic in the reference)
                    if (max_end <= i && i >= x.ref_prefix && before_suffix) { dropped = true; break; }
                }
            }
            if (!dropped) {
                for (size_t k = h0; k < hp; ++k) {
                    const FastHit &f = hits[k];
                    const int r = pair_read[f.pair] - win_read0[w], L = read_len[pair_read[f.pair]];
                    const int sc = (L - f.mm) * RA_MATCH - f.mm * RA_MISMATCH;
                    ReadAln &ra = A.reads[(size_t)r];
                    if (ra.score < sc) { A.score += sc - ra.score; ra.score = sc; ra.position = f.start; ra.cigar = std::to_string(L) + "="; }
                }
            }
            if (dropped || A.score == 0) { A.score = 0; A.reads.assign((size_t)x.n_reads, ReadAln()); }
            translate(x.haplotypes[h], (size_t)HL, hap_codes[(size_t)gh]);
        }
        for (int r = 0; r < x.n_reads; ++r) translate(x.seqs[r], (size_t)read_len[win_read0[w] + r], read_codes[(size_t)(win_read0[w] + r)]);
    }
    // ---- GPU: haplotype -> reference (realigner.cpp:325-349) ----
    {
        std::vector<const std::vector<int8_t> *> qs, rs;
        for (int w = 0; w < n_win; ++w)
            for (int h = 0; h < W[w].n_haps; ++h) { qs.push_back(&hap_codes[(size_t)(win_hap0[w] + h)]); rs.push_back(&ref_codes[(size_t)w]); }
        std::vector<SswOut> res;
        for (size_t i = 0; i < rs.size(); ++i) if (rs[i]->empty()) { set_error("realign: empty reference"); return -2; }
        if (ssw_many(qs, rs, res)) return -1;
        for (int gh = 0; gh < n_haps; ++gh) {
            HapAln &A = haps[(size_t)gh];
            const SswOut &a = res[(size_t)gh];
            if (a.score > 0) {
                A.is_ref = a.cigar == std::to_string(hap_len[gh]) + "=";
                A.cigar = a.cigar;
                parse_cigar(a.cigar, A.ops);
                A.ref_pos = a.ref_begin;
            }
            positions_map(hap_len[gh], A.cigar, A.pmap);
        }
    }
    // ---- GPU: reads without a placement -> every haplotype that kept a score (realigner.cpp:351-384) ----
    {
        double thr_d = RA_MATCH * 250 * 0.16934 - RA_MISMATCH * 250 * (1 - 0.16934);   // realigner.cpp:76-86 with set_options()
        int thr = (int)thr_d;
        if (thr < 0) thr = 1;
        std::vector<const std::vector<int8_t> *> qs, rs;
        std::vector<std::pair<int, int>> who;   // (global haplotype, read within its window)
        for (int w = 0; w < n_win; ++w)
            for (int r = 0; r < W[w].n_reads; ++r) {
                bool any = false;
                for (int h = 0; h < W[w].n_haps && !any; ++h) any = haps[(size_t)(win_hap0[w] + h)].reads[(size_t)r].score > 0;
                if (any) continue;
                if (read_len[win_read0[w] + r] == 0) continue;    // Aligner::Align returns false for an empty query
                for (int h = 0; h < W[w].n_haps; ++h) {
                    if (haps[(size_t)(win_hap0[w] + h)].score == 0) continue;
                    qs.push_back(&read_codes[(size_t)(win_read0[w] + r)]); rs.push_back(&hap_codes[(size_t)(win_hap0[w] + h)]);
                    who.emplace_back(win_hap0[w] + h, r);
                }
            }
        std::vector<SswOut> res;
        if (ssw_many(qs, rs, res)) return -1;
        for (size_t i = 0; i < who.size(); ++i) {
            ReadAln &ra = haps[(size_t)who[i].first].reads[(size_t)who[i].second];
            const SswOut &a = res[i];
            if (a.score > 0 && a.score >= thr && ra.score < a.score) { ra.score = a.score; ra.cigar = a.cigar; ra.position = a.ref_begin; }
        }
    }
    // ---- per window: haplotypes by score (std::sort, as the reference), best haplotype per read, CIGAR composition ----
    out_cig.assign((size_t)n_reads, std::string());
    std::vector<CigOp> ops;
    for (int w = 0; w < n_win; ++w) {
        const mpn_realign_window &x = W[w];
        std::sort(haps.begin() + win_hap0[w], haps.begin() + win_hap0[w + 1]);   // realigner.cpp:108
        for (int r = 0; r < x.n_reads; ++r) {
            const int gr = win_read0[w] + r;
            int best_score = 0;
            const HapAln *best = nullptr;
            for (int h = win_hap0[w]; h < win_hap0[w + 1]; ++h) {   // GetBestReadAlignment, realigner.cpp:516-538
                const int sc = haps[(size_t)h].reads[(size_t)r].score;
                if (sc > best_score || (best_score > 0 && sc == best_score && !haps[(size_t)h].is_ref)) { best_score = sc; best = &haps[(size_t)h]; }
            }
            out_pos[gr] = x.positions[r];
            out_cig[(size_t)gr] = x.cigars[r];
            if (!best) continue;
            const ReadAln &ra = best->reads[(size_t)r];
            const int p = ra.position;
            if (p < 0 || (size_t)p >= best->pmap.size()) continue;   // (outside the reference's defined behaviour)
            const int new_pos = x.ref_start + best->ref_pos + p + best->pmap[(size_t)p];
            if (!read_to_ref(read_len[gr], p, ra.cigar, best->ops, ops)) continue;
            if (!ops.empty()) { out_cig[(size_t)gr] = cigar_string(ops); out_pos[gr] = new_pos; }
        }
    }
    return 0;
}

}  // namespace mpn

using namespace mpn;

extern "C" {

int mpn_realign_batch(int32_t n_windows, const mpn_realign_window *windows, int32_t *out_position, char **out_cigar) {
    if (n_windows < 0 || (n_windows > 0 && (!windows || !out_position || !out_cigar))) { set_error("mpn_realign_batch: null argument"); return -2; }
    std::vector<std::string> cig;
    const int rc = realign_windows(n_windows, windows, out_position, cig);
    if (rc) return rc;
    for (size_t i = 0; i < cig.size(); ++i) {
        out_cigar[i] = (char *)malloc(cig[i].size() + 1);
        if (!out_cigar[i]) { for (size_t k = 0; k < i; ++k) free(out_cigar[k]); set_error("mpn_realign_batch: out of memory"); return -1; }
        memcpy(out_cigar[i], cig[i].c_str(), cig[i].size() + 1);
    }
    return 0;
}

void mpn_realign_free_cigars(char **cigars, int64_t n) {
    if (!cigars) return;
    for (int64_t i = 0; i < n; ++i) { free(cigars[i]); cigars[i] = nullptr; }
}

struct_str_arr *realign_reads(char *seqs[], int *positions, char *cigars[], char *reference, char *haplotypes, int ref_start,
                              int ref_prefix, int ref_suffix, int read_size) {
    if (read_size < 0 || read_size > 1000 || !reference || !haplotypes || (read_size > 0 && (!seqs || !positions || !cigars))) {
        fprintf(stderr, "realign_reads: invalid arguments (read_size %d; the reference's arrays hold 1000 reads)\n", read_size);
        return nullptr;
    }
    std::vector<std::string> hap_s;
    { std::istringstream in(haplotypes); std::string t; while (in >> t) hap_s.push_back(t); }   // realigner.cpp:787-792
    std::vector<const char *> hap_p;
    for (const std::string &s : hap_s) hap_p.push_back(s.c_str());
    std::vector<int32_t> pos(positions, positions + read_size);
    mpn_realign_window w;
    w.n_reads = read_size; w.seqs = seqs; w.positions = pos.data(); w.cigars = cigars; w.reference = reference;
    w.n_haps = (int32_t)hap_p.size(); w.haplotypes = hap_p.data(); w.ref_start = ref_start; w.ref_prefix = ref_prefix; w.ref_suffix = ref_suffix;
    std::vector<int32_t> out_pos((size_t)read_size);
    std::vector<std::string> cig;
    if (realign_windows(1, &w, out_pos.data(), cig)) { fprintf(stderr, "realign_reads: %s\n", mpn_last_error()); return nullptr; }
    struct_str_arr *res = new struct_str_arr();
    for (int i = 0; i < read_size; ++i) {
        res->cigar_string[i] = new char[cig[(size_t)i].size() + 1];
        strcpy(res->cigar_string[i], cig[(size_t)i].c_str());
        res->position[i] = out_pos[(size_t)i];
    }
    return res;
}

void free_memory(struct_str_arr *pointer, int size) {
    if (!pointer) return;
    for (int i = 0; i < size && i < 1000; ++i) delete[] pointer->cigar_string[i];
    delete pointer;
}

}  // extern "C"
