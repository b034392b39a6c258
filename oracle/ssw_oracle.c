/*
 * ssw_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle, never shipped, never measured
 * as the product).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this.
 *
 * Scalar CPU restatement of the SSW local aligner that MegaPath-Nano vendors at
 *   /root/reference/bin/realignment/realign/ssw.c  (ssw_init :733, ssw_align :762,
 *   sw_sse2_byte :123, sw_sse2_word :354, banded_sw :532, seq_reverse :720)
 * and calls through ctypes from bin/realignment/pyssw.py:30-48,137-147.
 *
 * The reference is SSE2 "striped" code.  This file does NOT re-implement the stripes; it
 * states, cell by cell, the values the striped code ends up computing, which are not quite
 * the textbook recurrences:
 *
 *   - the read is padded to P = segLen*LANES rows (LANES = 16 in the 8-bit pass, 8 in the
 *     16-bit pass, segLen = ceil(readLen/LANES)); padded rows score 0 against every
 *     reference base (ssw.c:108 `bias`, :346 `0`) and do take part in column maxima;
 *   - inside a column the vertical gap state F is first computed per stripe segment
 *     (restarting from 0 at every multiple of segLen, ssw.c:185,:219-220), giving H';
 *     E for the next column is derived from H' (ssw.c:213-216), i.e. BEFORE the lazy-F
 *     correction (ssw.c:226 comment); the lazy-F loop (:240-258 / :451-462) then lifts
 *     H' to H = max(H', F carried across segments) and only H feeds the next diagonal;
 *   - the best end is the first column (in scan order) whose column max strictly exceeds
 *     the running max (:269-272 / :474-476), end_read the smallest row holding that max
 *     (:285-293); 2nd best excludes +-maskLen with the byte pass skipping column `edge`
 *     (:318) and the word pass not (:520);
 *   - 8-bit pass reports 255 as soon as max+bias >= 255 (:271,:302) and the caller reruns
 *     the 16-bit pass (:789-792).
 *
 * banded_sw (traceback) is restated with the same rolling-row storage so that the
 * reference's out-of-band reads (including the `edge` zeroing at ssw.c:579-580 which can
 * clear a valid cell of the previous row) are reproduced bit for bit.
 *
 * Pinned against oracle/_ref/libssw.so (the reference's own ssw.c compiled in place) by
 * tests/test_ssw_oracle.py and against tests/golden/ssw_golden.json.
 *
 * Domain: gap_open > gap_extend (every reference call site uses 8/2: pyssw.py:52,
 * fast_align_reads2ref.py:4-9).  For gap_open <= gap_extend the reference's lazy-F loops exit
 * early in a way that depends on which SSE lanes are still "alive" (ssw.c:240,:460), which this
 * cell-wise statement does not model: status 3 is returned instead of a wrong answer.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint16_t score1, score2;
    int32_t ref_begin1, ref_end1, read_begin1, read_end1, ref_end2;
    int32_t cigar_len;
    int32_t status; /* 0 ok; 1 = reference would print error & return NULL; 2 = reference behaviour undefined */
} ssw_oracle_result;

typedef struct { int32_t score, ref, read, score2, ref2; } pass_out;

static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }
static inline int32_t sub0(int32_t a, int32_t b) { return a > b ? a - b : 0; }

/* One score pass.  lanes = 16 (byte) or 8 (word).  Follows ssw.c:123-328 / :354-530. */
static void score_pass(const int8_t *ref, int ref_dir, int32_t refLen, const int8_t *read, int32_t readLen,
                       const int8_t *mat, int32_t n, int32_t gapO, int32_t gapE, int lanes, int32_t bias,
                       int32_t terminate, int32_t maskLen, pass_out *out)
{
    const int is_byte = lanes == 16;
    int32_t segLen = (readLen + lanes - 1) / lanes, P = segLen * lanes;
    int32_t *H = (int32_t *)calloc(P > 0 ? P : 1, 4), *Hn = (int32_t *)calloc(P > 0 ? P : 1, 4);
    int32_t *E = (int32_t *)calloc(P > 0 ? P : 1, 4), *Hmax = (int32_t *)calloc(P > 0 ? P : 1, 4);
    int32_t *maxColumn = (int32_t *)calloc(refLen > 0 ? refLen : 1, 4);
    int32_t max = 0, end_read = readLen - 1, end_ref = is_byte ? -1 : 0; /* ssw.c:145 vs :371 */
    int32_t i, j, begin = 0, end = refLen, step = 1, overflow = 0;
    if (ref_dir == 1) begin = refLen - 1, end = -1, step = -1;

    for (i = begin; i != end; i += step) {
        int32_t colmax = 0, fseg = 0, f = 0;
        const int8_t *mrow = mat + (int32_t)ref[i] * n;
        /* inner pass: H' with segment-local F; E(next column) from H' */
        for (j = 0; j < P; ++j) {
            int32_t s = j < readLen ? mrow[read[j]] : 0;
            int32_t h = (j > 0 ? H[j - 1] : 0) + s, t;
            if (j % segLen == 0) fseg = 0;
            if (h < 0) h = 0;
            h = imax(h, E[j]);
            h = imax(h, fseg);
            Hn[j] = h;
            t = sub0(h, gapO);
            E[j] = imax(sub0(E[j], gapE), t);
            fseg = imax(sub0(fseg, gapE), t);
        }
        /* lazy-F: lift by the F carried across segment boundaries (opened from H' only) */
        for (j = 0; j < P; ++j) {
            int32_t hp = Hn[j];
            if (f > hp) Hn[j] = f;
            if (Hn[j] > colmax) colmax = Hn[j];
            f = imax(sub0(f, gapE), sub0(hp, gapO));
        }
        { int32_t *tmp = H; H = Hn; Hn = tmp; }
        if (colmax > max) {
            max = colmax;
            if (is_byte && max + bias >= 255) { overflow = 1; break; } /* ssw.c:271 */
            end_ref = i;
            memcpy(Hmax, H, (size_t)P * 4);
        }
        maxColumn[i] = colmax;
        if (colmax == terminate) break;
    }
    for (j = 0; j < P; ++j)
        if (Hmax[j] == max && j < end_read) end_read = j;

    out->score = overflow ? 255 : max;
    out->ref = end_ref;
    out->read = end_read;
    out->score2 = 0;
    out->ref2 = 0;
    {
        int32_t edge = (end_ref - maskLen) > 0 ? (end_ref - maskLen) : 0;
        for (i = 0; i < edge; ++i)
            if (maxColumn[i] > out->score2) out->score2 = maxColumn[i], out->ref2 = i;
        edge = (end_ref + maskLen) > refLen ? refLen : (end_ref + maskLen);
        for (i = is_byte ? edge + 1 : edge; i < refLen; ++i) /* ssw.c:318 vs :520 */
            if (maxColumn[i] > out->score2) out->score2 = maxColumn[i], out->ref2 = i;
    }
    free(H); free(Hn); free(E); free(Hmax); free(maxColumn);
}

/* Banded traceback DP, ssw.c:532-718.  Returns cigar length, or -1 (reference prints
 * "Trace back error" and returns 0), or -2 (band never reaches the score / undefined). */
static int32_t banded_traceback(const int8_t *ref, const int8_t *read, int32_t refLen, int32_t readLen, int32_t score,
                                int32_t gapO, int32_t gapE, int32_t bw, const int8_t *mat, int32_t n,
                                uint32_t *cig, int32_t cig_cap)
{
    int32_t max = 0, width, width_d, i, j, rounds = 0;
    int32_t *hb = 0, *eb = 0, *hc = 0;
    int8_t *dir = 0;
    do {
        int64_t dsz;
        width = bw * 2 + 3, width_d = bw * 2 + 1;
        dsz = (int64_t)width_d * readLen * 3;
        if (dsz > ((int64_t)1 << 31) - 1 || ++rounds > 40) { free(hb); free(eb); free(hc); free(dir); return -2; }
        free(hb); free(eb); free(hc); free(dir);
        hb = (int32_t *)calloc((size_t)(width + 2), 4);
        eb = (int32_t *)calloc((size_t)(width + 2), 4);
        hc = (int32_t *)calloc((size_t)(width + 2), 4);
        dir = (int8_t *)calloc((size_t)dsz + 16, 1);
        for (j = 1; j < width - 1; ++j) hb[j] = 0;
        for (i = 0; i < readLen; ++i) {
            int32_t x = i - bw > 0 ? i - bw : 0, xp = i - 1 - bw > 0 ? i - 1 - bw : 0;
            int32_t beg = x, end = refLen - 1 < i + bw ? refLen - 1 : i + bw;
            int32_t edge = end + 1 < width - 1 ? end + 1 : width - 1, f = 0, u = 0;
            int8_t *dl = dir + (int64_t)width_d * i * 3;
            hb[0] = eb[0] = hb[edge] = eb[edge] = hc[0] = 0; /* ssw.c:580, absolute `end` used as index */
            for (j = beg; j <= end; ++j) {
                int32_t up = j - xp + 1, lf, dg = j - 1 - xp + 1, c = (j - x) * 3;
                int32_t t1, t2, e1, f1, ev;
                u = j - x + 1; lf = u - 1;
                t1 = i == 0 ? -gapO : hb[up] - gapO;
                t2 = i == 0 ? -gapE : eb[up] - gapE;
                ev = t1 > t2 ? t1 : t2;
                eb[u] = ev;
                dl[c] = t1 > t2 ? 3 : 2;
                t1 = hc[lf] - gapO;
                t2 = f - gapE;
                f = t1 > t2 ? t1 : t2;
                dl[c + 1] = t1 > t2 ? 5 : 4;
                e1 = ev > 0 ? ev : 0;
                f1 = f > 0 ? f : 0;
                t1 = e1 > f1 ? e1 : f1;
                t2 = hb[dg] + mat[(int32_t)ref[j] * n + read[i]];
                hc[u] = t1 > t2 ? t1 : t2;
                if (hc[u] > max) max = hc[u];
                if (t1 <= t2) dl[c + 2] = 1;
                else dl[c + 2] = e1 > f1 ? dl[c] : dl[c + 1];
            }
            for (j = 1; j <= u; ++j) hb[j] = hc[j];
        }
        bw *= 2;
    } while (max < score);
    bw /= 2;
    free(hb); free(eb); free(hc);

    /* traceback, ssw.c:618-697 */
    {
        int32_t l = 0, e = 0, plane = 2, ncig = 0, ok = 1;
        char op = 'M', prev = 'M';
        uint32_t *tmp = (uint32_t *)malloc((size_t)(readLen + refLen + 4) * 4);
        i = readLen - 1; j = refLen - 1;
        while (i > 0) {
            int32_t x = i - bw > 0 ? i - bw : 0, cj = j - x, d;
            if (j < 0 || cj < 0 || cj >= width_d) { ok = 0; break; } /* reference would read outside its row */
            {
                int32_t hi = refLen - 1 < i + bw ? refLen - 1 : i + bw;
                if (j > hi) { ok = 0; break; }
            }
            d = dir[(int64_t)width_d * i * 3 + cj * 3 + plane];
            switch (d) {
            case 1: --i; --j; plane = 2; op = 'M'; break;
            case 2: --i; plane = 0; op = 'I'; break;
            case 3: --i; plane = 2; op = 'I'; break;
            case 4: --j; plane = 1; op = 'D'; break;
            case 5: --j; plane = 2; op = 'D'; break;
            default: free(tmp); free(dir); return -1;
            }
            if (op == prev) ++e;
            else {
                tmp[l++] = ((uint32_t)e << 4) | (prev == 'M' ? 0u : prev == 'I' ? 1u : 2u);
                prev = op; e = 1;
            }
        }
        if (!ok) { free(tmp); free(dir); return -2; }
        if (op == 'M') tmp[l++] = ((uint32_t)(e + 1) << 4);
        else {
            tmp[l++] = ((uint32_t)e << 4) | (op == 'I' ? 1u : 2u);
            tmp[l++] = (1u << 4);
        }
        ncig = l;
        for (i = 0; i < ncig && i < cig_cap; ++i) cig[i] = tmp[ncig - 1 - i];
        free(tmp); free(dir);
        return ncig;
    }
}

/* ssw_init + ssw_align in one call (ssw.c:733-852).  cigar written to cigar_buf (<= cigar_cap ops). */
int ssw_oracle_align(const int8_t *read, int32_t readLen, const int8_t *mat, int32_t n, int8_t score_size,
                     const int8_t *ref, int32_t refLen, uint8_t gapO, uint8_t gapE, uint8_t flag,
                     uint16_t filters, int32_t filterd, int32_t maskLen,
                     ssw_oracle_result *r, uint32_t *cigar_buf, int32_t cigar_cap)
{
    pass_out b;
    int32_t bias = 0, i, word = 0;
    int have_byte = score_size == 0 || score_size == 2, have_word = score_size == 1 || score_size == 2;
    memset(r, 0, sizeof(*r));
    r->ref_begin1 = -1; r->read_begin1 = -1;
    if (gapO <= gapE) { r->status = 3; return 3; }
    if (readLen <= 0 || refLen < 0) { r->status = 2; return 2; } /* ssw.c:189 indexes pvHStore[segLen-1] */
    if (have_byte) {
        for (i = 0; i < n * n; ++i) if (mat[i] < bias) bias = mat[i];
        bias = abs(bias);
        bias &= 0xff; /* stored in a uint8_t (ssw.c:85) */
    }
    if (have_byte) {
        score_pass(ref, 0, refLen, read, readLen, mat, n, gapO, gapE, 16, bias, 255 /* (uint8_t)-1 */, maskLen, &b);
        if (have_word && b.score == 255) {
            score_pass(ref, 0, refLen, read, readLen, mat, n, gapO, gapE, 8, 0, 65535 /* (uint16_t)-1 */, maskLen, &b);
            word = 1;
        } else if (b.score == 255) { r->status = 1; return 1; }
    } else if (have_word) {
        score_pass(ref, 0, refLen, read, readLen, mat, n, gapO, gapE, 8, 0, 65535, maskLen, &b);
        word = 1;
    } else { r->status = 1; return 1; }
    r->score1 = (uint16_t)b.score;
    r->ref_end1 = b.ref;
    r->read_end1 = b.read;
    if (maskLen >= 15) { r->score2 = (uint16_t)b.score2; r->ref_end2 = b.ref2; }
    else { r->score2 = 0; r->ref_end2 = -1; }
    if (flag == 0 || (flag == 2 && r->score1 < filters)) return 0;

    {   /* reverse pass for the begin position, ssw.c:820-832 */
        int32_t rl = r->read_end1 + 1;
        int8_t *rev;
        pass_out rb;
        if (rl <= 0 || r->ref_end1 + 1 < 0) { r->status = 2; return 2; }
        rev = (int8_t *)malloc((size_t)rl);
        for (i = 0; i < rl; ++i) rev[i] = read[rl - 1 - i];
        score_pass(ref, 1, r->ref_end1 + 1, rev, rl, mat, n, gapO, gapE, word ? 8 : 16, word ? 0 : bias,
                   r->score1, maskLen, &rb);
        free(rev);
        r->ref_begin1 = rb.ref;
        r->read_begin1 = r->read_end1 - rb.read;
    }
    if ((7 & flag) == 0 || ((2 & flag) != 0 && r->score1 < filters) ||
        ((4 & flag) != 0 && (r->ref_end1 - r->ref_begin1 > filterd || r->read_end1 - r->read_begin1 > filterd)))
        return 0;
    {
        int32_t rl = r->ref_end1 - r->ref_begin1 + 1, ql = r->read_end1 - r->read_begin1 + 1, nc;
        if (r->ref_begin1 < 0 || r->read_begin1 < 0 || rl <= 0 || ql <= 0) { r->status = 2; return 2; }
        nc = banded_traceback(ref + r->ref_begin1, read + r->read_begin1, rl, ql, r->score1, gapO, gapE,
                              abs(rl - ql) + 1, mat, n, cigar_buf, cigar_cap);
        if (nc == -1) { r->status = 1; return 1; }
        if (nc < 0) { r->status = 2; return 2; }
        r->cigar_len = nc;
    }
    return 0;
}
