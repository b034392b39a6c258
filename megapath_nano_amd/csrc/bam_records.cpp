// SAM lines -> BAM records (include/mpn_bam.h): the per-record work of the BAM writer in native code.  Written from the SAM/BAM
// specification (SAMv1 4.2) and htslib 1.13's choices where the specification leaves them open (smallest integer type for `i`
// tags, CG:B,I for CIGARs beyond 65535 operations); megapath_nano_amd/bam.py holds the same encoder in Python, and the tests
// compare the two byte for byte.
#include "mpn_common.h"
#include "../../include/mpn_bam.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

struct mpn_bam_encoder {
    std::unordered_map<std::string, int32_t> ref_id;
};

namespace {

struct Span { const char *p; int32_t n; };

inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    int s = 14, t = ((1 << 15) - 1) / 7;
    for (int l = 5; l > 0; --l) {
        if (beg >> s == end >> s) return t + (int)(beg >> s);
        s += 3;
        t -= 1 << (3 * (l - 1));
    }
    return 0;
}

inline bool parse_i64(Span s, long long *out) {
    if (s.n <= 0 || s.n > 20) return false;
    char buf[24];
    memcpy(buf, s.p, (size_t)s.n);
    buf[s.n] = 0;
    char *end = nullptr;
    *out = strtoll(buf, &end, 10);
    return end == buf + s.n;
}

template <typename T> inline void put(std::vector<uint8_t> &v, T x) { const size_t k = v.size(); v.resize(k + sizeof(T)); memcpy(v.data() + k, &x, sizeof(T)); }

const uint8_t *seq_table() {
    static uint8_t tab[256];
    static const bool init = []() {
        memset(tab, 15, sizeof(tab));
        const char *codes = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; ++i) { tab[(unsigned char)codes[i]] = (uint8_t)i; tab[(unsigned char)(codes[i] | 0x20)] = (uint8_t)i; }
        tab[(unsigned char)'='] = 0;
        return true;
    }();
    (void)init;
    return tab;
}

// one line -> record bytes (appended to rec); false + err on a malformed line
bool encode_line(const mpn_bam_encoder *e, const char *p, int32_t n, std::vector<uint8_t> &rec, int32_t *tid_o, int32_t *pos_o, int32_t *end_o,
                 int32_t *flag_o, std::string &err) {
    while (n > 0 && (p[n - 1] == '\n' || p[n - 1] == '\r')) --n;
    std::vector<Span> f;
    f.reserve(24);
    int32_t s0 = 0;
    for (int32_t i = 0; i <= n; ++i)
        if (i == n || p[i] == '\t') { f.push_back(Span{p + s0, i - s0}); s0 = i + 1; }
    if (f.size() < 11) { err = "SAM line with fewer than 11 fields"; return false; }
    long long flag, pos, mapq, pnext, tlen;
    if (!parse_i64(f[1], &flag) || !parse_i64(f[3], &pos) || !parse_i64(f[4], &mapq) || !parse_i64(f[7], &pnext) || !parse_i64(f[8], &tlen)) {
        err = "SAM line with a non-numeric FLAG/POS/MAPQ/PNEXT/TLEN";
        return false;
    }
    auto ref_of = [&](Span s) -> int32_t {
        if (s.n == 1 && s.p[0] == '*') return -1;
        const auto it = e->ref_id.find(std::string(s.p, (size_t)s.n));
        return it == e->ref_id.end() ? -1 : it->second;
    };
    const int32_t tid = ref_of(f[2]);
    const int32_t ntid = (f[6].n == 1 && f[6].p[0] == '=') ? tid : ref_of(f[6]);
    // CIGAR
    std::vector<uint32_t> cigar;
    int64_t ref_len = 0;
    if (!(f[5].n == 1 && f[5].p[0] == '*')) {
        uint64_t num = 0;
        for (int32_t i = 0; i < f[5].n; ++i) {
            const char c = f[5].p[i];
            if (c >= '0' && c <= '9') num = num * 10 + (uint64_t)(c - '0');
            else {
                const char *ops = "MIDNSHP=X", *w = strchr(ops, c);
                if (!w || c == 0) { err = "unknown CIGAR operation"; return false; }
                const uint32_t op = (uint32_t)(w - ops);
                cigar.push_back((uint32_t)num << 4 | op);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += (int64_t)num;
                num = 0;
            }
        }
    }
    const int64_t pos0 = pos - 1;
    const int64_t end0 = (ref_len > 0 && !(flag & 4)) ? pos0 + ref_len : pos0 + 1;
    const bool no_seq = f[9].n == 1 && f[9].p[0] == '*';
    const int32_t l_seq = no_seq ? 0 : f[9].n;
    // auxiliary fields
    std::vector<uint8_t> aux;
    for (size_t k = 11; k < f.size(); ++k) {
        const Span t = f[k];
        if (t.n < 5 || t.p[2] != ':' || t.p[4] != ':') { err = "malformed optional field"; return false; }
        const char typ = t.p[3];
        const Span val{t.p + 5, t.n - 5};
        aux.push_back((uint8_t)t.p[0]); aux.push_back((uint8_t)t.p[1]);
        if (typ == 'A') { aux.push_back('A'); aux.push_back(val.n > 0 ? (uint8_t)val.p[0] : 0); }
        else if (typ == 'i') {
            long long x;
            if (!parse_i64(val, &x)) { err = "malformed integer tag"; return false; }
            if (x >= 0) {
                if (x <= 0xff) { aux.push_back('C'); put<uint8_t>(aux, (uint8_t)x); }
                else if (x <= 0xffff) { aux.push_back('S'); put<uint16_t>(aux, (uint16_t)x); }
                else if (x <= 0xffffffffLL) { aux.push_back('I'); put<uint32_t>(aux, (uint32_t)x); }
                else { err = "integer tag out of range"; return false; }
            } else {
                if (x >= -0x80) { aux.push_back('c'); put<int8_t>(aux, (int8_t)x); }
                else if (x >= -0x8000) { aux.push_back('s'); put<int16_t>(aux, (int16_t)x); }
                else if (x >= -0x80000000LL) { aux.push_back('i'); put<int32_t>(aux, (int32_t)x); }
                else { err = "integer tag out of range"; return false; }
            }
        } else if (typ == 'f') {
            std::string sv(val.p, (size_t)val.n);
            aux.push_back('f'); put<float>(aux, (float)strtod(sv.c_str(), nullptr));   // (double, then rounded: what float(text) packed as <f gives)
        } else if (typ == 'Z' || typ == 'H') {
            aux.push_back((uint8_t)typ);
            aux.insert(aux.end(), (const uint8_t *)val.p, (const uint8_t *)val.p + val.n);
            aux.push_back(0);
        } else if (typ == 'B') {
            if (val.n < 1) { err = "malformed array tag"; return false; }
            const char sub = val.p[0];
            std::vector<std::string> items;
            int32_t a0 = 2;
            for (int32_t i = 2; i <= val.n; ++i)
                if (i == val.n || val.p[i] == ',') { if (i > a0 || i < val.n) items.emplace_back(val.p + a0, (size_t)(i - a0)); a0 = i + 1; }
            if (val.n <= 2) items.clear();
            aux.push_back('B'); aux.push_back((uint8_t)sub); put<uint32_t>(aux, (uint32_t)items.size());
            for (const std::string &it : items) {
                switch (sub) {
                case 'c': put<int8_t>(aux, (int8_t)atoll(it.c_str())); break;
                case 'C': put<uint8_t>(aux, (uint8_t)atoll(it.c_str())); break;
                case 's': put<int16_t>(aux, (int16_t)atoll(it.c_str())); break;
                case 'S': put<uint16_t>(aux, (uint16_t)atoll(it.c_str())); break;
                case 'i': put<int32_t>(aux, (int32_t)atoll(it.c_str())); break;
                case 'I': put<uint32_t>(aux, (uint32_t)atoll(it.c_str())); break;
                case 'f': put<float>(aux, (float)strtod(it.c_str(), nullptr)); break;
                default: err = "unknown array subtype"; return false;
                }
            }
        } else { err = "unknown tag type"; return false; }
    }
    if (cigar.size() > 65535) {
        aux.push_back('C'); aux.push_back('G'); aux.push_back('B'); aux.push_back('I');
        put<uint32_t>(aux, (uint32_t)cigar.size());
        const size_t k = aux.size();
        aux.resize(k + cigar.size() * 4);
        memcpy(aux.data() + k, cigar.data(), cigar.size() * 4);
        cigar.assign({(uint32_t)l_seq << 4 | 4u, (uint32_t)ref_len << 4 | 3u});
    }
    const int32_t l_name = f[0].n + 1;
    if (l_name > 255) { err = "QNAME longer than 254 characters"; return false; }
    rec.clear();
    rec.reserve(32 + (size_t)l_name + cigar.size() * 4 + (size_t)(l_seq + 1) / 2 + (size_t)l_seq + aux.size());
    put<int32_t>(rec, tid); put<int32_t>(rec, (int32_t)pos0); put<uint8_t>(rec, (uint8_t)l_name); put<uint8_t>(rec, (uint8_t)mapq);
    put<uint16_t>(rec, (uint16_t)reg2bin(pos0, end0)); put<uint16_t>(rec, (uint16_t)cigar.size()); put<uint16_t>(rec, (uint16_t)flag);
    put<int32_t>(rec, l_seq); put<int32_t>(rec, ntid); put<int32_t>(rec, (int32_t)(pnext - 1)); put<int32_t>(rec, (int32_t)tlen);
    rec.insert(rec.end(), (const uint8_t *)f[0].p, (const uint8_t *)f[0].p + f[0].n);
    rec.push_back(0);
    {
        const size_t k = rec.size();
        rec.resize(k + cigar.size() * 4);
        if (!cigar.empty()) memcpy(rec.data() + k, cigar.data(), cigar.size() * 4);
    }
    if (l_seq) {
        const uint8_t *tab = seq_table();
        const size_t k = rec.size();
        rec.resize(k + (size_t)(l_seq + 1) / 2);
        uint8_t *d = rec.data() + k;
        const unsigned char *sq = (const unsigned char *)f[9].p;
        for (int32_t i = 0; i + 1 < l_seq; i += 2) d[i >> 1] = (uint8_t)(tab[sq[i]] << 4 | tab[sq[i + 1]]);
        if (l_seq & 1) d[l_seq >> 1] = (uint8_t)(tab[sq[l_seq - 1]] << 4);
        const size_t kq = rec.size();
        rec.resize(kq + (size_t)l_seq);
        uint8_t *q = rec.data() + kq;
        if (f[10].n == 1 && f[10].p[0] == '*') memset(q, 0xff, (size_t)l_seq);
        else {
            if (f[10].n != l_seq) { err = "SEQ and QUAL of different lengths"; return false; }
            for (int32_t i = 0; i < l_seq; ++i) q[i] = (uint8_t)((unsigned char)f[10].p[i] - 33);
        }
    }
    rec.insert(rec.end(), aux.begin(), aux.end());
    *tid_o = tid; *pos_o = (int32_t)pos0; *end_o = (int32_t)end0; *flag_o = (int32_t)flag;
    return true;
}

}  // namespace

extern "C" mpn_bam_encoder *mpn_bam_encoder_create(const char *const *ref_names, int32_t n_ref) {
    if (n_ref < 0 || (n_ref > 0 && !ref_names)) { mpn::set_error("mpn_bam_encoder_create: bad arguments"); return nullptr; }
    mpn_bam_encoder *e = new mpn_bam_encoder();
    e->ref_id.reserve((size_t)n_ref * 2 + 1);
    for (int32_t i = 0; i < n_ref; ++i) e->ref_id.emplace(std::string(ref_names[i] ? ref_names[i] : ""), i);
    return e;
}
extern "C" void mpn_bam_encoder_destroy(mpn_bam_encoder *e) { delete e; }

extern "C" int64_t mpn_bam_encode(const mpn_bam_encoder *e, const char *text, const int64_t *line_off, const int32_t *line_len, int64_t n,
                                  uint8_t *out, int64_t out_cap, int64_t *rec_off, int32_t *tid, int32_t *pos0, int32_t *end0, int32_t *flag) {
    if (!e || n < 0 || (n > 0 && (!text || !line_off || !line_len || !rec_off || !tid || !pos0 || !end0 || !flag))) {
        mpn::set_error("mpn_bam_encode: null argument");
        return -1;
    }
    // lines are independent: threads take them in chunks, records are placed afterwards (sizes first, then one copy each)
    const int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>({16, (int64_t)std::thread::hardware_concurrency(), n / 256 + 1}));
    std::vector<std::vector<uint8_t>> recs((size_t)n);
    std::atomic<int64_t> next(0);
    std::atomic<int> failed(0);
    std::string err_text;
    std::mutex mu;
    auto work = [&]() {
        std::string err;
        for (;;) {
            const int64_t i0 = next.fetch_add(64);
            if (i0 >= n || failed) break;
            for (int64_t i = i0; i < std::min(n, i0 + 64); ++i)
                if (!encode_line(e, text + line_off[i], line_len[i], recs[(size_t)i], &tid[i], &pos0[i], &end0[i], &flag[i], err)) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!failed) err_text = err + " (line " + std::to_string(i) + " of the batch)";
                    failed = 1;
                    break;
                }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_thr; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed) { mpn::set_error("mpn_bam_encode: %s", err_text.c_str()); return -1; }
    int64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) { rec_off[i] = tot; tot += (int64_t)recs[(size_t)i].size(); }
    rec_off[n] = tot;
    if (tot > out_cap) return -3;
    for (int64_t i = 0; i < n; ++i) if (!recs[(size_t)i].empty()) memcpy(out + rec_off[i], recs[(size_t)i].data(), recs[(size_t)i].size());
    return tot;
}
