"""Independent checker of the mapper's PAF / SAM output (test infrastructure).

Written from the PAF and SAM format descriptions and from the published meaning of minimap2's tags (minimap2.1 man page:
NM, ms, AS, nn, tp, cm, s1, s2, de, rl, cg; SAM flags, clipping, SA), NOT from oracle/mm2_oracle.c or csrc/align.hip: it
recomputes every per-alignment number from the CIGAR and the two sequences, so that a misreading shared by the product's
host code and the oracle (which restate the same upstream functions) cannot pass unnoticed.
"""
import re

COMP = {ord(a): b for a, b in zip('ACGTNacgtn', 'TGCANtgcan')}
CIG_RE = re.compile(r'(\d+)([MIDNSHP=X])')


def revcomp(s):
    return s.translate(COMP)[::-1] if isinstance(s, str) else bytes(s).decode().translate(COMP)[::-1]


def parse_cigar(s):
    return [(int(n), op) for n, op in CIG_RE.findall(s)]


def base_code(c):
    return 'ACGT'.find(c.upper()) if c.upper() in 'ACGT' else (3 if c.upper() == 'U' else 4)


class Scoring:
    def __init__(self, a=2, b=4, q=4, e=2, q2=24, e2=1, sc_ambi=1):
        self.a, self.b, self.q, self.e, self.q2, self.e2, self.sc_ambi = a, b, q, e, q2, e2, sc_ambi

    def gap(self, length):
        return min(self.q + self.e * length, self.q2 + self.e2 * length)


def walk(cigar, qseq, tseq, sc):
    """Recompute the numbers of an alignment whose CIGAR (M/I/D only) spans qseq and tseq completely.
    -> dict(n_match, n_mismatch, n_ambi, n_ins, n_del, n_gapo, score (dual affine), ms (best local segment, first gap model))"""
    qi = ti = 0
    n_match = n_mis = n_ambi = n_ins = n_del = n_gapo = n_col = 0
    score, run, best = 0, 0, 0
    for n, op in cigar:
        if op == 'M':
            for k in range(n):
                cq, ct = base_code(qseq[qi + k]), base_code(tseq[ti + k])
                if cq > 3 or ct > 3:
                    n_ambi += 1
                    s = -sc.sc_ambi
                elif cq == ct:
                    n_match += 1
                    s = sc.a
                else:
                    n_mis += 1
                    s = -sc.b
                score += s
                run = max(0, run + s)
                best = max(best, run)
            qi += n
            ti += n
            n_col += n
        elif op in 'ID':
            n_gapo += 1
            score -= sc.gap(n)
            run = max(0, run - (sc.q + sc.e * n))
            if op == 'I':
                n_ambi_gap = sum(1 for k in range(n) if base_code(qseq[qi + k]) > 3)
                n_ins += n - n_ambi_gap
                n_ambi += n_ambi_gap
                qi += n
            else:
                n_ambi_gap = sum(1 for k in range(n) if base_code(tseq[ti + k]) > 3)
                n_del += n - n_ambi_gap
                n_ambi += n_ambi_gap
                ti += n
        else:
            raise AssertionError(f'unexpected CIGAR operation {op} inside an alignment')
    assert qi == len(qseq) and ti == len(tseq), 'CIGAR does not span the aligned intervals'
    return dict(n_match=n_match, n_mismatch=n_mis, n_ambi=n_ambi, n_ins=n_ins, n_del=n_del, n_gapo=n_gapo, n_col=n_col, score=score, ms=best)


_CODE = None


def _codes(seq):
    """ASCII (str / bytes) -> numpy codes 0..3, 4 for anything else (U counts as T)"""
    import numpy as np
    global _CODE
    if _CODE is None:
        _CODE = np.full(256, 4, dtype=np.int8)
        for k, ch in enumerate('ACGT'):
            _CODE[ord(ch)] = _CODE[ord(ch.lower())] = k
        _CODE[ord('U')] = _CODE[ord('u')] = 3
    b = seq.encode() if isinstance(seq, str) else bytes(seq)
    return _CODE[np.frombuffer(b, dtype=np.uint8)]


def walk_np(cigar, qseq, tseq, sc):
    """walk() with numpy: the same numbers from the same definitions, for the checks that run over thousands of
    multi-kilobase alignments (bench.py's correctness block).  The running local score `run = max(0, run + s)` is the
    Lindley recursion: run_t = S_t - min(0, min_{u <= t} S_u) over the prefix sums S of the per-event scores."""
    import numpy as np
    cq_all, ct_all = _codes(qseq), _codes(tseq)
    n_ops = len(cigar)
    lens = np.fromiter((n for n, _ in cigar), dtype=np.int64, count=n_ops)
    ops = np.fromiter((('MID'.index(op) if op in 'MID' else 9) for _, op in cigar), dtype=np.int8, count=n_ops)
    if (ops == 9).any():
        raise AssertionError('unexpected CIGAR operation inside an alignment')
    qadv = np.where(ops != 2, lens, 0)
    tadv = np.where(ops != 1, lens, 0)
    assert int(qadv.sum()) == len(cq_all) and int(tadv.sum()) == len(ct_all), 'CIGAR does not span the aligned intervals'
    qstart, tstart = np.cumsum(qadv) - qadv, np.cumsum(tadv) - tadv
    # events in alignment order: one per M column, one per gap
    n_ev = np.where(ops == 0, lens, 1)
    ev_start = np.cumsum(n_ev) - n_ev
    total = int(n_ev.sum())
    ev_score = np.zeros(total, dtype=np.int64)
    m = ops == 0
    rep = np.repeat(np.arange(n_ops)[m], lens[m])
    within = np.arange(int(lens[m].sum())) - np.repeat(np.cumsum(lens[m]) - lens[m], lens[m])
    cq, ct = cq_all[qstart[rep] + within], ct_all[tstart[rep] + within]
    amb = (cq > 3) | (ct > 3)
    eq = (cq == ct) & ~amb
    s_m = np.where(amb, -sc.sc_ambi, np.where(eq, sc.a, -sc.b)).astype(np.int64)
    ev_score[ev_start[rep] + within] = s_m
    g = ~m
    glen = lens[g]
    gap_cost = np.minimum(sc.q + sc.e * glen, sc.q2 + sc.e2 * glen)
    score = int(s_m.sum()) - int(gap_cost.sum())
    ev_score[ev_start[g]] = -(sc.q + sc.e * glen)
    S = np.cumsum(ev_score)
    floor = np.minimum.accumulate(np.minimum(S, 0))
    best = int((S - floor).max()) if total else 0
    # ambiguous bases inside gaps
    qa, ta = np.concatenate([[0], np.cumsum(cq_all > 3)]), np.concatenate([[0], np.cumsum(ct_all > 3)])
    ins, dele = ops == 1, ops == 2
    amb_ins = (qa[qstart[ins] + lens[ins]] - qa[qstart[ins]]).sum() if ins.any() else 0
    amb_del = (ta[tstart[dele] + lens[dele]] - ta[tstart[dele]]).sum() if dele.any() else 0
    return dict(n_match=int(eq.sum()), n_mismatch=int((~eq & ~amb).sum()), n_ambi=int(amb.sum() + amb_ins + amb_del),
                n_ins=int(lens[ins].sum() - amb_ins), n_del=int(lens[dele].sum() - amb_del), n_gapo=int(g.sum()),
                n_col=int(lens[m].sum()), score=score, ms=best)


def tags_of(fields):
    out = {}
    for f in fields:
        t, ty, v = f.split(':', 2)
        out[t] = int(v) if ty == 'i' else float(v) if ty == 'f' else v
    return out


def check_paf(paf_text, reads, genomes, best_n=5, with_cigar=True, sc=None, stats=None, fast=False):
    """reads: dict name -> sequence (str); genomes: dict name -> sequence (str), or any mapping whose values support len()
    and slicing (bench.py hands over slices fetched from the index in HBM).  Raises AssertionError on any violation.
    stats (optional dict) receives counters: lines, as_equal (AS equals the CIGAR's dual-affine score), primaries.
    fast: recompute with walk_np (numpy) instead of the per-base Python loop."""
    sc = sc or Scoring()
    stats = stats if stats is not None else {}
    per_read = {}
    for line in paf_text.splitlines():
        f = line.split('\t')
        assert len(f) >= 12, line[:80]
        name, qlen, qs, qe, strand, tname, tlen, ts, te, mlen, blen, mapq = f[0], int(f[1]), int(f[2]), int(f[3]), f[4], f[5], int(f[6]), int(f[7]), int(f[8]), int(f[9]), int(f[10]), int(f[11])
        assert name in reads and qlen == len(reads[name]) and 0 <= qs < qe <= qlen, line[:80]
        assert tname in genomes and tlen == len(genomes[tname]) and 0 <= ts < te <= tlen and strand in '+-', line[:80]
        assert 0 <= mapq <= 60
        t = tags_of(f[12:])
        assert t['tp'] in ('P', 'S', 'I', 'i') and t['rl'] >= 0
        # (an inversion hit is made from a local alignment, not from a chain: it carries no minimizers and no chaining score)
        assert (t['cm'] >= 1 and t['s1'] > 0) or (t['tp'] in 'Ii' and t['cm'] == 0 and t['s1'] == 0), line[:120]
        assert ('s2' in t) == (t['tp'] in 'PI'), 's2 is reported for primaries only'
        if with_cigar:
            # tag order relied on by the reference's awk: NM at column 13, AS at column 15 (bin/lib/aligner.py:271-273)
            assert f[12].startswith('NM:i:') and f[13].startswith('ms:i:') and f[14].startswith('AS:i:') and f[15].startswith('nn:i:')
            cig = parse_cigar(t['cg'])
            assert all(op in 'MID' for _, op in cig) and cig[0][1] == 'M' and cig[-1][1] == 'M', t['cg'][:60]
            assert all(cig[i][1] != cig[i + 1][1] for i in range(len(cig) - 1)), 'adjacent operations of the same kind'
            q = reads[name][qs:qe] if strand == '+' else revcomp(reads[name])[qlen - qe:qlen - qs]
            w = (walk_np if fast else walk)(cig, q, genomes[tname][ts:te], sc)
            assert mlen == w['n_match'], (line[:80], mlen, w)
            assert blen == w['n_match'] + w['n_mismatch'] + w['n_ins'] + w['n_del'], (line[:80], blen, w)
            assert t['nn'] == w['n_ambi'] and t['NM'] == w['n_mismatch'] + w['n_ins'] + w['n_del'] + w['n_ambi'], (line[:80], t['NM'], w)
            assert t['ms'] == w['ms'], (line[:80], t['ms'], w['ms'])
            assert t['ms'] <= t['AS'] or t['ms'] >= 0
            stats['as_equal'] = stats.get('as_equal', 0) + (t['AS'] == w['score'])
            # the DP maximised the score: no path (in particular not the reported one re-scored) may beat it by construction,
            # and the reported CIGAR is the DP's own path up to the left-shifting / merging of gaps, which never raises it
            assert w['score'] <= t['AS'] + 0 or abs(w['score'] - t['AS']) <= sc.q2, (line[:80], t['AS'], w['score'])
            ident = w['n_match'] / (w['n_col'] + w['n_gapo'])  # gap-compressed: every aligned column and every gap counts once
            want_de = 1.0 - ident
            assert abs(t['de'] - float('%.4f' % want_de)) < 1.1e-4, (line[:80], t['de'], want_de)
        per_read.setdefault(name, []).append((qs, qe, t['tp'], t['s1'], t.get('AS', t['s1']), mapq, tname, strand, ts, te))
        stats['lines'] = stats.get('lines', 0) + 1
    soft = 0
    for name, hits in per_read.items():
        assert hits[0][2] in 'PI', 'the first line of a read is a primary'
        prim = [h for h in hits if h[2] in 'PI']
        assert sum(1 for h in hits if h[2] in 'Si') <= best_n * len(prim), 'more secondaries than -N allows'
        assert all(h[5] == 0 for h in hits if h[2] in 'Si'), 'secondary alignments carry MAPQ 0'
        stats['primaries'] = stats.get('primaries', 0) + len(prim)
        for i in range(len(prim)):
            for j in range(i + 1, len(prim)):
                ol = min(prim[i][1], prim[j][1]) - max(prim[i][0], prim[j][0])
                if ol > 0.5 * min(prim[i][1] - prim[i][0], prim[j][1] - prim[j][0]):
                    soft += 1   # allowed only through the uncovered-length term of the overlap rule: must stay rare
    assert soft <= max(1, stats.get('primaries', 0) // 50), f'{soft} pairs of primaries overlap by more than the mask level'
    return stats


def check_sam(sam_text, paf_text, reads, quals=None):
    """SAM records of the same hits: same order, coordinates, strand, CIGAR (with clips), NM/AS; SEQ/QUAL rules; SA tags."""
    paf = [l.split('\t') for l in paf_text.splitlines()]
    recs = [l.split('\t') for l in sam_text.splitlines() if not l.startswith('@')]
    mapped = [r for r in recs if not int(r[1]) & 4]
    assert len(mapped) == len(paf)
    by_read = {}
    for r, p in zip(mapped, paf):
        flag, qlen = int(r[1]), int(p[1])
        qs, qe, rev = int(p[2]), int(p[3]), p[4] == '-'
        pt = tags_of(p[12:])
        assert r[0] == p[0] and r[2] == p[5] and int(r[3]) == int(p[7]) + 1 and int(r[4]) == int(p[11])
        assert bool(flag & 16) == rev and bool(flag & 256) == (pt['tp'] in 'Si') and not flag & ~(16 | 256 | 2048)
        cig = parse_cigar(r[5])
        clip = 'H' if flag & 2048 else 'S'
        lead, trail = (qlen - qe, qs) if rev else (qs, qlen - qe)
        core = [c for c in cig if c[1] in 'MID']
        assert ''.join('%d%s' % c for c in core) == pt['cg']
        assert (cig[0] == (lead, clip)) == (lead > 0) and (cig[-1] == (trail, clip)) == (trail > 0), (r[0], r[5][:30], lead, trail)
        st = tags_of(r[11:])
        assert st['NM'] == pt['NM'] and st['AS'] == pt['AS'] and st['tp'] == pt['tp']
        seq = reads[r[0]]
        if flag & 256:
            assert r[9] == '*' and r[10] == '*'
        else:
            want = (revcomp(seq) if rev else seq)
            if flag & 2048:
                want = want[lead:len(want) - trail]
            assert r[9] == want, (r[0], flag)
            if quals and quals.get(r[0]) is not None:
                wq = quals[r[0]][::-1] if rev else quals[r[0]]
                if flag & 2048:
                    wq = wq[lead:len(wq) - trail]
                assert r[10] == wq
            else:
                assert r[10] == '*'
        by_read.setdefault(r[0], []).append((r, st, flag))
    for name, lst in by_read.items():
        non_sec = [x for x in lst if not x[2] & 256]
        assert sum(1 for x in non_sec if not x[2] & 2048) == 1, 'exactly one record of a read is neither secondary nor supplementary'
        for r, st, flag in non_sec:
            others = [x for x in non_sec if x[0] is not r]
            assert ('SA' in st) == bool(others), (name, 'SA tag')
            if others:
                items = [s.split(',') for s in st['SA'].rstrip(';').split(';')]
                assert [(i[0], int(i[1]), i[2], int(i[4]), int(i[5])) for i in items] == \
                       [(o[0][2], int(o[0][3]), '-' if o[2] & 16 else '+', int(o[0][4]), o[1]['NM']) for o in others]
    for r in recs:
        if int(r[1]) & 4:
            assert r[2] == '*' and r[5] == '*' and r[9] == reads[r[0]]
    return len(recs)
