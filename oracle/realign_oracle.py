"""CPU restatement of the amplicon realigner's `realign_reads` (TEST INFRASTRUCTURE ONLY -- only tests/, smoke() and
bench legs may import this; the product path is csrc/realign.hip behind include/mpn_realign.h).

Follows /root/reference/bin/realignment/realign/realigner.cpp statement by statement (sequential loops, the k-mer index
as a dict, the same order of evaluation), with the SSW steps delegated to oracle/ssw_oracle.c and wrapped as
/root/reference/bin/realignment/realign/ssw_cpp.cpp does.  Pinned by tests/golden/realign_golden.json = outputs of the
reference's own sources compiled in place (oracle/_ref/librealigner.so, tests/golden/make_realign_golden.py).

Preconditions (the reference has undefined behaviour outside them): every haplotype has at least 32 bases, ASCII
input, at most 1000 reads, every SSW alignment scores > 0 whenever its result is used.
"""
import numpy as np

from oracle.ssw_bindings import oracle_align

KMER = 32                      # realigner.cpp:66
MAX_MM = 2                     # :68
MATCH, MISMATCH, GAP_O, GAP_E = 4, 6, 8, 2   # :70-73 (and ssw_cpp.cpp's default Aligner)
READ_SIZE, SIMILARITY = 250, 0.16934        # :67,:69
NOT_ALIGNED = -1
OP_UNSPEC, OP_MATCH, OP_INS, OP_DEL, OP_SKIP, OP_SOFT, OP_HARD = 0, 1, 2, 3, 4, 5, 6   # realigner.h:47-55

# ssw_cpp.cpp:8-25 kBaseTranslation: A/a/U/u 0, C/c 1, G/g 2, T/t 3, everything else 4
_TR = np.full(128, 4, dtype=np.int8)
for _c, _v in (('A', 0), ('a', 0), ('C', 1), ('c', 1), ('G', 2), ('g', 2), ('T', 3), ('t', 3), ('U', 0), ('u', 0)):
    _TR[ord(_c)] = _v
# ssw_cpp.cpp:27-47 BuildSwScoreMatrix: +match on the ACGT diagonal, -mismatch elsewhere (N against anything too)
_MAT = np.full((5, 5), -MISMATCH, dtype=np.int8)
for _i in range(4):
    _MAT[_i, _i] = MATCH


def translate(s):
    return _TR[np.frombuffer(s.encode(), dtype=np.uint8)]


def ssw_cpp_align(query, ref_codes):
    """ssw_cpp.cpp:268-300 Aligner::Align with the default Filter -> dict(sw_score, ref_begin, cigar_string) or None."""
    q = translate(query)
    if len(q) == 0 or len(ref_codes) == 0:
        return None
    res = oracle_align(read=q, ref=ref_codes, mat=_MAT.reshape(-1), gap_open=GAP_O, gap_extend=GAP_E, flag=0x0f, filters=0,
                       filterd=32767, mask=len(q), score_size=2)
    assert res is not None and not isinstance(res, str), res
    score1, _, ref_begin, _, q_begin, q_end, _, cigar = res
    # ConvertAlignment + CalculateNumberMismatch (ssw_cpp.cpp:50-203): M runs split into = / X, soft clips added
    out = []
    if cigar:
        if q_begin > 0:
            out.append('%dS' % q_begin)
        r, p = ref_begin, q_begin
        in_m = in_x = False
        len_m = len_x = 0

        def flush():
            nonlocal in_m, in_x, len_m, len_x
            if in_m:
                out.append('%d=' % len_m)
            elif in_x:
                out.append('%dX' % len_x)
            in_m = in_x = False
            len_m = len_x = 0
        for c in cigar:
            n, op = c >> 4, c & 15
            if op == 0:
                for _ in range(n):
                    if ref_codes[r] != q[p]:
                        if in_m:
                            out.append('%d=' % len_m)
                        len_m = 0
                        len_x += 1
                        in_m, in_x = False, True
                    else:
                        if in_x:
                            out.append('%dX' % len_x)
                        len_m += 1
                        len_x = 0
                        in_m, in_x = True, False
                    r += 1
                    p += 1
            elif op == 1:
                p += n
                flush()
                out.append('%dI' % n)
            elif op == 2:
                r += n
                flush()
                out.append('%dD' % n)
        flush()
        end = len(q) - q_end - 1
        if end > 0:
            out.append('%dS' % end)
    return dict(sw_score=score1, ref_begin=ref_begin, cigar_string=''.join(out))


def cigar_to_ops(cigar):
    """CigarStringToVector (realigner.cpp:272-291): regex (\\d+)([XIDS=]), case-insensitive."""
    import re
    ops = []
    for m in re.finditer(r'(\d+)([XIDS=])', cigar, flags=re.I):
        n, ch = int(m.group(1)), m.group(2)
        op = {'=': OP_MATCH, 'X': OP_MATCH, 'S': OP_SOFT, 'D': OP_DEL, 'I': OP_INS}.get(ch, OP_UNSPEC)
        ops.append([op, n])
    return ops


def ops_to_string(ops):
    """CigarVectorToString (realigner.cpp:294-317): a match is written 'X'; other codes print the length only."""
    return ''.join('%d%s' % (n, {OP_MATCH: 'X', OP_INS: 'I', OP_DEL: 'D', OP_SOFT: 'S'}.get(op, '')) for op, n in ops)


def fast_align_strings(s1, s2, max_mismatches):
    """realigner.cpp:232-253 -> (score, mismatches); 'N' on either side counts as a match."""
    matches = mm = 0
    for c1, c2 in zip(s1, s2):
        if c1 != c2 and c1 != 'N' and c2 != 'N':
            mm += 1
            if mm == max_mismatches:
                return 0, mm
        else:
            matches += 1
    return matches * MATCH - mm * MISMATCH, mm


def libstdcxx_sort(keys):
    """Permutation std::sort (libstdc++ introsort: median-of-3 quicksort above 16 elements, then insertion sort) applies
    to a sequence compared by `keys` with operator< -- realigner.cpp:108 sorts the haplotypes by score with it, and the
    order of equal scores decides ties in GetBestReadAlignment."""
    a = list(range(len(keys)))

    def less(x, y):
        return keys[x] < keys[y]

    def insertion(first, last, guarded=True):
        for i in range(first + (1 if guarded else 0), last):
            v = a[i]
            if guarded and less(v, a[first]):
                a[first + 1:i + 1] = a[first:i]
                a[first] = v
            else:
                j = i
                while less(v, a[j - 1]):
                    a[j] = a[j - 1]
                    j -= 1
                a[j] = v

    def heap_sort(first, last):  # __partial_sort(first, last, last): make_heap + sort_heap
        n = last - first

        def adjust(hole, length, value):
            top = hole
            child = hole
            while child < (length - 1) // 2:
                child = 2 * (child + 1)
                if less(a[first + child], a[first + child - 1]):
                    child -= 1
                a[first + hole] = a[first + child]
                hole = child
            if (length & 1) == 0 and child == (length - 2) // 2:
                child = 2 * (child + 1)
                a[first + hole] = a[first + child - 1]
                hole = child - 1
            parent = (hole - 1) // 2
            while hole > top and less(a[first + parent], value):
                a[first + hole] = a[first + parent]
                hole = parent
                parent = (hole - 1) // 2
            a[first + hole] = value
        if n >= 2:
            parent = (n - 2) // 2
            while True:
                adjust(parent, n, a[first + parent])
                if parent == 0:
                    break
                parent -= 1
        end = last
        while end - first > 1:
            end -= 1
            v = a[end]
            a[end] = a[first]
            adjust(0, end - first, v)

    def introsort(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                return
            depth -= 1
            mid = first + (last - first) // 2
            x, y, z = first + 1, mid, last - 1   # __move_median_to_first(first, first+1, mid, last-1)
            if less(a[x], a[y]):
                if less(a[y], a[z]):
                    m = y
                elif less(a[x], a[z]):
                    m = z
                else:
                    m = x
            elif less(a[x], a[z]):
                m = x
            elif less(a[y], a[z]):
                m = z
            else:
                m = y
            a[first], a[m] = a[m], a[first]
            lo, hi = first + 1, last             # __unguarded_partition(first+1, last, first)
            while True:
                while less(a[lo], a[first]):
                    lo += 1
                hi -= 1
                while less(a[first], a[hi]):
                    hi -= 1
                if not lo < hi:
                    break
                a[lo], a[hi] = a[hi], a[lo]
                lo += 1
            introsort(lo, last, depth)
            last = lo

    n = len(a)
    if n > 1:
        introsort(0, n, 2 * (n.bit_length() - 1))
        if n > 16:
            insertion(0, 16)
            insertion(16, n, guarded=False)
        else:
            insertion(0, n)
    return a


def _merge_op(op, length, read_len, cigar):
    """MergeCigarOp (realigner.cpp:551-574)."""
    last = cigar[-1][0] if cigar else OP_UNSPEC
    before = sum(n for o, n in cigar if o != OP_DEL)
    new_len = min(length, read_len - before) if op != OP_DEL else length
    if new_len <= 0 or before == read_len:
        return
    if op == last:
        cigar[-1][1] += new_len
    else:
        cigar.append([op, new_len])


def _aligned_len(cigar):
    return sum(n for o, n in cigar if o != OP_DEL)


def _left_trim(hap_ops_in, read_to_hap_pos):
    """LeftTrimHaplotypeToRefAlignment (realigner.cpp:578-607)."""
    ops = [list(x) for x in hap_ops_in]
    cur = 0
    while cur != read_to_hap_pos:
        op, n = ops.pop(0)
        if op in (OP_MATCH, OP_HARD, OP_SOFT, OP_INS):
            if n + cur > read_to_hap_pos:
                ops.insert(0, [op, n - (read_to_hap_pos - cur)])
            cur = min(n + cur, read_to_hap_pos)
    if ops[0][0] == OP_DEL:
        ops.pop(0)
    return ops


def read_to_ref_alignment(read_len, position, read_cigar, hap_ops_in):
    """CalculateReadToRefAlignment (realigner.cpp:653-777) -> list of [op, len] (empty = keep the original alignment)."""
    r2h = cigar_to_ops(read_cigar)
    h2r = _left_trim(hap_ops_in, position)
    out = []
    is_m = lambda o: o in (OP_MATCH, OP_SOFT)
    if r2h and r2h[0][0] == OP_SOFT:
        _merge_op(OP_SOFT, r2h[0][1], read_len, out)
        r2h.pop(0)
    while (r2h or h2r) and _aligned_len(out) < read_len:
        if r2h and not h2r:
            _merge_op(r2h[0][0], r2h[0][1], read_len, out)
            r2h.pop(0)
            continue
        if not r2h and h2r:
            break
        a = r2h.pop(0)
        b = h2r.pop(0)
        if is_m(a[0]) and is_m(b[0]):
            n = min(a[1], b[1])
            _merge_op(OP_SOFT if OP_SOFT in (a[0], b[0]) else OP_MATCH, n, read_len, out)
            a[1] -= n
            if a[1] > 0:
                r2h.insert(0, a)
            b[1] -= n
            if b[1] > 0:
                h2r.insert(0, b)
        elif a[0] == OP_DEL and is_m(b[0]):
            _merge_op(OP_DEL, a[1], read_len, out)
            b[1] -= a[1]
            if b[1] > 0:
                h2r.insert(0, b)
        elif b[0] == OP_DEL and is_m(a[0]):
            _merge_op(OP_DEL, b[1], read_len, out)
            if a[1] > 0:
                r2h.insert(0, a)
        elif a[0] == OP_DEL and b[0] == OP_DEL:
            _merge_op(OP_DEL, a[1] + b[1], read_len, out)
        elif a[0] == OP_INS and is_m(b[0]):
            a[1] = min(read_len - _aligned_len(out), a[1])
            _merge_op(OP_INS, a[1], read_len, out)
            if b[1] > 0:
                h2r.insert(0, b)
        elif b[0] == OP_INS and is_m(a[0]):
            b[1] = min(read_len - _aligned_len(out), b[1])
            _merge_op(OP_INS, b[1], read_len, out)
            a[1] = max(0, a[1] - b[1])
            if a[1] > 0:
                r2h.insert(0, a)
        elif a[0] == OP_INS and b[0] == OP_INS:
            _merge_op(OP_INS, a[1] + b[1], read_len, out)
        else:
            return []
    return out


def positions_map(hap_len, cigar):
    """SetPositionsMap (realigner.cpp:453-507)."""
    import re
    pm = [0] * hap_len
    shift = pos = 0
    for m in re.finditer(r'(\d+)([XIDS=])', cigar, flags=re.I):
        n, op = int(m.group(1)), m.group(2)
        if op in '=X':
            for _ in range(n):
                pm[pos] = shift
                pos += 1
        elif op == 'S':
            shift -= n
            for _ in range(n):
                pm[pos] = shift
                pos += 1
        elif op == 'D':
            shift += n
        elif op == 'I':
            for _ in range(n):
                pm[pos] = shift
                shift -= 1
                pos += 1
    return pm


def realign_reads(seqs, positions, cigars, reference, haplotypes, ref_start, ref_prefix, ref_suffix):
    """-> [(position, cigar_string), ...] per read (realigner.cpp:782-850, AlignReads :88-117)."""
    n = len(seqs)
    thr = MATCH * READ_SIZE * SIMILARITY - MISMATCH * READ_SIZE * (1 - SIMILARITY)
    thr = 1 if thr < 0 else int(thr)
    # BuildIndex (:429-451)
    index = {}
    for rid, read in enumerate(seqs):
        if len(read) <= KMER:
            continue
        for i in range(len(read) - KMER + 1):
            index.setdefault(read[i:i + KMER], []).append((rid, i))
    # FastAlignReadsToHaplotypes (:147-230)
    haps = []
    for hi, hap in enumerate(haplotypes):
        assert len(hap) >= KMER
        al = [dict(position=NOT_ALIGNED, cigar='', score=0) for _ in range(n)]
        hap_score = 0
        is_ref = hap == reference
        coverage = [0] * len(hap)
        for i in range(len(hap) - KMER + 1):
            occs = index.get(hap[i:i + KMER])
            if occs is None:   # (:181-183) `continue`: the coverage test below is skipped too
                continue
            for rid, rpos in occs:
                start = max(0, i - rpos)
                size = len(seqs[rid])
                if start + size > len(hap):
                    continue
                ra = al[rid]
                if ra['position'] != NOT_ALIGNED and ra['position'] == start:
                    continue
                sc, mm = fast_align_strings(hap[start:start + size], seqs[rid], MAX_MM + 1)
                if mm <= MAX_MM:
                    old = ra['score']
                    for p in range(start, start + size):
                        coverage[p] += 1
                    if old < sc:
                        ra['score'] = sc
                        hap_score += sc - old
                        ra['position'] = start
                        ra['cigar'] = '%d=' % size
            if coverage[i] == 0 and i >= ref_prefix and i < len(hap) - ref_suffix and not is_ref:
                hap_score = 0
                break
        if hap_score == 0:
            al = [dict(position=NOT_ALIGNED, cigar='', score=0) for _ in range(n)]
        haps.append(dict(index=hi, score=hap_score, reads=al, cigar='', ops=[], ref_pos=0, is_ref=False, pmap=[]))
    # AlignHaplotypesToReference (:325-349)
    ref_codes = translate(reference)
    for h in haps:
        a = ssw_cpp_align(haplotypes[h['index']], ref_codes)
        if a is not None and a['sw_score'] > 0:
            hap_len = len(haplotypes[h['index']])
            h['is_ref'] = a['cigar_string'] == '%d=' % hap_len
            h['cigar'] = a['cigar_string']
            h['ops'] = cigar_to_ops(a['cigar_string'])
            h['ref_pos'] = a['ref_begin']
    # CalculatePositionMaps (:509-514)
    for h in haps:
        h['pmap'] = positions_map(len(haplotypes[h['index']]), h['cigar'])
    # SswAlignReadsToHaplotypes (:351-384)
    hap_codes = {}
    for i in range(n):
        if any(h['reads'][i]['score'] > 0 for h in haps):
            continue
        for h in haps:
            if h['score'] == 0:
                continue
            if h['index'] not in hap_codes:
                hap_codes[h['index']] = translate(haplotypes[h['index']])
            a = ssw_cpp_align(seqs[i], hap_codes[h['index']])
            if a is not None and a['sw_score'] > 0 and a['sw_score'] >= thr and h['reads'][i]['score'] < a['sw_score']:
                h['reads'][i] = dict(score=a['sw_score'], cigar=a['cigar_string'], position=a['ref_begin'])
    # sort by haplotype score (:108) and pick per read (:386-427, :516-538)
    haps = [haps[j] for j in libstdcxx_sort([h['score'] for h in haps])]
    out = []
    for i in range(n):
        best_score, best = 0, None
        for h in haps:
            sc = h['reads'][i]['score']
            if sc > best_score or (best_score > 0 and sc == best_score and not h['is_ref']):
                best_score, best = sc, h
        res = (positions[i], cigars[i])
        if best is not None:
            ra = best['reads'][i]
            p = ra['position']
            new_pos = ref_start + best['ref_pos'] + p + best['pmap'][p]
            ops = read_to_ref_alignment(len(seqs[i]), p, ra['cigar'], best['ops'])
            if ops:
                res = (new_pos, ops_to_string(ops))
        out.append(res)
    return out
