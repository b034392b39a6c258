"""SAM text -> coordinate-sorted BGZF-compressed BAM + BAI index (host side).

Replaces the shell pipeline the reference starts after the species-placement alignment
(/root/reference/bin/lib/aligner.py:246-252):   samtools view -F1796 -b x.sam | samtools sort -o x.bam; samtools index x.bam
Written from the SAM/BAM format specification (SAMv1 sections 4.2, 5.2, 5.3) and, for what the specification leaves
open, from htslib 1.13's behaviour as vendored by the reference (bin/samtools-1.13/htslib-1.13: sam.c sam_parse1 /
bam_write1, bgzf.c, hts.c hts_idx_push / hts_idx_finish): smallest integer type for `i` tags, a record never straddles
BGZF blocks unless it is larger than one, the index's pseudo-bin 37450, bins with less than 64 KiB of compressed span
folded into their parents, chunks that start in the block the previous one ends in merged.  The reference's own htslib
test data pin this module (tests/golden/htslib).
"""
import struct
import zlib

BGZF_BLOCK = 0xff00          # payload bytes per block (htslib BGZF_BLOCK_SIZE)
BGZF_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
_SEQ_CODE = {c: i for i, c in enumerate('=ACMGRSVTWYHKDBN')}
_CIGAR_CODE = {c: i for i, c in enumerate('MIDNSHP=X')}
_REF_CONSUMING = {0, 2, 3, 7, 8}
MIN_SHIFT, N_LVLS = 14, 5
META_BIN = ((1 << (3 * N_LVLS + 3)) - 1) // 7 + 1   # 37450
MIN_MARKER_DIST = 0x10000


def reg2bin(beg, end):
    """Bin of the zero-based half-open interval [beg, end) in the UCSC binning scheme (SAMv1 section 5.3)."""
    end -= 1
    s, t = MIN_SHIFT, ((1 << (3 * N_LVLS)) - 1) // 7
    for l in range(N_LVLS, 0, -1):
        if beg >> s == end >> s:
            return t + (beg >> s)
        s += 3
        t -= 1 << (3 * (l - 1))
    return 0


class BgzfWriter:
    """BGZF stream: tell() is the virtual offset the next byte will get."""

    def __init__(self, fileobj, level=-1):
        self.f, self.level = fileobj, level
        self.buf = bytearray()
        self.block_address = 0

    def tell(self):
        return self.block_address << 16 | len(self.buf)

    def _emit(self, payload):
        if self.level == 0:   # htslib writes level 0 as one stored deflate block
            body = b'\x01' + struct.pack('<HH', len(payload), len(payload) ^ 0xffff) + bytes(payload)
        else:
            c = zlib.compressobj(self.level, zlib.DEFLATED, -15)
            body = c.compress(bytes(payload)) + c.flush()
        block = (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(body) + 25) + body +
                 struct.pack('<II', zlib.crc32(bytes(payload)) & 0xffffffff, len(payload)))
        self.f.write(block)
        self.block_address += len(block)

    def flush(self):
        while self.buf:
            self._emit(self.buf[:BGZF_BLOCK])
            del self.buf[:BGZF_BLOCK]

    def flush_try(self, size):
        """Start a new block if `size` more bytes do not fit the current one (a BAM record stays in one block)."""
        if len(self.buf) + size > BGZF_BLOCK:
            self.flush()

    def write(self, data):
        self.buf += data
        while len(self.buf) >= BGZF_BLOCK:
            self._emit(self.buf[:BGZF_BLOCK])
            del self.buf[:BGZF_BLOCK]

    def close(self):
        self.flush()
        self.f.write(BGZF_EOF)


def _aux_bytes(field):
    tag, typ, val = field[:2], field[3], field[5:]
    t = tag.encode()
    if typ == 'A':
        return t + b'A' + val[:1].encode()
    if typ == 'i':
        x = int(val)
        for code, fmt, lo, hi in (('C', '<B', 0, 0xff), ('S', '<H', 0, 0xffff), ('I', '<I', 0, 0xffffffff)) if x >= 0 else \
                (('c', '<b', -0x80, -1), ('s', '<h', -0x8000, -1), ('i', '<i', -0x80000000, -1)):
            if lo <= x <= hi:
                return t + code.encode() + struct.pack(fmt, x)
        raise ValueError(f'integer tag out of range: {field}')
    if typ == 'f':
        return t + b'f' + struct.pack('<f', float(val))
    if typ in 'ZH':
        return t + typ.encode() + val.encode() + b'\0'
    if typ == 'B':
        sub, *items = val.split(',')
        fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}[sub]
        conv = float if sub == 'f' else int
        return t + b'B' + sub.encode() + struct.pack('<I', len(items)) + struct.pack('<%d%s' % (len(items), fmt), *map(conv, items))
    raise ValueError(f'unknown tag type in {field}')


def parse_cigar(text):
    if text == '*':
        return []
    ops, num = [], 0
    for ch in text:
        if ch.isdigit():
            num = num * 10 + ord(ch) - 48
        else:
            ops.append(num << 4 | _CIGAR_CODE[ch])
            num = 0
    return ops


def encode_record(fields, ref_id):
    """One SAM line (already split on tabs) -> (refID, pos0, end0, flag, BAM record bytes without the block_size word)."""
    qname, flag, rname, pos, mapq, cigar_s, rnext, pnext, tlen, seq, qual = fields[:11]
    flag, pos0, mapq = int(flag), int(pos) - 1, int(mapq)
    tid = ref_id.get(rname, -1) if rname != '*' else -1
    ntid = tid if rnext == '=' else (ref_id.get(rnext, -1) if rnext != '*' else -1)
    cigar = parse_cigar(cigar_s)
    ref_len = sum(c >> 4 for c in cigar if (c & 0xf) in _REF_CONSUMING)
    end0 = pos0 + ref_len if (ref_len > 0 and not flag & 4) else pos0 + 1
    l_seq = 0 if seq == '*' else len(seq)
    packed = bytearray((l_seq + 1) // 2)
    for i in range(l_seq):
        packed[i >> 1] |= _SEQ_CODE.get(seq[i].upper(), 15) << (4 if i % 2 == 0 else 0)
    if l_seq == 0:
        q = b''
    elif qual == '*':
        q = b'\xff' * l_seq
    else:
        q = bytes(ord(c) - 33 for c in qual)
    name = qname.encode() + b'\0'
    rec = (struct.pack('<iiBBHHHiiii', tid, pos0, len(name), mapq, reg2bin(pos0, end0), len(cigar), flag, l_seq, ntid, int(pnext) - 1,
                       int(tlen)) + name + struct.pack('<%dI' % len(cigar), *cigar) + bytes(packed) + q +
           b''.join(_aux_bytes(f) for f in fields[11:]))
    return tid, pos0, end0, flag, rec


def parse_header(header_lines):
    """-> (reference names, lengths) from the @SQ lines"""
    names, lens = [], []
    for line in header_lines:
        if line.startswith('@SQ'):
            tags = dict(f.split(':', 1) for f in line.rstrip('\n').split('\t')[1:])
            names.append(tags['SN'])
            lens.append(int(tags['LN']))
    return names, lens


class BaiBuilder:
    """The state machine of htslib's hts_idx_push / hts_idx_finish for the BAI flavour (min_shift 14, 5 levels)."""

    def __init__(self, n_ref, offset0):
        self.bins = [dict() for _ in range(n_ref)]     # per reference: bin -> list of [beg, end] virtual offsets
        self.lin = [dict() for _ in range(n_ref)]      # per reference: 16 kb window -> smallest virtual offset
        self.lin_n = [0] * n_ref
        self.n_no_coor = 0
        self.last_off = self.save_off = self.off_beg = offset0
        self.last_bin = self.save_bin = 0xffffffff
        self.last_tid = self.save_tid = -1
        self.n_mapped = self.n_unmapped = 0
        self.first = True

    def _chunk(self, tid, b, beg, end):
        self.bins[tid].setdefault(b, []).append([beg, end])

    def push(self, tid, beg, end, offset_after, is_mapped):
        if tid < 0:
            beg, end = -1, 0
        if self.last_tid != tid or self.first:
            self.last_tid = tid
            self.last_bin = 0xffffffff
            self.first = False
        if tid >= 0:
            if is_mapped:
                b0, e0 = max(beg, 0), (end if end > 0 else 1)
                for wdw in range(b0 >> MIN_SHIFT, ((e0 - 1) >> MIN_SHIFT) + 1):
                    self.lin[tid].setdefault(wdw, self.last_off)
                self.lin_n[tid] = max(self.lin_n[tid], ((e0 - 1) >> MIN_SHIFT) + 1)
        else:
            self.n_no_coor += 1
        b = reg2bin(beg, end) if tid >= 0 else reg2bin(-1, 0)
        if self.last_bin != b:
            if self.save_bin != 0xffffffff:
                self._chunk(self.save_tid, self.save_bin, self.save_off, self.last_off)
            if self.last_bin == 0xffffffff and self.save_bin != 0xffffffff:   # change of reference: close its pseudo-bin
                self._chunk(self.save_tid, META_BIN, self.off_beg, self.last_off)
                self._chunk(self.save_tid, META_BIN, self.n_mapped, self.n_unmapped)
                self.n_mapped = self.n_unmapped = 0
                self.off_beg = self.last_off
            self.save_off = self.last_off
            self.save_bin = self.last_bin = b
            self.save_tid = tid
        if is_mapped:
            self.n_mapped += 1
        else:
            self.n_unmapped += 1
        self.last_off = offset_after

    def finish(self, final_offset):
        if self.save_tid >= 0:
            self._chunk(self.save_tid, self.save_bin, self.save_off, final_offset)
            self._chunk(self.save_tid, META_BIN, self.off_beg, final_offset)
            self._chunk(self.save_tid, META_BIN, self.n_mapped, self.n_unmapped)
        n_bins = META_BIN - 1
        for tid, bins in enumerate(self.bins):
            # bins whose chunks span less than 64 KiB of the file are folded into their parents, bottom level first
            for lvl in range(N_LVLS, 0, -1):
                start = ((1 << (3 * lvl)) - 1) // 7
                for b in sorted(k for k in bins if start <= k < n_bins):
                    lst = bins[b]
                    if lvl < N_LVLS and len(lst) > 1:
                        lst.sort(key=lambda c: c[0])
                    parent = (b - 1) >> 3
                    if (lst[-1][1] >> 16) - (lst[0][0] >> 16) < MIN_MARKER_DIST and parent in bins:
                        bins[parent].extend(lst)
                        del bins[b]
            if 0 in bins:
                bins[0].sort(key=lambda c: c[0])
            for b, lst in bins.items():
                if b >= n_bins:
                    continue
                merged = [lst[0]]
                for c in lst[1:]:
                    if merged[-1][1] >> 16 >= c[0] >> 16:
                        merged[-1][1] = max(merged[-1][1], c[1])
                    else:
                        merged.append(c)
                bins[b] = merged

    def linear(self, tid):
        n = self.lin_n[tid]
        out = [self.lin[tid].get(w) for w in range(n)]
        for w in range(n - 2, -1, -1):
            if out[w] is None:
                out[w] = out[w + 1]
        return [x if x is not None else 0 for x in out]

    def write(self, path):
        with open(path, 'wb') as f:
            f.write(b'BAI\1' + struct.pack('<i', len(self.bins)))
            for tid, bins in enumerate(self.bins):
                f.write(struct.pack('<i', len(bins)))
                for b in sorted(bins):
                    f.write(struct.pack('<Ii', b, len(bins[b])))
                    for beg, end in bins[b]:
                        f.write(struct.pack('<QQ', beg, end))
                lin = self.linear(tid)
                f.write(struct.pack('<i', len(lin)) + struct.pack('<%dQ' % len(lin), *lin))
            f.write(struct.pack('<Q', self.n_no_coor))


def write_bam(path, header_text, ref_names, ref_lens, records, level=-1, index_path=None):
    """records: iterable of (tid, pos0, end0, flag, record bytes) in file order.  The index (if asked for) requires them
    to be coordinate sorted with the unplaced ones last."""
    with open(path, 'wb') as f:
        w = BgzfWriter(f, level)
        text = header_text.encode()
        hdr = bytearray(b'BAM\1' + struct.pack('<i', len(text)) + text + struct.pack('<i', len(ref_names)))
        for nm, ln in zip(ref_names, ref_lens):
            hdr += struct.pack('<i', len(nm) + 1) + nm.encode() + b'\0' + struct.pack('<i', ln)
        w.write(hdr)
        w.flush()
        bai = BaiBuilder(len(ref_names), w.tell()) if index_path else None
        for tid, pos0, end0, flag, rec in records:
            w.flush_try(4 + len(rec))
            w.write(struct.pack('<i', len(rec)) + rec)
            if bai:
                bai.push(tid, pos0, end0, w.tell(), not flag & 4)
        if bai:
            w.flush()
            bai.finish(w.tell())
        w.close()
    if bai:
        bai.write(index_path)


def sam_to_sorted_bam(sam_path, bam_path, exclude_flags=0, level=-1, index=True):
    """`samtools view -F <exclude_flags> -b | samtools sort; samtools index`: keep the records without any of the flags, order them
    by (reference, position, strand) with the unplaced ones last -- a stable sort, like samtools' -- and write BAM + .bai."""
    header, recs = [], []
    with open(sam_path) as f:
        for line in f:
            if line.startswith('@'):
                header.append(line)
    names, lens = parse_header(header)
    ref_id = {n: i for i, n in enumerate(names)}
    with open(sam_path) as f:
        for line in f:
            if line.startswith('@') or not line.strip():
                continue
            fields = line.rstrip('\n').split('\t')
            if int(fields[1]) & exclude_flags:
                continue
            recs.append(encode_record(fields, ref_id))
    recs.sort(key=lambda r: ((r[0] if r[0] >= 0 else 1 << 40), r[1] + 1, (r[3] >> 4) & 1))
    hd = '@HD\tVN:1.6\tSO:coordinate\n'
    body = ''.join(l for l in header if not l.startswith('@HD'))
    write_bam(bam_path, hd + body, names, lens, recs, level=level, index_path=bam_path + '.bai' if index else None)
    return len(recs)
