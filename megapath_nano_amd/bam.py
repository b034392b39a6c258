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


def _bgzf_block(payload, level):
    if level == 0:   # htslib writes level 0 as one stored deflate block
        body = b'\x01' + struct.pack('<HH', len(payload), len(payload) ^ 0xffff) + payload
    else:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(payload) + c.flush()
    return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', len(body) + 25) + body +
            struct.pack('<II', zlib.crc32(payload) & 0xffffffff, len(payload)))


class BgzfWriter:
    """BGZF stream.  Blocks are compressed by a thread pool (zlib releases the GIL), PENDING blocks at a time, so the file
    address of a block is not known when it is closed: tell() returns a provisional virtual offset (block NUMBER << 16 |
    offset in the block) and resolve() turns it into the real one (block ADDRESS << 16 | offset) once the blocks before it
    have been written (after drain())."""
    PENDING = 256

    def __init__(self, fileobj, level=-1, threads=None):
        import os
        self.f, self.level = fileobj, level
        self.buf = bytearray()
        self.pending = []            # closed blocks that are not compressed yet
        self.addr = [0]              # addr[i] = file address of block i, for every block whose predecessors are written
        self.block_no = 0            # number of the block being filled
        self.threads = threads if threads is not None else max(1, min(16, os.cpu_count() or 1))
        self.pool = None

    def tell(self):
        return self.block_no << 16 | len(self.buf)

    def resolve(self, voff):
        return self.addr[voff >> 16] << 16 | (voff & 0xffff)

    def _close_block(self, payload):
        self.pending.append(bytes(payload))
        self.block_no += 1
        if len(self.pending) >= self.PENDING:
            self.drain()

    def drain(self):
        if not self.pending:
            return
        if self.threads > 1 and len(self.pending) > 4 and self.level != 0:
            if self.pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self.pool = ThreadPoolExecutor(self.threads)
            blocks = self.pool.map(lambda p: _bgzf_block(p, self.level), self.pending)
        else:
            blocks = (_bgzf_block(p, self.level) for p in self.pending)
        for b in blocks:
            self.f.write(b)
            self.addr.append(self.addr[-1] + len(b))
        self.pending = []

    def flush(self):
        while self.buf:
            self._close_block(self.buf[:BGZF_BLOCK])
            del self.buf[:BGZF_BLOCK]

    def flush_try(self, size):
        """Start a new block if `size` more bytes do not fit the current one (a BAM record stays in one block)."""
        if len(self.buf) + size > BGZF_BLOCK:
            self.flush()

    def write(self, data):
        self.buf += data
        while len(self.buf) >= BGZF_BLOCK:
            self._close_block(self.buf[:BGZF_BLOCK])
            del self.buf[:BGZF_BLOCK]

    def close(self):
        self.flush()
        self.drain()
        self.f.write(BGZF_EOF)
        if self.pool is not None:
            self.pool.shutdown()


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


_SEQ_TABLE = bytes(_SEQ_CODE.get(chr(c).upper(), 15) for c in range(256))
_QUAL_TABLE = bytes((c - 33) & 0xff for c in range(256))
MAX_CIGAR_OPS = 65535          # n_cigar_op is 16 bits wide in a BAM record


def pack_seq(seq):
    """SEQ text -> 4-bit codes, two bases per byte, high nibble first (SAMv1 4.2); bytes.translate + numpy, no per-base loop."""
    import numpy as np
    codes = np.frombuffer(seq.encode().translate(_SEQ_TABLE), dtype=np.uint8)
    if len(codes) & 1:
        codes = np.concatenate([codes, np.zeros(1, dtype=np.uint8)])
    return ((codes[0::2] << 4) | codes[1::2]).tobytes()


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
    packed = pack_seq(seq) if l_seq else b''
    if l_seq == 0:
        q = b''
    elif qual == '*':
        q = b'\xff' * l_seq
    else:
        q = qual.encode().translate(_QUAL_TABLE)
    name = qname.encode() + b'\0'
    aux = b''.join(_aux_bytes(f) for f in fields[11:])
    if len(cigar) > MAX_CIGAR_OPS:
        # htslib (sam.c bam_write1 / SAMv1 4.2.2): a CIGAR that does not fit 16 bits is replaced by <l_seq>S<ref_len>N and
        # the real one travels in a CG:B,I tag (ultra-long ONT reads have such CIGARs)
        aux += b'CGBI' + struct.pack('<I', len(cigar)) + struct.pack('<%dI' % len(cigar), *cigar)
        cigar = [l_seq << 4 | 4, ref_len << 4 | 3]
    rec = (struct.pack('<iiBBHHHiiii', tid, pos0, len(name), mapq, reg2bin(pos0, end0), len(cigar), flag, l_seq, ntid, int(pnext) - 1,
                       int(tlen)) + name + struct.pack('<%dI' % len(cigar), *cigar) + packed + q + aux)
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

    def remap(self, fn):
        """Every virtual offset collected so far through fn (BgzfWriter.resolve: provisional -> real); the count pairs of the
        pseudo-bins are not offsets and stay."""
        for bins in self.bins:
            for b, lst in bins.items():
                for k, c in enumerate(lst):
                    if b == META_BIN and k % 2 == 1:
                        continue
                    c[0], c[1] = fn(c[0]), fn(c[1])
        for lin in self.lin:
            for w in lin:
                lin[w] = fn(lin[w])
        self.last_off, self.save_off, self.off_beg = fn(self.last_off), fn(self.save_off), fn(self.off_beg)

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
            w.drain()
            bai.remap(w.resolve)
            bai.finish(w.resolve(w.tell()))
        w.close()
    if bai:
        bai.write(index_path)


def sam_to_sorted_bam(sam_path, bam_path, exclude_flags=0, level=-1, index=True, sort_keys=None, native=True, batch_bytes=64 << 20):
    """`samtools view -F <exclude_flags> -b | samtools sort; samtools index`: keep the records without any of the flags, order them
    by (reference, position, strand) with the unplaced ones last -- a stable sort, like samtools' -- and write BAM + .bai.

    The SAM text of a run is tens of GB: only a key (reference, position, strand) and the line's place in the file are kept per
    record; the lines are read back in sorted order and encoded as they are written.  sort_keys(tid, pos, rev) -> order may
    replace the host sort (megapath_nano_amd.abundance.device_sort_order runs it on the GPU).  native: the records are encoded by
    libmpn.so (mpn_bam_encode, all host cores) instead of encode_record; the two are compared byte for byte in tests/test_bam.py."""
    import numpy as np
    header = []
    tids, poss, revs, offs, lens_ = [], [], [], [], []
    ref_id = None
    names = lens = None
    with open(sam_path, 'rb') as f:
        off = 0
        for raw in f:
            n = len(raw)
            if raw[:1] == b'@':
                header.append(raw.decode())
            elif raw.strip():
                if ref_id is None:
                    names, lens = parse_header(header)
                    ref_id = {nm.encode(): i for i, nm in enumerate(names)}
                # QNAME FLAG RNAME POS: the first four fields are all the key needs
                f4 = raw.split(b'\t', 4)
                flag = int(f4[1])
                if not flag & exclude_flags:
                    tids.append(ref_id.get(f4[2], -1) if f4[2] != b'*' else -1)
                    poss.append(int(f4[3]))
                    revs.append((flag >> 4) & 1)
                    offs.append(off)
                    lens_.append(n)
            off += n
    if ref_id is None:
        names, lens = parse_header(header)
    tid = np.asarray(tids, dtype=np.int64)
    pos = np.asarray(poss, dtype=np.int64)
    rev = np.asarray(revs, dtype=np.int64)
    tid_key = np.where(tid >= 0, tid, np.int64(1) << 40)
    if sort_keys is not None and len(tid):
        order = np.asarray(sort_keys(tid_key, pos, rev), dtype=np.int64)
    else:
        order = np.lexsort((rev, pos, tid_key))   # stable: equal keys stay in file order
    ref_str = {n_: i for i, n_ in enumerate(names)}
    offs_a, lens_a = np.asarray(offs, dtype=np.int64), np.asarray(lens_, dtype=np.int64)

    def records_python():
        with open(sam_path, 'rb') as f:
            for k in order:
                f.seek(int(offs_a[k]))
                yield encode_record(f.read(int(lens_a[k])).decode().rstrip('\n').split('\t'), ref_str)

    def records():
        # the lines of a batch, in sorted order, through the native encoder (include/mpn_bam.h: the same encoding as
        # encode_record above, in C, over all host cores): a run has millions of records
        enc = NativeEncoder(names)
        try:
            with open(sam_path, 'rb') as f:
                k0 = 0
                while k0 < len(order):
                    k1, size = k0, 0
                    while k1 < len(order) and (size < batch_bytes or k1 == k0):
                        size += int(lens_a[order[k1]])
                        k1 += 1
                    lines = []
                    for k in order[k0:k1]:
                        f.seek(int(offs_a[k]))
                        lines.append(f.read(int(lens_a[k])))
                    yield from enc.encode(lines)
                    k0 = k1
        finally:
            enc.close()

    hd = '@HD\tVN:1.6\tSO:coordinate\n'
    body = ''.join(l for l in header if not l.startswith('@HD'))
    write_bam(bam_path, hd + body, names, lens, records() if native else records_python(), level=level,
              index_path=bam_path + '.bai' if index else None)
    return len(order)


class NativeEncoder:
    """SAM lines -> BAM records through libmpn.so (mpn_bam_encode); yields what encode_record yields."""

    def __init__(self, ref_names):
        import ctypes as ct
        from . import _ffi
        self.ct, self._ffi = ct, _ffi
        lib = _ffi.lib()
        lib.mpn_bam_encoder_create.argtypes = [ct.POINTER(ct.c_char_p), ct.c_int32]
        lib.mpn_bam_encoder_create.restype = ct.c_void_p
        lib.mpn_bam_encoder_destroy.argtypes = [ct.c_void_p]
        lib.mpn_bam_encoder_destroy.restype = None
        lib.mpn_bam_encode.argtypes = [ct.c_void_p, ct.c_char_p] + [ct.c_void_p] * 2 + [ct.c_int64, ct.c_void_p, ct.c_int64] + [ct.c_void_p] * 5
        lib.mpn_bam_encode.restype = ct.c_int64
        self.lib = lib
        arr = (ct.c_char_p * max(1, len(ref_names)))(*[n_.encode() for n_ in ref_names])
        self.h = lib.mpn_bam_encoder_create(arr, len(ref_names))
        if not self.h:
            raise _ffi.MpnError('mpn_bam_encoder_create: ' + _ffi.last_error())

    def encode(self, lines):
        """lines: list of bytes (one SAM line each) -> list of (tid, pos0, end0, flag, record bytes)"""
        import numpy as np
        n = len(lines)
        if n == 0:
            return []
        text = b''.join(lines)
        lens = np.fromiter((len(l) for l in lines), dtype=np.int32, count=n)
        offs = np.zeros(n, dtype=np.int64)
        np.cumsum(lens[:-1], out=offs[1:])
        rec_off = np.zeros(n + 1, dtype=np.int64)
        tid, pos0, end0, flag = (np.zeros(n, dtype=np.int32) for _ in range(4))
        cap = len(text) + 64 * n + 1024
        while True:
            out = np.empty(cap, dtype=np.uint8)
            r = self.lib.mpn_bam_encode(self.h, text, offs.ctypes.data, lens.ctypes.data, n, out.ctypes.data, cap, rec_off.ctypes.data,
                                        tid.ctypes.data, pos0.ctypes.data, end0.ctypes.data, flag.ctypes.data)
            if r == -3:
                cap = int(rec_off[n]) + 1024
                continue
            if r < 0:
                raise ValueError('mpn_bam_encode: ' + self._ffi.last_error())
            break
        buf = out.tobytes()
        ro = rec_off.tolist()
        return [(int(tid[i]), int(pos0[i]), int(end0[i]), int(flag[i]), buf[ro[i]:ro[i + 1]]) for i in range(n)]

    def close(self):
        if self.h:
            self.lib.mpn_bam_encoder_destroy(self.h)
            self.h = None
