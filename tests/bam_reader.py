"""Minimal BAM / BAI readers for the tests (written from the SAM/BAM specification; test infrastructure only)."""
import struct
import zlib


def read_bgzf(path):
    """-> (uncompressed bytes, list of (compressed offset of block, uncompressed offset of its first byte))"""
    raw = open(path, 'rb').read()
    out, blocks, p = bytearray(), [], 0
    while p < len(raw):
        assert raw[p:p + 4] == b'\x1f\x8b\x08\x04', 'not a BGZF block'
        xlen = struct.unpack_from('<H', raw, p + 10)[0]
        assert raw[p + 12:p + 16] == b'BC\x02\x00'
        bsize = struct.unpack_from('<H', raw, p + 16)[0] + 1
        data = zlib.decompress(raw[p + 12 + xlen:p + bsize - 8], -15)
        crc, isize = struct.unpack_from('<II', raw, p + bsize - 8)
        assert isize == len(data) and crc == zlib.crc32(data) & 0xffffffff
        blocks.append((p, len(out)))
        out += data
        p += bsize
    return bytes(out), blocks


def virtual_offset(blocks, uoff):
    """Virtual offset of uncompressed position uoff (the position right after the last byte of a block maps to the NEXT block)."""
    lo, hi = 0, len(blocks)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if blocks[mid][1] <= uoff:
            lo = mid
        else:
            hi = mid
    return blocks[lo][0] << 16 | (uoff - blocks[lo][1])


def read_bam(path):
    """-> dict(text, refs [(name, len)], records [dict], offsets [(voff_start, voff_end)])"""
    data, blocks = read_bgzf(path)
    assert data[:4] == b'BAM\1'
    l_text = struct.unpack_from('<i', data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from('<i', data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        l = struct.unpack_from('<i', data, p)[0]
        refs.append((data[p + 4:p + 4 + l - 1].decode(), struct.unpack_from('<i', data, p + 4 + l)[0]))
        p += 8 + l
    recs, offs = [], []
    while p < len(data):
        bs = struct.unpack_from('<i', data, p)[0]
        tid, pos, l_name, mapq, bin_, n_cig, flag, l_seq, ntid, npos, tlen = struct.unpack_from('<iiBBHHHiiii', data, p + 4)
        q = p + 36
        name = data[q:q + l_name - 1].decode()
        q += l_name
        cigar = list(struct.unpack_from('<%dI' % n_cig, data, q))
        q += 4 * n_cig
        seq = ''.join('=ACMGRSVTWYHKDBN'[data[q + (i >> 1)] >> (4 if i % 2 == 0 else 0) & 15] for i in range(l_seq))
        q += (l_seq + 1) // 2
        qual = data[q:q + l_seq]
        q += l_seq
        recs.append(dict(tid=tid, pos=pos, name=name, mapq=mapq, bin=bin_, cigar=cigar, flag=flag, seq=seq, qual=qual, ntid=ntid,
                         npos=npos, tlen=tlen, aux=data[q:p + 4 + bs], raw=data[p + 4:p + 4 + bs]))
        offs.append((p, p + 4 + bs))
        p += 4 + bs
    return dict(text=text, refs=refs, records=recs, offsets=[(virtual_offset(blocks, a), b) for a, b in offs], blocks=blocks,
                data_len=len(data))


def read_bai(path):
    """-> (list per reference of (dict bin -> [(beg, end)], linear index list), n_no_coor or None)"""
    d = open(path, 'rb').read()
    assert d[:4] == b'BAI\1'
    n_ref = struct.unpack_from('<i', d, 4)[0]
    p, refs = 8, []
    for _ in range(n_ref):
        n_bin = struct.unpack_from('<i', d, p)[0]
        p += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from('<Ii', d, p)
            p += 8
            bins[b] = [struct.unpack_from('<QQ', d, p + 16 * i) for i in range(n_chunk)]
            p += 16 * n_chunk
        n_intv = struct.unpack_from('<i', d, p)[0]
        lin = list(struct.unpack_from('<%dQ' % n_intv, d, p + 4))
        p += 4 + 8 * n_intv
        refs.append((bins, lin))
    no_coor = struct.unpack_from('<Q', d, p)[0] if p + 8 <= len(d) else None
    return refs, no_coor
