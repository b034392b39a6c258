"""Host-side mirror of the reference's `nanofastq` read filter on top of libmpn.so.

/root/reference/bin/tools/nanofastq.c reads FASTA/FASTQ (kseq) on stdin and writes the reads that pass to stdout --
cropped by -h/-t, kept if the cropped length >= -l and (FASTQ only) the mean error-probability Phred score after cropping
>= -q -- and one statistics line per read to stderr: `read_id, length, avgQ, length_after_crop, avgQ_after_crop, passed`
(:228-238; read by /root/reference/bin/megapath_nano.py:1075).  `-r PREFIX` renames reads PREFIX1, PREFIX2, ...

Same contract here: `filter_fastx()` returns the two byte strings (and the surviving reads packed for the mapper, so that
they can go straight to the GPU); `bin/mpn-nanofastq` is the executable drop-in.  The error-probability sums come from
`mpn_fastq_qsum_batch` (bit-identical doubles, include/mpn_fastq.h); log10 and the `%.2f` formatting use the same libm /
printf rules as the C program.  Reproduced quirks: `-0.00` for a perfect-zero score, and the unsigned wrap-around of
`length - headcrop - tailcrop` in the fourth column when the crops exceed the read (:231-234).
"""
import ctypes as ct
import gzip
import math

import numpy as np

from . import _ffi

PHRED_TABLE = np.array([math.pow(10.0, -i / 10.0) for i in range(128)], dtype=np.float64)   # nanofastq.c:147-149


def parse_fastx(data):
    """kseq-style records from a bytes object -> list of (name, comment or None, seq, qual or None), all bytes."""
    if data[:2] == b'\x1f\x8b':
        data = gzip.decompress(data)
    lines = data.split(b'\n')
    n, i, out = len(lines), 0, []
    while i < n:
        head = lines[i]
        if not head or head[:1] not in (b'>', b'@'):
            i += 1
            continue
        head = head[1:].rstrip(b'\r')
        cut = -1
        for k, ch in enumerate(head):
            if ch in (32, 9):
                cut = k
                break
        name, comment = (head, None) if cut < 0 else (head[:cut], head[cut + 1:])
        i += 1
        seq = []
        while i < n and lines[i][:1] not in (b'>', b'@', b'+'):
            seq.append(lines[i].rstrip(b'\r'))
            i += 1
        seq = b''.join(seq)
        qual = None
        if i < n and lines[i][:1] == b'+':
            i += 1
            q, got = [], 0
            while i < n and got < len(seq):
                ln = lines[i].rstrip(b'\r')
                q.append(ln)
                got += len(ln)
                i += 1
            qual = b''.join(q)
            if len(qual) != len(seq):
                raise ValueError(f'truncated quality string for read {name.decode(errors="replace")}')
        out.append((name, comment if comment else None, seq, qual))
    return out


def qsums(quals, head_crop, tail_crop, min_len):
    """-> (total, cropped) float64 arrays for a list of quality byte strings, from the HIP kernel."""
    lib = _ffi.lib()
    n = len(quals)
    total, cropped = np.zeros(n, dtype=np.float64), np.zeros(n, dtype=np.float64)
    if n == 0:
        return total, cropped
    lens = np.array([len(q) for q in quals], dtype=np.int32)
    off = np.zeros(n, dtype=np.int64)
    np.cumsum(lens[:-1], out=off[1:])
    buf = np.frombuffer(b''.join(quals) + b'\0', dtype=np.uint8)
    status = np.zeros(n, dtype=np.uint8)
    P = ct.c_void_p
    lib.mpn_fastq_qsum_batch.argtypes = [ct.c_int32, P, P, P, ct.c_int32, ct.c_int32, ct.c_int32, P, P, P, P]
    lib.mpn_fastq_qsum_batch.restype = ct.c_int
    _ffi.check(lib.mpn_fastq_qsum_batch(n, buf.ctypes.data, off.ctypes.data, lens.ctypes.data, head_crop, tail_crop, min_len,
                                        PHRED_TABLE.ctypes.data, total.ctypes.data, cropped.ctypes.data, status.ctypes.data),
               'mpn_fastq_qsum_batch')
    if status.any():
        raise ValueError(f'read {int(np.flatnonzero(status)[0])} holds a quality character outside Phred+33')
    return total, cropped


def _phred(total_err, n):
    return -10 * math.log10(total_err / n)


def assemble(records, total, cropped, min_quality=0, min_length=0, head_crop=0, tail_crop=0, read_id_prefix=None):
    """The program's outputs from the parsed records and their error sums (shared with the oracle's checker).
    -> (stdout bytes, stderr bytes, kept: list of (name, cropped seq))"""
    if min_length == 0:
        min_length = 1                                                           # nanofastq.c:127-130
    out, info, kept = [], [], []
    for k, (name, comment, seq, qual) in enumerate(records):
        rid = name if read_id_prefix is None else read_id_prefix.encode() + str(k + 1).encode()   # :158-163
        l = len(seq)
        fq = qual is not None
        avg = _phred(total[k], l) if fq else 0.0                                  # :166-173
        avg_crop = 0.0
        start, end = head_crop, l - tail_crop
        passed = 1
        if end - start < min_length:                                              # :182
            passed = 0
        else:
            if fq:
                avg_crop = _phred(cropped[k], l - head_crop - tail_crop)          # :196
            if fq and avg_crop < min_quality:                                     # :199
                passed = 0
            else:
                out.append((b'@' if fq else b'>') + rid + (b' ' + comment if comment else b'') + b'\n' + seq[start:end] + b'\n')
                if fq:
                    out.append(b'+\n' + qual[start:end] + b'\n')
                kept.append((rid.decode(errors='replace'), seq[start:end]))
        n_crop = l - head_crop - tail_crop
        col4 = n_crop % (1 << 64) if n_crop != 0 else 0                           # size_t arithmetic, :231-234
        info.append(b'%s\t%d\t%s\t%d\t%s\t%d\n' % (rid, l, ('%.2f' % avg).encode(), col4, ('%.2f' % avg_crop).encode(), passed))
    return b''.join(out), b''.join(info), kept


def filter_fastx(data, min_quality=0, min_length=0, head_crop=0, tail_crop=0, read_id_prefix=None):
    """nanofastq on a bytes object.  -> (stdout bytes, stderr bytes, kept reads [(name, seq bytes)])"""
    if min(min_quality, min_length, head_crop, tail_crop) < 0:
        raise ValueError('negative option')                                       # nanofastq.c:122-126 prints the usage and exits
    records = parse_fastx(data)
    for name, _, seq, _ in records:
        if len(seq) == 0:
            raise ValueError(f'empty read {name.decode(errors="replace")}: the reference divides by its length')
    idx = [k for k, r in enumerate(records) if r[3] is not None]
    total, cropped = np.zeros(len(records)), np.zeros(len(records))
    t, c = qsums([records[k][3] for k in idx], head_crop, tail_crop, max(min_length, 1))
    total[idx], cropped[idx] = t, c
    return assemble(records, total, cropped, min_quality, min_length, head_crop, tail_crop, read_id_prefix)
