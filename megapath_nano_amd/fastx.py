"""FASTA/FASTQ ingestion for the aligner boundary (host side).

The reference hands minimap2 its target as a FIFO that `cat` fills with concatenated `.fna.gz` members
(/root/reference/bin/lib/aligner.py:143-144,209-217) or as one `.fna.gz` (:221).  A FIFO can be opened ONCE: closing the
read end makes the writer fail with EPIPE and the head of the stream is gone.  So every path is opened exactly once,
the format (saved index / gzip / plain) is sniffed from the first bytes of that one stream (`peek`, nothing consumed),
and records are parsed incrementally from it; concatenated gzip members are handled by the decompressor.
"""
import gzip
import io
import os
import stat

INDEX_MAGIC = b'MPNIDX01'


def is_fifo(path):
    try:
        return stat.S_ISFIFO(os.stat(path).st_mode)
    except OSError:
        return False


def open_once(path):
    """-> (kind, binary stream); kind is 'index' (a saved mpn index: the stream is positioned at its first byte),
    'gzip' or 'plain' (both: a stream of decompressed bytes).  The path is opened exactly once."""
    raw = io.BufferedReader(open(path, 'rb', buffering=0), buffer_size=1 << 20)
    head = raw.peek(8)[:8]
    if head == INDEX_MAGIC:
        return 'index', raw
    if head[:2] == b'\x1f\x8b':
        return 'gzip', io.BufferedReader(gzip.GzipFile(fileobj=raw, mode='rb'), buffer_size=1 << 20)
    return 'plain', raw


def iter_fastx(stream):
    """Yield (name, sequence bytes, quality bytes or None) from a binary FASTA / FASTQ stream (multi-line records, CRLF,
    mixed FASTA and FASTQ records as kseq accepts them).  The name is the first word of the header."""
    line = stream.readline()
    while line:
        line = line.rstrip(b'\r\n')
        if not line:
            line = stream.readline()
            continue
        tag = line[:1]
        if tag not in (b'>', b'@'):
            raise ValueError('neither FASTA nor FASTQ: record header expected, got %r' % line[:20])
        words = line[1:].split()
        name = words[0].decode() if words else ''
        seq = []
        line = stream.readline()
        while line and line[:1] not in (b'>', b'@', b'+'):
            seq.append(line.strip())
            line = stream.readline()
        seq = b''.join(seq)
        qual = None
        if line[:1] == b'+':
            got, parts = 0, []
            line = stream.readline()
            while line and got < len(seq):
                q = line.rstrip(b'\r\n')
                parts.append(q)
                got += len(q)
                line = stream.readline()
            qual = b''.join(parts)
            if len(qual) != len(seq):
                qual = None  # truncated quality string: kseq reports an error; the bases are still usable
        yield name, seq, qual


def read_fastx(path, with_qual=False):
    """-> list of (name, bytes) -- or (name, bytes, qual) with with_qual -- from FASTA or FASTQ, plain or gzip
    (concatenated members included), regular file or FIFO."""
    kind, stream = open_once(path)
    try:
        if kind == 'index':
            raise ValueError(f'{path}: a saved index, not sequences')
        recs = list(iter_fastx(stream))
    finally:
        stream.close()
    return recs if with_qual else [(n, s) for n, s, _ in recs]


def iter_fastx_full(stream):
    """Like iter_fastx but keeps the comment: yields (name, comment or None, sequence, quality or None)."""
    line = stream.readline()
    while line:
        line = line.rstrip(b'\r\n')
        if not line:
            line = stream.readline()
            continue
        if line[:1] not in (b'>', b'@'):
            raise ValueError('neither FASTA nor FASTQ: record header expected, got %r' % line[:20])
        head = line[1:]
        cut = len(head)
        for k, ch in enumerate(head):
            if ch in b' \t':
                cut = k
                break
        name, comment = head[:cut], (head[cut + 1:] if cut < len(head) else None)
        seq = []
        line = stream.readline()
        while line and line[:1] not in (b'>', b'@', b'+'):
            seq.append(line.strip())
            line = stream.readline()
        seq = b''.join(seq)
        qual = None
        if line[:1] == b'+':
            got, parts = 0, []
            line = stream.readline()
            while line and got < len(seq):
                q = line.rstrip(b'\r\n')
                parts.append(q)
                got += len(q)
                line = stream.readline()
            qual = b''.join(parts)
            if len(qual) != len(seq):
                qual = None
        yield name.decode(), comment, seq, qual


def subseq(in_path, names, out):
    """`seqtk subseq <in.fq> <name.lst>` (the reference writes the human/decoy-filtered reads with it,
    bin/megapath_nano.py:1221-1233): the records whose name is listed, in input order, FASTQ records as
    `@name[ comment]`, sequence, `+`, quality on single lines, FASTA records as `>name[ comment]` and the sequence.
    names: iterable of read names; out: binary stream.  -> number of records written."""
    wanted = {n if isinstance(n, str) else n.decode() for n in names}
    kind, stream = open_once(in_path)
    n = 0
    try:
        if kind == 'index':
            raise ValueError(f'{in_path}: a saved index, not sequences')
        for name, comment, seq, qual in iter_fastx_full(stream):
            if name not in wanted:
                continue
            head = name.encode() + ((b' ' + comment) if comment is not None else b'')
            if qual is not None:
                out.write(b'@' + head + b'\n' + seq + b'\n+\n' + qual + b'\n')
            else:
                out.write(b'>' + head + b'\n' + seq + b'\n')
            n += 1
    finally:
        stream.close()
    return n
