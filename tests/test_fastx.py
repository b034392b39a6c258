"""CPU tests of the FASTA/FASTQ ingestion (host logic): one open per path, FIFO targets fed with concatenated gzip
members in chunks (the reference's own way of handing the target to the aligner, aligner.py:143-144,209-217)."""
import gzip
import os
import threading
import time

import pytest

from megapath_nano_amd import fastx


def feed_fifo(pipe, files, chunk=7):
    """What the reference's `cat a.fna.gz b.fna.gz ... > FIFO` does, in small writes."""
    def run():
        with open(pipe, 'wb') as p:
            for fn in files:
                data = open(fn, 'rb').read()
                for i in range(0, len(data), chunk):
                    p.write(data[i:i + chunk])
                    p.flush()
                time.sleep(0.01)
    t = threading.Thread(target=run, daemon=True)
    t.start()
    return t


def test_fifo_of_concatenated_gzip_members(tmp_path):
    files, want = [], []
    for i in range(4):
        fn = tmp_path / f'g{i}.fna.gz'
        with gzip.open(fn, 'wb') as f:
            f.write(b'>seq%d some description\nACGTACGT\nAC\n\n>other%d\nTTTTGGGG\n' % (i, i))
        files.append(str(fn))
        want += [(f'seq{i}', b'ACGTACGTAC'), (f'other{i}', b'TTTTGGGG')]
    pipe = str(tmp_path / 'pipe')
    os.mkfifo(pipe)
    t = feed_fifo(pipe, files)
    done = {}
    r = threading.Thread(target=lambda: done.setdefault('recs', fastx.read_fastx(pipe)), daemon=True)
    r.start()
    r.join(20)
    assert not r.is_alive(), 'reader blocked on the FIFO'
    t.join(5)
    assert done['recs'] == want


def test_plain_fifo_and_magic_sniff(tmp_path):
    pipe = str(tmp_path / 'pipe')
    os.mkfifo(pipe)
    fa = tmp_path / 'a.fa'
    fa.write_bytes(b'>x\nAC\nGT\n')
    t = feed_fifo(pipe, [str(fa)], chunk=2)
    kind, stream = fastx.open_once(pipe)
    assert kind == 'plain'
    assert list(fastx.iter_fastx(stream)) == [('x', b'ACGT', None)]
    stream.close()
    t.join(5)
    idx = tmp_path / 'i.mpi'
    idx.write_bytes(fastx.INDEX_MAGIC + b'\0' * 64)
    kind, stream = fastx.open_once(str(idx))
    assert kind == 'index' and stream.read(8) == fastx.INDEX_MAGIC
    stream.close()
    assert fastx.is_fifo(pipe) and not fastx.is_fifo(str(fa)) and not fastx.is_fifo(str(tmp_path / 'missing'))


def test_fastq_records_and_qualities(tmp_path):
    fq = tmp_path / 'r.fq.gz'
    with gzip.open(fq, 'wb') as f:
        f.write(b'@r1 comment\nACGT\n+\n@III\n@r2\nAC\nGT\n+r2\n>I\nI@\n>fa\nAAA\r\n@r3\nNN\n+\n!!\n')
    assert fastx.read_fastx(str(fq), with_qual=True) == [('r1', b'ACGT', b'@III'), ('r2', b'ACGT', b'>II@'), ('fa', b'AAA', None),
                                                           ('r3', b'NN', b'!!')]
    assert fastx.read_fastx(str(fq)) == [('r1', b'ACGT'), ('r2', b'ACGT'), ('fa', b'AAA'), ('r3', b'NN')]
    empty = tmp_path / 'e.fa'
    empty.write_bytes(b'')
    assert fastx.read_fastx(str(empty)) == []
    bad = tmp_path / 'b.txt'
    bad.write_bytes(b'hello\n')
    with pytest.raises(ValueError):
        fastx.read_fastx(str(bad))


def test_subseq_like_seqtk(tmp_path):
    import io
    fq = tmp_path / 'r.fq.gz'
    with gzip.open(fq, 'wb') as f:
        f.write(b'@r1 first read\nACGT\n+\nIIII\n@r2\nAC\nGT\n+\nII\nII\n>fa1 desc here\nAAAA\nCC\n@r3\tx\nGG\n+\n##\n')
    out = io.BytesIO()
    assert fastx.subseq(str(fq), ['r3', 'r2', b'fa1', 'missing'], out) == 3
    assert out.getvalue() == b'@r2\nACGT\n+\nIIII\n>fa1 desc here\nAAAACC\n@r3 x\nGG\n+\n##\n'
    out = io.BytesIO()
    assert fastx.subseq(str(fq), [], out) == 0 and out.getvalue() == b''
