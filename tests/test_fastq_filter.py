"""Read filter: the oracle's arithmetic + the host mirror's assembly reproduce the reference binary's
stdout/stderr byte for byte (CPU, golden fixture made by the reference's own prebuilt nanofastq); the HIP kernel's sums
are bit-identical to the oracle's and the product path reproduces the same bytes (-m gpu)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'nanofastq_golden.json')))


def kw_of(args):
    keys = {'-q': 'min_quality', '-l': 'min_length', '-h': 'head_crop', '-t': 'tail_crop', '-r': 'read_id_prefix'}
    kw = {}
    for a, v in zip(args[::2], args[1::2]):
        kw[keys[a]] = v if a == '-r' else int(v)
    return kw


def oracle_outputs(text, kw):
    """parse (host mirror) -> oracle sums -> assemble (host mirror)"""
    from megapath_nano_amd import fastq_filter as ff
    from oracle import fastq_oracle as fo
    recs = ff.parse_fastx(text.encode('latin-1'))
    total, cropped = np.zeros(len(recs)), np.zeros(len(recs))
    for k, r in enumerate(recs):
        if r[3] is not None:
            total[k], cropped[k] = fo.qsums(r[3], kw.get('head_crop', 0), kw.get('tail_crop', 0), max(kw.get('min_length', 0), 1))
    out, info, _ = ff.assemble(recs, total, cropped, **kw)
    return out.decode('latin-1'), info.decode('latin-1')


def test_oracle_reproduces_reference_binary():
    assert len(GOLD['cases']) >= 28
    wraps = 0
    for c in GOLD['cases']:
        out, info = oracle_outputs(GOLD['inputs'][c['input']], kw_of(c['args']))
        assert out == c['stdout'], (c['input'], c['args'])
        assert info == c['stderr'], (c['input'], c['args'])
        wraps += '\t1844674407370955' in info
    assert wraps > 0, 'the size_t wrap-around quirk is not exercised'
    assert any('-0.00' in c['stderr'] for c in GOLD['cases'])


def test_parser_kseq_conventions():
    from megapath_nano_amd.fastq_filter import parse_fastx
    recs = parse_fastx(b'@a  two spaces\nAC\nGT\n+anything\nII\nII\n>b\tc\nAAAA\n\n@c\nA\n+\n@\n')
    assert recs == [(b'a', b' two spaces', b'ACGT', b'IIII'), (b'b', b'c', b'AAAA', None), (b'c', None, b'A', b'@')]
    with pytest.raises(ValueError):
        parse_fastx(b'@a\nACGT\n+\nII\n')


@pytest.mark.gpu
def test_hip_sums_are_bit_identical_and_outputs_match(libmpn):
    from megapath_nano_amd import fastq_filter as ff
    from oracle import fastq_oracle as fo
    rng = np.random.default_rng(5)
    quals = [bytes(rng.integers(33, 33 + 61, size=int(rng.integers(1, 9000)), dtype=np.uint8)) for _ in range(300)]
    for h, t, m in ((0, 0, 1), (50, 30, 100), (8000, 8000, 1), (0, 1, 1)):
        total, cropped = ff.qsums(quals, h, t, m)
        for k, q in enumerate(quals):
            a, b = fo.qsums(q, h, t, m)
            assert total[k] == a and cropped[k] == b, (h, t, m, k)          # exact doubles, not a tolerance
    for c in GOLD['cases']:
        out, info, kept = ff.filter_fastx(GOLD['inputs'][c['input']].encode('latin-1'), **kw_of(c['args']))
        assert out.decode('latin-1') == c['stdout'] and info.decode('latin-1') == c['stderr'], (c['input'], c['args'])
        assert len(kept) == sum(1 for l in info.decode('latin-1').splitlines() if l.endswith('\t1'))
    with pytest.raises(ValueError):
        ff.qsums([b'II\x1fI'], 0, 0, 1)
    assert ff.filter_fastx(b'')[:2] == (b'', b'')


@pytest.mark.gpu
def test_executable_drop_in(libmpn):
    c = next(c for c in GOLD['cases'] if c['input'] == 'ont_like' and '-r' in c['args'])
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'mpn-nanofastq')] + c['args'], input=GOLD['inputs']['ont_like'].encode('latin-1'),
                       capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr[-500:]
    assert p.stdout.decode('latin-1') == c['stdout'] and p.stderr.decode('latin-1') == c['stderr']
