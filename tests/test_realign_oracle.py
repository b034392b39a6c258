"""f4 oracle: the Python restatement of realigner.cpp (oracle/realign_oracle.py) against the golden vectors produced by the
reference's own sources compiled in place, and -- where that build is present -- against it directly on random windows."""
import json
import os

import numpy as np
import pytest

from oracle import realign_oracle as ro
from oracle.realign_bindings import have_ref, ref_realign
from realign_cases import make_window

pytestmark = pytest.mark.usefixtures('oracle_built')
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'realign_golden.json')
KEYS = ('seqs', 'positions', 'cigars', 'reference', 'haplotypes', 'ref_start', 'ref_prefix', 'ref_suffix')


def test_oracle_matches_golden():
    cases = json.load(open(GOLD))
    assert len(cases) >= 15
    n = 0
    for c in cases:
        got = ro.realign_reads(**{k: c[k] for k in KEYS})
        assert [list(x) for x in got] == c['expected']
        n += len(got)
    assert n > 600


def test_golden_covers_the_paths():
    """the fixtures exercise the fast (k-mer + Hamming) path, the SSW path, untouched reads and clipped results"""
    cases = json.load(open(GOLD))
    flat = [(e, p, c) for cs in cases for e, p, c in zip(cs['expected'], cs['positions'], cs['cigars'])]
    kinds = {''.join(ch for ch in e[1] if not ch.isdigit()) for e, _, _ in flat}
    assert {'X', 'XDX', 'XIX'} <= kinds and any('S' in k for k in kinds)
    assert any(e == [p, c] for e, p, c in flat)          # a read the realigner leaves alone
    assert any(len(cs['haplotypes']) > 16 for cs in cases)   # std::sort leaves its insertion-sort regime


def test_libstdcxx_sort_is_a_permutation_and_sorted():
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 15, 16, 17, 40, 200):
        keys = [int(x) for x in rng.integers(0, 5, n)]
        perm = ro.libstdcxx_sort(keys)
        assert sorted(perm) == list(range(n))
        assert [keys[i] for i in perm] == sorted(keys)
        if n <= 16:   # insertion sort only: stable
            assert perm == sorted(range(n), key=lambda i: keys[i])


@pytest.mark.skipif(not have_ref(), reason='oracle/_ref/librealigner.so not built (needs /root/reference)')
def test_oracle_matches_compiled_reference_on_random_windows():
    rng = np.random.default_rng(11)
    for k in range(40):
        kw = dict(n_reads=int(rng.integers(1, 50)), n_haps=int(rng.integers(1, 30)), prefix=int(rng.integers(33, 200)),
                  center=int(rng.integers(8, 100)), suffix=int(rng.integers(33, 200)), read_len=int(rng.integers(25, 151)),
                  include_ref=bool(rng.integers(0, 2)), uncovered_hap=bool(rng.integers(0, 3) == 0),
                  repeat=bool(rng.integers(0, 4) == 0))
        c = make_window(5000 + k, **kw)
        assert ro.realign_reads(**c) == ref_realign(**c), kw
