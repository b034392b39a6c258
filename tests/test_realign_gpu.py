"""f4 parity on the GPU: the amplicon realigner behind the reference's own C entry points (realign_reads / free_memory) and
the batched form, against the golden vectors made by the reference's compiled sources and against the oracle."""
import json
import os

import numpy as np
import pytest

from megapath_nano_amd import _ffi, realigner
from oracle import realign_oracle as ro
from realign_cases import make_window

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('libmpn', 'oracle_built')]
GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'realign_golden.json')
KEYS = ('seqs', 'positions', 'cigars', 'reference', 'haplotypes', 'ref_start', 'ref_prefix', 'ref_suffix')


def _inputs(c):
    return {k: c[k] for k in KEYS}


def test_reference_abi_matches_golden():
    """every window through realign_reads(), bound as realign_illumina_reads.py binds the reference's library"""
    for c in json.load(open(GOLD)):
        got = realigner.realign_reads(**_inputs(c))
        assert [list(x) for x in got] == c['expected']


def test_batched_form_matches_golden_and_single_calls():
    cases = json.load(open(GOLD))
    got = realigner.realign_batch([_inputs(c) for c in cases])
    assert [[list(x) for x in w] for w in got] == [c['expected'] for c in cases]


def test_random_windows_match_oracle():
    rng = np.random.default_rng(21)
    wins = []
    for k in range(60):
        kw = dict(n_reads=int(rng.integers(1, 60)), n_haps=int(rng.integers(1, 30)), prefix=int(rng.integers(33, 200)),
                  center=int(rng.integers(8, 100)), suffix=int(rng.integers(33, 200)), read_len=int(rng.integers(25, 151)),
                  include_ref=bool(rng.integers(0, 2)), uncovered_hap=bool(rng.integers(0, 3) == 0),
                  repeat=bool(rng.integers(0, 4) == 0))
        wins.append(make_window(7000 + k, **kw))
    got = realigner.realign_batch(wins)
    for w, g in zip(wins, got):
        assert g == ro.realign_reads(**w)


def test_illumina_sized_window():
    """250-base reads, a 1 kb window and 10 haplotypes: the shape realign_illumina_reads.py produces"""
    w = make_window(99, n_reads=300, n_haps=10, prefix=400, center=200, suffix=400, read_len=250)
    assert realigner.realign_reads(**w) == ro.realign_reads(**w)


def test_edge_cases():
    w = make_window(5, n_reads=6, n_haps=2, read_len=30)           # every read <= 32 bases: only the SSW path can place them
    assert realigner.realign_reads(**w) == ro.realign_reads(**w)
    w0 = dict(w, seqs=[], positions=[], cigars=[])
    assert realigner.realign_reads(**w0) == []
    assert realigner.realign_batch([]) == []
    assert realigner.realign_batch([w0, w]) == [[], ro.realign_reads(**w)]
    short = dict(w, haplotypes=['ACGT' * 5])                        # < 32 bases: undefined in the reference, refused here
    with pytest.raises(_ffi.MpnError):
        realigner.realign_reads(**short)
