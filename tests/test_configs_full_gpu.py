"""BASELINE.json configs[1] and configs[2] at FULL size on the GPU box (-m gpu), so that the driver -- not only the builder's
own bench runs -- exercises them:

  C2  3.1 Gbp human-like genome (24 x 129 Mbp, 45 % interspersed repeats) + 8 decoys, 100 000 reads, -x map-ont -N 5 -p 0.8
      (megapath_nano.py:1124), human / decoy / microbe classification (:1135-1200): every class >= 97 % correct for reads >= 2 kb,
      the independent PAF checker over the lines of 1024 reads, and a 60-read sample identical to the CPU oracle indexing the
      SAME 3.1 Gbp (the oracle's -f cut-off must agree with the GPU's).
  C3  20 Gbp of targets (5000 x 4 Mbp incl. 10 strain copies), 262 144 reads, -N 50 -p 1 (megapath_nano.py:1270), reassignment
      on: bench.py's correctness block at the full index -- primaries on the sampled locus, PAF checker over slices fetched
      from HBM, per-name counts against the sampled composition.

The workloads are bench.py's own (`bench.build_workload`), generated on the GPU; about two minutes together."""
import argparse
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def bench_args(config, genomes, reads):
    return argparse.Namespace(config=config, genomes=genomes, genome_len=4000000, strain_pairs=10, reads_per_step=reads, mean_len=8000,
                              distinct_batches=1, warmup=0, steps=1, no_cpu_baseline=True, cpu_index_genomes=250, mapping_only=False, parts=4)


def free_workload(W):
    import torch
    for i in (W['idx'] if isinstance(W['idx'], (list, tuple)) else [W['idx']]):
        i.close()
    W.clear()
    torch.cuda.empty_cache()


def test_c3_full_size_correctness_block(libmpn):
    import torch
    import bench
    from megapath_nano_amd.pipeline import align_and_assign
    args = bench_args('c3', 5000, 262144)
    W = bench.build_workload(args, torch.device('cuda', 0), 0, 1)
    try:
        assert W['index_bp'] == 20_000_000_000 and W['batches'][0].n == 262144
        b = W['batches'][0]
        out = align_and_assign(W['idx'], W['opt'], b, W['tax'], rng=random.Random(12345), use_device=False)
        sampled = np.bincount(b.truth['genome'], minlength=W['n'])
        res = bench.correctness_block(W['idx'], W['opt'], b, args, W['members'], W['twin_of'], out['read_count'], sampled)
        assert res['ok'], res['failures']
        assert res['reads_ge_1kb'] >= 15000 and res['primary_on_true_locus_frac_ge_1kb'] >= 0.99
        assert res['paf_check']['lines'] >= 2000 and res['paf_check']['as_equals_cigar_score'] == res['paf_check']['lines']
        assert res['counts']['reads_assigned'] >= 0.97 * 262144 and res['counts']['assigned_outside_community'] == 0
    finally:
        free_workload(W)


def test_c2_full_size_classification_and_oracle_sample(libmpn, oracle_built):
    import torch
    import bench
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    args = bench_args('c2', 0, 100000)
    dev = torch.device('cuda', 0)
    W = bench.build_workload(args, dev, 0, 1)
    try:
        idx, opt, b, kind = W['idx'], W['opt'], W['batches'][0], W['kind']
        assert 3.0e9 < W['index_bp'] < 3.2e9 and b.n == 100000
        rnd = random.Random(12345)
        step = bench.human_decoy_step(idx, opt, b, kind, rnd, use_device=False)
        assert int(step['read_count'].sum()) == 100000
        res = bench.c2_correctness(idx, opt, b, kind, rnd)
        assert res['ok'], res['failures']
        for cls in ('human', 'decoy', 'microbe'):
            assert res[f'{cls}_classified_frac'] >= 0.97 and res[f'{cls}_reads_ge_2kb'] > 100, (cls, res)
        assert res['paf_check']['lines'] >= 500
        # the oracle on the SAME targets (fetched back from the 2-bit copy in HBM: the ASCII left with the generator)
        genomes = [(name, np.frombuffer(idx.fetch_seq(i, 0, int(idx.lens[i])), dtype=np.uint8)) for i, name in enumerate(idx.names)]
        os.environ.setdefault('OMP_NUM_THREADS', str(max(1, min(16, os.cpu_count() or 1))))
        oidx = mb.Index(genomes)
        oopt = mb.default_opt(best_n=5, pri_ratio=0.8)
        oopt.mid_occ = oidx.mid_occ()
        assert int(oopt.mid_occ) == int(opt.mid_occ), 'the -f cut-off of the oracle and of the GPU index differ'
        rng = np.random.default_rng(5)
        pick = sorted(int(i) for i in rng.choice(b.n, size=60, replace=False))
        sub = mapper.PackedReads([b.names[i] for i in pick], [b.seq(i) for i in pick])
        paf, _ = mapper.map_batch_ex(idx, opt, sub, want_paf=True, want_cols=False, use_device=False)
        by = {}
        for line in paf.splitlines(keepends=True):
            by.setdefault(line.split('\t', 1)[0], []).append(line)
        n_lines = 0
        for i in pick:
            want = mb.map_read(oidx, oopt, b.names[i], b.seq(i))[2]
            assert ''.join(by.get(b.names[i], [])) == want, b.names[i]
            n_lines += want.count('\n')
        assert n_lines >= 30
        oidx.close()
    finally:
        free_workload(W)
