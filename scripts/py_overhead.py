"""Where does a step spend time outside mpn_map_batch_ex?  cProfile of pipeline.align_and_assign on a small world."""
import argparse
import cProfile
import os
import pstats
import random
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from megapath_nano_amd import mapper
    from megapath_nano_amd.pipeline import align_and_assign
    args = argparse.Namespace(genomes=60, genome_len=2000000, strain_pairs=10, reads_per_step=int(sys.argv[1]) if len(sys.argv) > 1 else 65536,
                              mean_len=8000)
    dev = torch.device('cuda', 0)
    genomes, weights, tax = bench.build_world(args, 0)
    idx = mapper.Index(genomes)
    opt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    opt.mid_occ = idx.mid_occ()
    batches = [bench.make_batch(genomes, weights, args, 1000 + s, dev) for s in range(2)]
    rnd = random.Random(1)
    align_and_assign(idx, opt, batches[0][0], tax, rng=rnd)
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    align_and_assign(idx, opt, batches[1][0], tax, rng=rnd)
    pr.disable()
    print('step ms', (time.perf_counter() - t0) * 1e3, 'mapper wall ms', mapper.last_stats()['wall_total_ns'] / 1e6)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)


main()
