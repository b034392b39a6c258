"""Throughput of the read-filter sums (mpn_fastq_qsum_batch, host pointers in, PCIe inclusive) on synthetic qualities."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megapath_nano_amd import fastq_filter as ff  # noqa: E402

rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
lens = np.maximum(200, rng.gamma(1.6, 8000 / 1.6, size=n)).astype(np.int64)
blob = rng.integers(33, 33 + 41, size=int(lens.sum()), dtype=np.uint8).tobytes()
offs = np.concatenate([[0], np.cumsum(lens)])
quals = [blob[offs[i]:offs[i + 1]] for i in range(n)]
ff.qsums(quals[:1000], 50, 30, 100)
t0 = time.perf_counter()
total, cropped = ff.qsums(quals, 50, 30, 100)
dt = time.perf_counter() - t0
print(f'{n} reads, {lens.sum() / 1e9:.2f} Gbase of qualities: {dt * 1e3:.0f} ms incl. packing + H2D + D2H = {lens.sum() / dt / 1e9:.2f} GB/s '
      f'({lens.sum() / dt * 60 / 1e9:.0f} Gbp/min); checksum {float(total.sum()):.6f}')
