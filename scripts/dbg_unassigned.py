"""debug: which reads end up without a name (run on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from megapath_nano_amd import mapper, synth
import bench
class A: pass
args = A(); args.genome_len = 4000000; args.reads_per_step = 65536; args.mean_len = 8000
n, sp = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 10
dev = torch.device('cuda', 0)
members, weights = bench.community(n, sp)
names, flat, lens = synth.make_genomes_device(20240901, n, args.genome_len, sp, dev)
idx = mapper.Index.from_device(names, flat.data_ptr(), lens)
opt = mapper.default_opt(best_n=50, pri_ratio=1.0); opt.mid_occ = idx.mid_occ()
b = bench.make_batch(flat, weights, args, 1000, dev)
_, c = mapper.map_batch_ex(idx, opt, b, want_paf=False, want_cols=True)
mapped = np.zeros(b.n, bool); mapped[c['read_idx']] = True
un = np.flatnonzero(~mapped)
print('unmapped', len(un), 'of', b.n, 'mid_occ', opt.mid_occ)
g = b.truth['genome'][un]
print('by genome', dict(zip(*np.unique(g, return_counts=True))))
print('length quantiles', np.quantile(b.lens[un], [0, .25, .5, .75, 1]) if len(un) else None)
sh = b.lens < 500
print('reads < 500bp:', int(sh.sum()), 'unmapped among them', int((~mapped & sh).sum()))
for gg in (0, 1, 2, 7):
    m = b.truth['genome'] == gg
    print('genome', gg, 'reads', int(m.sum()), 'unmapped', int((~mapped & m).sum()))
