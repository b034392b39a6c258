import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from megapath_nano_amd import synth, mapper
import argparse
sys.argv=[sys.argv[0]]
import bench
class A: genomes=250; genome_len=4000000; strain_pairs=10; reads_per_step=6000; mean_len=8000
genomes, weights, tax = bench.build_world(A, 0)
idx=mapper.Index(genomes)
reads=synth.make_reads(5,genomes,A.reads_per_step,mean_len=A.mean_len,weights=weights)
opt=mapper.default_opt(best_n=50,pri_ratio=1.0); opt.mid_occ=idx.mid_occ()
print('mid_occ',opt.mid_occ)
for rep in range(2):
    t=time.time(); res=mapper.seed_chain_batch(idx,opt,[r['seq'] for r in reads]); dt=time.time()-t
    st=mapper.last_stats()
    na=np.array([r['n_anchor'] for r in res]); L=np.array([len(r['seq']) for r in reads])
    print('wall',round(dt,3),'chain_dp ms',st['ev_chain_dp_ns']/1e6,'anchors total',na.sum(),'max',na.max(),'p99',np.percentile(na,99),'mean',na.mean(),'maxlen',L.max(), 'anchors/len max', (na/L).max())
