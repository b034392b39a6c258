import os, sys, time, random
sys.path.insert(0, os.getcwd())
os.environ.setdefault('GPU_MAX_HW_QUEUES', '20')
import numpy as np, torch
import bench
from megapath_nano_amd import mapper, synth
from megapath_nano_amd.pipeline import Taxonomy, align_and_assign, sharded_tiebreak
from megapath_nano_amd.reassignment import ReassignPlan
class A: pass
args = A(); args.config='c3'; args.genomes=2000; args.genome_len=4000000; args.strain_pairs=10; args.reads_per_step=262144; args.mean_len=8000
args.distinct_batches=1; args.warmup=1; args.steps=1; args.no_cpu_baseline=True; args.cpu_index_genomes=250; args.mapping_only=False
dev = torch.device('cuda', 0)
W = bench.build_workload(args, dev, 0, 1)
idx, tax, opt, b = W['idx'], W['tax'], W['opt'], W['batches'][0]
rnd = random.Random(1)
for it in range(3):
    t0, c0 = time.perf_counter(), time.thread_time()
    _, c = mapper.map_batch_ex(idx, opt, b, want_paf=False, want_cols=True, use_device=True)
    t1, c1 = time.perf_counter(), time.thread_time()
    keep = c['as_'] >= 0
    read_idx = c['read_idx'][keep]; rid = c['rid'][keep]; score = c['as_'][keep]
    aligned_bp = (c['re'][keep] - c['rs'][keep]).astype(np.int64)
    tb = sharded_tiebreak(rnd, len(read_idx), (0, 1), None)
    t2, c2 = time.perf_counter(), time.thread_time()
    plan = ReassignPlan(read_idx, tax.name_code[rid], score, tb, aligned_bp, tax.species_code[rid], tax.n_names, tax.n_species)
    a_, u_, m_ = plan.counts()
    out = plan.apply(a_, u_, np.arange(tax.n_names, dtype=np.int32), 0.05, 0.05, 0.0)
    plan.close()
    t3, c3 = time.perf_counter(), time.thread_time()
    print(f'map wall {t1-t0:.3f} main-cpu {c1-c0:.3f} | numpy wall {t2-t1:.3f} cpu {c2-c1:.3f} | reassign wall {t3-t2:.3f} cpu {c3-c2:.3f}', flush=True)
