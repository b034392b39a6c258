import sys, os, time, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
print('cpus',os.cpu_count(),flush=True)
from megapath_nano_amd import synth, mapper
ng=int(sys.argv[1]); glen=int(sys.argv[2]); nr=int(sys.argv[3]); mean=int(sys.argv[4]); strains=int(sys.argv[5]) if len(sys.argv)>5 else 2
reps=int(sys.argv[6]) if len(sys.argv)>6 else 2
t=time.time(); gen=synth.make_genomes(1,ng,glen,strain_pairs=strains); print('genomes',time.time()-t,flush=True)
t=time.time(); idx=mapper.Index(gen); print('index build',time.time()-t,'n_mz',idx.n_minimizers,'keys',idx.n_keys,'mid_occ',idx.mid_occ(),flush=True)
t=time.time(); reads=synth.make_reads(2,gen,nr,mean_len=mean); print('reads',time.time()-t, 'bases', sum(len(r['seq']) for r in reads),flush=True)
opt=mapper.default_opt(best_n=50,pri_ratio=1.0)
names=[r['name'] for r in reads]; seqs=[r['seq'] for r in reads]
for rep in range(reps):
    t=time.time(); paf=mapper.map_batch(idx,opt,names,seqs); dt=time.time()-t
    st=mapper.last_stats()
    print('map_batch wall',round(dt,3),'s  Gbp/min',round(st['bases']/dt*60/1e9,3), 'lines',paf.count('\n'),flush=True)
    print(json.dumps({k:(round(v/1e6,2) if k.endswith('_ns') else v) for k,v in st.items()}),flush=True)
