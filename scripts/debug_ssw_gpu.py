import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import subprocess
subprocess.check_call(['make','-s','-C',os.path.join(ROOT,'oracle')],stdout=subprocess.DEVNULL)
from ssw_cases import make_cases
from test_ssw_gpu import run_batch_grouped
from oracle.ssw_bindings import oracle_align
n=int(sys.argv[1]) if len(sys.argv)>1 else 300
cases=make_cases(777,n)
got=run_batch_grouped(cases)
bad=0; cat={}
for i,(c,g) in enumerate(zip(cases,got)):
    w=oracle_align(read=c['read'],ref=c['ref'],mat=c['mat'],gap_open=c['gap_open'],gap_extend=c['gap_extend'],flag=c['flag'],filters=c['filters'],filterd=c['filterd'],mask=c['mask'],score_size=c['score_size'])
    if g!=w:
        bad+=1
        if isinstance(g,tuple) and isinstance(w,tuple):
            k='score' if g[:2]!=w[:2] else 'ends' if (g[3],g[5],g[6])!=(w[3],w[5],w[6]) else 'begin' if (g[2],g[4])!=(w[2],w[4]) else 'cigar'
        else: k=f'{type(g).__name__ if not isinstance(g,str) else g}->{type(w).__name__ if not isinstance(w,str) else w}'
        cat[k]=cat.get(k,0)+1
        if bad<=10: print(i,k,'L',len(c['read']),len(c['ref']),'flag',c['flag'],'mask',c['mask'],'ss',c['score_size'],'go',c['gap_open'],c['gap_extend'],'\n  got ',g if not isinstance(g,tuple) else (g[:7],g[7][:8]),'\n  want',w if not isinstance(w,tuple) else (w[:7],w[7][:8]))
print('total',n,'bad',bad,cat)
