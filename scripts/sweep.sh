for W in 4 6 8; do for SB in 32000000 48000000; do
MPN_PIPE_WORKERS=$W MPN_SUB_BATCH_BP=$SB timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/sw.log 2>&1; python -c "
import json;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print('W=$W SB=$SB', round(d['value'],2), round(d['ms_per_step'],1), {k[5:-3]:round(v) for k,v in d['per_step'].items() if k.startswith('wall')}, flush=True)"; done; done
