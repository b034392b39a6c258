run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), {k[5:-3]:round(v) for k,v in d['per_step'].items() if k.startswith('wall')}, flush=True)" "$1"; }
export MPN_NO_STAGGER=1 GPU_MAX_HW_QUEUES=16
for W in 8 10 12; do for T in 16 32; do for SB in 32000000 48000000; do
MPN_PIPE_WORKERS=$W MPN_HOST_THREADS=$T MPN_SUB_BATCH_BP=$SB timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/sw.log 2>&1; run W${W}_T${T}_SB$SB
done; done; done
