run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), round(d['per_step']['wall_total_ns'],1), flush=True)" "$1"; }
for W in 12 16; do for SB in 16000000 24000000; do
MPN_PIPE_WORKERS=$W MPN_SUB_BATCH_BP=$SB MPN_HOST_THREADS=48 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run W${W}_SB${SB}_T48
done; done
MPN_HOST_THREADS=48 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run W8_SB24_T48
