run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
MPN_SUB_BATCH_BP=16000000 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run SB16
MPN_SUB_BATCH_BP=32000000 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run SB32
MPN_PIPE_WORKERS=10 MPN_HOST_THREADS=40 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run W10_T40
