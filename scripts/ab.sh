run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
GPU_MAX_HW_QUEUES=32 MPN_PIPE_WORKERS=12 MPN_HOST_THREADS=48 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run Q32_W12_T48
GPU_MAX_HW_QUEUES=32 MPN_PIPE_WORKERS=8 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run Q32_W8
GPU_MAX_HW_QUEUES=24 MPN_PIPE_WORKERS=10 MPN_HOST_THREADS=40 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/sw.log 2>&1; run Q24_W10_T40
