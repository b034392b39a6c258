run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 2 --pcie-steps 0"
$B > gpurun_out/sw.log 2>/dev/null; run base
MPN_PIPE_WORKERS=12 $B > gpurun_out/sw.log 2>/dev/null; run W12
MPN_PIPE_WORKERS=16 MPN_HOST_THREADS=64 $B > gpurun_out/sw.log 2>/dev/null; run W16_T64
MPN_HOST_THREADS=64 $B > gpurun_out/sw.log 2>/dev/null; run T64
MPN_SUB_BATCH_BP=12000000 MPN_PIPE_WORKERS=16 MPN_HOST_THREADS=64 $B > gpurun_out/sw.log 2>/dev/null; run SB12_W16_T64
MPN_SUB_BATCH_BP=48000000 $B > gpurun_out/sw.log 2>/dev/null; run SB48
GPU_MAX_HW_QUEUES=24 MPN_PIPE_WORKERS=12 $B > gpurun_out/sw.log 2>/dev/null; run Q24_W12
