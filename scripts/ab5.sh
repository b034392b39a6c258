run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
B="timeout -k 10 400 python bench.py --no-cpu-baseline --warmup 2 --pcie-steps 0"
$B --steps 6 > gpurun_out/sw.log 2>/dev/null; run base131k
$B --steps 4 --reads-per-step 262144 > gpurun_out/sw.log 2>/dev/null; run reads262k
MPN_PIPE_WORKERS=16 MPN_SEED_SLOTS=8 MPN_HOST_THREADS=64 GPU_MAX_HW_QUEUES=32 $B --steps 6 > gpurun_out/sw.log 2>/dev/null; run W16_S8_T64_Q32
MPN_HOST_THREADS=64 $B --steps 6 > gpurun_out/sw.log 2>/dev/null; run T64
