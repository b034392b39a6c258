run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --pcie-steps 0"
$B > gpurun_out/sw.log 2>/dev/null; run base
MPN_SUB_BATCH_BP=16000000 $B > gpurun_out/sw.log 2>/dev/null; run SB16
MPN_SUB_BATCH_BP=32000000 $B > gpurun_out/sw.log 2>/dev/null; run SB32
MPN_SUB_BATCH_BP=16000000 MPN_PIPE_WORKERS=12 MPN_SEED_SLOTS=6 $B > gpurun_out/sw.log 2>/dev/null; run SB16_W12_S6
MPN_PIPE_WORKERS=10 MPN_SEED_SLOTS=5 $B > gpurun_out/sw.log 2>/dev/null; run W10_S5
MPN_PIPE_WORKERS=6 MPN_SEED_SLOTS=3 $B > gpurun_out/sw.log 2>/dev/null; run W6_S3
