run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
B="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --pcie-steps 0"
for k in 8 4 3 2; do MPN_SEED_SLOTS=$k $B > gpurun_out/sw.log 2>/dev/null; run SLOTS$k; done
MPN_SEED_SLOTS=3 MPN_PIPE_WORKERS=12 $B > gpurun_out/sw.log 2>/dev/null; run SLOTS3_W12
MPN_SEED_SLOTS=4 MPN_PIPE_WORKERS=12 $B > gpurun_out/sw.log 2>/dev/null; run SLOTS4_W12
