run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), 'cpu_s/step', d['host_cpu_s_per_step'], flush=True)" "$1"; }
B="timeout -k 10 400 python bench.py --no-cpu-baseline --warmup 2 --pcie-steps 0 --steps 4"
MPN_HOST_THREADS=2 $B > gpurun_out/sw.log 2>/dev/null; run T2
MPN_HOST_THREADS=16 MPN_PIPE_WORKERS=2 $B > gpurun_out/sw.log 2>/dev/null; run T16_W2
MPN_HOST_THREADS=16 HIP_LAUNCH_BLOCKING=0 HSA_ENABLE_INTERRUPT=1 $B > gpurun_out/sw.log 2>/dev/null; run T16_int
