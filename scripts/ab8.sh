run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), 'cpu_s/step', d['host_cpu_s_per_step'], flush=True)" "$1"; grep "worker-thread" gpurun_out/cpu.err | tail -1 | cut -c1-200; }
B="timeout -k 10 400 python bench.py --no-cpu-baseline --warmup 2 --pcie-steps 0 --steps 4"
MPN_DEBUG_CPU=1 $B > gpurun_out/sw.log 2>gpurun_out/cpu.err; run base
ROC_ACTIVE_WAIT_TIMEOUT=0 MPN_DEBUG_CPU=1 $B > gpurun_out/sw.log 2>gpurun_out/cpu.err; run ACTIVE_WAIT0
HSA_ENABLE_INTERRUPT=1 ROC_ACTIVE_WAIT_TIMEOUT=0 ROC_SIGNAL_POOL_SIZE=256 MPN_DEBUG_CPU=1 $B > gpurun_out/sw.log 2>gpurun_out/cpu.err; run INT_AW0
