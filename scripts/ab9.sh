run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), 'cpu_s/step', d['host_cpu_s_per_step'], flush=True)" "$1"; grep "worker-thread" gpurun_out/cpu.err | tail -1 | cut -c1-200; }
B="timeout -k 10 400 python bench.py --no-cpu-baseline --warmup 2 --pcie-steps 0 --steps 5"
MPN_DEBUG_CPU=1 $B > gpurun_out/sw.log 2>gpurun_out/cpu.err; run poll
$B > gpurun_out/sw.log 2>gpurun_out/cpu.err; run poll_nodebug
MPN_PIPE_WORKERS=12 MPN_SEED_SLOTS=6 $B > gpurun_out/sw.log 2>gpurun_out/cpu.err; run poll_W12_S6
