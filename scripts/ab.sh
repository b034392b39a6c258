# A/B helper for the GPU box: two runs of the default bench, value and ms/step
run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), 'cpu_s/step', d['host_cpu_s_per_step'], flush=True)" "$1"; }
B="timeout -k 10 400 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --pcie-steps 0"
$B > gpurun_out/sw.log 2>/dev/null; run A1
$B > gpurun_out/sw.log 2>/dev/null; run A2
