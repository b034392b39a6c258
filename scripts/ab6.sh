run() { python -c "
import json,sys;d=json.loads(open('gpurun_out/sw.log').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), flush=True)" "$1"; }
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc
B="timeout -k 10 400 python bench.py --no-cpu-baseline --warmup 2 --pcie-steps 0 --steps 6"
for t in 8 12 16 24 32; do MPN_HOST_THREADS=$t $B > gpurun_out/sw.log 2>/dev/null; run T$t; done
