# is the GPU saturated by one process?  one bench alone, then two at once (each its own index), aggregate compared
B="timeout -k 10 400 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --pcie-steps 0 --genomes 2500"
$B > gpurun_out/tp_single.log 2>/dev/null
$B > gpurun_out/tp_a.log 2>/dev/null &
PA=$!
$B > gpurun_out/tp_b.log 2>/dev/null &
PB=$!
wait $PA; wait $PB
python - <<'PY'
import json
v=lambda f: json.loads(open(f).read().strip().splitlines()[-1])['value']
s,a,b=v('gpurun_out/tp_single.log'),v('gpurun_out/tp_a.log'),v('gpurun_out/tp_b.log')
print('single %.1f   two at once %.1f + %.1f = %.1f Gbp/min' % (s,a,b,a+b))
PY
