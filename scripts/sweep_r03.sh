# usage (GPU box): bash scripts/sweep_r03.sh "VAR=a VAR2=b" "VAR=c" ...  -> one short bench per environment setting
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 --no-correctness 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_step']; print(round(d['value'],2), round(d['ms_per_step'],1), d['host_cpu_s_per_step'], 'filter ms', p['k_seed_filter_ns'], 'strips', p['k_strip16_ns'], 'xstrips', p['k_xstrip_ns'], 'emitted M', round(p['anchors_emitted']/1e6), 'kept M', round(p['anchors_kept']/1e6), 'sort ms', p['ev_sort_ns'], 'seed ms', p['ev_seed_ns'])"
done
