for cfg in "MPN_SEED_SLOTS=4" "MPN_SEED_SLOTS=8" "MPN_SEED_SLOTS=12" "MPN_PIPE_WORKERS=16 MPN_SEED_SLOTS=16"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python3 bench.py --gpus 1 --steps 6 --warmup 2 --pcie-steps 0 --mapping-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_step']; print(round(d['value'],2), round(d['ms_per_step'],1), d['host_cpu_s_per_step'], 'filter', p['k_seed_filter_ns'], 'sort', p['ev_sort_ns'], 'seed', p['ev_seed_ns'], 'chain', p['k_chain_dp_ns'])"
done
