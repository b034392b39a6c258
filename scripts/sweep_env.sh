# usage: scripts/sweep_env.sh "VAR=a VAR2=b" "VAR=c" ...  -> one short bench per environment setting (GPU box)
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 300 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 > gpurun_out/sw_$tag.log 2> gpurun_out/sw_$tag.err || exit 1
  echo "$cfg: $(python scripts/show_bench.py gpurun_out/sw_$tag.log | head -1)"
done
