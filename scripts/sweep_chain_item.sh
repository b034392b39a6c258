set -e
export TMPDIR=/tmp
for w in 1024 256 64; do
  MPN_CHAIN_ITEM=$w timeout -k 10 300 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-correctness > gpurun_out/ci_${w}.json 2> gpurun_out/ci_${w}.err
  echo $w done
done
for w in 1024 128; do
MPN_CHAIN_ITEM=$w timeout -k 10 300 python3 bench.py --config c3 --steps 8 --warmup 3 --no-cpu-baseline --no-correctness > gpurun_out/ci_c3_${w}.json 2> gpurun_out/ci_c3_${w}.err
done
echo c3 done
