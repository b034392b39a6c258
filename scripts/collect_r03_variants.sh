# run on the GPU box: the bench variants VERDICT r2 asked for beside the headline (each prints one JSON line)
set -e
O=gpurun_out
timeout -k 10 500 python3 bench.py --config strain --reads-per-step 65536 --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --pcie-steps 0 > $O/r03_variant_strain.log 2> $O/r03_variant_strain.err; echo strain rc=$?
timeout -k 10 500 python3 bench.py --config parts --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 > $O/r03_variant_parts.log 2> $O/r03_variant_parts.err; echo parts rc=$?
timeout -k 10 500 python3 bench.py --config c2 --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 > $O/r03_variant_c2.log 2> $O/r03_variant_c2.err; echo c2 rc=$?
timeout -k 10 600 python3 bench.py --config big --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 > $O/r03_variant_big.log 2> $O/r03_variant_big.err; echo big rc=$?
