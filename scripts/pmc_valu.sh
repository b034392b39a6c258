# run on the GPU box: one PMC pass (VALU instructions, waves, cycles) of a short bench run -> gpurun_out/<tag>_valu.json
set -e
export TMPDIR=/tmp
T=${1:-valu}; O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/pmc1 -- python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 > $O/pmc1.log 2>&1
python scripts/pmc_summary.py gpurun_out/${T}.json 8 $(ls $O/pmc1/*/*counter_collection.csv)
rm -rf $O/pmc1
