# run on the GPU box: kernel trace of a short bench run, reduced by scripts/trace_overlap.py -> gpurun_out/<tag>_overlap.txt
set -e
export TMPDIR=/tmp
T=${1:-r03}; shift || true; O=/tmp/prof_trace_$T
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 --no-correctness "$@" > $O/run.log 2>&1
python scripts/trace_overlap.py $(ls $O/*/*kernel_trace.csv | head -1) gpurun_out/${T}_overlap.txt
rm -rf $O
