# run on the GPU box: rocprofv3 kernel stats of a short bench run with extra bench arguments -> gpurun_out/<tag>_kernel_stats.csv
set -e
export TMPDIR=/tmp
T=$1; shift
O=/tmp/prof_stats_$T
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --pcie-steps 0 --no-correctness "$@" > $O/run.log 2>&1
cp $(ls $O/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
grep "^{" $O/run.log | tail -1 | cut -c1-300 > gpurun_out/${T}_bench.log
rm -rf $O
