# run on the GPU box: the strain-rich variant's worker/slot report, its rocprofv3 kernel stats, and the single-worker kernel stats
export TMPDIR=/tmp
T=${1:-r04_strain}; O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
B="python3 bench.py --config strain --reads-per-step 65536 --no-cpu-baseline --pcie-steps 0 --no-correctness"
MPN_DEBUG_WORKERS=1 $B --steps 2 --warmup 1 > $O/workers.json 2> $O/workers.log
grep -E "^\[slot|^\[call|out of device" $O/workers.log | tail -30 > gpurun_out/${T}_workers.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 3 --warmup 1 > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
grep "^{" $O/stats.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.log
rm -rf $O/stats
MPN_PIPE_WORKERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 1 --warmup 1 > $O/w1.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_w1_kernel_stats.csv
grep "^{" $O/w1.log | tail -1 > gpurun_out/${T}_w1_bench.log
rm -rf $O/stats
echo done
