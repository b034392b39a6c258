# run on the GPU box: single-worker kernel trace (kernels alone, no pipeline overlap) of a short bench
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_w1
rm -rf $O; mkdir -p $O
export MPN_PIPE_WORKERS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --pcie-steps 0 --reads-per-step ${READS:-32768} > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${TAG:-w1}_kernel_stats.csv
rm -rf $O/stats
head -40 gpurun_out/${TAG:-w1}_kernel_stats.csv | cut -c1-160
