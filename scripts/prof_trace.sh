# run on the GPU box: kernel trace (no stats) of the default bench; analysis of the timed steps
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_tr
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 > $O/tr.log 2>&1
python scripts/trace_share.py $(ls $O/tr/*/*kernel_trace.csv | head -1) 0.55 > gpurun_out/${TAG:-r2}_trace_share.txt
rm -rf $O/tr
cat gpurun_out/${TAG:-r2}_trace_share.txt
