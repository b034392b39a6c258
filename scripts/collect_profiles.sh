# run on the GPU box: kernel stats + three PMC passes of the same bench command, and a one-worker run (kernels alone)
set -e
export TMPDIR=/tmp
CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --reads-per-step 32768"
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc1 -- $CMD > $O/pmc1.log 2>&1
echo pmc1 done
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O/pmc2 -- $CMD > $O/pmc2.log 2>&1
echo pmc2 done
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -- $CMD > $O/pmc3.log 2>&1
echo pmc3 done
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_r01_final.csv
python scripts/pmc_summary.py gpurun_out/pmc_summary_r01_final.json 2 $(ls $O/pmc*/*/*counter_collection.csv)
MPN_PIPE_WORKERS=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --reads-per-step 32768 > gpurun_out/bench_w1.log 2>&1
tail -1 gpurun_out/bench_w1.log | cut -c1-200
rm -rf $O/stats $O/pmc1 $O/pmc2 $O/pmc3
