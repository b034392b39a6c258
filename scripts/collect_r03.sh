# run on the GPU box: the driver's bench command, its rocprofv3 kernel stats, PMC passes, and the single-worker variants
set -e
export TMPDIR=/tmp
T=${1:-r03}; O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5"
timeout -k 10 600 $CMD > gpurun_out/${T}_bench_driver_cmd.log 2> gpurun_out/${T}_bench_driver_cmd.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD --no-cpu-baseline > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
grep "^{" $O/stats.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.log
rm -rf $O/stats; echo stats done
P="python3 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --pcie-steps 0"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc1 -- $P > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O/pmc2 -- $P > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -- $P > $O/pmc3.log 2>&1
python scripts/pmc_summary.py gpurun_out/${T}_pmc_summary.json 6 $(ls $O/pmc*/*/*counter_collection.csv)
rm -rf $O/pmc1 $O/pmc2 $O/pmc3; echo pmc done
export MPN_PIPE_WORKERS=1
W="python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --pcie-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $W > $O/w1.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_w1_kernel_stats.csv
grep "^{" $O/w1.log | tail -1 > gpurun_out/${T}_w1_bench.log
rm -rf $O/stats
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc1 -- $W > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc2 -- $W > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -- $W > $O/pmc3.log 2>&1
python scripts/pmc_summary.py gpurun_out/${T}_w1_pmc_summary.json 3 $(ls $O/pmc*/*/*counter_collection.csv)
rm -rf $O/pmc1 $O/pmc2 $O/pmc3; echo w1 done
