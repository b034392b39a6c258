# run on the GPU box: the driver's bench command (default config: refseq), its rocprofv3 kernel stats, PMC passes, and the c3 variant
set -e
export TMPDIR=/tmp
T=${1:-r04}; O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5"
if [ "${2:-all}" != "stats-only" ]; then
  timeout -k 10 900 $CMD > gpurun_out/${T}_bench_driver_cmd.log 2> gpurun_out/${T}_bench_driver_cmd.err
  echo bench done
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD --no-cpu-baseline --no-correctness > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_kernel_stats.csv
grep "^{" $O/stats.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.log
rm -rf $O/stats; echo stats done
if [ "${2:-all}" = "stats-only" ]; then exit 0; fi
P="python3 bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-correctness --resident-steps 0 --alone-reads -1"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc1 -- $P > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O/pmc2 -- $P > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -- $P > $O/pmc3.log 2>&1
MPN_PMC_CONFIG=refseq python scripts/pmc_summary.py gpurun_out/${T}_pmc_summary.json 5 $(ls $O/pmc*/*/*counter_collection.csv)
rm -rf $O/pmc1 $O/pmc2 $O/pmc3; echo pmc done
C3="python3 bench.py --config c3 --gpus 1 --steps 20 --warmup 5"
timeout -k 10 600 $C3 > gpurun_out/${T}_variant_c3.json 2> gpurun_out/${T}_variant_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $C3 --no-cpu-baseline --no-correctness > $O/stats_c3.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) gpurun_out/${T}_c3_kernel_stats.csv
rm -rf $O/stats; echo c3 done
