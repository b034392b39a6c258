# run on the GPU box: PMC passes (separate runs, --kernel-trace only) of a short bench; summary -> gpurun_out/${TAG}_pmc.json
set -e
export TMPDIR=/tmp
O=gpurun_out/prof_pmc
rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --pcie-steps 0 --reads-per-step ${READS:-32768}"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d $O/pmc1 -- $CMD > $O/pmc1.log 2>&1
echo pmc1 done
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O/pmc2 -- $CMD > $O/pmc2.log 2>&1
echo pmc2 done
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc3 -- $CMD > $O/pmc3.log 2>&1
echo pmc3 done
python scripts/pmc_summary.py gpurun_out/${TAG:-r2}_pmc.json 2 $(ls $O/pmc*/*/*counter_collection.csv)
rm -rf $O/pmc1 $O/pmc2 $O/pmc3
