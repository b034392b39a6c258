# run on the GPU box: VALU / wave-cycle counters and GPU-active cycles of the kernels of a short refseq run (two PMC passes)
set -e
export TMPDIR=/tmp
T=${1:-pmcs}; O=gpurun_out/prof_$T
rm -rf $O; mkdir -p $O
P="python3 bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-correctness --resident-steps 0 --alone-reads -1"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/pmc1 -- $P > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O/pmc2 -- $P > $O/pmc2.log 2>&1
MPN_PMC_CONFIG=refseq python scripts/pmc_summary.py gpurun_out/${T}_pmc_summary.json 5 $(ls $O/pmc*/*/*counter_collection.csv)
rm -rf $O/pmc1 $O/pmc2; echo pmc done
