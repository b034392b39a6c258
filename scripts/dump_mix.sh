set -e
export TMPDIR=/tmp
MPN_DUMP_JOBS=/tmp/jobs.bin MPN_DUMP_STRIPS=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-correctness --alone-reads -1 --resident-steps 0 > gpurun_out/mix_bench.json 2> gpurun_out/mix_bench.err
ls -la /tmp/jobs.bin
python scripts/job_mix.py /tmp/jobs.bin > gpurun_out/mix_refseq.txt
