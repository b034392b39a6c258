# run on the GPU box: the -m gpu suite, then short benches of c3 and the default (refseq) config
set -e
export TMPDIR=/tmp
T=${1:-quick}
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1
tail -3 gpurun_out/${T}_pytest.log
timeout -k 10 300 python3 bench.py --config c3 --steps 8 --warmup 3 --no-cpu-baseline --no-correctness > gpurun_out/${T}_c3.json 2> gpurun_out/${T}_c3.err
echo c3 done
timeout -k 10 400 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${T}_refseq.json 2> gpurun_out/${T}_refseq.err
echo refseq done
