# run on the GPU box: sample clocks / power / utilisation twice a second while a short bench runs -> gpurun_out/<tag>_smi.txt
T=${1:-r03}
( for i in $(seq 1 70); do rocm-smi --showclocks --showpower --showuse --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|busy|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/${T}_smi.txt &
SMI=$!
python3 bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --pcie-steps 0 --no-correctness > gpurun_out/${T}_smi_bench.log 2>&1
wait $SMI
