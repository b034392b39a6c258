# Is the strip kernel slowed by the few-wave kernels that share the GPU with it?  The bench's single-worker pass (alone_ms) with
# and without AMD_SERIALIZE_KERNEL=3, and the engine clock sampled during the runs.
set -e
export TMPDIR=/tmp
B="python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-correctness --resident-steps 0"
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average" ; sleep 1; done ) > gpurun_out/probe_clk.log &
CLK=$!
timeout -k 10 500 $B > gpurun_out/probe_normal.json 2> gpurun_out/probe_normal.err
echo normal done
AMD_SERIALIZE_KERNEL=3 timeout -k 10 500 $B > gpurun_out/probe_serial.json 2> gpurun_out/probe_serial.err
echo serial done
kill $CLK
