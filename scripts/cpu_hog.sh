# which threads burn host CPU during the timed steps?
python bench.py --no-cpu-baseline --warmup 2 --pcie-steps 0 --steps 40 > gpurun_out/hog.log 2> gpurun_out/hog.err &
PID=$!
sleep 22
python - $PID <<'PY'
import os, sys, time
pid = int(sys.argv[1])
def snap():
    out = {}
    for tid in os.listdir(f'/proc/{pid}/task'):
        try:
            f = open(f'/proc/{pid}/task/{tid}/stat').read()
            comm = f[f.index('(') + 1:f.rindex(')')]
            rest = f[f.rindex(')') + 2:].split()
            out[tid] = (comm, int(rest[11]) + int(rest[12]))
        except Exception:
            pass
    return out
a = snap(); time.sleep(4); b = snap()
hz = os.sysconf('SC_CLK_TCK')
rows = sorted(((b[t][1] - a.get(t, (0, 0))[1]) / hz / 4, b[t][0], t) for t in b)
print('threads', len(b), 'total cores busy %.1f' % sum(r[0] for r in rows))
for r in rows[-14:]:
    print('%5.2f cores  %-20s tid %s' % r)
PY
which gdb perf 2>/dev/null
wait $PID
tail -1 gpurun_out/hog.err
