"""Who occupies the GPU?  From a rocprofv3 --kernel-trace CSV: over the window of the last `frac` of the trace (the timed
steps), the fraction of time any kernel runs, the mean number of kernels in flight, and per kernel its duration sum and
its SHARE = integral of 1/(kernels in flight) while it runs (shares add up to the busy time).
usage: trace_share.py <kernel_trace.csv> [window_start_fraction=0.5]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mpn::', '')[:44]))
t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
w0 = t0 + (t1 - t0) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
rows = [r for r in rows if r[1] > w0]
ev = []
for i, (s, e, n) in enumerate(rows):
    ev.append((max(s, w0), 1, i))
    ev.append((e, -1, i))
ev.sort()
active = set()
share = defaultdict(float)
dur = defaultdict(float)
calls = defaultdict(int)
busy = 0.0
conc_t = defaultdict(float)
prev = ev[0][0]
for t, d, i in ev:
    dt = t - prev
    if dt > 0:
        k = len(active)
        conc_t[min(k, 12)] += dt
        if k:
            busy += dt
            for j in active:
                share[rows[j][2]] += dt / k
    prev = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
for s, e, n in rows:
    dur[n] += e - max(s, w0)
    calls[n] += 1
span = t1 - w0
print(f'window {span/1e6:.1f} ms  busy {busy/span:.1%}  mean kernels in flight while busy {sum(dur.values())/busy:.2f}')
print('time with k kernels in flight:', {k: f'{v/span:.1%}' for k, v in sorted(conc_t.items())})
for n, v in sorted(share.items(), key=lambda kv: -kv[1])[:26]:
    print(f'{v/span:7.1%} share  {dur[n]/1e6:9.1f} ms dur-sum {calls[n]:6d} calls  avg {dur[n]/1e3/calls[n]:9.1f} us  {n}')
