"""Union coverage of kernel intervals in a rocprofv3 --kernel-trace CSV: how busy the GPU was, and per-kernel sums."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# restrict to the last `frac` of the trace (the timed steps) if asked
lo = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
rows = [r for r in rows if r[0] >= lo]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(r[1] for r in rows) - rows[0][0]
print(f'span {span/1e6:.1f} ms  busy(any kernel) {busy/1e6:.1f} ms = {busy/span:.1%}')
tot = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    tot[n][0] += e - s
    tot[n][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f'{d/1e6:9.1f} ms {c:7d}  {n}')
