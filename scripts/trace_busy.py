"""GPU busy fraction and per-kernel sums over the mapping steps of a rocprofv3 --kernel-trace CSV.

The region analysed starts at the first sketch kernel (the first mapping step; index build is before it) and ends at
the last kernel.  'busy' is the union of all kernel intervals; 'conc' = sum of kernel durations / busy time.
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:48]))
rows.sort()
t_first = min(s for s, e, n in rows if 'sketch_chunk_kernel' in n)
rows = [r for r in rows if r[0] >= t_first]
n_steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
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
ksum = sum(e - s for s, e, _ in rows)
print(f'steps {n_steps:g}: span {span/1e6/n_steps:.1f} ms/step  busy(any kernel) {busy/1e6/n_steps:.1f} ms/step = {busy/span:.1%}  '
      f'kernel-sum {ksum/1e6/n_steps:.1f} ms/step  conc {ksum/busy:.2f}')
tot = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    tot[n][0] += e - s
    tot[n][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f'{d/1e6/n_steps:9.1f} ms/step {c:7d} calls  avg {d/1e3/c:9.1f} us  {n}')
