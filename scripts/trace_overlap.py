"""Concurrency picture of a rocprofv3 --kernel-trace run of bench.py (run on the GPU box: the trace is too large to bring back).
usage: python scripts/trace_overlap.py <kernel_trace.csv> [out.txt]
Over the middle half of the mapping (between the first and last seed_lookup_kernel), reports: the share of time with a strip
kernel resident, the average number of resident kernels, and which kernels are resident while no strip kernel is."""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mpn::', '')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name))
look = [r for r in rows if 'seed_lookup_kernel' in r[2]]
t0, t1 = min(r[0] for r in look), max(r[1] for r in look)
lo, hi = t0 + (t1 - t0) // 4, t1 - (t1 - t0) // 4
ev = []
for s, e, n in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
live = collections.Counter()
prev = lo
tot = hi - lo
acc = collections.Counter()       # time by predicate
alone = collections.Counter()     # time a kernel is resident while no strip kernel is
conc_hist = collections.Counter()
def is_strip(n): return 'ext_dp_strip_kernel' in n
self_t = collections.Counter()     # time a kernel name is resident at all
self_c = collections.Counter()     # integral of its simultaneous instances
for t, d, n in ev:
    dt = t - prev
    if dt > 0:
        for m, c in live.items():
            if c:
                self_t[m] += dt; self_c[m] += dt * c
        k = sum(live.values())
        strips = sum(c for m, c in live.items() if is_strip(m))
        acc['any'] += dt if k else 0
        acc['strip'] += dt if strips else 0
        acc['conc'] += dt * k
        acc['strip_conc'] += dt * strips
        conc_hist[min(k, 16)] += dt
        if not strips:
            for m, c in live.items():
                if c: alone[m] += dt
    live[n] += d
    if live[n] == 0: del live[n]
    prev = t
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
print(f'window {tot / 1e6:.1f} ms; some kernel resident {acc["any"] / tot:.3f} of the time; a strip kernel resident {acc["strip"] / tot:.3f}; '
      f'average resident kernels {acc["conc"] / tot:.2f}, of which strips {acc["strip_conc"] / tot:.2f}', file=out)
print('resident kernels -> share of time: ' + ', '.join(f'{k}{"+" if k == 16 else ""}: {v / tot:.3f}' for k, v in sorted(conc_hist.items())), file=out)
print('resident while NO strip kernel is (share of the window):', file=out)
for m, v in alone.most_common(14):
    print(f'  {v / tot:.3f}  {m[:70]}', file=out)
print('kernel: share of the window it is resident, simultaneous instances while resident:', file=out)
for m, v in self_t.most_common(16):
    print(f'  {v / tot:.3f}  x{self_c[m] / v:.2f}  {m[:70]}', file=out)
