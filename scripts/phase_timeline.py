"""Timeline of the pipeline workers from MPN_DEBUG_PHASES=1 output (last call in the log).
Phases by stats slot: 17 seed+chain (GPU), 18 chain download, 19 hits (host), 20 plan (host), 27 ext prep (host), 28 enqueue,
29 ext GPU wait, 30 ext finish, 22 stitch (host), 23 final (host).  (21 = whole extension stage: skipped, it contains 27-30.)"""
import sys
from collections import defaultdict
calls, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith('[phase] '):
        w, s, a, b = map(int, line.split()[1:])
        cur.append((w, s, a, b))
    elif line.startswith('[phase-end]'):
        calls.append(cur); cur = []
recs = [r for r in calls[-1] if r[1] != 21 and r[0] >= 0]
GPU = {17, 29}
names = {17: 'seed+chain', 18: 'd2h', 19: 'hits', 20: 'plan', 27: 'prep', 28: 'enq', 29: 'ext-wait', 30: 'ext-fin', 22: 'stitch', 23: 'final'}
T = max(r[3] for r in recs)
ev = []
for w, s, a, b in recs:
    ev.append((a, 1, s in GPU)); ev.append((b, -1, s in GPU))
ev.sort()
ng = nh = 0; prev = 0; hist = defaultdict(float)
for t, d, g in ev:
    hist[ng] += t - prev; prev = t
    if g: ng += d
print('call length %.1f ms' % (T / 1e6))
print('time with k workers in a GPU phase:', {k: '%.1f%%' % (100 * v / T) for k, v in sorted(hist.items())})
tot = defaultdict(float)
for w, s, a, b in recs: tot[s] += b - a
print({names.get(s, s): '%.0f ms' % (v / 1e6) for s, v in sorted(tot.items(), key=lambda kv: -kv[1])})
# per worker sequence of the first two sub-batches
for w in range(2):
    seq = sorted([r for r in recs if r[0] == w], key=lambda r: r[2])[:24]
    print('worker', w, ' '.join('%s:%.0f-%.0f' % (names.get(s, s), a / 1e6, b / 1e6) for _, s, a, b in seq))
