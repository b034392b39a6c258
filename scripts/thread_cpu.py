"""debug: CPU seconds of this process's threads by name (Linux /proc), e.g. around bench.py's timed steps"""
import os


DETAIL = {}   # tid -> (user s, system s, (voluntary, involuntary) context switches) at the last snapshot


def ctx_switches(tid):
    v = n = 0
    try:
        for line in open(f'/proc/self/task/{tid}/status'):
            if line.startswith('voluntary_ctxt_switches'):
                v = int(line.split()[1])
            elif line.startswith('nonvoluntary_ctxt_switches'):
                n = int(line.split()[1])
    except OSError:
        pass
    return v, n


def snapshot():
    out = {}
    tck = os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            s = open(f'/proc/self/task/{tid}/stat').read()
        except OSError:
            continue
        name = s[s.index('(') + 1:s.rindex(')')]
        f = s[s.rindex(')') + 2:].split()
        out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tck)
        DETAIL[int(tid)] = (int(f[11]) / tck, int(f[12]) / tck, ctx_switches(tid))
    return out


def diff(a, b):
    agg = {}
    for tid, (name, t) in b.items():
        d = t - a.get(tid, (name, 0.0))[1]
        n, tot = agg.get(name, (0, 0.0))
        agg[name] = (n + 1, tot + d)
    return sorted(agg.items(), key=lambda kv: -kv[1][1])


def top_threads(a, b, k=8):
    """the k busiest single threads: (tid, name, CPU seconds, is_main)"""
    rows = [(tid, name, t - a.get(tid, (name, 0.0))[1], tid == os.getpid()) for tid, (name, t) in b.items()]
    return sorted(rows, key=lambda r: -r[2])[:k]
