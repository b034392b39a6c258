"""f4 measurement: windows per second of the batched amplicon realigner on the GPU, beside the reference's own compiled
realigner (oracle/_ref/librealigner.so, one CPU thread, one window per call as realign_illumina_reads.py drives it).
    python scripts/bench_realign.py [n_windows] [reads_per_window] [haplotypes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from megapath_nano_amd import realigner  # noqa: E402
from realign_cases import make_window  # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 400
nh = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wins = [make_window(300 + k, n_reads=nr, n_haps=nh, prefix=400, center=200, suffix=400, read_len=250) for k in range(nw)]
realigner.realign_batch(wins[:2])  # warm-up (HIP init, code objects)
t = time.perf_counter()
got = realigner.realign_batch(wins)
dt = time.perf_counter() - t
print('GPU batched: %d windows x %d reads x %d haplotypes in %.3f s = %.1f windows/s, %.0f reads/s' % (nw, nr, nh, dt, nw / dt, nw * nr / dt))
try:
    from oracle.realign_bindings import have_ref, ref_realign
    if have_ref():
        k = min(nw, 8)
        t = time.perf_counter()
        want = [ref_realign(**w) for w in wins[:k]]
        dc = time.perf_counter() - t
        print('reference (compiled in place, 1 thread): %d windows in %.3f s = %.1f windows/s; identical: %s' % (k, dc, k / dc, want == got[:k]))
except ImportError:
    pass
