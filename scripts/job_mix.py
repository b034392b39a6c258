"""Window mix of the extension stage (debug): reads the dump written with MPN_DUMP_JOBS=<file> (with MPN_DUMP_STRIPS set too)."""
import sys
import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.int32).reshape(-1, 6)
cls, q, t, w, flag, ncol = (r[:, k].astype(np.int64) for k in range(6))
lst = cls & 0xff
EXTZ, RIGHT, APPROX = 0x40, 0x02, 0x08   # printed for orientation only: see ext_kernels.h for the flag values
L_BAND = 164   # plan_kernels.h: L_STRIP + 16 * N_STRIP_CLASS
fam = np.where(lst < 5, 0, np.where(lst < 20, 1, np.where(lst < L_BAND, 2, 3)))
names = ['lds', 'wg', 'strip', 'band']
print('windows', len(r))
for f in range(4):
    m = (fam == f) & (cls >= 0)
    if not m.any():
        continue
    nominal = ((q + t - 1) * ncol)[m].sum()
    print(f'{names[f]:6s} n={m.sum():9d} q*t={(q * t)[m].sum() / 1e9:8.2f} G  n_r*n_col={nominal / 1e9:8.2f} G  mean q {q[m].mean():7.1f} t {t[m].mean():7.1f}')
    if f == 3:
        for l in range(L_BAND, L_BAND + 16):
            ml = m & (lst == l)
            if ml.any():
                qq, tt = q[ml], t[ml]
                print(f'   list {l} (variant {(l - L_BAND) // 4}, lds class {(l - L_BAND) % 4}): n={ml.sum():8d}  q*t {(qq * tt).sum() / 1e9:7.2f} G  '
                      f'n_r*n_col {((qq + tt - 1) * ncol[ml]).sum() / 1e9:7.2f} G  mean q {qq.mean():7.1f} t {tt.mean():7.1f}  max q {qq.max()} t {tt.max()}  '
                      f'flags {np.unique(flag[ml] & 0xff)[:8]}')

# the band lists by window kind: gap fill (approximate maximum), right extension, left extension (right-aligned gaps), exact fill
kind = np.where(flag & 0x02, 0, np.where(flag & 0x40, np.where(flag & 0x08, 2, 1), 3))
knames = ['gap fill (approx)', 'right extension', 'left extension', 'exact fill']
mb = (fam == 3) & (cls >= 0)
for k in range(4):
    m = mb & (kind == k)
    if m.any():
        print(f'band windows, {knames[k]:18s}: n={m.sum():8d}  n_r*n_col {((q + t - 1) * ncol)[m].sum() / 1e9:7.2f} G  mean q {q[m].mean():7.1f} t {t[m].mean():7.1f}  '
              f'max q {q[m].max()} t {t[m].max()}  t>1024: {(t[m] > 1024).sum()}  w<max(q,t): {(w[m] < np.maximum(q[m], t[m])).sum()}')
