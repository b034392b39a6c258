"""Slot utilisation of the strip launch lists (debug): reads the dump written with MPN_DUMP_STRIPS=<file> and reports, per
lane-group class, how many of the cell slots the waves issue (64 lanes x S rows x steps) hold a real cell."""
import sys
import numpy as np

rec = np.fromfile(sys.argv[1], dtype=np.int32).reshape(-1, 4)
for glc in range(3):
    r = rec[rec[:, 0] == glc]
    if not len(r):
        continue
    per = 4 >> glc
    gl = 16 << glc
    n = len(r) // per * per
    q, t, s = (r[:n, k].reshape(-1, per).astype(np.int64) for k in (1, 2, 3))
    S = s.max(axis=1)
    lanes = np.where(s > 0, -(-t // np.maximum(s, 1)), 0)
    steps = np.where(q > 0, q + lanes - 1, 0).max(axis=1)
    slots = (64 * S * steps).sum()
    cells = (q * t).sum()
    ramp = (np.where(q > 0, lanes - 1, 0) * lanes * s).sum()          # slots of the groups' own ramps
    rowpad = (q * (gl * S[:, None] - t)).sum()                        # rows the group's lanes do not cover, over its own columns
    imb = ((steps[:, None] - np.where(q > 0, q + lanes - 1, 0)) * gl * S[:, None]).sum()
    print(f'class {glc}: windows {np.count_nonzero(q)}, waves {len(S)}, cells {cells / 1e9:.2f} G, slots {slots / 1e9:.2f} G, util {cells / slots:.3f}; '
          f'lost to ramp {ramp / slots:.3f}, unused rows/lanes {rowpad / slots:.3f}, wave imbalance {imb / slots:.3f}; mean S {S.mean():.1f}, mean steps {steps.mean():.0f}')
    for lo, hi in ((1, 4), (4, 8), (8, 12), (12, 16), (16, 17)):
        m = (S >= lo) & (S < hi)
        if m.any():
            print(f'   S in [{lo},{hi}): waves {m.sum()}, slots share {(64 * S * steps)[m].sum() / slots:.3f}, util {(q * t)[m].sum() / (64 * S * steps)[m].sum():.3f}')
