"""Summarise rocprofv3 --pmc counter_collection.csv files (separate passes) into profiles/<round>/pmc_summary.json.
usage: python scripts/pmc_summary.py <out.json> <steps_profiled> <csv> [<csv> ...]"""
import collections
import csv
import json
import sys

out, steps = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sys.argv[3:]:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mpn::', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
res = {}
for k, v in agg.items():
    d = {c: x / steps for c, x in v.items()}
    # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE under-reports wide
    # coalesced reads by 2x on gfx950 -- these kernels read bytes/words, so the raw value is kept and flagged
    if 'WRITE_SIZE' in d or 'FETCH_SIZE' in d:
        d['hbm_bytes_per_step'] = (d.get('WRITE_SIZE', 0) + d.get('FETCH_SIZE', 0)) * 1024
    res[k] = d
json.dump(dict(steps_profiled=steps, note='per bench step (warm-up steps included in the average); FETCH_SIZE uncorrected',
               kernels=res), open(out, 'w'), indent=1, sort_keys=True)
print('wrote', out, len(res), 'kernels')
