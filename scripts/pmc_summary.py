"""Summarise rocprofv3 --pmc counter_collection.csv files (separate passes) into profiles/<round>/pmc_summary.json.
usage: python scripts/pmc_summary.py <out.json> <steps_profiled> <csv> [<csv> ...]"""
import collections
import csv
import json
import os
import sys

out, steps = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sys.argv[3:]:
    rows = list(csv.DictReader(open(f)))
    # everything dispatched before the first seed lookup is set-up (synthetic genomes and reads, index build): kept apart
    # under "setup:<kernel>" and not divided into the per-step figures of the mapping kernels it shares code with (sketch)
    first_map = min((int(r['Dispatch_Id']) for r in rows if 'seed_lookup_kernel' in r['Kernel_Name']), default=0)
    for r in rows:
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mpn::', '')
        if int(r['Dispatch_Id']) < first_map:
            k = 'setup:' + k
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
res = {}
for k, v in agg.items():
    d = {c: x / steps for c, x in v.items()}
    # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE under-reports wide
    # coalesced reads by 2x on gfx950 -- these kernels read bytes/words, so the raw value is kept and flagged
    if 'WRITE_SIZE' in d or 'FETCH_SIZE' in d:
        d['hbm_bytes_per_step'] = (d.get('WRITE_SIZE', 0) + d.get('FETCH_SIZE', 0)) * 1024
    res[k] = d
json.dump(dict(steps_profiled=steps, config=os.environ.get('MPN_PMC_CONFIG', 'refseq'), note='per bench step (warm-up steps included in the average; set-up dispatches -- before the first seed lookup -- under setup:<kernel>, also divided by the steps); FETCH_SIZE uncorrected',
               kernels=res), open(out, 'w'), indent=1, sort_keys=True)
print('wrote', out, len(res), 'kernels')
