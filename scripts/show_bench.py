"""Print the interesting fields of a bench.py JSON line (scripts/show_bench.py gpurun_out/x.log)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', round(d['value'], 2), 'ms/step', round(d['ms_per_step'], 1), 'pcie', d.get('pcie_inclusive_gbp_per_min'))
r = d['roofline']
print('dominant', r['kernel'], r['achieved'], 'GB/s frac', r['frac'])
for k, v in r['candidates'].items():
    print('  %-40s' % k, v)
ps = d['per_step']
print({k: v for k, v in ps.items() if k.endswith('_ns')})
print({k: v for k, v in ps.items() if not k.endswith('_ns')})
if 'cpu_baseline' in d:
    print('cpu', round(d['cpu_baseline']['value'], 3), d['cpu_baseline']['cores'])
