"""Rank the kernels of a pmc_summary JSON by VALU wave-instructions per step.  usage: python scripts/pmc_rank.py <json> [n]"""
import json
import sys

d = json.load(open(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = sorted(((k.get('SQ_INSTS_VALU', 0), name, k) for name, k in d['kernels'].items()), key=lambda r: -r[0])
tot = sum(r[0] for r in rows if not r[1].startswith('setup:'))
print(f'mapping kernels: {tot / 1e9:.1f} G VALU wave-instructions per step; set-up: {sum(r[0] for r in rows if r[1].startswith("setup:")) / 1e9:.1f} G (per step-equivalent)')
for v, name, k in rows[:n]:
    wc, wa = k.get('SQ_WAVE_CYCLES', 0), k.get('SQ_WAIT_INST_ANY', 0)
    print(f'{v / 1e9:8.2f} G {100 * v / tot:5.1f}%  wait/cycles {wa / wc if wc else 0:4.2f}  {name[:80]}')
