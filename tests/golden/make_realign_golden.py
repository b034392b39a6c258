"""Generates tests/golden/realign_golden.json from the reference's own amplicon realigner: realigner.cpp + ssw_cpp.cpp + ssw.c
compiled in place by oracle/Makefile (oracle/_ref/librealigner.so), called through its C entry points `realign_reads` /
`free_memory` exactly as /root/reference/bin/realignment/realign_illumina_reads.py:596-629 does, on the seeded windows of
tests/realign_cases.py.  Only inputs and outputs are stored.

    make -C oracle && python tests/golden/make_realign_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle.realign_bindings import have_ref, ref_realign  # noqa: E402
from realign_cases import make_cases  # noqa: E402

assert have_ref(), 'oracle/_ref/librealigner.so is missing: run make -C oracle in a container that has /root/reference'
out = []
for c in make_cases():
    res = ref_realign(**c)
    out.append(dict(c, expected=[[p, cg] for p, cg in res]))
with open(os.path.join(HERE, 'realign_golden.json'), 'w') as f:
    json.dump(out, f, separators=(',', ':'))
print('wrote', len(out), 'windows,', sum(len(c['seqs']) for c in out), 'reads')
