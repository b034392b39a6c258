"""Regenerates tests/golden/nanofastq_golden.json: stdin -> (stdout, stderr) of the REFERENCE's own prebuilt read filter,
/root/reference/bin/tools/nanofastq (an x86-64 ELF shipped in the reference tree; its source nanofastq.c does not build
here because it needs seqtk's kseq.h).  Run in the build container only: /root/reference does not exist on the GPU box.
The fixture is data (inputs and expected outputs), not source."""
import json
import os
import subprocess

import numpy as np

REF = '/root/reference/bin/tools/nanofastq'
HERE = os.path.dirname(os.path.abspath(__file__))


def fastq(rng, n, lo, hi, qlo, qhi, comments=True, multiline=False, fasta_every=0):
    out = []
    for i in range(n):
        l = int(rng.integers(lo, hi + 1))
        seq = ''.join('ACGTN'[int(x)] for x in rng.choice(5, size=l, p=[.245, .245, .245, .245, .02]))
        head = f'read{i}' + (f' runid={rng.integers(1 << 30):x} ch={i % 512}' if comments and i % 3 else '')
        if fasta_every and i % fasta_every == 0:
            out.append(f'>{head}\n{seq}\n')
            continue
        mean_q = rng.uniform(qlo, qhi)
        q = np.clip(np.rint(rng.normal(mean_q, 6.0, size=l)), 0, 60).astype(int)
        if i % 7 == 0:
            q[: l // 4] = rng.integers(0, 6, size=l // 4)        # a bad head: cropping changes the verdict
        qual = ''.join(chr(33 + int(x)) for x in q)
        if multiline and l > 150:
            seq = '\n'.join(seq[k:k + 70] for k in range(0, l, 70))
            qual = '\n'.join(qual[k:k + 70] for k in range(0, l, 70))
        out.append(f'@{head}\n{seq}\n+\n{qual}\n')
    return ''.join(out)


def main():
    rng = np.random.default_rng(20240907)
    cases = []
    inputs = {
        'mixed_small': fastq(rng, 40, 1, 250, 3, 25, fasta_every=9),
        'ont_like': fastq(rng, 16, 200, 2500, 6, 16),
        'multiline': fastq(rng, 8, 100, 600, 5, 20, multiline=True),
        'perfect_and_awful': '@p\nACGTACGT\n+\n~~~~~~~~\n@z\nACGTACGT\n+\n!!!!!!!!\n@one\nA\n+\n5\n>f only\nACGT\n',
    }
    argsets = [[], ['-q', '7'], ['-l', '500', '-q', '10'], ['-h', '50', '-t', '30', '-l', '100', '-q', '9'], ['-h', '400', '-t', '400'],
               ['-q', '12', '-r', 'sample.fq_'], ['-l', '0', '-q', '0', '-h', '0', '-t', '1']]
    for name, text in inputs.items():
        for args in argsets:
            p = subprocess.run([REF] + args, input=text.encode(), capture_output=True, timeout=60)
            assert p.returncode == 0, (name, args, p.stderr[-200:])
            cases.append(dict(input=name, args=args, stdout=p.stdout.decode('latin-1'), stderr=p.stderr.decode('latin-1')))
    json.dump(dict(generator='tests/golden/make_nanofastq_golden.py', reference_binary='bin/tools/nanofastq (prebuilt, reference tree)',
                   inputs=inputs, cases=cases), open(os.path.join(HERE, 'nanofastq_golden.json'), 'w'))
    print(len(cases), 'cases')


if __name__ == '__main__':
    main()
