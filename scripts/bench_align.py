"""End-to-end rate of the drop-in `Align()` (FASTQ in -> PAF + SAM + sorted BAM/BAI + DataFrame; the reference's
bin/lib/aligner.py:239-335) beside the mapping-only rate of the same reads, on a C1-shaped input (1 000 reads x 8 kb against
5 genomes) and on a 100 000-read input.  Files are written to a scratch directory first (not timed).

  python scripts/bench_align.py [--reads 1000,100000] [--genomes 5] [--genome-len 5000000] [--profile]
"""
import argparse
import cProfile
import gzip
import io
import json
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_inputs(d, n_genomes, genome_len, n_reads, mean_len, seed):
    import torch
    from megapath_nano_amd import synth
    dev = torch.device('cuda', 0)
    names, flat, lens = synth.make_genomes_device(seed, n_genomes, genome_len, 1, dev)
    w = np.ones(n_genomes)
    buf, off, ln = synth.make_reads_device(seed + 1, flat, genome_len, n_reads, w, dev, mean_len=mean_len)
    torch.cuda.synchronize()
    hb, ho, hl = buf.cpu().numpy(), off.cpu().numpy(), ln.cpu().numpy()
    view = flat.view(n_genomes, genome_len).cpu().numpy()
    paths = []
    for g in range(n_genomes):
        p = os.path.join(d, f'asm{g}.fna.gz')
        with gzip.open(p, 'wb', compresslevel=1) as f:
            f.write(b'>' + names[g].encode() + b' synthetic\n')
            f.write(view[g].tobytes())
            f.write(b'\n')
        paths.append(p)
    fq = os.path.join(d, f'reads_{n_reads}.fq')
    rng = np.random.default_rng(seed)
    with open(fq, 'wb') as f:
        out = io.BytesIO()
        for i in range(n_reads):
            s = hb[ho[i]:ho[i] + hl[i]].tobytes()
            q = (rng.integers(5, 30, size=hl[i], dtype=np.uint8) + 33).tobytes()
            out.write(b'@read%07d\n' % i + s + b'\n+\n' + q + b'\n')
            if out.tell() > (64 << 20):
                f.write(out.getvalue())
                out = io.BytesIO()
        f.write(out.getvalue())
    del flat, buf
    torch.cuda.empty_cache()
    return paths, fq, int(hl.sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', default='1000,100000')
    ap.add_argument('--genomes', type=int, default=5)
    ap.add_argument('--genome-len', type=int, default=5000000)
    ap.add_argument('--mean-len', type=int, default=8000)
    ap.add_argument('--profile', action='store_true')
    args = ap.parse_args()
    import pandas as pd
    from megapath_nano_amd import aligner, build
    build.build()
    opts = ['-t', '16', '-I', '8G', '-N', '50', '-p', '1', '-x', 'map-ont', '--split-prefix', 'tmp']   # megapath_nano.py:1270
    go = {'min_alignment_score': 0}
    for n_reads in (int(x) for x in args.reads.split(',')):
        with tempfile.TemporaryDirectory(dir=os.environ.get('TMPDIR', '/tmp')) as d:
            paths, fq, bases = write_inputs(d, args.genomes, args.genome_len, n_reads, args.mean_len, 77)
            tf, qf = pd.DataFrame({'path': paths}), pd.DataFrame({'path': [fq]})

            def run(prefix, mapping_only=False):
                return aligner.Align(assembly_metadata=None, global_options=go, temp_dir_name=d, log_file=sys.stderr, query_filename_list=qf,
                                     target_filename_list=tf, aligner_options=opts, paf_path_and_prefix=prefix, mapping_only=mapping_only)
            run(None)   # warm-up: library, index cache, scratch pools
            t0 = time.perf_counter()
            table = run(None)
            t_cols = time.perf_counter() - t0
            prof = cProfile.Profile() if args.profile else None
            t0 = time.perf_counter()
            if prof:
                prof.enable()
            table2 = run(os.path.join(d, 'out'))
            if prof:
                prof.disable()
            t_full = time.perf_counter() - t0
            sizes = {e: os.path.getsize(os.path.join(d, 'out.' + e)) for e in ('paf', 'sam', 'bam', 'bam.bai')}
            print(json.dumps({'reads': n_reads, 'read_bases': bases, 'rows': int(table.shape[0]), 'rows_with_files': int(table2.shape[0]),
                              'align_table_only_s': round(t_cols, 3), 'align_table_only_gbp_per_min': round(bases / t_cols * 60 / 1e9, 2),
                              'align_paf_sam_bam_s': round(t_full, 3), 'align_paf_sam_bam_gbp_per_min': round(bases / t_full * 60 / 1e9, 2),
                              'output_bytes': sizes}), flush=True)
            if prof:
                s = io.StringIO()
                pstats.Stats(prof, stream=s).sort_stats('cumulative').print_stats(22)
                print(s.getvalue(), file=sys.stderr)


if __name__ == '__main__':
    main()
