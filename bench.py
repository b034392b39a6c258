#!/usr/bin/env python
"""bench.py -- Gbp/min of ONT reads aligned + species-assigned on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself, one per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (the driver's form for N > 1)

A "step" is one pass of the hot path over one batch of synthetic reads that is already resident in HBM:
seed-chain-extend against the resident index (HIP kernels), host hit bookkeeping, read reassignment (HIP kernels),
best hit per read and the per-species / per-name counters, followed for N > 1 by the RCCL all-reduce of those counters.
Workload (--config, default refseq) = BASELINE.json configs[2] at the scale and composition north_star names: a 10-species
community sampled against a STRAIN-RICH target set (100 assemblies of every community species at 97-99.9 % identity) of
18000 genomes = 72 Gbp, held as RESIDENT index parts built one after another (minimap2 -I; 288 GB of HBM keep all of them,
where a CPU host streams them), every read mapped against every part and the hits merged per read like --split-prefix
(mpn_hits); every rank holds all parts and maps its own reads (weak scaling, no data-path collective).  --config c3 is the
round-1..3 headline (20 Gbp of random genomes + 10 strain copies in one index); strain / big / parts / c2 are the other
variants; each names itself in config.workload.  Genomes and reads are generated ON THE GPU (torch); a few distinct read batches are generated
once and rotated over the steps (the mapper keeps no state between calls, so a repeated batch costs what a fresh one does).

Rank 0 prints ONE JSON line.  `value` = read bases / wall time of the K timed steps fed from HOST buffers (the read H2D is
inside the timed call: SURVEY 8d's wall time includes it); `resident_input_gbp_per_min` = a few more steps with the reads
already in HBM, as a side figure.  `roofline` is for the kernel with the largest EXCLUSIVE device time: after the timed region
a slice of a batch runs through ONE pipeline worker (nothing else on the GPU), which gives every candidate's launch duration
alone (`alone_ms`) beside its HIP-event span inside the 12-worker pipeline (`in_pipeline_ms`); fractions are given on the HBM
axis (8 TB/s) and on the VALU axis (1.229e12 wave-instructions/s = 1024 SIMDs x 2.4 GHz / 2 cycles, the guide's figure; the
instruction-mix ceiling is a side note).  `cpu_baseline` times the CPU oracle (oracle/mm2_oracle.c, a port)
on a bounded sample of the same reads on this host's cores against an index of >= 1 Gbp; `correctness` checks the output of
THIS run at the full index size (outside the timed region): truth hit rate from the generator's read origins, the
independent PAF checker (tests/paf_check.py) over target slices fetched from the index in HBM, per-name counts against the
sampled composition.  Progress goes to stderr.
"""
import argparse
import json
import os
import random
import subprocess
import sys
import time

import numpy as np

os.environ.setdefault('GPU_MAX_HW_QUEUES', '20')  # before the HIP runtime starts: see megapath_nano_amd/__init__.py

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
# Issue rates of the strip cell's instruction types on gfx950 (profiles/r03/valu_microbench2.txt, 4 waves per SIMD): two-operand 32-bit
# add / sub / and / or / mov and the 16-bit two-operand max take ~2.7 cycles per wave-instruction per SIMD, packed 16-bit, three-operand,
# permute, DPP ones ~4.4.  The S = 16 loop body of the strip kernel (two windows per lane group since round 4) is 599 VALU instructions
# per step of 16 rows x 2 windows, ~60 of them per-step overhead: 16.8 per cell and lane (21.5 with one window per group), of which 7
# per cell are of the fast kind: ~3.7 cycles per instruction on average -> the ceiling of THIS mix for the chip.
STRIP_INSTR_PER_CELL = 16.8
STRIP_FAST_PER_CELL = 7.0
STRIP_CYCLES_PER_INSTR = (STRIP_FAST_PER_CELL * 2.7 + (STRIP_INSTR_PER_CELL - STRIP_FAST_PER_CELL) * 4.4) / STRIP_INSTR_PER_CELL
VALU_CEIL_WAVE_INSTR = 1024 * 2.4e9 / STRIP_CYCLES_PER_INSTR   # 1024 SIMDs at 2.4 GHz
T00 = time.time()


def pmc_traffic(kernel, launches_per_step, config='refseq'):
    """HBM bytes per launch of `kernel` from the PMC summary committed under profiles/ (collected by scripts/collect_r03.sh on this
    workload in separate --pmc passes, as MI355X_MICROARCH.md prescribes); None when the file is absent."""
    ks, rel = None, None
    for rel in ('profiles/r04/r04_pmc_summary.json', 'profiles/r03/r03_pmc_summary.json'):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                doc = json.load(f)
            if doc.get('config', 'c3') != config:   # (a PMC summary belongs to the workload it was collected on; r03's was c3)
                continue
            ks = doc['kernels']
            break
        except (OSError, ValueError, KeyError):
            continue
    if ks is None:
        return None, None
    stem = kernel.split('<')[0].split(' ')[0]
    want_exact = '<true>' in kernel
    rows = [v for k, v in ks.items() if k.split('<')[0] == stem and not k.startswith('setup:')
            and not (stem == 'ext_dp_strip_kernel' and ('true' in k) != want_exact)]
    b = sum(v.get('hbm_bytes_per_step', 0) for v in rows)
    return (int(b / max(launches_per_step, 1)) if b else None), rel + ' (bytes per step / launches per step)'


def log(msg):
    print(f'[bench {time.time() - T00:7.1f}s] {msg}', file=sys.stderr, flush=True)


def cpu_quota():
    """CPUs this container may use (cgroup quota), or None when unlimited: the library sizes its host pool by it."""
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        return None if q == 'max' else round(int(q) / int(p), 2)
    except (OSError, ValueError):
        return None


def launch_ranks(n):
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU) BEFORE this process touches the GPU,
    wait for them, exit with the worst return code.  Rank 0 prints the JSON line to the inherited stdout."""
    import socket
    import torch
    single = os.environ.get('MPN_SINGLE_DEVICE') == '1'
    ndev = torch.cuda.device_count()  # does not initialise HIP
    if ndev < n and not single:
        sys.exit(f'bench.py: --gpus {n} but only {ndev} GPU(s) are visible (MPN_SINGLE_DEVICE=1 + MPN_DIST_BACKEND=gloo '
                 f'rehearses the N-rank path on one GPU)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


# ---- workloads -------------------------------------------------------------------------------------------------------
def community(n, strain_pairs):
    """10 members with log-normal abundance, including a close-relative pair (genome 0 and its 99 % copy)."""
    rng = np.random.default_rng(7)
    base = n - strain_pairs
    members = list(range(min(9, base))) + ([base] if strain_pairs > 0 else [])
    weights = np.zeros(n)
    weights[members] = rng.lognormal(0.0, 1.0, size=len(members))
    return members, weights


def make_batch(flat, weights, args, seed, device):
    """One batch of reads generated on the GPU -> PackedReads (device tensors + the host copy the C-ABI also takes) with the
    generator's read origins attached (`truth`)."""
    import torch
    from megapath_nano_amd import synth, mapper
    buf, off, lens, truth = synth.make_reads_device(seed, flat, args.genome_len, args.reads_per_step, weights, device, mean_len=args.mean_len,
                                                    return_truth=True)
    torch.cuda.synchronize()
    names = [f'read{r:07d}' for r in range(args.reads_per_step)]
    b = mapper.PackedReads.from_arrays(names, buf.cpu().numpy(), off.cpu().numpy(), lens.cpu().numpy(), dev=(buf, off, lens))
    b.truth = truth
    return b


def build_workload(args, device, rank, world):
    """Targets and read batches of --config, generated in HBM and indexed from HBM.  -> dict (see main)."""
    import torch
    from megapath_nano_amd import mapper, synth
    from megapath_nano_amd.pipeline import Taxonomy
    cfg = args.config
    W = dict(kind=None, cpu_genomes=None)
    n_distinct = max(1, min(args.distinct_batches, args.warmup + args.steps))
    if cfg == 'c2':
        # configs[1]: 24 chromosomes x 129 Mbp = 3.1 Gbp, 45 % interspersed repeats, + 8 plasmid-like decoys; the reads: 70 % human-like,
        # 5 % decoy, 25 % from 5 microbial genomes that are NOT in the index (they must come out as microbe reads)
        n_chrom, chrom_len = (24, 129_000_000) if args.genomes == 0 else (args.genomes, args.genome_len)
        names, flat, lens, kind = synth.make_humanlike_device(20240901, n_chrom, chrom_len, device)
        mic_names, mic, mic_lens = synth.make_genomes_device(4242, 5, 4_000_000, 0, device)
        torch.cuda.synchronize()
        t0 = time.time()
        idx = mapper.Index.from_device(names, flat.data_ptr(), lens)
        W['index_s'] = time.time() - t0
        opt_kw = dict(best_n=5, pri_ratio=0.8)   # megapath_nano.py:1124: -x map-ont defaults
        all_flat = torch.cat([flat, mic])
        all_lens = np.concatenate([lens, mic_lens])
        w = np.concatenate([0.70 * lens[kind == 0] / lens[kind == 0].sum(), 0.05 * lens[kind == 1] / lens[kind == 1].sum(), np.full(5, 0.05)])
        batches = []
        for s_ in range(n_distinct):
            buf, off, ln, truth = synth.make_reads_from_targets_device(1000 * (rank + 1) + s_, all_flat, all_lens, args.reads_per_step, w, device,
                                                                       mean_len=args.mean_len)
            torch.cuda.synchronize()
            b = mapper.PackedReads.from_arrays([f'read{r:07d}' for r in range(args.reads_per_step)], buf.cpu().numpy(), off.cpu().numpy(),
                                               ln.cpu().numpy(), dev=(buf, off, ln))
            b.truth = truth
            batches.append(b)
        n = len(names)
        W.update(idx=idx, tax=None, kind=np.concatenate([kind, np.full(5, 2, dtype=np.int32)]), n=n, members=list(range(n)),
                 twin_of=np.full(n + 5, -1, dtype=np.int64), index_bp=int(lens.sum()), n_index_genomes=n,
                 workload=f'configs[1] (100k ONT reads vs human+decoy, decoy-filter alignment stage) at a synthetic human-like genome of {n_chrom} x {chrom_len} '
                          f'bp = {n_chrom * chrom_len / 1e9:.2f} Gbp with 45% interspersed repeat families + 8 plasmid-like decoys; reads: 70% human-like, 5% decoy, '
                          f'25% from 5 microbial genomes outside the index; {args.reads_per_step} reads/step, -x map-ont -c (-N 5 -p 0.8), human/decoy '
                          f'classification (megapath_nano.py:1135-1200) inside the step')
        del flat, mic, all_flat
    elif cfg == 'refseq':
        # configs[2] as north_star names it: a strain-rich target set of args.genomes x genome_len (default 18000 x 4 Mbp = 72 Gbp)
        # as args.parts RESIDENT index parts, generated and indexed one after another (the ASCII of a part leaves HBM before the
        # next one is made); reads come from the 10 base genomes of the community; every step maps every part, hits merged per read
        n_parts, n_fam, copies = args.parts, 10, 100
        per = args.genomes // n_parts
        base_names, base_flat, _ = synth.make_genomes_device(20240901, n_fam, args.genome_len, 0, device)
        base = base_flat.view(n_fam, args.genome_len)
        weights = np.random.default_rng(7).lognormal(0.0, 1.0, size=n_fam)
        batches = [make_batch(base_flat, weights, args, 1000 * (rank + 1) + s_, device) for s_ in range(n_distinct)]
        idx, fam_all, part_s, n_tot = [], [], [], 0
        t_all = time.time()
        for p_ in range(n_parts):
            names, flat, lens, fam = synth.make_refseq_part_device(20240901, p_, n_parts, per, args.genome_len, base, copies, 0.97, 0.999, device)
            torch.cuda.synchronize()
            if p_ == 0 and rank == 0 and world == 1 and not args.no_cpu_baseline:
                # host copies of the genomes the CPU baseline indexes: the community + fillers (the first genomes of part 0)
                view = flat.view(per, args.genome_len)
                W['cpu_genomes'] = [(names[g], view[g].cpu().numpy()) for g in range(min(per, max(n_fam, args.cpu_index_genomes)))]
            t0 = time.time()
            idx.append(mapper.Index.from_device(names, flat.data_ptr(), lens))
            part_s.append(round(time.time() - t0, 2))
            fam_all.append(fam)
            n_tot += len(names)
            del flat
            torch.cuda.empty_cache()
            if rank == 0:
                log(f'part {p_ + 1}/{n_parts}: {per} genomes indexed in {part_s[-1]:.1f} s ({idx[-1].n_minimizers} minimizers)')
        W['index_s'] = sum(part_s)
        fam_all = np.concatenate(fam_all)
        n = n_tot
        gid = np.arange(n, dtype=np.int64)
        # the species name of an assembly is its base genome's (sequence_name -> species, reassignment.py:69-71); truth genome f = target f
        name_code = np.where(fam_all >= 0, fam_all, gid).astype(np.int32)
        twin_of = np.where((fam_all >= 0) & (gid >= n_fam), fam_all, -1).astype(np.int64)
        opt_kw = dict(best_n=50, pri_ratio=1.0)  # megapath_nano.py:1270  -N 50 -p 1 -x map-ont
        W.update(idx=idx, tax=Taxonomy(name_code, n, name_code, n), n=n, members=list(range(n_fam)), twin_of=twin_of,
                 index_bp=n * args.genome_len, n_index_genomes=n, part_build_s=part_s, parts=n_parts,
                 workload=f'configs[2] (1M-read 10-species community vs full RefSeq bacterial, reassignment on) at N_g = {n} synthetic genomes x '
                          f'{args.genome_len} bp = {n * args.genome_len / 1e9:.1f} Gbp of targets, STRAIN-RICH ({copies} assemblies of each of the {n_fam} community '
                          f'species at 97-99.9% identity, spread over the parts), held as {n_parts} RESIDENT index parts of {per * args.genome_len / 1e9:.1f} Gbp '
                          f'(minimap2 -I) on every GPU, every read mapped against every part and the hits merged per read (mpn_hits = --split-prefix); '
                          f'{args.reads_per_step} synthetic ONT-like reads/step/GPU (Gamma lengths, mean {args.mean_len} bp, 12% errors, homopolymer-biased '
                          f'indels; in-repo stand-in for badread, generated on the GPU; {n_distinct} distinct batches rotated over the steps), -N 50 -p 1 -x map-ont -c')
        del base_flat
    else:
        n = args.genomes
        families = (10, 100, 0.97, 0.999) if cfg == 'strain' else None
        strain_pairs = 0 if families else args.strain_pairs
        if families:
            members = list(range(10))
            weights = np.zeros(n)
            weights[members] = np.random.default_rng(7).lognormal(0.0, 1.0, size=10)
            twin_of = np.full(n, -1, dtype=np.int64)
            twin_of[n - 1000:] = np.repeat(np.arange(10), 100)
        else:
            members, weights = community(n, strain_pairs)
            twin_of = np.full(n, -1, dtype=np.int64)       # strain copy -> the genome it was copied from
            twin_of[n - strain_pairs:] = np.arange(strain_pairs)
        names, flat, lens = synth.make_genomes_device(20240901, n, args.genome_len, strain_pairs, device, families=families)
        torch.cuda.synchronize()
        t0 = time.time()
        if cfg == 'parts':
            # two index parts, both resident (288 GB of HBM hold what a CPU host must stream through -I): the hits of the parts
            # are merged per read like minimap2 --split-prefix merges its dumps
            h = n // 2
            idx = [mapper.Index.from_device(names[:h], flat.data_ptr(), lens[:h]),
                   mapper.Index.from_device(names[h:], flat.data_ptr() + h * args.genome_len, lens[h:])]
        else:
            idx = mapper.Index.from_device(names, flat.data_ptr(), lens)
        W['index_s'] = time.time() - t0
        opt_kw = dict(best_n=50, pri_ratio=1.0)  # megapath_nano.py:1270  -N 50 -p 1 -x map-ont
        batches = [make_batch(flat, weights, args, 1000 * (rank + 1) + s_, device) for s_ in range(n_distinct)]
        # host copies of the genomes the CPU baseline indexes (the community + fillers), then the ASCII targets leave HBM
        if rank == 0 and world == 1 and not args.no_cpu_baseline and cfg == 'c3':
            want = max(len(members), min(n, args.cpu_index_genomes))
            ms = set(members)
            pick = sorted(members + [g for g in range(n) if g not in ms][:want - len(members)])
            view = flat.view(n, args.genome_len)
            W['cpu_genomes'] = [(names[g], view[g].cpu().numpy()) for g in pick]
        del flat
        what = {'c3': f'incl. {strain_pairs} 99%-identity strain copies', 'big': f'incl. {strain_pairs} 99%-identity strain copies (the largest one-piece index)',
                'parts': f'incl. {strain_pairs} 99%-identity strain copies, held as TWO resident index parts whose hits are merged per read (mpn_hits, minimap2 --split-prefix)',
                'strain': 'of which 1000 are 100 assemblies of each of the 10 community species at 97-99.9% identity (strain-rich)'}[cfg]
        W.update(idx=idx, tax=Taxonomy(np.arange(n, dtype=np.int32), n, np.arange(n, dtype=np.int32), n), n=n, members=members, twin_of=twin_of,
                 index_bp=n * args.genome_len, n_index_genomes=n,
                 workload=f'configs[2] (1M-read 10-species community vs full RefSeq bacterial, reassignment on) at N_g = {n} synthetic genomes x '
                          f'{args.genome_len} bp = {n * args.genome_len / 1e9:.1f} Gbp of targets {what}, resident on every GPU; '
                          f'{args.reads_per_step} synthetic ONT-like reads/step/GPU (Gamma lengths, mean {args.mean_len} bp, 12% errors; in-repo stand-in '
                          f'for badread, generated on the GPU; {n_distinct} distinct batches rotated over the steps), -N 50 -p 1 -x map-ont -c')
    torch.cuda.empty_cache()
    idx0 = W['idx'][0] if isinstance(W['idx'], (list, tuple)) else W['idx']
    if getattr(args, 'mapping_only', False):
        opt_kw = dict(opt_kw, with_cigar=0)
    opt = mapper.default_opt(**opt_kw)
    if not isinstance(W['idx'], (list, tuple)):
        opt.mid_occ = idx0.mid_occ()      # (index parts: every part applies its own -f cut-off, like minimap2)
    W.update(opt=opt, opt_kw=opt_kw, batches=batches)
    if rank == 0:
        log(f'index built in {W["index_s"]:.1f} s: {sum(i.n_minimizers for i in (W["idx"] if isinstance(W["idx"], (list, tuple)) else [W["idx"]]))} minimizers')
    return W


def human_decoy_step(idx, opt, b, kind, rnd, use_device):
    """configs[1]'s step: map against human+decoy, then the human / decoy / microbe classification on integer columns."""
    from megapath_nano_amd import mapper
    from megapath_nano_amd.filters import classify_codes
    from megapath_nano_amd.pipeline import random_block
    _, c = mapper.map_batch_ex(idx, opt, b, want_paf=False, want_cols=True, use_device=use_device)
    tb = random_block(rnd, len(c['read_idx']))
    cls = classify_codes(c['read_idx'], kind[c['rid']], c['as_'], tb, b.lens, b.n)
    return dict(read_count=np.bincount(cls, minlength=3).astype(np.int64), classes=cls, n_rows=len(tb))


def cpu_baseline(genomes, packed, opt_kw, seconds_target=15.0):
    """Oracle (port of the minimap2 path) on a bounded sample of the step batch, all host cores up to the CPU quota, against
    an index of >= 1 Gbp (so that the port, too, meets stray seed hits and the -f cut-off)."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'libmm2_oracle.so'], stdout=subprocess.DEVNULL)
    from oracle import mm2_bindings as mb
    cores = max(1, min(os.cpu_count() or 1, int(cpu_quota() or os.cpu_count() or 1)))
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    t0 = time.time()
    oidx = mb.Index(genomes)
    idx_s = time.time() - t0
    oopt = mb.default_opt(**opt_kw)
    oopt.mid_occ = oidx.mid_occ()
    reads = [(packed.names[i], packed.seq(i)) for i in range(min(packed.n, 65536))]
    t0 = time.time()
    for nm, s in reads[:8]:
        mb.map_read(oidx, oopt, nm, s)
    per_read = (time.time() - t0) / 8
    n = int(max(8, min(len(reads), seconds_target * cores / max(per_read, 1e-4))))
    sample = reads[:n]
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda r: mb.map_read(oidx, oopt, r[0], r[1])[1], sample))
    dt = time.time() - t0
    bases = sum(len(s) for _, s in sample)
    mid = int(oopt.mid_occ)
    oidx.close()
    idx_bp = sum(len(g[1]) for g in genomes)
    return dict(value=bases / dt * 60 / 1e9, unit='Gbp/min', cores=cores, host_cpus=os.cpu_count(), kind='port',
                sample=f'{n} reads ({bases} bp) of the step batch, oracle/mm2_oracle.c seed-chain-extend on {cores} threads, {dt:.1f} s wall; '
                       f'its index holds {len(genomes)} of the genomes (the community + fillers, {idx_bp} bp, built in {idx_s:.1f} s, '
                       f'mid_occ {mid}): {idx_bp / 1e9:.2f} Gbp against the GPU\'s full index, so the CPU still sees fewer stray seed hits',
                index_bp=idx_bp, index_build_s=round(idx_s, 1), mid_occ=mid, same_inputs=False,   # (same reads, a smaller index: host RAM and the time budget do not hold the GPU's)
                minimap2_on_box=shutil.which('minimap2'))   # a real binary would be timed beside the port; none ships in the image


class TargetSlices:
    """name -> sequence view for tests/paf_check.py over targets that live only in HBM: len() from the index, slices fetched
    through mpn_index_fetch_seq (the 2-bit targets decoded on the device).  Takes one index or a list of index parts."""

    class _Seq:
        def __init__(self, idx, i):
            self.idx, self.i = idx, i

        def __len__(self):
            return int(self.idx.lens[self.i])

        def __getitem__(self, sl):
            assert isinstance(sl, slice) and sl.step in (None, 1)
            lo, hi, _ = sl.indices(len(self))
            return self.idx.fetch_seq(self.i, lo, max(0, hi - lo)).decode()

    def __init__(self, idx):
        self.by_name = {}
        for part in (idx if isinstance(idx, (list, tuple)) else [idx]):
            for i, n in enumerate(part.names):
                self.by_name[n] = (part, i)

    def __contains__(self, name):
        return name in self.by_name

    def __getitem__(self, name):
        return TargetSlices._Seq(*self.by_name[name])


def map_text_and_cols(idx, opt, sub):
    """PAF text + columns of a batch against one index or against index parts (merged like --split-prefix)"""
    from megapath_nano_amd import mapper
    if isinstance(idx, (list, tuple)):
        h = mapper.Hits(sub)
        try:
            h.add_parts(list(idx), opt, use_device=False)
            paf, _, c = h.finish(opt, want_paf=True, want_cols=True)
        finally:
            h.close()
        return paf, c
    return mapper.map_batch_ex(idx, opt, sub, want_paf=True, want_cols=True, use_device=False)


def sub_batch(batch, n):
    from megapath_nano_amd import mapper
    end = int(batch.off[n - 1] + batch.lens[n - 1])
    return mapper.PackedReads.from_arrays(batch.names[:n], np.concatenate([batch.buf[:end], np.full(16, ord('A'), dtype=np.uint8)]),
                                          batch.off[:n], batch.lens[:n])


def c2_correctness(idx, opt, batch, kind, rnd, n_check=16384, n_paf_reads=1024):
    """configs[1]: the class of every read against the target it was sampled from (reads >= 2 kb: at 12 % errors their alignment
    score clears the reference's thresholds, AS >= 1000 or AS >= read length), and the independent PAF checker."""
    from megapath_nano_amd.filters import classify_codes
    from megapath_nano_amd.pipeline import random_block
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import paf_check
    out = {'ok': True, 'failures': []}
    n = min(n_check, batch.n)
    sub = sub_batch(batch, n)
    paf, c = map_text_and_cols(idx, opt, sub)
    cls = classify_codes(c['read_idx'], kind[c['rid']], c['as_'], random_block(rnd, len(c['read_idx'])), sub.lens, n)
    want = np.array([1, 2, 0], dtype=np.int8)[kind[batch.truth['genome'][:n]]]     # human target -> 1, decoy -> 2, microbe (not indexed) -> 0
    longr = sub.lens >= 2000
    for name, code in (('human', 1), ('decoy', 2), ('microbe', 0)):
        m = longr & (want == code)
        frac = float((cls[m] == code).mean()) if m.any() else 1.0
        out[f'{name}_reads_ge_2kb'] = int(m.sum())
        out[f'{name}_classified_frac'] = round(frac, 5)
        if frac < 0.97:
            out['ok'] = False
            out['failures'].append(f'fewer than 97 % of the {name} reads >= 2 kb were classified {name}')
    names_m = set(batch.names[:min(n_paf_reads, n)])
    text = ''.join(l for l in paf.splitlines(keepends=True) if l.split('\t', 1)[0] in names_m)
    reads = {batch.names[i]: bytes(batch.seq(i)).decode() for i in range(min(n_paf_reads, n))}
    st = {}
    try:
        paf_check.check_paf(text, reads, TargetSlices(idx), best_n=opt.best_n, stats=st, fast=True)
        out['paf_check'] = dict(lines=st.get('lines', 0), primaries=st.get('primaries', 0), as_equals_cigar_score=st.get('as_equal', 0))
    except AssertionError as e:
        out['ok'] = False
        out['paf_check'] = dict(error=str(e)[:300])
        out['failures'].append('paf_check failed: ' + str(e)[:200])
    return out


def correctness_block(idx, opt, batch, args, members, twin_of, counts, sampled, n_check=16384, n_paf_reads=2048):
    """Checks of THIS run's output at the full index size (VERDICT r2 #1b).  Nothing here is timed."""
    from megapath_nano_amd import mapper
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import paf_check
    out = {'ok': True, 'failures': []}

    def need(cond, what):
        if not cond:
            out['ok'] = False
            out['failures'].append(what)

    n = min(n_check, batch.n)
    sub = sub_batch(batch, n)
    t0 = time.time()
    paf, c = map_text_and_cols(idx, opt, sub)
    map_s = time.time() - t0
    # (1) truth: the first line of a read is its primary; it must lie on the genome the read was sampled from (or on that
    # genome's 99 % strain twin, which is an equally good locus) and overlap the sampled interval, for reads >= 1 kb
    tr = batch.truth
    first = np.ones(len(c['read_idx']), dtype=bool)
    first[1:] = c['read_idx'][1:] != c['read_idx'][:-1]
    ri = c['read_idx'][first]
    rid, rs, re_, rev = c['rid'][first], c['rs'][first], c['re'][first], c['rev'][first]
    tg, ts, te, trev = tr['genome'][ri], tr['start'][ri], tr['end'][ri], tr['rev'][ri]
    same = (rid == tg) | (twin_of[rid] == tg) | (rid == twin_of[tg])
    on_truth = same & (rs < te) & (re_ > ts) & (rev.astype(bool) == trev)
    long_reads = np.flatnonzero(batch.lens[:n] >= 1000)
    hit = np.zeros(n, dtype=bool)
    hit[ri] = on_truth
    mapped = np.zeros(n, dtype=bool)
    mapped[ri] = True
    out['reads_checked'] = int(n)
    out['reads_ge_1kb'] = int(len(long_reads))
    out['mapped_frac_ge_1kb'] = round(float(mapped[long_reads].mean()), 5)
    out['primary_on_true_locus_frac_ge_1kb'] = round(float(hit[long_reads].mean()), 5)
    need(out['primary_on_true_locus_frac_ge_1kb'] >= 0.97, 'fewer than 97 % of the reads >= 1 kb have their primary on the sampled locus')
    # aligned span of the primaries against the sampled span (an alignment that covers a tenth of its read is not a placement)
    cov = (re_ - rs)[on_truth] / np.maximum(1, (te - ts)[on_truth])
    out['median_primary_span_over_sampled_span'] = round(float(np.median(cov)), 4) if len(cov) else None
    need(len(cov) and np.median(cov) >= 0.9, 'primaries cover less than 90 % of the sampled interval (median)')
    # (2) the independent checker over the lines of the first reads: every per-alignment number recomputed from CIGAR + sequences
    m = min(n_paf_reads, n)
    lines = paf.splitlines(keepends=True)
    names_m = set(batch.names[:m])
    text = ''.join(l for l in lines if l.split('\t', 1)[0] in names_m)
    reads = {batch.names[i]: bytes(batch.seq(i)).decode() for i in range(m)}
    st = {}
    t0 = time.time()
    try:
        paf_check.check_paf(text, reads, TargetSlices(idx), best_n=opt.best_n, stats=st, fast=True)
        out['paf_check'] = dict(lines=st.get('lines', 0), primaries=st.get('primaries', 0), as_equals_cigar_score=st.get('as_equal', 0),
                                seconds=round(time.time() - t0, 1))
        need(st.get('lines', 0) >= min(2000, m), f'paf_check saw only {st.get("lines", 0)} lines')
        need(st.get('as_equal', 0) >= 0.98 * st.get('lines', 1), 'AS differs from the CIGAR\'s dual-affine score on more than 2 % of the lines')
    except AssertionError as e:
        out['paf_check'] = dict(error=str(e)[:300])
        need(False, 'paf_check failed: ' + str(e)[:200])
    # (3) per-name read counts of the timed steps against the composition the generator sampled (strain twins pooled: a read of
    # genome g is a correct assignment to g or to its 99 % copy)
    pooled = lambda v: {int(g): int(v[g] + (v[np.flatnonzero(twin_of == g)].sum() if (twin_of == g).any() else 0)) for g in members if twin_of[g] < 0}
    got, want = pooled(counts), pooled(sampled)
    tot_got, tot_want = int(counts.sum()), int(sampled.sum())
    off = tot_got - sum(got.values())
    dev = max(abs(got[g] - want[g]) / max(want[g], 1) for g in want)
    out['counts'] = dict(reads_sampled=tot_want, reads_assigned=tot_got, assigned_outside_community=int(off),
                         max_rel_dev_per_member=round(float(dev), 5),
                         per_member={str(g): [got[g], want[g]] for g in want})
    need(tot_got >= 0.97 * tot_want, 'fewer than 97 % of the sampled reads were assigned a name')
    need(off <= 0.005 * tot_want, 'more than 0.5 % of the reads were assigned outside the community')
    need(dev <= 0.03, 'a community member\'s count deviates by more than 3 % from the sampled composition')
    out['map_seconds'] = round(map_s, 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='refseq', choices=['refseq', 'c3', 'strain', 'big', 'parts', 'c2'],
                    help='refseq = BASELINE configs[2] at north_star\'s scale (the headline): a strain-rich 72 Gbp target set as resident index '
                         'parts, hits merged per read; c3 = the round-1..3 headline (20 Gbp of random genomes + 10 strain copies, one index); '
                         'strain = c3\'s size with 100 assemblies of every community species; big = the largest one-piece index (36 Gbp); '
                         'parts = the 20 Gbp target set as two index parts merged like minimap2 --split-prefix; c2 = BASELINE configs[1] '
                         '(100k reads vs a repeat-rich human-like genome + decoys, -N 5 -p 0.8, human/decoy classification)')
    ap.add_argument('--parts', type=int, default=4, help='refseq: resident index parts the target set is cut into (9 = the reference\'s -I 8G on a 512 GiB host)')
    ap.add_argument('--genomes', type=int, default=None)
    ap.add_argument('--genome-len', type=int, default=4000000)
    ap.add_argument('--strain-pairs', type=int, default=10)
    ap.add_argument('--reads-per-step', type=int, default=None)
    ap.add_argument('--mean-len', type=int, default=8000)
    ap.add_argument('--distinct-batches', type=int, default=3)
    ap.add_argument('--resident-steps', '--pcie-steps', dest='resident_steps', type=int, default=6,
                    help='extra steps with the reads already resident in HBM (the side figure resident_input_gbp_per_min; `value` is fed from host buffers)')
    ap.add_argument('--alone-reads', type=int, default=0, help='reads of the single-worker pass that times every kernel alone (0 = an eighth of a batch; -1 = skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-correctness', action='store_true')
    ap.add_argument('--mapping-only', action='store_true', help='diagnostic: no base-level extension (the reference\'s mapping_only mode); implies --no-correctness, not the metric')
    ap.add_argument('--cpu-index-genomes', type=int, default=250, help='genomes of the CPU baseline\'s index (250 x 4 Mbp = 1 Gbp)')
    args = ap.parse_args()
    if args.mapping_only:
        args.no_correctness = True
        args.no_cpu_baseline = True
    cfg_defaults = {'refseq': (18000, 65536), 'c3': (5000, 262144), 'strain': (5000, 65536), 'big': (9000, 262144), 'parts': (5000, 262144), 'c2': (0, 100000)}
    if args.genomes is None:
        args.genomes = cfg_defaults[args.config][0]
    if args.reads_per_step is None:
        args.reads_per_step = cfg_defaults[args.config][1]

    env_world = os.environ.get('WORLD_SIZE')
    if args.gpus > 1 and env_world is None:
        launch_ranks(args.gpus)       # does not return
    if env_world is not None and int(env_world) != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}')
    # the library sizes its host thread pool by the CPU quota divided by the ranks that share this node
    os.environ.setdefault('MPN_RANKS_ON_NODE', os.environ.get('LOCAL_WORLD_SIZE', env_world or '1'))

    import torch
    from megapath_nano_amd import dist as mdist, mapper, synth
    from megapath_nano_amd.pipeline import Taxonomy, align_and_assign
    # MPN_DIST_BACKEND=gloo + MPN_SINGLE_DEVICE=1 rehearses the N>1 code path on a one-GPU box (all ranks on cuda:0,
    # counters reduced over gloo); the driver's real runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get('MPN_DIST_BACKEND') or None
    single = os.environ.get('MPN_SINGLE_DEVICE') == '1'
    rank, world, local = mdist.init_from_env(backend=backend)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    device = torch.device('cuda', 0 if (world == 1 or single) else local)
    torch.cuda.set_device(device)
    red_device = None if backend == 'gloo' else device
    from megapath_nano_amd import build
    if rank == 0:
        build.build()
    mdist.barrier()
    if rank == 0:
        log(f'torch + libmpn ready; generating {args.genomes} x {args.genome_len} bp on the GPU')

    # ---- targets + reads of the chosen workload: generated in HBM, indexed from HBM ----------------------------------------
    W = build_workload(args, device, rank, world)
    idx, tax, opt, opt_kw, batches, members, twin_of, n = W['idx'], W['tax'], W['opt'], W['opt_kw'], W['batches'], W['members'], W['twin_of'], W['n']
    index_s, n_distinct, cpu_genomes = W['index_s'], len(W['batches']), W['cpu_genomes']
    idx0 = idx[0] if isinstance(idx, (list, tuple)) else idx
    if rank == 0:
        log(f'{n_distinct} read batches of {args.reads_per_step} reads ready ({batches[0].bases} bp each); mid_occ = {opt.mid_occ}')

    allreduce = mdist.make_allreduce(red_device)
    rnd = random.Random(12345)  # the same stream on every rank: tiebreakers are drawn in global row order (pipeline.sharded_tiebreak)

    def run(b, use_device=True):
        if args.config == 'c2':
            return human_decoy_step(idx, opt, b, W['kind'], rnd, use_device)
        return align_and_assign(idx, opt, b, tax, allreduce=allreduce, rng=rnd, shard=(rank, world), use_device=use_device)

    for s in range(args.warmup):
        run(batches[s % n_distinct], use_device=False)
        if rank == 0:
            log(f'warmup step {s + 1}/{args.warmup} done')
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    thr0 = None
    if os.environ.get('MPN_DEBUG_THREADS'):
        sys.path.insert(0, os.path.join(ROOT, 'scripts'))
        import thread_cpu
        thr0 = thread_cpu.snapshot()
        det0 = dict(thread_cpu.DETAIL)
    stats_acc = {}
    counts = None
    sampled = np.zeros(n, dtype=np.int64)   # what this rank's generator sampled in the timed steps, per genome
    bases = 0
    for s in range(args.steps):
        b = batches[(args.warmup + s) % n_distinct]
        out = run(b, use_device=False)   # fed from host buffers: the read H2D is inside the timed call (SURVEY 8d's wall time)
        bases += b.bases
        counts = out['read_count'] if counts is None else counts + out['read_count']
        for k, v in mapper.last_stats().items():
            stats_acc[k] = stats_acc.get(k, 0) + v
        if rank == 0 and (s % 4 == 3 or s == args.steps - 1):
            log(f'timed step {s + 1}/{args.steps} done ({(time.perf_counter() - t0) / (s + 1):.2f} s/step)')
    torch.cuda.synchronize()
    mdist.barrier()
    dt = time.perf_counter() - t0
    host_cpu_s = time.process_time() - cpu0
    if thr0 is not None and rank == 0:
        thr1 = thread_cpu.snapshot()
        for name, (cnt, sec) in thread_cpu.diff(thr0, thr1)[:12]:
            log(f'threads {name!r} x{cnt}: {sec / max(1, args.steps):.3f} CPU-s per step')
        for tid, name, sec, is_main in thread_cpu.top_threads(thr0, thr1):
            u1, s1, (v1, n1) = thread_cpu.DETAIL.get(tid, (0, 0, (0, 0)))
            u0, s0, (v0, n0) = det0.get(tid, (0, 0, (0, 0)))
            log(f'  thread {tid} {name!r}{" (main)" if is_main else ""}: {sec / max(1, args.steps):.3f} CPU-s per step '
                f'(user {(u1 - u0) / max(1, args.steps):.3f}, system {(s1 - s0) / max(1, args.steps):.3f}; context switches per step: '
                f'{(v1 - v0) / max(1, args.steps):.0f} voluntary, {(n1 - n0) / max(1, args.steps):.0f} involuntary)')
    if args.config != 'c2':
        for s in range(args.steps):
            sampled += np.bincount(batches[(args.warmup + s) % n_distinct].truth['genome'], minlength=n)
    host_cpu_max = host_cpu_s
    if world > 1:
        import torch.distributed as dist
        rd = red_device if red_device is not None else 'cpu'
        t = torch.tensor([dt, host_cpu_s], dtype=torch.float64, device=rd)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, host_cpu_max = float(t[0].item()), float(t[1].item())
        bt = torch.tensor([bases], dtype=torch.int64, device=rd)
        dist.all_reduce(bt, op=dist.ReduceOp.SUM)
        bases = int(bt.item())
        if args.config != 'c2':
            allreduce(sampled)   # counts are already global (align_and_assign all-reduces them)
    # side figure: the same steps with the reads already resident in HBM (no H2D in the call)
    resident = None
    if args.resident_steps > 0:
        mdist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pb = 0
        for s in range(args.resident_steps):
            b = batches[s % n_distinct]
            run(b, use_device=True)
            pb += b.bases
        torch.cuda.synchronize()
        mdist.barrier()
        resident = pb * world / (time.perf_counter() - t1) * 60 / 1e9
        if rank == 0:
            log(f'{args.resident_steps} resident-input steps done')
    if rank != 0:
        return

    K = max(1, args.steps)
    st = {k: v / K for k, v in stats_acc.items()}  # per step (rank 0), summed over the index parts a step maps
    nsub = max(st['sub_batches'], 1)
    rounds = max(st['dp_rounds'], 1)

    # ---- every kernel ALONE: a slice of a batch through ONE pipeline worker (nothing else on the GPU), after the timed region ----
    alone = None
    if args.alone_reads >= 0:
        na = args.alone_reads or max(256, batches[0].n // 8)
        sub = sub_batch(batches[0], min(na, batches[0].n))
        os.environ['MPN_PIPE_WORKERS'] = '1'
        try:
            acc = {}
            for part in (idx if isinstance(idx, (list, tuple)) else [idx]):
                mapper.map_batch_ex(part, opt, sub, want_paf=False, want_cols=True, use_device=False)
                for k, v in mapper.last_stats().items():
                    acc[k] = acc.get(k, 0) + v
            alone = acc
        finally:
            del os.environ['MPN_PIPE_WORKERS']
        log(f'single-worker pass over {sub.n} reads done ({int(alone["sub_batches"])} sub-batches)')

    # Candidate kernels: (stat key(s) of the HIP-event span, launches per sub-batch or round, ALGORITHMIC bytes per step, cells per
    # step for the DP kernels).  Algorithmic bytes:
    #   sketch (count pass + fill)   : 2 x 1 B per read base in + 16 B per minimizer out
    #   seed lookup                  : 16 B per minimizer in + 12 B (count, first position) out
    #   stray-hit filter (2 passes)  : 2 x 8 B per index position gathered + 1 keep bit per position out
    #   anchor emission              : 8 B per EMITTED position gathered + 16 B per emitted anchor out
    #   anchor partition (MSD)       : 8 B key read for the histogram + 16 B record in + 16 B out per emitted anchor
    #   anchor window sort           : 16 B in + 16 B out per emitted anchor
    #   anchor compaction (2 passes) : 2 x 16 B per emitted anchor in + 16 B per kept anchor out
    #   chain DP                     : 16 B per kept anchor in + 16 B (f, p, t, v) out
    #   strip DP (gap fill / exact)  : 1 direction byte out per DP cell (qlen x tlen per window)
    #   alignment finishing          : 4 B per CIGAR op in + 4 B out, ~1 B per aligned query base + 0.25 B per target base in
    def candidates(s):
        strip_cells = s['strip16_cells'] + s['strip32_cells'] + s['strip64_cells']
        return {
            'sketch_fast_kernel + sketch_fill_kernel': (('k_sketch_count_ns', 'k_sketch_fill_ns'), ('sub', 2), 2 * s['bases'] + 16 * s['minimizers'], 0),
            'seed_lookup_kernel': (('k_seed_lookup_ns',), ('sub', 1), 28 * s['minimizers'], 0),
            'seed_filter_kernel': (('k_seed_filter_ns',), ('sub', 1), 16 * s['anchors'] + s['anchors'] / 8, 0),
            'seed_emit_kernel': (('k_seed_fill_ns',), ('sub', 1), 24 * s['anchors_emitted'], 0),
            'anchor_msd_kernel': (('k_sort_msd_ns',), ('sub', 1), 40 * s['anchors_emitted'], 0),
            'anchor_window_sort_kernel': (('k_sort_chunk_ns',), ('sub', 1), 32 * s['anchors_emitted'], 0),
            'anchor_compact_kernel<count|write>': (('k_compact_ns',), ('sub', 2), 32 * s['anchors_emitted'] + 16 * s['anchors_kept'], 0),
            'chain_dp_kernel': (('k_chain_dp_ns',), ('sub', 1), 32 * s['anchors_kept'], 0),
            # (one launch per round: 16-, 32- and 64-lane groups are segments of its grid)
            'ext_dp_strip_kernel<false> (gap fills)': (('k_strip16_ns', 'k_strip32_ns', 'k_strip64_ns'), ('round', 1), strip_cells, strip_cells),
            'ext_dp_strip_kernel<true> (end extensions, exact fills)': (('k_xstrip_ns',), ('round', 1), s['xstrip_cells'], s['xstrip_cells']),
            'aln_finish_wave_kernel': (('k_finish_ns',), ('round', 4), 8 * s['cigar_ops'] + 2 * s['bases'], 0),
        }
    cand = candidates(st)
    # The single-worker pass maps a slice of a batch part by part: its launches are not the timed region's launches (the sub-batches
    # of one worker with the whole scratch are larger), so its fractions are formed from ITS OWN algorithmic bytes and cells, and
    # its time is scaled to a step by the kernel's algorithmic bytes (a slice maps the same reads against every part, like a step).
    cand_alone = candidates(alone) if alone is not None else None
    VALU_PEAK = 1024 * 2.4e9 / 2.0    # MI355X_MICROARCH.md: 1024 SIMDs, a wave64 VALU instruction issues over 2 cycles at 2.4 GHz
    wi_per_cell = STRIP_INSTR_PER_CELL / 64.0
    rows = {}
    n_launch = lambda unit, stats: unit[1] * max(stats['sub_batches'] if unit[0] == 'sub' else stats['dp_rounds'], 1)
    for name, (keys, unit, abytes, cells) in cand.items():
        launches = n_launch(unit, st)
        ns_pipe = sum(st[k] for k in keys)
        row = {'launches_per_step': round(launches, 1), 'alg_GB_per_step': round(abytes / 1e9, 3),
               'in_pipeline_ms': round(ns_pipe / 1e6 / max(launches, 1), 3), 'in_pipeline_ms_per_step': round(ns_pipe / 1e6, 2)}
        if alone is not None:
            ns_alone = sum(alone[k] for k in keys)
            launches_a = n_launch(unit, alone)
            abytes_a = cand_alone[name][2]
            row['alone_ms'] = round(ns_alone / 1e6 / max(launches_a, 1), 3)
            row['alone_alg_GB_per_launch'] = round(abytes_a / 1e9 / max(launches_a, 1), 4)
            row['alone_ms_per_step'] = round(ns_alone / 1e6 * abytes / max(abytes_a, 1), 2)
        rows[name] = row
    # the dominant kernel: largest EXCLUSIVE device time per step (alone), else largest in-pipeline span
    keyf = (lambda k: rows[k].get('alone_ms_per_step', 0.0)) if alone is not None else (lambda k: rows[k]['in_pipeline_ms_per_step'])
    dom = max(rows, key=keyf)
    keys, unit, abytes, cells = cand[dom]
    launches = n_launch(unit, st)
    ns = sum(st[k] for k in keys)
    achieved = abytes / max(ns, 1)  # bytes per ns == GB/s, from the HIP-event spans of the timed region (the contract's figure)
    traffic, traffic_src = pmc_traffic(dom, launches, args.config)
    hits_per_mz = st['anchors'] / max(st['minimizers'], 1)
    roof = {
        'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_src,
        'launches_per_step': round(launches, 1), 'launch_ms_avg': round(ns / 1e6 / max(launches, 1), 3),
        'algorithmic_bytes_per_launch': int(abytes / max(launches, 1)),
        'chosen_by': 'largest exclusive device time per step (single-worker pass)' if alone is not None else 'largest in-pipeline HIP-event span (no single-worker pass)',
    }
    if alone is not None:
        ns_alone_dom = max(sum(alone[k] for k in keys), 1)
        roof['alone_ms'] = rows[dom]['alone_ms']
        roof['alone_algorithmic_bytes_per_launch'] = int(cand_alone[dom][2] / max(n_launch(unit, alone), 1))
        roof['frac_alone'] = round(cand_alone[dom][2] / ns_alone_dom / HBM_PEAK_GBS, 5)
    if cells:
        wi_launch = cells * wi_per_cell / max(launches, 1)
        roof['valu'] = {
            'cells_per_step': int(cells), 'wave_instr_per_cell': round(wi_per_cell, 4), 'wave_instr_per_launch': int(wi_launch),
            'peak_wave_instr_per_s': VALU_PEAK,
            'achieved_wave_instr_per_s': round(wi_launch / max(ns / max(launches, 1), 1) * 1e9, 0),
            'frac': round(wi_launch / max(ns / max(launches, 1), 1) * 1e9 / VALU_PEAK, 4),
            'frac_alone': round(cand_alone[dom][3] * wi_per_cell / ns_alone_dom * 1e9 / VALU_PEAK, 4) if alone is not None else None,
            'mix_ceiling_wave_instr_per_s': VALU_CEIL_WAVE_INSTR,
            'gcups_whole_step': round(st['dp_cells'] / (dt / K * 1e9), 2),
            'note': 'the DP kernels are bound by VALU issue, not by HBM. wave-instructions = cells x 16.8 (ISA count of the loop body per cell and lane, two windows '
                    'per lane group) / 64: the ALGORITHMIC count -- ramps of the systolic array, padding rows, unequal pairs and per-step overhead are not in it (PMC: '
                    '22.2 per cell). peak = 1024 SIMDs x 2.4 GHz / 2 cycles (the guide); mix_ceiling = what the cell\'s own instruction mix can issue '
                    '(profiles/r03/valu_microbench2.txt: 7 instructions at 2.7 cycles + 9.8 packed / three-operand ones at 4.4 per cell).',
        }
    roof['note'] = ('in_pipeline_ms / frac: HIP-event span per launch on its own stream over the timed region, while the other pipeline workers share the GPU '
                    '(the rocprofv3 average of the same command, profiles/, is the figure to compare). alone_ms / frac_alone: the kernel with nothing else on '
                    'the GPU (single-worker pass over a slice of a batch after the timed region; its launches are larger than the pipeline\'s, so the fractions use the '
                    'slice\'s own bytes and cells: alone_algorithmic_bytes_per_launch). traffic: HBM bytes per launch (FETCH_SIZE + WRITE_SIZE) from committed PMC passes of the '
                    'same workload (separate rocprofv3 runs), null when none is committed for this --config.')
    roof['candidates'] = rows
    roof['whole_path_alg_bytes_per_bp'] = round(9.73 + 13.1 * hits_per_mz + 1.75 * st['alignments'] / max(args.reads_per_step, 1), 2)
    n_parts = len(idx) if isinstance(idx, (list, tuple)) else 1
    line = {
        'metric': 'Gbp/min ONT reads aligned+species-assigned vs RefSeq, 1/2/4/8 MI355X',
        'value': bases / dt * 60 / 1e9,
        'unit': 'Gbp/min',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt / K * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'int32',
        'data': 'synthetic',
        'config': {
            'workload': W['workload'] + f' (index built on the GPU in {index_s:.1f} s)', 'name': args.config,
            'reads_per_step_per_gpu': args.reads_per_step, 'index_genomes': W['n_index_genomes'], 'index_bp': W['index_bp'],
            'index_parts': n_parts, 'part_build_s': W.get('part_build_s'),
            'index_build_s': round(index_s, 2), 'mid_occ': [int(i.mid_occ()) for i in idx] if isinstance(idx, (list, tuple)) else int(opt.mid_occ),
            'index_minimizers': int(sum(i.n_minimizers for i in (idx if isinstance(idx, (list, tuple)) else [idx]))),
            'hits_per_read_minimizer_per_part': round(hits_per_mz, 2),                                     # h of SURVEY 8d (per part mapped)
            'alignments_per_read': round(st['alignments'] / max(args.reads_per_step, 1) / 1.0, 3),         # c (after the merge over parts)
            'parallelism': f'reads sharded over {world} GPU(s), index replicated', 'host_cpus': os.cpu_count(), 'cpu_quota': cpu_quota(),
            'ranks_on_node': int(os.environ.get('MPN_RANKS_ON_NODE', '1')),
        },
        'value_inputs': 'reads in HOST buffers when the timed region starts: the read H2D is inside the timed calls (SURVEY 8d); index resident in HBM',
        'resident_input_gbp_per_min': None if resident is None else round(resident, 2),
        'resident_steps': args.resident_steps,
        'host_cpu_s_per_step': round(host_cpu_s / K, 3),           # rank 0's process CPU time per step
        'host_cpu_s_per_step_max_rank': round(host_cpu_max / K, 3),
        'host_cpu_s_per_gbp': round(host_cpu_s / max(bases / world, 1) * 1e9, 3),
        'roofline': roof,
        'per_step': {k: (round(v / 1e6, 2) if k.endswith('_ns') else int(v)) for k, v in st.items()},
        'reads_per_name_top': sorted(((int(c), int(i)) for i, c in enumerate(counts) if c), reverse=True)[:5],
    }
    if args.config == 'c2':
        line['reads_per_class'] = dict(zip(('microbe', 'human', 'decoy'), (int(x) for x in counts)))
        line.pop('reads_per_name_top', None)
        if not args.no_correctness:
            line['correctness'] = c2_correctness(idx, opt, batches[0], W['kind'], rnd)
            log(f'correctness: ok={line["correctness"]["ok"]} {line["correctness"]["failures"]}')
    elif not args.no_correctness:
        log('checking the output of this run (truth hit rate, paf_check, counts)')
        line['correctness'] = correctness_block(idx, opt, batches[0], args, members, twin_of, counts, sampled)
        log(f'correctness: ok={line["correctness"]["ok"]} {line["correctness"]["failures"]}')
    if world == 1 and not args.no_cpu_baseline and cpu_genomes is not None:
        log('timing the CPU oracle on a sample')
        line['cpu_baseline'] = cpu_baseline(cpu_genomes, batches[0], opt_kw)
    print(json.dumps(line), flush=True)
    log('done')
    if not args.no_correctness and not line['correctness']['ok']:
        sys.exit(3)


if __name__ == '__main__':
    main()
