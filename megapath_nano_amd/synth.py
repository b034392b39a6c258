"""Seeded synthetic genomes and ONT-like reads.

`badread` is not available in the build image: this generator is its declared stand-in.  Genomes are random
DNA with a per-genome GC content, planted IS-like repeats and optional near-identical strain pairs so that
multi-mapping and reassignment trigger; reads have Gamma-distributed lengths and substitution / insertion /
deletion errors.  Pure numpy; used by tests, bench.py and the smoke test (data, not algorithm).
"""
import numpy as np

ALPHA = np.frombuffer(b'ACGT', dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b'ACGTNacgtn', b'TGCANtgcan'):
    COMP[_a] = _b


def random_genome(rng, length, gc=0.5):
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return ALPHA[rng.choice(4, size=length, p=p)]


def plant_repeats(rng, g, n_copies=10, rep_len=1500):
    rep = ALPHA[rng.integers(0, 4, size=rep_len)]
    for _ in range(n_copies):
        if len(g) <= rep_len + 1:
            break
        s = int(rng.integers(0, len(g) - rep_len))
        g[s:s + rep_len] = rep
    return g


def mutate_strain(rng, g, identity=0.99):
    g = g.copy()
    n = int(len(g) * (1 - identity))
    pos = rng.integers(0, len(g), size=n)
    g[pos] = ALPHA[(np.searchsorted(ALPHA, g[pos]) + rng.integers(1, 4, size=n)) % 4]
    return g


def make_genomes(seed, n_genomes, length, strain_pairs=1, repeats=True):
    """-> list of (name, uint8 ASCII array).  The last `strain_pairs` genomes are 99%-identity copies of the first ones."""
    rng = np.random.default_rng(seed)
    out = []
    n_base = n_genomes - strain_pairs
    for i in range(n_base):
        g = random_genome(rng, length, gc=float(rng.uniform(0.35, 0.65)))
        if repeats:
            g = plant_repeats(rng, g, 10, min(1500, max(50, length // 20)))
        out.append((f'NZ_SYN{i:05d}.1', g))
    for i in range(strain_pairs):
        out.append((f'NZ_STR{i:05d}.1', mutate_strain(rng, out[i][1], 0.99)))
    return out


HP_IN, HP_OUT = 1.75, 0.75   # indel rate inside / outside a homopolymer context (a base equal to its predecessor: a quarter of
                             # the positions of random sequence, so the mean rates stay sub / ins / dele)


def ont_errors(rng, seq, sub=0.04, ins=0.03, dele=0.05, hp_bias=True):
    """ONT-like errors (vectorised), SURVEY 8d: substitutions are independent per base; deletions and insertions are
    homopolymer-biased -- a base that repeats its predecessor is deleted or followed by an insertion HP_IN / HP_OUT times as often
    as another one, and what is inserted after it is a further copy of it (a homopolymer read too long); elsewhere inserted
    bases are random.  Deletion drops the base, insertion adds a geometric number of bases after it."""
    n = len(seq)
    hp = np.zeros(n, dtype=bool)
    if hp_bias and n > 1:
        hp[1:] = seq[1:] == seq[:-1]
    scale = np.where(hp, HP_IN, HP_OUT) if hp_bias else 1.0
    d_eff, i_eff = dele * scale, ins * scale
    r = rng.random(n)
    keep = r >= d_eff
    subm = keep & (r < d_eff + sub)
    s = seq.copy()
    s[subm] = ALPHA[(np.searchsorted(ALPHA, s[subm]) + rng.integers(1, 4, size=int(subm.sum()))) % 4]
    n_ins = np.where(rng.random(n) < i_eff, rng.geometric(0.6, size=n), 0)
    n_ins[~keep] = 0
    counts = keep.astype(np.int64) + n_ins
    out = np.repeat(s, counts)
    # positions of inserted bases: all but the first copy of each kept base
    starts = np.cumsum(counts) - counts
    first = np.zeros(len(out), dtype=bool)
    first[starts[counts > 0]] = True
    ins_mask = ~first & ~np.repeat(hp, counts)   # (insertions in a homopolymer context stay copies of the base)
    out[ins_mask] = ALPHA[rng.integers(0, 4, size=int(ins_mask.sum()))]
    return out


def make_reads(seed, genomes, n_reads, mean_len=8000, min_len=200, max_len=None, weights=None,
               sub=0.04, ins=0.03, dele=0.05, random_frac=0.0):
    """-> list of dict(name, seq (uint8 ASCII), genome index, start, end, strand)."""
    rng = np.random.default_rng(seed)
    w = np.ones(len(genomes)) if weights is None else np.asarray(weights, dtype=float)
    w = w / w.sum()
    reads = []
    for r in range(n_reads):
        L = int(max(min_len, rng.gamma(1.6, mean_len / 1.6)))
        if max_len:
            L = min(L, max_len)
        if rng.random() < random_frac:
            reads.append(dict(name=f'read{r:07d}', seq=ALPHA[rng.integers(0, 4, size=L)], genome=-1, start=0, end=0,
                              strand='+'))
            continue
        gi = int(rng.choice(len(genomes), p=w))
        g = genomes[gi][1]
        L = min(L, len(g))
        s = int(rng.integers(0, len(g) - L + 1))
        frag = g[s:s + L]
        strand = '+'
        if rng.random() < 0.5:
            frag = COMP[frag[::-1]]
            strand = '-'
        reads.append(dict(name=f'read{r:07d}', seq=ont_errors(rng, frag, sub, ins, dele), genome=gi, start=s, end=s + L,
                          strand=strand))
    return reads


# ---- the same generators on the GPU (torch): bench.py's workload is tens of Gbp of targets and ~1 Gbp of reads per batch,
# which numpy cannot produce inside the bench's time budget.  Same distributions as above (not the same random streams).
def make_genomes_device(seed, n_genomes, length, strain_pairs, device, repeats=True, chunk=64, families=None):
    """-> (names, uint8 tensor [n_genomes * length] of concatenated ASCII on `device`, int32 lens).
    families = (n_fam, copies, id_lo, id_hi): a strain-RICH target set -- the last n_fam * copies genomes (strain_pairs must be
    0) are `copies` assemblies of each of the genomes 0 .. n_fam-1, every copy mutated on its own at an identity drawn
    uniformly from [id_lo, id_hi] (RefSeq holds hundreds of near-identical assemblies of the common species)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    n_fam_copies = families[0] * families[1] if families else 0
    assert not (families and strain_pairs), 'families replace the strain pairs'
    n_base = n_genomes - strain_pairs - n_fam_copies
    out = torch.empty(n_genomes * length, dtype=torch.uint8, device=device)
    alpha = torch.tensor(list(b'ACGT'), dtype=torch.uint8, device=device)
    view = out.view(n_genomes, length)
    for g0 in range(0, n_base, chunk):
        g1 = min(n_base, g0 + chunk)
        gc = torch.rand(g1 - g0, 1, generator=gen, device=device) * 0.3 + 0.35
        u = torch.rand(g1 - g0, length, generator=gen, device=device)
        # A | C | G | T with P(C) = P(G) = gc / 2
        code = (u >= (1 - gc) / 2).to(torch.uint8) + (u >= 0.5).to(torch.uint8) + (u >= 0.5 + gc / 2).to(torch.uint8)
        view[g0:g1] = alpha[code.long()]
        del u, code
        if repeats and length > 3000:
            rep_len, n_copies = min(1500, max(50, length // 20)), 10
            rep = alpha[torch.randint(0, 4, (g1 - g0, 1, rep_len), generator=gen, device=device)]
            starts = torch.randint(0, length - rep_len, (g1 - g0, n_copies, 1), generator=gen, device=device)
            cols = (starts + torch.arange(rep_len, device=device)).view(g1 - g0, -1)
            view[g0:g1].scatter_(1, cols, rep.expand(-1, n_copies, -1).reshape(g1 - g0, -1))
    for i in range(strain_pairs):  # 99 %-identity copies of the first genomes
        g = view[i].clone()
        n_mut = length // 100
        pos = torch.randint(0, length, (n_mut,), generator=gen, device=device)
        cur = (g[pos] == alpha[1]).long() + 2 * (g[pos] == alpha[2]).long() + 3 * (g[pos] == alpha[3]).long()
        g[pos] = alpha[(cur + torch.randint(1, 4, (n_mut,), generator=gen, device=device)) % 4]
        view[n_base + i] = g
    names = [f'NZ_SYN{i:05d}.1' for i in range(n_base)] + [f'NZ_STR{i:05d}.1' for i in range(strain_pairs)]
    if families:
        n_fam, copies, lo, hi = families
        for f in range(n_fam):
            base = view[f]
            cur = (base == alpha[1]).long() + 2 * (base == alpha[2]).long() + 3 * (base == alpha[3]).long()
            for c in range(copies):
                ident = lo + (hi - lo) * float(torch.rand(1, generator=gen, device=device))
                mut = torch.rand(length, generator=gen, device=device) < (1.0 - ident)
                shift = torch.randint(1, 4, (length,), generator=gen, device=device)
                view[n_base + f * copies + c] = torch.where(mut, alpha[(cur + shift) % 4], base)
                names.append(f'NZ_FAM{f:02d}_{c:03d}.1')
    return names, out, np.full(n_genomes, length, dtype=np.int32)


def make_refseq_part_device(seed, part, n_parts, n_genomes_part, length, base, copies, id_lo, id_hi, device, chunk=64):
    """One part of a strain-RICH target set that is generated and indexed part by part (the whole set need not fit beside the
    resident index parts): `n_genomes_part` genomes of `length` bp = [the community's base genomes `base` (part 0 only)] +
    [random genomes] + [the assemblies c = part, part + n_parts, ... < copies of every base genome, each mutated on its own at
    an identity drawn uniformly from [id_lo, id_hi]] -- the hundreds of near-identical assemblies RefSeq holds of a common
    species are spread over the parts of a `-I`-split index like this.
    -> (names, uint8 tensor of the concatenated ASCII, int32 lens, family) with family[i] = the base genome an assembly was
    copied from (the base genomes themselves included), -1 for the random genomes."""
    import torch
    n_fam = base.shape[0]
    mine = [c for c in range(copies) if c % n_parts == part]
    n_base = n_fam if part == 0 else 0
    n_rand = n_genomes_part - n_base - n_fam * len(mine)
    assert n_rand >= 0
    rn, rflat, _ = make_genomes_device(int(seed) * 1000 + part, n_rand, length, 0, device, chunk=chunk) if n_rand else ([], None, None)
    out = torch.empty(n_genomes_part * length, dtype=torch.uint8, device=device)
    view = out.view(n_genomes_part, length)
    names, family = [], []
    k = 0
    if n_base:
        view[:n_fam] = base
        names += [f'NZ_COM{f:02d}.1' for f in range(n_fam)]
        family += list(range(n_fam))
        k = n_fam
    if n_rand:
        view[k:k + n_rand] = rflat.view(n_rand, length)
        names += [f'NZ_P{part:02d}R{i:05d}.1' for i in range(n_rand)]
        family += [-1] * n_rand
        k += n_rand
        del rflat
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) * 7919 + 31 * part + 5)
    alpha = torch.tensor(list(b'ACGT'), dtype=torch.uint8, device=device)
    for f in range(n_fam):
        b = base[f]
        cur = (b == alpha[1]).long() + 2 * (b == alpha[2]).long() + 3 * (b == alpha[3]).long()
        for c in mine:
            ident = id_lo + (id_hi - id_lo) * float(torch.rand(1, generator=gen, device=device))
            mut = torch.rand(length, generator=gen, device=device) < (1.0 - ident)
            shift = torch.randint(1, 4, (length,), generator=gen, device=device)
            view[k] = torch.where(mut, alpha[(cur + shift) % 4], b)
            names.append(f'NZ_FAM{f:02d}_{c:03d}.1')
            family.append(f)
            k += 1
    assert k == n_genomes_part
    return names, out, np.full(n_genomes_part, length, dtype=np.int32), np.array(family, dtype=np.int64)


def make_humanlike_device(seed, n_chrom, chrom_len, device, repeat_frac=0.45, n_decoys=8):
    """configs[1]'s target set: a repeat-rich "human-like" genome -- n_chrom chromosomes of random sequence in
    which `repeat_frac` of the bases are diverged copies of a few interspersed repeat families (a short high-copy family and
    longer low-copy ones, 5-20 % divergence per copy) -- followed by n_decoys plasmid-like decoys of 50-200 kb.
    -> (names, uint8 tensor of the concatenated ASCII, int32 lens, kind) with kind[i] = 0 human-like, 1 decoy."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    alpha = torch.tensor(list(b'ACGT'), dtype=torch.uint8, device=device)
    fams = [(300, 0.5), (1200, 0.2), (3000, 0.15), (6000, 0.15)]            # (length, share of the repeat bases)
    cons = [torch.randint(0, 4, (L,), generator=gen, device=device) for L, _ in fams]
    dec_lens = [int(x) for x in torch.randint(50000, 200001, (n_decoys,), generator=gen, device=device).cpu()]
    lens = [chrom_len] * n_chrom + dec_lens
    out = torch.empty(sum(lens), dtype=torch.uint8, device=device)
    off = 0
    for c in range(n_chrom):
        code = torch.randint(0, 4, (chrom_len,), generator=gen, device=device)
        for (L, share), con in zip(fams, cons):
            n_copies = int(repeat_frac * share * chrom_len / L)
            for k0 in range(0, n_copies, 20000):                             # bounded temporaries
                k1 = min(n_copies, k0 + 20000)
                div = torch.rand(k1 - k0, 1, generator=gen, device=device) * 0.15 + 0.05
                copy = con.unsqueeze(0).expand(k1 - k0, L)
                mut = torch.rand(k1 - k0, L, generator=gen, device=device) < div
                copy = torch.where(mut, (copy + torch.randint(1, 4, (k1 - k0, L), generator=gen, device=device)) % 4, copy)
                start = torch.randint(0, chrom_len - L, (k1 - k0, 1), generator=gen, device=device)
                code.scatter_(0, (start + torch.arange(L, device=device)).reshape(-1), copy.reshape(-1))
        out[off:off + chrom_len] = alpha[code]
        off += chrom_len
        del code
    for L in dec_lens:
        out[off:off + L] = alpha[torch.randint(0, 4, (L,), generator=gen, device=device)]
        off += L
    names = [f'chrH{c + 1:02d}' for c in range(n_chrom)] + [f'decoy{d:02d}' for d in range(n_decoys)]
    kind = np.array([0] * n_chrom + [1] * n_decoys, dtype=np.int32)
    return names, out, np.array(lens, dtype=np.int32), kind


def make_reads_from_targets_device(seed, flat, lens, n_reads, weights, device, **kw):
    """Like make_reads_device for targets of unequal lengths: reads are sampled from target t with probability weights[t] at a
    uniform start (the sampled length is clipped to the target).  -> (buf, offsets, lengths, truth)."""
    import torch
    lens = np.asarray(lens, dtype=np.int64)
    offs = np.zeros(len(lens), dtype=np.int64)
    offs[1:] = np.cumsum(lens[:-1])
    # reuse the equal-length sampler on a virtual genome length: sample target and start here, then gather per read
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    mean_len, min_len = kw.get('mean_len', 8000), kw.get('min_len', 200)
    torch.manual_seed(int(seed) * 2654435761 % (1 << 31) + 17)
    gamma = torch.distributions.Gamma(torch.tensor(1.6, device=device), torch.tensor(1.6 / mean_len, device=device))
    L = gamma.sample((n_reads,)).clamp_(min=min_len).long()
    w = torch.as_tensor(np.asarray(weights, dtype=np.float64) / float(np.sum(weights)), device=device, dtype=torch.float32)
    ti = torch.multinomial(w, n_reads, replacement=True, generator=gen)
    tl = torch.as_tensor(lens, device=device)[ti]
    L = torch.minimum(L, tl)
    start = (torch.rand(n_reads, generator=gen, device=device, dtype=torch.float64) * (tl - L + 1).double()).long()
    g0 = torch.as_tensor(offs, device=device)[ti] + start
    return _reads_from_spans(gen, flat, g0, L, n_reads, device, dict(genome=ti, start=start, end=start + L), **kw)


def _reads_from_spans(gen, genomes_flat, g0_all, L, n_reads, device, truth_t, sub=0.04, ins=0.03, dele=0.05, chunk_reads=8192, **_):
    """ONT-like errors on the spans [g0, g0 + L) of the concatenated targets (strand chosen here)."""
    import torch
    rev = torch.rand(n_reads, generator=gen, device=device) < 0.5
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b'ACGT', b'TGCA'):
        comp[a] = b
    alpha = torch.tensor(list(b'ACGT'), dtype=torch.uint8, device=device)
    code_of = torch.zeros(256, dtype=torch.long, device=device)
    for k, a in enumerate(b'ACGT'):
        code_of[a] = k
    pieces, out_lens = [], []
    for r0 in range(0, n_reads, chunk_reads):
        r1 = min(n_reads, r0 + chunk_reads)
        Lc = L[r0:r1]
        off = torch.cumsum(Lc, 0) - Lc
        T = int(Lc.sum())
        rid = torch.repeat_interleave(torch.arange(r1 - r0, device=device), Lc)
        p = torch.arange(T, device=device) - off[rid]
        g0 = g0_all[r0:r1]
        rv = rev[r0:r1][rid]
        src = torch.where(rv, (g0 + Lc - 1)[rid] - p, g0[rid] + p)
        frag = genomes_flat[src]
        frag = torch.where(rv, comp[frag.long()], frag)
        del src, p
        # homopolymer-biased indels (see ont_errors): a base that repeats its predecessor is deleted / followed by an insertion
        # HP_IN / HP_OUT times as often as another one, and what is inserted after it is a further copy of it
        hp = torch.zeros(T, dtype=torch.bool, device=device)
        if T > 1:
            hp[1:] = frag[1:] == frag[:-1]
        scale = torch.where(hp, HP_IN, HP_OUT)
        r = torch.rand(T, generator=gen, device=device)
        keep = r >= dele * scale
        subm = keep & (r < dele * scale + sub)
        n_sub = int(subm.sum())
        frag[subm] = alpha[(code_of[frag[subm].long()] + torch.randint(1, 4, (n_sub,), generator=gen, device=device)) % 4]
        n_ins = torch.where(torch.rand(T, generator=gen, device=device) < ins * scale,
                            torch.empty(T, device=device).geometric_(0.6, generator=gen).long(), torch.zeros((), dtype=torch.long, device=device))
        counts = keep.long() + torch.where(keep, n_ins, torch.zeros((), dtype=torch.long, device=device))
        out = torch.repeat_interleave(frag, counts)
        cstart = torch.cumsum(counts, 0) - counts
        first = torch.zeros(out.numel(), dtype=torch.bool, device=device)
        first[cstart[counts > 0]] = True
        rnd_ins = ~first & ~torch.repeat_interleave(hp, counts)
        n_insd = int(rnd_ins.sum())
        out[rnd_ins] = alpha[torch.randint(0, 4, (n_insd,), generator=gen, device=device)]
        del hp, scale, rnd_ins
        csum = torch.cumsum(counts, 0)
        tot_at_end = csum[off + Lc - 1]
        new_len = tot_at_end - torch.cat([torch.zeros(1, dtype=torch.long, device=device), tot_at_end[:-1]])
        pieces.append(out)
        out_lens.append(new_len)
        del rid, rv, frag, r, keep, subm, n_ins, counts, cstart, first, csum
    lens_o = torch.cat(out_lens)
    buf = torch.cat(pieces + [torch.full((16,), ord('A'), dtype=torch.uint8, device=device)])
    offs_o = torch.cumsum(lens_o, 0) - lens_o
    truth = {k: v.cpu().numpy() for k, v in truth_t.items()}
    truth['rev'] = rev.cpu().numpy()
    return buf, offs_o.contiguous(), lens_o.to(torch.int32).contiguous(), truth


def make_reads_device(seed, genomes_flat, genome_len, n_reads, weights, device, mean_len=8000, min_len=200,
                      sub=0.04, ins=0.03, dele=0.05, chunk_reads=8192, return_truth=False):
    """ONT-like reads sampled from device-resident genomes.
    -> (uint8 tensor of the concatenated reads (padded by 16 bytes), int64 offsets, int32 lengths) on `device`;
    with return_truth also the origin of every read as host arrays: dict(genome, start, end, rev)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    w = torch.as_tensor(np.asarray(weights, dtype=np.float64) / float(np.sum(weights)), device=device, dtype=torch.float32)
    gamma = torch.distributions.Gamma(torch.tensor(1.6, device=device), torch.tensor(1.6 / mean_len, device=device))
    # torch.distributions draws from the default generator; a seed of its own, or the lengths correlate with the genome
    # choice below (identical Philox streams: the shortest reads would all come from the first community member)
    torch.manual_seed(int(seed) * 2654435761 % (1 << 31) + 17)
    L = gamma.sample((n_reads,)).clamp_(min=min_len, max=genome_len).long()
    gi = torch.multinomial(w, n_reads, replacement=True, generator=gen)
    start = (torch.rand(n_reads, generator=gen, device=device, dtype=torch.float64) * (genome_len - L + 1).double()).long()
    rev = torch.rand(n_reads, generator=gen, device=device) < 0.5
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip(b'ACGT', b'TGCA'):
        comp[a] = b
    alpha = torch.tensor(list(b'ACGT'), dtype=torch.uint8, device=device)
    code_of = torch.zeros(256, dtype=torch.long, device=device)
    for k, a in enumerate(b'ACGT'):
        code_of[a] = k
    pieces, out_lens = [], []
    for r0 in range(0, n_reads, chunk_reads):
        r1 = min(n_reads, r0 + chunk_reads)
        Lc = L[r0:r1]
        off = torch.cumsum(Lc, 0) - Lc
        T = int(Lc.sum())
        rid = torch.repeat_interleave(torch.arange(r1 - r0, device=device), Lc)
        p = torch.arange(T, device=device) - off[rid]
        g0 = gi[r0:r1] * genome_len + start[r0:r1]
        rv = rev[r0:r1][rid]
        src = torch.where(rv, (g0 + Lc - 1)[rid] - p, g0[rid] + p)
        frag = genomes_flat[src]
        frag = torch.where(rv, comp[frag.long()], frag)
        del src, p
        # homopolymer-biased indels (see ont_errors): a base that repeats its predecessor is deleted / followed by an insertion
        # HP_IN / HP_OUT times as often as another one, and what is inserted after it is a further copy of it
        hp = torch.zeros(T, dtype=torch.bool, device=device)
        if T > 1:
            hp[1:] = frag[1:] == frag[:-1]
        scale = torch.where(hp, HP_IN, HP_OUT)
        r = torch.rand(T, generator=gen, device=device)
        keep = r >= dele * scale
        subm = keep & (r < dele * scale + sub)
        n_sub = int(subm.sum())
        frag[subm] = alpha[(code_of[frag[subm].long()] + torch.randint(1, 4, (n_sub,), generator=gen, device=device)) % 4]
        n_ins = torch.where(torch.rand(T, generator=gen, device=device) < ins * scale,
                            torch.empty(T, device=device).geometric_(0.6, generator=gen).long(), torch.zeros((), dtype=torch.long, device=device))
        counts = keep.long() + torch.where(keep, n_ins, torch.zeros((), dtype=torch.long, device=device))
        out = torch.repeat_interleave(frag, counts)
        cstart = torch.cumsum(counts, 0) - counts
        first = torch.zeros(out.numel(), dtype=torch.bool, device=device)
        first[cstart[counts > 0]] = True
        rnd_ins = ~first & ~torch.repeat_interleave(hp, counts)
        n_insd = int(rnd_ins.sum())
        out[rnd_ins] = alpha[torch.randint(0, 4, (n_insd,), generator=gen, device=device)]
        del hp, scale, rnd_ins
        # new length of every read: sum of its counts
        csum = torch.cumsum(counts, 0)
        ends = off + Lc - 1
        tot_at_end = csum[ends]
        new_len = tot_at_end - torch.cat([torch.zeros(1, dtype=torch.long, device=device), tot_at_end[:-1]])
        pieces.append(out)
        out_lens.append(new_len)
        del rid, rv, frag, r, keep, subm, n_ins, counts, cstart, first, csum
    lens = torch.cat(out_lens)
    buf = torch.cat(pieces + [torch.full((16,), ord('A'), dtype=torch.uint8, device=device)])
    offs = torch.cumsum(lens, 0) - lens
    if return_truth:
        truth = dict(genome=gi.cpu().numpy(), start=start.cpu().numpy(), end=(start + L).cpu().numpy(), rev=rev.cpu().numpy())
        return buf, offs.contiguous(), lens.to(torch.int32).contiguous(), truth
    return buf, offs.contiguous(), lens.to(torch.int32).contiguous()
