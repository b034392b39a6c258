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


def ont_errors(rng, seq, sub=0.04, ins=0.03, dele=0.05):
    """Per-base independent errors (vectorised): deletion drops the base, insertion adds random bases after it."""
    n = len(seq)
    r = rng.random(n)
    keep = r >= dele
    subm = (r >= dele) & (r < dele + sub)
    s = seq.copy()
    s[subm] = ALPHA[(np.searchsorted(ALPHA, s[subm]) + rng.integers(1, 4, size=int(subm.sum()))) % 4]
    n_ins = np.where(rng.random(n) < ins, rng.geometric(0.6, size=n), 0)
    n_ins[~keep] = 0
    counts = keep.astype(np.int64) + n_ins
    out = np.repeat(s, counts)
    # positions of inserted bases: all but the first copy of each kept base
    starts = np.cumsum(counts) - counts
    first = np.zeros(len(out), dtype=bool)
    first[starts[counts > 0]] = True
    ins_mask = ~first
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
