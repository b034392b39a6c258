"""Seeded genomes + reads for the mapper parity tests (data only)."""
import numpy as np

from megapath_nano_amd import synth


def small_world(seed=1, n_genomes=5, glen=120000, n_reads=40, mean_len=3000):
    gen = synth.make_genomes(seed, n_genomes, glen, strain_pairs=1)
    reads = synth.make_reads(seed + 1, gen, n_reads, mean_len=mean_len, random_frac=0.05)
    rng = np.random.default_rng(seed + 2)
    # edge cases: very short read, read with N run, low-complexity read, exact copy
    reads.append(dict(name='short', seq=gen[0][1][100:130].copy(), genome=0, start=100, end=130, strand='+'))
    withn = gen[1][1][5000:7000].copy()
    withn[700:760] = ord('N')
    reads.append(dict(name='with_n', seq=withn, genome=1, start=5000, end=7000, strand='+'))
    reads.append(dict(name='polyA', seq=np.full(500, ord('A'), dtype=np.uint8), genome=-1, start=0, end=0, strand='+'))
    reads.append(dict(name='at_repeat', seq=np.frombuffer(b'AT' * 300, dtype=np.uint8).copy(), genome=-1, start=0, end=0, strand='+'))
    reads.append(dict(name='exact', seq=gen[2][1][30000:36000].copy(), genome=2, start=30000, end=36000, strand='+'))
    reads.append(dict(name='lower', seq=np.frombuffer(bytes(gen[2][1][40000:42000]).lower(), dtype=np.uint8).copy(), genome=2,
                      start=40000, end=42000, strand='+'))
    reads.append(dict(name='tiny', seq=gen[0][1][0:5].copy(), genome=0, start=0, end=5, strand='+'))
    return gen, reads


def hard_reads(gen, seed=9):
    """Reads that force the rarely taken branches: z-drop inside a gap fill (second exact pass + split hit), chimeras,
    inversions, a long deletion (long-join, very wide DP window), a long read, N runs."""
    rng = np.random.default_rng(seed)
    comp = synth.COMP
    g0, g1, g2 = gen[0][1], gen[1][1], gen[2][1]
    out = []

    def add(name, parts, err=True):
        seq = np.concatenate(parts)
        if err:
            seq = synth.ont_errors(rng, seq, 0.03, 0.02, 0.03)
        out.append(dict(name=name, seq=seq, genome=-1, start=0, end=0, strand='+'))

    junk = lambda n: synth.ALPHA[rng.integers(0, 4, size=n)]  # noqa: E731
    add('junk_mid', [g0[10000:13000], junk(600), g0[13600:17000]])
    add('junk_mid2', [g1[30000:32500], junk(900), g1[33400:36000]])
    add('junk_small', [g1[5000:7500], junk(250), g1[7750:10000]])
    add('chimera', [g0[40000:44000], g2[20000:23500]])
    add('chimera_rc', [g0[50000:53000], comp[g1[60000:63000][::-1]]])
    add('inversion', [g2[70000:73000], comp[g2[73000:74200][::-1]], g2[74200:77000]])
    add('deletion_2k', [g0[60000:64000], g0[66000:70000]])
    add('deletion_6k', [g1[70000:74000], g1[80000:84000]])
    add('insertion_1500', [g2[30000:33000], junk(1500), g2[33000:36000]])
    add('long_60k', [g0[20000:80000]])
    withn = g1[90000:96000].copy()
    withn[2000:2200] = ord('N')
    withn[4000] = ord('n')
    add('n_run', [withn])
    add('exact_long', [g2[1000:21000]], err=False)
    add('dup_tandem', [g0[5000:6500], g0[5000:6500], g0[5000:6500]])
    return out
