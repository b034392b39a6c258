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
