"""GPU tests (-m gpu) shaped like BASELINE.json's configs.

C1  = configs[0]: 1000 synthetic ONT reads (8 kb mean) against a 5-genome mini-RefSeq: EVERY read against the oracle, then
      Align -> Reassign -> best alignment per read -> .read_count_by_name through the mirrors of the reference's own
      functions, against the pinned reassignment oracle fed with the mapping oracle's rows.
C2' = configs[1] at a stated reduced size: reads from a repeat-rich "human-like" genome, plasmid-like decoys and microbes
      against a human+decoy target handed over as a FIFO of .fna.gz members (the reference's own call,
      bin/megapath_nano.py:1116-1126), `-x map-ont` defaults (-N 5 -p 0.8), then the human / decoy classification.
"""
import gzip
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pandas as pd
import pytest

import paf_check
from test_align_mirror_gpu import FakeMetadata
from test_fastx import feed_fifo

pytestmark = pytest.mark.gpu


def write_assemblies(d, gen, species_of=None, prefix='GCF'):
    rows = []
    for i, (name, seq) in enumerate(gen):
        p = d / f'{prefix}_{i:05d}.fna.gz'
        with gzip.open(p, 'wb', compresslevel=1) as f:
            f.write(b'>' + name.encode() + b' synthetic\n')
            b = bytes(seq)
            for k in range(0, len(b), 80):
                f.write(b[k:k + 80] + b'\n')
        rows.append(dict(assembly_id=f'{prefix}_{i:09d}.1', path=p.name, assembly_length=len(seq), tax_id=10000 + i,
                         species_tax_id=(species_of[i] if species_of else 5000 + i), genus_tax_id=77, sequence_id=name))
    return pd.DataFrame(rows)


def write_fastq(path, reads):
    with open(path, 'wb') as f:
        for r in reads:
            f.write(b'@' + r['name'].encode() + b'\n' + bytes(r['seq']) + b'\n+\n' + b'5' * len(r['seq']) + b'\n')


def test_c1_every_read_then_align_reassign_counts(libmpn, oracle_built, tmp_path, monkeypatch):
    from megapath_nano_amd import mapper, report, synth
    from megapath_nano_amd.aligner import Align
    from megapath_nano_amd.reassignment import Reassign
    from oracle import mm2_bindings as mb, reassign_oracle as ro
    gen = synth.make_genomes(20240901, 5, 5_000_000, strain_pairs=1)       # 4 species + a 99 % strain of the first
    w = np.array([5, 3, 2, 1, 0.06], dtype=float)                          # the strain is rare: its few unique reads are 'explained'
    reads = synth.make_reads(1, gen, 1000, mean_len=8000, weights=w)
    names, seqs = [r['name'] for r in reads], [r['seq'] for r in reads]
    # (1) every read against the oracle, -N 50 -p 1 (megapath_nano.py:1270)
    gidx, oidx = mapper.Index(gen), mb.Index(gen)
    gopt, oopt = mapper.default_opt(best_n=50, pri_ratio=1.0), mb.default_opt(best_n=50, pri_ratio=1.0)
    oopt.mid_occ = oidx.mid_occ()
    sp = mb.SplitIndex([oidx])
    got = mapper.map_batch(gidx, gopt, names, seqs)
    with ThreadPoolExecutor(16) as ex:
        want = list(ex.map(lambda r: mb.map_read(oidx, oopt, r['name'], r['seq'])[2], reads))
        want_split = list(ex.map(lambda r: sp.map_read(oopt, r['name'], r['seq']), reads))
    by = {}
    for line in got.splitlines(keepends=True):
        by.setdefault(line.split('\t', 1)[0], []).append(line)
    for r, w_ in zip(reads, want):
        assert ''.join(by.get(r['name'], [])) == w_, r['name']
    assert sum(1 for w_ in want if w_) >= 990
    rd = {n_: bytes(s_).decode() for n_, s_ in zip(names, seqs)}
    gd = {n_: bytes(s_).decode() for n_, s_ in gen}
    st = paf_check.check_paf(got, rd, gd, best_n=50)
    assert st['as_equal'] >= 0.98 * st['lines']
    gidx.close()
    # (2) the stage through the mirrors: Align (FASTA files of the assemblies, --split-prefix) -> Reassign -> consumers
    d = tmp_path
    species = [501, 502, 503, 504, 505]
    table = write_assemblies(d, gen, species_of=species)
    write_fastq(d / 'reads.fq', reads)
    (d / 'db').mkdir()
    descr = ['Escherichia coli K-12', 'Bacillus subtilis 168', 'Listeria monocytogenes EGD', 'Pseudomonas sp. XYZ 12', 'Escherichia albertii KF1']
    with open(d / 'db' / 'sequence_name', 'w') as f:
        for (nm, _), ds in zip(gen, descr):
            f.write(f'{nm}\t{ds}\n')
    monkeypatch.chdir(d)
    meta = FakeMetadata(table)
    opts = dict(assembly_folder=str(d), min_alignment_score=0, debug=False, db_folder=str(d / 'db'))
    al = Align(assembly_metadata=meta, global_options=opts, temp_dir_name=str(d), log_file=None,
               query_filename_list=pd.DataFrame({'path': [str(d / 'reads.fq')]}), target_assembly_list=table[['assembly_id']],
               aligner_options=['-t', '16', '-I', '4G', '-N', '50', '-p', '1', '-x', 'map-ont', '--split-prefix', 'tmp'],
               paf_path_and_prefix=str(d / 'c1.species'))
    assert open(d / 'c1.species.paf').read() == ''.join(want_split)
    assert os.path.getsize(d / 'c1.species.bam') > 0 and os.path.getsize(d / 'c1.species.bam.bai') > 0
    stats = {}
    out = Reassign(al, str(d / 'db'), stats=stats)
    best = report.best_align_list(out)
    counts = report.read_count_by_name(best, reassignment=True)
    report.write_read_count_by_name(counts, str(d / 'c1.read_count_by_name'))
    # expected: the mapping oracle's rows through the awk projection, then the pinned reassignment oracle
    cols = {c: [] for c in ('read_id', 'sequence_id', 'alignment_score', 'alignment_score_tiebreaker', 'sequence_from', 'sequence_to',
                            'species_tax_id')}
    sp_of = dict(zip([g[0] for g in gen], species))
    for text in want_split:
        for line in text.splitlines():
            f = line.split('\t')
            cols['read_id'].append(f[0]); cols['sequence_id'].append(f[5]); cols['alignment_score'].append(int(f[14][5:]))
            cols['sequence_from'].append(int(f[7])); cols['sequence_to'].append(int(f[8])); cols['species_tax_id'].append(sp_of[f[5]])
    assert list(al['read_id']) == cols['read_id'] and list(al['alignment_score']) == cols['alignment_score']
    cols['alignment_score_tiebreaker'] = list(al['alignment_score_tiebreaker'])
    exp = ro.reassign_oracle(cols, [(g[0], ds) for g, ds in zip(gen, descr)])
    assert stats['explains'] == exp['explains'] and stats['explains'], 'the strain pair must trigger an explains relation'
    assert stats['read_count_by_name'] == ro.read_count_by_name(cols, exp['rows']) == counts.to_dict()
    assert stats['aligned_bp_by_species'] == ro.aligned_bp_by_species(cols, exp['rows'])
    lines = open(d / 'c1.read_count_by_name').read().splitlines()
    assert lines[0] == 'name\tread_id' and sum(int(l.split('\t')[1]) for l in lines[1:]) == len(set(cols['read_id']))
    # without --reassignment the names come from db/sequence_name at species level; an unknown sequence keeps its id
    plain = report.read_count_by_name(report.best_align_list(al), db_folder=str(d / 'db'), reassignment=False)
    assert set(plain.index) <= {'Escherichia coli', 'Bacillus subtilis', 'Listeria monocytogenes', 'Pseudomonas sp. XYZ', 'Escherichia albertii'}
    assert int(plain.sum()) == len(set(cols['read_id'])) and list(plain.values) == sorted(plain.values, reverse=True)
    sp.close()
    oidx.close()


def humanlike(rng, length, synth):
    """Random sequence in which ~45 % is covered by copies of a few interspersed repeat families at 80-95 % identity."""
    g = synth.random_genome(rng, length, gc=0.41)
    fams = [synth.random_genome(rng, int(L)) for L in (300, 300, 1200, 6000, 2500)]
    covered = 0
    while covered < 0.45 * length:
        fam = fams[int(rng.integers(0, len(fams)))]
        cut = int(rng.integers(len(fam) // 3, len(fam) + 1))
        piece = synth.mutate_strain(rng, fam[:cut], float(rng.uniform(0.80, 0.95)))
        s = int(rng.integers(0, length - cut))
        g[s:s + cut] = piece
        covered += cut
    return g


def test_c2_shaped_human_decoy_stage_through_fifo(libmpn, oracle_built, tmp_path):
    """configs[1] reduced: 12 Mbp human-like + 8 plasmid decoys (50-200 kb) as targets, 1 500 reads (600 human, 200 decoy,
    700 from 3 microbes that are NOT in the target set).  Sizes stated here because the full config (3.1 Gbp, 100k reads)
    is bench-sized, not test-sized."""
    from megapath_nano_amd import synth
    from megapath_nano_amd.aligner import AlignerOptions, map_files, _frame_of
    from megapath_nano_amd.filters import human_and_decoy_classify
    from megapath_nano_amd.pipeline import random_block
    import random
    rng = np.random.default_rng(2024)
    human = [('chrH1', humanlike(rng, 8_000_000, synth)), ('chrH2', humanlike(rng, 4_000_000, synth))]
    decoys = [(f'plasmid{i}', synth.random_genome(rng, int(rng.integers(50_000, 200_000)))) for i in range(8)]
    microbes = synth.make_genomes(9, 3, 1_000_000, strain_pairs=0)
    r_h = synth.make_reads(31, human, 600, mean_len=8000)
    r_d = synth.make_reads(32, decoys, 200, mean_len=4000)
    r_m = synth.make_reads(33, microbes, 700, mean_len=8000)
    reads = []
    for tag, lst in (('hum', r_h), ('dec', r_d), ('mic', r_m)):
        for r in lst:
            r['name'] = f"{tag}_{r['name']}"
            reads.append(r)
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    d = tmp_path
    t_h = write_assemblies(d, human, prefix='HUM')
    t_d = write_assemblies(d, decoys, prefix='DEC')
    write_fastq(d / 'reads.fq', reads)
    pipe = str(d / 'temp_pipe_target_fasta')
    os.mkfifo(pipe)
    files = [str(d / p) for p in list(t_h['path']) + list(t_d['path'])]
    writer = feed_fifo(pipe, files, chunk=1 << 20)
    options = AlignerOptions(['-t', '16', '-I', '4G', '-x', 'map-ont', '--split-prefix', 'tmp'], False)  # megapath_nano.py:1124
    batches, _ = map_files([pipe], [str(d / 'reads.fq')], options, want_paf=True, want_sam=False, want_cols=True)
    writer.join(30)
    paf = ''.join(b.paf for b in batches)
    rd = {r['name']: bytes(r['seq']).decode() for r in reads}
    gd = {n_: bytes(s_).decode() for n_, s_ in human + decoys}
    st = paf_check.check_paf(paf, rd, gd, best_n=5)
    assert st['as_equal'] >= 0.97 * st['lines']
    al = pd.concat([_frame_of(b) for b in batches], ignore_index=True)
    to_asm = dict(zip(pd.concat([t_h, t_d])['sequence_id'], pd.concat([t_h, t_d])['assembly_id']))
    al['assembly_id'] = al['sequence_id'].map(to_asm)
    al['alignment_score_tiebreaker'] = random_block(random.Random(3), len(al))
    read_ids = pd.DataFrame({'read_id': [r['name'] for r in reads], 'read_length': [len(r['seq']) for r in reads]})
    out = human_and_decoy_classify(al, t_h[['assembly_id']], t_d[['assembly_id']], read_ids)
    hum, dec, mic = set(out['human_read_id_list']['read_id']), set(out['decoy_read_id_list']['read_id']), set(out['microbe_read_id_list']['read_id'])
    assert hum.isdisjoint(dec) and hum.isdisjoint(mic) and dec.isdisjoint(mic) and len(hum | dec | mic) == len(reads)
    long_h = [r['name'] for r in reads if r['name'].startswith('hum_') and len(r['seq']) >= 1500]
    long_d = [r['name'] for r in reads if r['name'].startswith('dec_') and len(r['seq']) >= 1500]
    assert sum(n_ in hum for n_ in long_h) >= 0.97 * len(long_h), 'human-derived reads must be classified human'
    assert sum(n_ in dec for n_ in long_d) >= 0.97 * len(long_d), 'decoy-derived reads must be classified decoy'
    mics = [r['name'] for r in reads if r['name'].startswith('mic_')]
    assert sum(n_ in mic for n_ in mics) >= 0.99 * len(mics), 'reads of organisms outside the target set stay microbe'
    # a sample of the reads against the oracle (same targets, the merge path of --split-prefix)
    from oracle import mm2_bindings as mb
    oidx = mb.Index(human + decoys)
    sp = mb.SplitIndex([oidx])
    oopt = mb.default_opt()
    by = {}
    for line in paf.splitlines(keepends=True):
        by.setdefault(line.split('\t', 1)[0], []).append(line)
    for i in rng.choice(len(reads), size=60, replace=False):
        r = reads[int(i)]
        assert ''.join(by.get(r['name'], [])) == sp.map_read(oopt, r['name'], r['seq']), r['name']
    sp.close()
    oidx.close()
