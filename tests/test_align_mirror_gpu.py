"""GPU tests (-m gpu) of the Python mirror of the reference's Align() (bin/lib/aligner.py:93-342) and of the
`--aligner` executable drop-in: argument handling, DataFrame schema, taxonomy join, tiebreakers, PAF side file."""
import gzip
import hashlib
import os
import random
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from map_cases import small_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeMetadata:
    """The three joins Align() uses (assembly_metadata.py:33-66), over an in-memory table."""

    def __init__(self, table):
        self.t = table

    def get_assembly_path(self, *, assembly_list, how='inner'):
        return assembly_list.merge(self.t[['assembly_id', 'path']].drop_duplicates(), on='assembly_id', how=how)

    def get_assembly_length(self, *, assembly_list, how='inner'):
        return assembly_list.merge(self.t[['assembly_id', 'assembly_length']].drop_duplicates(), on='assembly_id', how=how)

    def get_sequence_tax_id(self, *, assembly_list, how='inner'):
        return assembly_list.merge(self.t[['assembly_id', 'tax_id', 'species_tax_id', 'genus_tax_id', 'sequence_id']],
                                   on='assembly_id', how=how)


@pytest.fixture(scope='module')
def files(tmp_path_factory, libmpn, oracle_built):
    d = tmp_path_factory.mktemp('align')
    gen, reads = small_world(seed=11, n_genomes=4, glen=80000, n_reads=25, mean_len=2500)
    rows = []
    for i, (name, seq) in enumerate(gen):
        p = d / f'asm{i}.fna.gz'
        with gzip.open(p, 'wb') as f:
            f.write(b'>' + name.encode() + b' some description\n')
            s = bytes(seq)
            for o in range(0, len(s), 80):
                f.write(s[o:o + 80] + b'\n')
        rows.append(dict(assembly_id=f'GCF_{i:09d}.1', path=p.name, assembly_length=len(seq), tax_id=1000 + i,
                         species_tax_id=500 + i, genus_tax_id=50, sequence_id=name))
    fq = d / 'reads.fq'
    with open(fq, 'wb') as f:
        for r in reads:
            f.write(b'@' + r['name'].encode() + b' extra\n' + bytes(r['seq']) + b'\n+\n' + b'I' * len(r['seq']) + b'\n')
    return d, gen, reads, pd.DataFrame(rows), fq


def test_align_mirror_schema_join_and_tiebreak(files):
    from megapath_nano_amd.aligner import Align
    from oracle import mm2_bindings as mb
    d, gen, reads, table, fq = files
    meta = FakeMetadata(table)
    target = table[['assembly_id']].iloc[:3].copy()  # the 4th genome is indexed? no: only 3 assemblies are targets
    opts = dict(assembly_folder=str(d), min_alignment_score=0, debug=False)
    out = Align(assembly_metadata=meta, global_options=opts, temp_dir_name=str(d), log_file=None,
                query_filename_list=pd.DataFrame({'path': [str(fq)]}), target_assembly_list=target,
                aligner_options=['-t', '4', '-I', '1G', '-N', '50', '-p', '1', '-x', 'map-ont', '--split-prefix', 'tmp'],
                paf_path_and_prefix=str(d / 'out.species'))
    assert list(out.columns) == ['read_id', 'read_length', 'read_from', 'read_to', 'strand', 'sequence_id', 'sequence_length',
                                 'sequence_from', 'sequence_to', 'match', 'alignment_block_length', 'mapq', 'edit_dist',
                                 'alignment_score', 'assembly_id', 'tax_id', 'species_tax_id', 'genus_tax_id',
                                 'alignment_score_tiebreaker']
    assert out['alignment_score'].dtype == np.int64 and out['alignment_score_tiebreaker'].dtype == np.float64
    # tiebreakers: Python's random seeded with md5 of the query basename (aligner.py:160-168), SURVEY section 8c values
    assert list(out['alignment_score_tiebreaker'][:3]) == [0.2507641374631844, 0.3100258391581491, 0.8083515623068414]
    rnd = random.Random()
    rnd.seed(hashlib.md5(b'reads.fq').hexdigest())
    assert list(out['alignment_score_tiebreaker']) == [rnd.random() for _ in range(len(out))]
    # rows == oracle PAF through the reference's awk projection ($1..$13,$15 with NM:i:/AS:i: stripped)
    oidx = mb.Index(gen[:3])
    oopt = mb.default_opt(best_n=50, pri_ratio=1.0)
    sp = mb.SplitIndex([oidx])  # the call passes --split-prefix: minimap2's merge path (oracle: mmo_map_read_split)
    want = []
    for r in reads:
        for line in sp.map_read(oopt, r['name'], r['seq']).splitlines():
            f = line.split('\t')
            want.append((f[0], int(f[1]), int(f[2]), int(f[3]), f[4], f[5], int(f[6]), int(f[7]), int(f[8]), int(f[9]),
                         int(f[10]), int(f[11]), int(f[12][5:]), int(f[14][5:])))
    got = [tuple(x) for x in out[list(out.columns[:14])].itertuples(index=False, name=None)]
    assert got == want
    assert (out['assembly_id'] == out['sequence_id'].map(dict(zip(table['sequence_id'], table['assembly_id'])))).all()
    # side file: PAF text identical to the oracle's
    paf = open(d / 'out.species.paf').read()
    assert paf == ''.join(sp.map_read(oopt, r['name'], r['seq']) for r in reads)
    # ... and <prefix>.sam (the reference's -a run, aligner.py:183-184,219-227): header + the oracle's records
    sam = open(d / 'out.species.sam').read().splitlines(keepends=True)
    hdr = [l for l in sam if l.startswith('@')]
    assert len(hdr) == len(target) + 1 and hdr[-1].startswith('@PG')
    assert ''.join(sam[len(hdr):]) == ''.join(sp.map_read(oopt, r['name'], r['seq'], sam=True, qual=b'I' * len(r['seq'])) for r in reads)
    # ... and <prefix>.bam / .bam.bai: primary + supplementary records only (-F 1796), coordinate sorted, indexed
    from bam_reader import read_bai, read_bam
    b = read_bam(str(d / 'out.species.bam'))
    kept = [l.split('\t') for l in sam[len(hdr):] if not int(l.split('\t')[1]) & 1796]
    assert len(b['records']) == len(kept) > 0 and b['text'].startswith('@HD\tVN:1.6\tSO:coordinate\n')
    order = [(r['tid'], r['pos']) for r in b['records']]
    assert order == sorted(order) and sorted(r['name'] for r in b['records']) == sorted(f[0] for f in kept)
    refs, _ = read_bai(str(d / 'out.species.bam.bai'))
    assert len(refs) == len(target)
    sp.close()
    oidx.close()


def test_align_mirror_argument_errors(files):
    from megapath_nano_amd.aligner import Align
    d, gen, reads, table, fq = files
    meta = FakeMetadata(table)
    opts = dict(assembly_folder=str(d), min_alignment_score=0, debug=False)
    with pytest.raises(SystemExit) as e:
        Align(assembly_metadata=meta, global_options=opts, temp_dir_name=str(d), log_file=None,
              query_filename_list=pd.DataFrame({'path': [str(fq)]}), aligner_options=[])
    assert 'Exactly one of target_filename_list and target_assembly_list' in str(e.value)
    with pytest.raises(SystemExit) as e:
        Align(assembly_metadata=meta, global_options=opts, temp_dir_name=str(d), log_file=None,
              target_assembly_list=table[['assembly_id']], aligner_options=[])
    assert 'Exactly one of query_filename_list and query_assembly_list' in str(e.value)
    with pytest.raises(SystemExit) as e:
        Align(assembly_metadata=meta, global_options=opts, temp_dir_name=str(d), log_file=None,
              query_filename_list=pd.DataFrame({'path': [str(d / 'missing.fq')]}), target_assembly_list=table[['assembly_id']],
              aligner_options=[])
    assert 'not exists' in str(e.value)


def test_aligner_executable_dropin(files):
    """aligner.py:187-206: argv[0] + minimap2 options + target + queries, PAF on stdout."""
    from oracle import mm2_bindings as mb
    d, gen, reads, table, fq = files
    target = d / 'all.fna'
    with open(target, 'wb') as f:
        for name, seq in gen:
            f.write(b'>' + name.encode() + b'\n' + bytes(seq) + b'\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bin', 'mpn-aligner'), '-c', '-t', '8', '-I', '0G', '-x', 'map-ont',
                          str(target), str(fq), '--split-prefix', 'tmp'], check=True, capture_output=True, text=True).stdout
    # `-I 0G` is what the reference passes on a host with less than 64 GiB of free memory (megapath_nano.py:4019-4022): minimap2
    # then reads min(mini-batch, 0) bases at a time, i.e. every target sequence becomes an index part of its own
    parts = [mb.Index([g]) for g in gen]
    oopt = mb.default_opt()
    sp = mb.SplitIndex(parts)
    assert out == ''.join(sp.map_read(oopt, r['name'], r['seq']) for r in reads)
    sp.close()
    for p_ in parts:
        p_.close()


def test_align_mirror_amplicon_module_keeps_secondaries_and_exits(files):
    """aligner.py:246-259: the amplicon filter's BAM is made with -F4 (secondary alignments stay) and Align() exits after it;
    with a PAF prefix the call carries -a, so the SAM has CIGARs even for mapping_only."""
    from megapath_nano_amd.aligner import Align
    from bam_reader import read_bam
    d, gen, reads, table, fq = files
    target = d / 'amplicon_ref.fna'
    with open(target, 'wb') as f:
        for name, seq in gen:
            f.write(b'>' + name.encode() + b'\n' + bytes(seq) + b'\n')
    opts = dict(assembly_folder=str(d), min_alignment_score=0, debug=False, nano_dir=str(d))
    with pytest.raises(SystemExit) as e:
        Align(assembly_metadata=FakeMetadata(table), global_options=opts, temp_dir_name=str(d), log_file=None,
              query_filename_list=pd.DataFrame({'path': [str(fq)]}), target_filename_list=pd.DataFrame({'path': [str(target)]}),
              aligner_options=['-t', '4', '-N', '50', '-p', '1', '-x', 'map-ont'], paf_path_and_prefix=str(d / 'amp'),
              mapping_only=True, module_option='amplicon_filter_module', align_concat_fa=True)
    assert e.value.code is None                                   # os.sys.exit() without a message
    sam = [l.split('\t') for l in open(d / 'amp.sam') if not l.startswith('@')]
    mapped = [f for f in sam if not int(f[1]) & 4]
    assert mapped and all(f[5] != '*' for f in mapped)            # -a implies CIGARs
    assert any(int(f[1]) & 256 for f in mapped)                   # the world has a 99 % strain pair: secondaries exist
    b = read_bam(str(d / 'amp.bam'))
    assert len(b['records']) == len(mapped) and any(r['flag'] & 256 for r in b['records'])
    assert os.path.exists(d / 'amp.bam.bai')
