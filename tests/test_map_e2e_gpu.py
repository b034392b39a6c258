"""GPU parity tests (-m gpu): mpn_map_batch (seed-chain-extend + hit bookkeeping + PAF) against the CPU oracle
oracle/mm2_oracle.c, line by line.  Parity unpinned against a real minimap2 (not vendored by the reference);
bit-exact against the oracle on coordinates, CIGAR, scores, MAPQ and tags."""
import os
import subprocess
import sys

import numpy as np
import pytest

from map_cases import small_world
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def oracle_paf(oidx, oopt, reads):
    from oracle import mm2_bindings as mb
    out = []
    for r in reads:
        _, _, paf = mb.map_read(oidx, oopt, r['name'], r['seq'])
        out.append(paf)
    return out


def split_by_read(paf, names):
    by = {n: [] for n in names}
    for line in paf.splitlines():
        by[line.split('\t', 1)[0]].append(line)
    return ['\n'.join(by[n]) + ('\n' if by[n] else '') for n in names]


@pytest.fixture(scope='module')
def world(libmpn, oracle_built):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads = small_world(seed=3, n_genomes=6, glen=150000, n_reads=60, mean_len=4000)
    gidx = mapper.Index(gen)
    oidx = mb.Index(gen)
    yield gen, reads, gidx, oidx
    gidx.close()
    oidx.close()


@pytest.mark.parametrize('best_n,pri_ratio,with_cigar', [(5, 0.8, 1), (50, 1.0, 1), (5, 0.8, 0)])
def test_paf_matches_oracle(world, best_n, pri_ratio, with_cigar):
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    gopt = mapper.default_opt(best_n=best_n, pri_ratio=pri_ratio, with_cigar=with_cigar)
    oopt = mb.default_opt(best_n=best_n, pri_ratio=pri_ratio, with_cigar=with_cigar)
    names = [r['name'] for r in reads]
    got = split_by_read(mapper.map_batch(gidx, gopt, names, [r['seq'] for r in reads]), names)
    want = oracle_paf(oidx, oopt, reads)
    n_lines = 0
    for r, g, w in zip(reads, got, want):
        assert g == w, (r['name'], len(r['seq']))
        n_lines += w.count('\n')
    assert n_lines >= len(reads) // 2
    if with_cigar:
        # column contract relied on by the reference's awk (aligner.py:271-273): NM at 13, AS at 15
        for line in ''.join(want).splitlines():
            f = line.split('\t')
            assert f[12].startswith('NM:i:') and f[14].startswith('AS:i:')


def test_mapping_lands_on_truth(world):
    """size-independent sanity: the primary hit of a simulated read overlaps its true origin"""
    from megapath_nano_amd import mapper
    gen, reads, gidx, _ = world
    gopt = mapper.default_opt()
    names = [r['name'] for r in reads]
    paf = mapper.map_batch(gidx, gopt, names, [r['seq'] for r in reads])
    prim = {}
    for line in paf.splitlines():
        f = line.split('\t')
        if 'tp:A:P' in f and f[0] not in prim:
            prim[f[0]] = f
    ok = tot = 0
    name_to_idx = {g[0]: i for i, g in enumerate(gen)}
    for r in reads:
        if r['genome'] < 0 or len(r['seq']) < 500:
            continue
        tot += 1
        f = prim.get(r['name'])
        if f is None:
            continue
        gi = name_to_idx[f[5]]
        same = gi == r['genome'] or {gen[gi][0][:6], gen[r['genome']][0][:6]} == {'NZ_SYN', 'NZ_STR'}
        if same and int(f[7]) < r['end'] and int(f[8]) > r['start'] and f[4] == r['strand']:
            ok += 1
    assert ok >= tot * 0.9, (ok, tot)


def test_pipelined_workers_are_deterministic(world, monkeypatch):
    """The batch is cut into sub-batches that run through 1..4 worker threads (own stream + arena each):
    the PAF must not depend on the cut or on the number of workers, and must equal the unsplit result."""
    from megapath_nano_amd import mapper
    gen, reads, gidx, _ = world
    gopt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    names = [r['name'] for r in reads]
    seqs = [r['seq'] for r in reads]
    monkeypatch.setenv('MPN_SUB_BATCH_BP', '1000000000')
    monkeypatch.setenv('MPN_PIPE_WORKERS', '1')
    base = mapper.map_batch(gidx, gopt, names, seqs)
    assert base.count('\n') > len(reads) // 2
    for sb, w in (('20000', '1'), ('20000', '4'), ('7000', '3'), ('150000', '2')):
        monkeypatch.setenv('MPN_SUB_BATCH_BP', sb)
        monkeypatch.setenv('MPN_PIPE_WORKERS', w)
        for _ in range(2):
            assert mapper.map_batch(gidx, gopt, names, seqs) == base, (sb, w)


def test_hard_reads_match_oracle(world):
    """z-drop + second exact pass + split hits, chimeras, inversions, long-join windows, a 60 kb read, N runs."""
    from map_cases import hard_reads
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, _, gidx, oidx = world
    reads = hard_reads(gen)
    names = [r['name'] for r in reads]
    for best_n, pri in ((5, 0.8), (50, 1.0)):
        gopt = mapper.default_opt(best_n=best_n, pri_ratio=pri)
        oopt = mb.default_opt(best_n=best_n, pri_ratio=pri)
        got = split_by_read(mapper.map_batch(gidx, gopt, names, [r['seq'] for r in reads]), names)
        st = mapper.last_stats()
        want = oracle_paf(oidx, oopt, reads)
        for r, g, w in zip(reads, got, want):
            assert g == w, r['name']
        text = ''.join(want)
        assert 'zd:i:' in text, 'no split hit in the hard set'
        assert st['dp_rounds'] >= 2 and st['second_pass_jobs'] >= 1, st
        # the read with a 1.2 kb inverted segment: the z-drop test's inversion probe cuts the hit there (second pass with
        # zdrop_inv, the remainder marked split_inv) and mm_align1_inv aligns the segment on the opposite strand: a tp:A:I line
        # between the two pieces, on the same target, on the other strand
        inv = [l.split('\t') for l in got[names.index('inversion')].splitlines()]
        lines_i = [f for f in inv if 'tp:A:I' in f]
        assert len(lines_i) == 1 and len([f for f in inv if 'tp:A:P' in f]) == 2, inv
        fi, prim = lines_i[0], sorted((f for f in inv if 'tp:A:P' in f), key=lambda f: int(f[2]))
        assert fi[4] != prim[0][4] and fi[5] == prim[0][5] == prim[1][5]
        assert int(prim[0][3]) <= int(fi[2]) < int(fi[3]) <= int(prim[1][2]) and int(prim[0][8]) <= int(fi[7]) < int(fi[8]) <= int(prim[1][7])
        assert 1000 <= int(fi[3]) - int(fi[2]) <= 1300 and int(fi[11]) == 0 and 'cm:i:0' in fi
    assert text.count('\n') >= len(reads)


def test_index_save_load_round_trip(world, tmp_path):
    """A saved index loaded back (SURVEY 8f1 / minimap2 -d) has the same keys, positions and targets, maps the hard reads
    (N runs included) to the same PAF, and the `--aligner` drop-in accepts it as the target."""
    from map_cases import hard_reads
    from megapath_nano_amd import mapper
    gen, reads, gidx, _ = world
    path = tmp_path / 'world.mpi'
    gidx.save(str(path))
    assert mapper.Index.is_index_file(str(path))
    idx2 = mapper.Index.load(str(path))
    assert idx2.names == gidx.names and list(idx2.lens) == list(gidx.lens) and (idx2.k, idx2.w) == (gidx.k, gidx.w)
    for a, b in zip(gidx.export(), idx2.export()):
        assert np.array_equal(a, b)
    assert idx2.mid_occ() == gidx.mid_occ()
    hard = hard_reads(gen)
    opt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    names, seqs = [r['name'] for r in hard], [r['seq'] for r in hard]
    assert mapper.map_batch(idx2, opt, names, seqs) == mapper.map_batch(gidx, opt, names, seqs)
    with pytest.raises(Exception):
        bad = tmp_path / 'bad.mpi'
        bad.write_bytes(b'MPNIDX01' + b'\0' * 40)
        mapper.Index.load(str(bad))
    # B1: the executable with a saved index as the target
    fq = tmp_path / 'r.fa'
    with open(fq, 'w') as f:
        for r in hard[:4]:
            f.write(f">{r['name']}\n{bytes(r['seq']).decode()}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bin', 'mpn-aligner'), '-c', '-t', '4', '-I', '1G', '-N', '50', '-p', '1',
                          '-x', 'map-ont', str(path), str(fq), '--split-prefix', 'tmp'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    # --split-prefix: minimap2 takes its merge path even for one part (oracle: mmo_map_read_split)
    from oracle import mm2_bindings as mb
    oidx = mb.Index(gen)
    sp = mb.SplitIndex([oidx])
    oopt = mb.default_opt(best_n=50, pri_ratio=1.0)
    assert out.stdout == ''.join(sp.map_read(oopt, n_, s_) for n_, s_ in zip(names[:4], seqs[:4]))
    sp.close()
    oidx.close()


def test_dropin_with_fifo_target(world, tmp_path):
    """B1 as the reference really calls it for the human/decoy stage (aligner.py:143-144,187-217): the target is a FIFO
    that a writer fills with concatenated .fna.gz members in chunks; reads from a FASTQ; `--split-prefix tmp` trailing."""
    import gzip
    from test_fastx import feed_fifo
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    files = []
    for i, (name, seq) in enumerate(gen):
        fn = tmp_path / f'asm{i}.fna.gz'
        with gzip.open(fn, 'wb') as f:
            f.write(b'>' + name.encode() + b' synthetic assembly\n')
            b = bytes(seq)
            for k in range(0, len(b), 80):
                f.write(b[k:k + 80] + b'\n')
        files.append(str(fn))
    pipe = str(tmp_path / 'temp_pipe_target_fasta')
    os.mkfifo(pipe)
    sub = reads[:24]
    fq = tmp_path / 'reads.fq'
    with open(fq, 'w') as f:
        for r in sub:
            f.write(f"@{r['name']}\n{bytes(r['seq']).decode()}\n+\n{'I' * len(r['seq'])}\n")
    writer = feed_fifo(pipe, files, chunk=65536)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bin', 'mpn-aligner'), '-c', '-t', '4', '-I', '4G', '-x', 'map-ont', pipe, str(fq),
                          '--split-prefix', 'tmp'], capture_output=True, text=True, timeout=300)
    writer.join(10)
    assert out.returncode == 0, out.stderr[-2000:]
    oopt = mb.default_opt()  # -x map-ont defaults: -N 5 -p 0.8
    sp = mb.SplitIndex([oidx])  # --split-prefix: the merge path
    assert out.stdout == ''.join(sp.map_read(oopt, r['name'], r['seq']) for r in sub)
    sp.close()
    assert out.stdout.count('\n') >= len(sub) // 2


def test_split_index_parts_merge_matches_oracle(world, tmp_path):
    """minimap2 -I parts + --split-prefix (aligner.py:199): three index parts, hits merged per read.  The C-ABI accumulator
    (mpn_hits_*) against the oracle's restatement of mm_split_merge; then the same through the executable, which cuts the
    target into the same parts itself; then invariants against the one-part run."""
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    parts = [gen[:2], gen[2:4], gen[4:]]
    gparts, oparts = [mapper.Index(p) for p in parts], [mb.Index(p) for p in parts]
    sp = mb.SplitIndex(oparts)
    names, seqs = [r['name'] for r in reads], [r['seq'] for r in reads]
    quals = [bytes(33 + (7 * i + k) % 40 for k in range(len(s))) for i, s in enumerate(seqs)]
    packed = mapper.PackedReads(names, seqs, quals=quals)
    for best_n, pri in ((5, 0.8), (50, 1.0)):
        gopt = mapper.default_opt(best_n=best_n, pri_ratio=pri, out_sam=2)
        oopt = mb.default_opt(best_n=best_n, pri_ratio=pri)
        h = mapper.Hits(packed)
        for gp in gparts:
            h.add_part(gp, gopt)
        paf, sam, cols = h.finish(gopt, want_paf=True, want_cols=True)
        # all parts resident, one call (mpn_map_batch_parts: (sub-batch, part) pairs in one pipeline): the same accumulator
        h2 = mapper.Hits(packed)
        h2.add_parts(gparts, gopt)
        paf2, sam2, cols2 = h2.finish(gopt, want_paf=True, want_cols=True)
        h2.close()
        assert paf2 == paf and sam2 == sam and all(np.array_equal(cols[k], cols2[k]) for k in cols)
        tnames, tlens = h.targets()
        assert tnames == [g[0] for g in gen] and list(tlens) == [len(g[1]) for g in gen]
        want = [sp.map_read(oopt, n_, s_) for n_, s_ in zip(names, seqs)]
        assert split_by_read(paf, names) == want
        assert sam == ''.join(sp.map_read(oopt, n_, s_, sam=True, qual=q_) for n_, s_, q_ in zip(names, seqs, quals))
        lines = paf.splitlines()
        assert len(lines) == len(cols['rid']) and [tnames[i] for i in cols['rid']] == [l.split('\t')[5] for l in lines]
        if best_n == 50:
            # against the one-part index: the best hit of a read (first line) keeps target, strand, coordinates and CIGAR
            one = split_by_read(mapper.map_batch(gidx, mapper.default_opt(best_n=50, pri_ratio=1.0), names, seqs), names)
            same = 0
            for a, b in zip(one, want):
                if a and b:
                    fa, fb = a.splitlines()[0].split('\t'), b.splitlines()[0].split('\t')
                    if fa[14] != fb[14] or fa[11] == '0' or fb[11] == '0':
                        continue  # (equal-score hits on the strain pair, or on two copies of a planted repeat, may swap)
                    assert fa[:11] == fb[:11] and fa[-1] == fb[-1], (fa[0],)
                    same += 1
            assert same >= len(reads) // 2
        h.close()
    # the executable: MPN_IDX_MINI_BATCH makes every genome a mini-batch, -I 200K closes a part after two of them
    fa = tmp_path / 'targets.fa'
    with open(fa, 'w') as f:
        for name, seq in gen:
            f.write(f'>{name}\n{bytes(seq).decode()}\n')
    fq = tmp_path / 'reads.fq'
    with open(fq, 'w') as f:
        for n_, s_, q_ in zip(names, seqs, quals):
            f.write(f'@{n_}\n{bytes(s_).decode()}\n+\n{q_.decode()}\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPN_IDX_MINI_BATCH='100000')
    out = subprocess.run([sys.executable, os.path.join(root, 'bin', 'mpn-aligner'), '-c', '-a', '-t', '4', '-I', '200K', '-x', 'map-ont', str(fa),
                          str(fq), '--split-prefix', 'tmp'], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    body = ''.join(l for l in out.stdout.splitlines(keepends=True) if not l.startswith('@'))
    oopt = mb.default_opt()
    assert body == ''.join(sp.map_read(oopt, n_, s_, sam=True, qual=q_) for n_, s_, q_ in zip(names, seqs, quals))
    assert out.stdout.count('@SQ\t') == len(gen)
    # -d with a target set of several parts (bin/megapath_nano.py:1641-1645): every part goes into the one file, and the file
    # as the target gives what the FASTA gave
    mpi = tmp_path / 'three_parts.mpi'
    cmd = [sys.executable, os.path.join(root, 'bin', 'mpn-aligner'), '-c', '-a', '-t', '4', '-I', '200K', '-x', 'map-ont']
    out_d = subprocess.run(cmd + ['-d', str(mpi), str(fa), str(fq), '--split-prefix', 'tmp'], capture_output=True, text=True, timeout=300, env=env)
    assert out_d.returncode == 0, out_d.stderr[-2000:]
    n_loaded, off, seen = 0, 0, []
    while off >= 0:
        part, off = mapper.Index.load_at(str(mpi), off)
        seen += part.names
        n_loaded += 1
        part.close()
    assert n_loaded == 3 and seen == [g[0] for g in gen]
    out_l = subprocess.run(cmd + [str(mpi), str(fq), '--split-prefix', 'tmp'], capture_output=True, text=True, timeout=300, env=env)
    assert out_l.returncode == 0, out_l.stderr[-2000:]
    assert ''.join(l for l in out_l.stdout.splitlines(keepends=True) if not l.startswith('@')) == body == \
        ''.join(l for l in out_d.stdout.splitlines(keepends=True) if not l.startswith('@'))
    sp.close()
    for x in gparts + oparts:
        x.close()


def test_independent_checker_on_gpu_output(world):
    """Every number of the PAF / SAM lines recomputed from the CIGAR and the sequences by tests/paf_check.py (written from
    the format and tag definitions, not from the oracle): breaks the twin relation between csrc/align.hip's host half and
    oracle/mm2_oracle.c for NM, ms, AS, nn, de, mlen, blen, clipping, flags, SA, one-primary-per-read, MAPQ range."""
    import paf_check
    from map_cases import hard_reads
    from megapath_nano_amd import mapper
    gen, reads, gidx, _ = world
    allr = reads + hard_reads(gen)
    names, seqs = [r['name'] for r in allr], [r['seq'] for r in allr]
    quals = [bytes(33 + (3 * i + k) % 41 for k in range(len(s))) for i, s in enumerate(seqs)]
    rd = {n_: bytes(s_).decode() for n_, s_ in zip(names, seqs)}
    gd = {n_: bytes(s_).decode() for n_, s_ in gen}
    qd = {n_: q_.decode() for n_, q_ in zip(names, quals)}
    packed = mapper.PackedReads(names, seqs, quals=quals)
    for best_n, pri in ((5, 0.8), (50, 1.0)):
        opt = mapper.default_opt(best_n=best_n, pri_ratio=pri, out_sam=2)
        paf, sam, _ = mapper.map_batch_full(gidx, opt, packed, want_paf=True, want_cols=False)
        st = paf_check.check_paf(paf, rd, gd, best_n=best_n)
        assert st['lines'] >= len(reads) // 2 and st['as_equal'] >= 0.98 * st['lines'], st
        assert paf_check.check_sam(sam, paf, rd, qd) >= len(allr)


def test_mapq_grows_with_the_divergence_of_the_second_copy(libmpn):
    """MAPQ reflects how much better the best locus is than the next one: the same read against targets that hold its locus
    plus a copy of it at growing divergence must not lose mapping quality (constructed pairs; independent of the oracle)."""
    from megapath_nano_amd import mapper, synth
    rng = np.random.default_rng(123)
    a = synth.random_genome(rng, 120000)
    read = synth.ont_errors(rng, a[40000:46000].copy(), 0.03, 0.02, 0.03)
    mapqs = []
    for div in (0.0, 0.01, 0.03, 0.08, 0.2, None):
        gens = [('A', a)]
        if div is not None:
            b = synth.random_genome(rng, 60000)
            b[20000:30000] = synth.mutate_strain(rng, a[38000:48000], 1.0 - div) if div > 0 else a[38000:48000]
            gens.append(('B', b))
        idx = mapper.Index(gens)
        paf = mapper.map_batch(idx, mapper.default_opt(), ['r'], [read])
        idx.close()
        first = paf.splitlines()[0].split('\t')
        assert first[5] in ('A', 'B') and (div == 0.0 or first[5] == 'A')
        mapqs.append(int(first[11]))
    assert mapqs == sorted(mapqs) and mapqs[0] <= 3 and mapqs[-1] == 60, mapqs


def test_edge_inputs_match_oracle(world):
    """Empty and degenerate batches through the C-ABI: no reads, reads shorter than k / shorter than one window, all-N,
    lower-case and IUPAC bases, duplicated names, a read that is a whole target, reads that map nowhere."""
    from megapath_nano_amd import mapper
    from megapath_nano_amd import synth
    gen, _, gidx, oidx = world
    from oracle import mm2_bindings as mb
    opt, oopt = mapper.default_opt(best_n=50, pri_ratio=1.0), mb.default_opt(best_n=50, pri_ratio=1.0)
    assert mapper.map_batch(gidx, opt, [], []) == ''
    paf, cols = mapper.map_batch_ex(gidx, opt, mapper.PackedReads([], []), want_paf=True, want_cols=True)
    assert paf == '' and all(len(v) == 0 for v in cols.values())
    rng = np.random.default_rng(77)
    g0 = np.frombuffer(bytes(gen[0][1]), dtype=np.uint8)
    g1 = np.frombuffer(bytes(gen[1][1]), dtype=np.uint8)
    seg = g0[20000:26000].copy()
    iupac = g1[5000:9000].copy()
    iupac[rng.integers(0, len(iupac), size=60)] = np.frombuffer(b'RYKMSWBDHVN', dtype=np.uint8)[rng.integers(0, 11, size=60)]
    reads = [
        ('len1', g0[100:101]), ('len14', g0[100:114]), ('len15', g0[100:115]), ('len24', g0[100:124]), ('len60', g0[100:160]),
        ('all_n', np.full(500, ord('N'), dtype=np.uint8)), ('lower', np.frombuffer(bytes(seg).lower(), dtype=np.uint8)),
        ('upper', seg), ('upper', seg),  # the same name twice
        ('iupac', iupac), ('whole_target', g1), ('nowhere', synth.ALPHA[rng.integers(0, 4, size=7000)]),
        ('homopolymer', np.full(3000, ord('A'), dtype=np.uint8)),
    ]
    rs = [dict(name=n, seq=s) for n, s in reads]
    got = mapper.map_batch(gidx, opt, [r['name'] for r in rs], [r['seq'] for r in rs])
    want = ''.join(oracle_paf(oidx, oopt, rs))
    assert got == want
    assert 'whole_target\t' in got and 'lower\t' in got and 'nowhere\t' not in got and 'len14\t' not in got
    # a batch in which nothing maps
    none = [dict(name=f'x{i}', seq=synth.ALPHA[rng.integers(0, 4, size=900)]) for i in range(5)]
    assert mapper.map_batch(gidx, opt, [r['name'] for r in none], [r['seq'] for r in none]) == ''.join(oracle_paf(oidx, oopt, none)) == ''


def test_sam_output_matches_oracle(world):
    """-a: SAM records (flags, soft/hard clips, SEQ on the hit's strand, '*' for secondaries, SA tags, flag-4 records for
    unmapped reads) against the oracle's restatement of minimap2's writer; the header lists every target."""
    from map_cases import hard_reads
    from megapath_nano_amd import mapper, synth
    from oracle import mm2_bindings as mb
    gen, reads, gidx, oidx = world
    rng = np.random.default_rng(4)
    extra = [dict(name='nowhere', seq=synth.ALPHA[rng.integers(0, 4, size=3000)]), dict(name='tiny', seq=synth.ALPHA[rng.integers(0, 4, size=9)])]
    rs = list(reads[:25]) + hard_reads(gen) + extra
    names, seqs = [r['name'] for r in rs], [r['seq'] for r in rs]
    for best_n, pri in ((5, 0.8), (50, 1.0)):
        gopt = mapper.default_opt(best_n=best_n, pri_ratio=pri, out_sam=1)
        oopt = mb.default_opt(best_n=best_n, pri_ratio=pri)
        got = mapper.map_batch(gidx, gopt, names, seqs)
        want = ''.join(mb.map_read_sam(oidx, oopt, r['name'], r['seq']) for r in rs)
        assert got == want
    # a world with a 99 %-identity strain copy: secondary records (flag 0x100, SEQ '*')
    gen2 = synth.make_genomes(5, 4, 60000, strain_pairs=1)
    reads2 = synth.make_reads(6, gen2, 12, mean_len=2000)
    gidx2, oidx2 = mapper.Index(gen2), mb.Index(gen2)
    got2 = mapper.map_batch(gidx2, mapper.default_opt(best_n=50, pri_ratio=1.0, out_sam=1), [r['name'] for r in reads2], [r['seq'] for r in reads2])
    assert got2 == ''.join(mb.map_read_sam(oidx2, mb.default_opt(best_n=50, pri_ratio=1.0), r['name'], r['seq']) for r in reads2)
    gidx2.close()
    oidx2.close()
    lines = got.splitlines() + got2.splitlines()
    flags = [int(l.split('\t')[1]) for l in lines]
    assert any(f & 0x800 for f in flags) and any(f & 0x100 for f in flags) and any(f & 0x10 for f in flags) and flags.count(4) >= 2
    assert all(l.split('\t')[9] == '*' for l, f in zip(lines, flags) if f & 0x100)
    assert any('\tSA:Z:' in l for l in lines)
    for l in lines:
        f = l.split('\t')
        if f[5] != '*' and f[9] != '*':   # CIGAR query length == SEQ length (hard clips excluded)
            import re
            qlen = sum(int(n) for n, op in re.findall(r'(\d+)([MIDNSH])', f[5]) if op in 'MIS')
            assert qlen == len(f[9]), f[0]
    hdr = gidx.sam_header('mpn-aligner -a test')
    assert hdr.count('@SQ\t') == len(gen) and hdr.splitlines()[-1].startswith('@PG\tID:mpn-aligner') and f'SN:{gen[0][0]}\tLN:{len(gen[0][1])}' in hdr


@pytest.mark.parametrize('k,w,kw', [
    (15, 15, dict(a=1, b=4, q=1, e=2, min_dp_max=50, zdrop=50, zdrop_inv=50, best_n=1000, pri_ratio=0.0)),   # megapath_nano.py:221-241
    (13, 8, dict(a=3, b=5, q=6, e=3, q2=30, e2=1, best_n=5, pri_ratio=0.8)),
    (15, 10, dict(a=2, b=40, q=4, e=2, best_n=5, pri_ratio=0.8)),   # mismatch score outside the strip kernel's 6-bit fields
])
def test_other_index_and_scoring_options(world, k, w, kw):
    """The genome-similarity option sets of the reference (-k15 -w15 -A1 -B4 -O1 -E2 -s50 -z50 -N 1000 -p 0) and other k/w/scores:
    same PAF as the oracle (the strip kernel's score table takes 6-bit scores; beyond that the dispatcher uses the band kernel)."""
    from megapath_nano_amd import mapper
    from oracle import mm2_bindings as mb
    gen, reads, _, _ = world
    gidx, oidx = mapper.Index(gen, k=k, w=w), mb.Index(gen, k=k, w=w)
    try:
        gopt, oopt = mapper.default_opt(**kw), mb.default_opt(**kw)
        sub = reads[:30]
        names = [r['name'] for r in sub]
        got = split_by_read(mapper.map_batch(gidx, gopt, names, [r['seq'] for r in sub]), names)
        want = oracle_paf(oidx, oopt, sub)
        for r, g, x in zip(sub, got, want):
            assert g == x, (r['name'], k, w)
        assert sum(x.count('\n') for x in want) >= len(sub) // 2
    finally:
        gidx.close()
        oidx.close()


def test_columns_without_text_equal_columns_with_text(world):
    """With columns only the stitched CIGARs never leave the GPU (csrc/stitch_kernels.h); the numbers must not depend on it."""
    from megapath_nano_amd import mapper
    gen, reads, gidx, _ = world
    gopt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    packed = mapper.PackedReads([r['name'] for r in reads], [r['seq'] for r in reads])
    paf, c1 = mapper.map_batch_ex(gidx, gopt, packed, want_paf=True, want_cols=True)
    _, c2 = mapper.map_batch_ex(gidx, gopt, packed, want_paf=False, want_cols=True)
    assert len(c1['rid']) == paf.count('\n') > 0
    for k in c1:
        assert np.array_equal(c1[k], c2[k]), k


def test_dp_groups_share_the_round_pools():
    """A round whose direction matrices exceed the scratch budget runs in several groups; job records, results and CIGARs of
    all groups stay in the round's device pools for the stitching kernel.  MPN_DP_BUDGET (read once per process) forces many
    groups on a small world; the PAF must equal the oracle's."""
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from map_cases import small_world, hard_reads
from megapath_nano_amd import mapper
from oracle import mm2_bindings as mb
gen, reads = small_world(seed=5, n_genomes=5, glen=120000, n_reads=40, mean_len=4000)
reads += hard_reads(gen)
gidx, oidx = mapper.Index(gen), mb.Index(gen)
gopt, oopt = mapper.default_opt(), mb.default_opt()
names = [r['name'] for r in reads]
got = mapper.map_batch(gidx, gopt, names, [r['seq'] for r in reads])
want = ''.join(mb.map_read(oidx, oopt, r['name'], r['seq'])[2] for r in reads)
by = {}
for line in got.splitlines(keepends=True):
    by.setdefault(line.split('\t', 1)[0], []).append(line)
assert ''.join(''.join(by.get(n_, [])) for n_ in names) == want
assert mapper.last_stats()['dp_jobs'] > 200
print('OK')
''' % (ROOT, os.path.join(ROOT, 'tests'))
    env = dict(os.environ, MPN_DP_BUDGET=str(2 << 20))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and 'OK' in out.stdout, out.stderr[-2000:]


_SHED_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
from map_cases import small_world
from megapath_nano_amd import mapper
gen, reads = small_world(seed=3, n_genomes=6, glen=150000, n_reads=60, mean_len=4000)
idx = mapper.Index(gen)
opt = mapper.default_opt(best_n=50, pri_ratio=1.0)
names = [r['name'] for r in reads]
paf = mapper.map_batch(idx, opt, names, [r['seq'] for r in reads])
st = mapper.last_stats()
print(json.dumps(dict(paf=paf, shed=st['workers_shed'], workers=st['workers'], sub_batches=st['sub_batches'])))
'''


def test_worker_sheds_on_arena_oom(world):
    """A worker whose device ARENA cannot grow (MPN_TEST_ARENA_BUDGET: all arenas share a budget that only some of the workers fit
    in) gives its scratch back and leaves; its sub-batch goes to the others and the output is what the unconstrained run gives.
    The decision rests on the allocation's out-of-memory flag, not on the error text."""
    import json
    from megapath_nano_amd import mapper
    gen, reads, gidx, oidx = world
    gopt = mapper.default_opt(best_n=50, pri_ratio=1.0)
    names = [r['name'] for r in reads]
    want = mapper.map_batch(gidx, gopt, names, [r['seq'] for r in reads])
    env = dict(os.environ, MPN_PIPE_WORKERS='4', MPN_SUB_BATCH_BP='15000', MPN_TEST_ARENA_BUDGET=str(40 << 20))
    out = subprocess.run([sys.executable, '-c', _SHED_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got['sub_batches'] >= 8 and got['workers'] == 4
    assert got['shed'] >= 1, got['shed']
    assert got['paf'] == want
