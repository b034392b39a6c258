"""Seeded synthetic align_list tables for the Reassign parity tests (pure data generation).

Columns follow the align_list wire format of /root/reference/bin/lib/aligner.py:27-32,291-294.
No (read_id, name) group ever holds two rows with the same alignment_score: the reference resolves
such ties with an unstable sort (reassignment.py:73), see SURVEY.md Appendix B-6.
"""
import numpy as np

SPECIES = [
    ('NZ_ECOLI1.1', 'Escherichia coli strain K-12 substr. MG1655, complete genome'),
    ('NZ_ECOLI2.1', 'Escherichia coli O157:H7 str. Sakai chromosome'),
    ('NZ_SFLEX1.1', 'Shigella flexneri 2a str. 301 chromosome, complete genome'),
    ('NZ_SSONN1.1', 'Shigella sonnei strain ATCC 29930 chromosome'),
    ('NZ_KPNEU1.1', 'Klebsiella pneumoniae subsp. pneumoniae HS11286 chromosome'),
    ('NZ_BSP001.1', 'Bacillus sp. FJAT-27231 chromosome, complete genome'),
    ('NZ_BSP002.1', 'Bacillus sp. X1(2014) chromosome'),
    ('NZ_SAURE1.1', 'Staphylococcus aureus subsp. aureus NCTC 8325 chromosome'),
    ('NZ_PAERU1.1', 'Pseudomonas aeruginosa PAO1 chromosome, complete genome'),
    ('NZ_SINGLE.1', 'Plasmid'),
    ('NZ_ABAUM1.1', 'Acinetobacter baumannii strain AB30 chromosome'),
    ('NZ_UNUSED.1', 'Zymomonas mobilis subsp. mobilis ZM4 chromosome'),
]


def table(rows):
    cols = ['read_id', 'read_length', 'read_from', 'read_to', 'strand', 'sequence_id', 'sequence_length',
            'sequence_from', 'sequence_to', 'match', 'alignment_block_length', 'mapq', 'edit_dist',
            'alignment_score', 'assembly_id', 'tax_id', 'species_tax_id', 'genus_tax_id',
            'alignment_score_tiebreaker']
    return {c: [r[c] for r in rows] for c in cols}


def make_row(rng, read_id, read_length, seq_idx, score):
    sid = SPECIES[seq_idx][0]
    rf = int(rng.integers(0, 50))
    rt = read_length - int(rng.integers(0, 50))
    sf = int(rng.integers(0, 4000000))
    return dict(read_id=read_id, read_length=read_length, read_from=rf, read_to=rt,
                strand='+' if rng.random() < 0.5 else '-', sequence_id=sid, sequence_length=4600000, sequence_from=sf,
                sequence_to=sf + (rt - rf) + int(rng.integers(-20, 20)), match=int((rt - rf) * 0.88),
                alignment_block_length=int((rt - rf) * 1.05), mapq=int(rng.integers(0, 61)),
                edit_dist=int((rt - rf) * 0.1), alignment_score=int(score), assembly_id=f'GCF_{seq_idx:09d}.1',
                tax_id=1000 + seq_idx, species_tax_id=500 + seq_idx // 2 if seq_idx < 2 else 600 + seq_idx,
                genus_tax_id=50 + seq_idx // 4, alignment_score_tiebreaker=float(rng.random()))


def community(seed, n_reads, weights, ambiguity, extra_rows=0.0, unknown_seq=False):
    """weights[i]: relative number of reads truly from SPECIES[i]; ambiguity[i]: list of (other, prob) that a
    read from i also aligns to `other` with a slightly lower or higher score."""
    rng = np.random.default_rng(seed)
    w = np.array(weights, dtype=float)
    w /= w.sum()
    rows = []
    for r in range(n_reads):
        src = int(rng.choice(len(w), p=w))
        rid = f'read_{seed}_{r:06d}'
        rl = int(rng.integers(500, 20000))
        base = int(rl * 1.6)
        used = set()
        score = base + int(rng.integers(-50, 50))
        rows.append(make_row(rng, rid, rl, src, score))
        used.add(score)
        for other, prob in ambiguity.get(src, []):
            if rng.random() < prob:
                sc = base + int(rng.integers(-120, 60))
                while sc in used:
                    sc += 1
                used.add(sc)
                rows.append(make_row(rng, rid, rl, other, sc))
        if rng.random() < extra_rows:  # a second, lower-scoring hit on the same sequence (supplementary-like)
            sc = base // 3 + int(rng.integers(0, 40))
            while sc in used:
                sc += 1
            used.add(sc)
            rows.append(make_row(rng, rid, rl, src, sc))
    if unknown_seq:
        r0 = make_row(rng, 'read_unknown', 900, 0, 777)
        r0['sequence_id'] = 'NZ_NOT_IN_DB.1'
        rows.append(r0)
    order = rng.permutation(len(rows))
    return table([rows[i] for i in order])


def cases():
    out = []
    # 1 basic: E. coli explains Shigella flexneri (few unique Shigella reads)
    out.append(dict(name='basic_explain', level='species', params={},
                    table=community(1, 400, [60, 0, 1, 0, 20, 0, 0, 10, 0, 0, 0, 0],
                                    {0: [(2, 0.5)], 2: [(0, 0.9)]}, extra_rows=0.1)))
    # 2 several relations, strain pairs, ' sp. ' names, reads hitting two explainers, unknown sequence id
    out.append(dict(name='multi_explainers', level='species', params={},
                    table=community(2, 900, [50, 30, 1, 1, 40, 30, 1, 25, 0, 1, 1, 0],
                                    {0: [(2, 0.4), (3, 0.3), (1, 0.5), (4, 0.1)], 1: [(0, 0.5), (2, 0.3)],
                                     2: [(0, 0.9), (4, 0.5)], 3: [(0, 0.8), (4, 0.6)], 4: [(2, 0.2), (3, 0.2), (10, 0.1)],
                                     5: [(6, 0.5)], 6: [(5, 0.9), (7, 0.5)], 7: [(6, 0.1)], 9: [(7, 0.9)],
                                     10: [(4, 0.9), (0, 0.5), (7, 0.5)]}, extra_rows=0.2, unknown_seq=True)))
    # 3 same data at strain level (names = full descriptions)
    out.append(dict(name='strain_level', level='strain', params={},
                    table=community(3, 500, [50, 30, 1, 1, 40, 0, 0, 0, 0, 0, 0, 0],
                                    {0: [(1, 0.6), (2, 0.4)], 1: [(0, 0.6)], 2: [(0, 0.9)], 3: [(1, 0.9)]})))
    # 4 non-default thresholds incl. AS_threshold > 0
    out.append(dict(name='thresholds', level='species', params=dict(error_rate=0.2, ratio=0.5, AS_threshold=1.02),
                    table=community(4, 600, [50, 0, 8, 5, 30, 0, 0, 0, 0, 0, 0, 0],
                                    {0: [(2, 0.5), (3, 0.5)], 2: [(0, 0.9)], 3: [(0, 0.9), (4, 0.9)], 4: [(3, 0.3)]})))
    # 5 no relation at all -> early return of the deduplicated table
    out.append(dict(name='no_relation', level='species', params={},
                    table=community(5, 200, [10, 0, 10, 0, 10, 0, 0, 0, 0, 0, 0, 0], {0: [(2, 0.3)], 2: [(4, 0.3)]})))
    # 6 no multi-mapped read -> functools.reduce over an empty iterable raises TypeError (reassignment.py:91)
    out.append(dict(name='no_multimapped', level='species', params={},
                    table=community(6, 50, [10, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0], {})))
    return out


def boundary_case():
    """UCount_j == error_rate * UCount_i exactly (20 unique E. coli K-12 reads -> threshold 1.0; S. flexneri has exactly 1
    unique read): strict `<` means NOT explained; K. pneumoniae (0 unique) IS explained."""
    rng = np.random.default_rng(7)
    rows = []
    for r in range(20):
        rows.append(make_row(rng, f'u_ecoli_{r}', 3000, 0, 4000 + r))
    rows.append(make_row(rng, 'u_sflex_0', 3000, 2, 4100))
    for r in range(6):
        rid = f'amb_{r}'
        rows.append(make_row(rng, rid, 3000, 0, 4200 + 3 * r))
        rows.append(make_row(rng, rid, 3000, 2, 4201 + 3 * r))
        rows.append(make_row(rng, rid, 3000, 4, 4202 + 3 * r))
    return dict(name='boundary_strict_less', level='species', params={}, table=table(rows))
