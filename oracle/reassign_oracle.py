"""reassign_oracle.py -- TEST INFRASTRUCTURE ONLY (parity oracle; never imported by the product).

Plain-Python restatement of the reference's "global multi-alignment scoring pass"
    /root/reference/bin/lib/reassignment.py   Reassign :66-108, build_i_explains_j_dict :27-36,
                                              reassign_alignment :38-64
plus the two consumers that turn its output into the parity artefacts
    /root/reference/bin/megapath_nano.py:1287  best alignment per read
    /root/reference/bin/megapath_nano.py:1289  aligned bp per species_tax_id
    /root/reference/bin/megapath_nano.py:3664-3667  .read_count_by_name

It is written over dicts/lists (no pandas) and is O(N); the reference is an O(N^2) pandas row-apply.
Pinned by tests/golden/reassign_golden.json, which holds outputs of the reference module itself
imported in the build container (pandas 2.3.3).

Reference behaviours kept on purpose:
  B-1  MCount is looked up with a frozenset in a Counter keyed by tuples (:16 vs :31) => always 0.
  B-3  no read with >= 2 (read, name) rows => functools.reduce on an empty list => TypeError (:91).
  B-4  a relabelled row keeps its own score/coordinates/taxids; only name and sequence_id change (:56).
  B-5  explainer rows are processed in ascending order of their (snapshot) name: on pandas >= 1.5 the
       Categorical assignment at :61 leaves an object column, so :62 sorts alphabetically
       (explainer_order='alphabetical', what the goldens pin); 'frequency' is the pandas <= 1.1 order.
  B-6  ties of alignment_score inside one (read_id, name) group are resolved by an unstable sort in the
       reference; here the row that comes LAST in input order wins (stable).  Tests avoid such ties.
"""
from collections import Counter, OrderedDict


def species_name(desc, level):
    """reassignment.py:69-70"""
    if level != 'species':
        return desc
    if ' sp. ' not in desc:
        return ' '.join(desc.split(' ', 2)[0:2])
    return ' '.join(desc.split(' ', 3)[0:3])


def reassign_oracle(table, sequence_name, error_rate=0.05, ratio=0.05, AS_threshold=0, level='species',
                    explainer_order='alphabetical'):
    """table: dict column -> list (align_list wire format).  sequence_name: [(sequence_id, description)].
    Returns dict(explains, rows) where rows = [dict(index, name, sequence_id, is_in_explain_other, src)] in
    ascending `index` (index = row number after the inner merge with sequence_name, as in the reference;
    src = row number in the input table).  explains is None when the relation is empty (early return :100)."""
    # :68-71 inner merge on sequence_id, left order preserved
    by_seq = OrderedDict()
    for sid, desc in sequence_name:
        by_seq.setdefault(sid, []).append(species_name(desc, level))
    merged = []  # (index, src_row, name)
    n_in = len(table['read_id'])
    for r in range(n_in):
        for nm in by_seq.get(table['sequence_id'][r], ()):
            merged.append((len(merged), r, nm))
    # :73 sort by score, keep the last row of every (read_id, name)
    order = sorted(range(len(merged)), key=lambda k: table['alignment_score'][merged[k][1]])
    last = {}
    for pos, k in enumerate(order):
        _, r, nm = merged[k]
        last[(table['read_id'][r], nm)] = pos
    kept = [order[pos] for pos in sorted(last.values())]  # merged-row ids in sorted-by-score order
    rows = [dict(index=merged[k][0], src=merged[k][1], name=merged[k][2],
                 sequence_id=table['sequence_id'][merged[k][1]]) for k in kept]
    # :75-81 counts
    all_count = Counter(x['name'] for x in rows)
    per_read = Counter(table['read_id'][x['src']] for x in rows)
    u_count = Counter(x['name'] for x in rows if per_read[table['read_id'][x['src']]] == 1)
    # :83-91
    if not any(v > 1 for v in per_read.values()):
        raise TypeError('reduce() of empty iterable with no initial value')
    mcount = 0  # B-1
    species_list = [nm for nm, _ in sorted(all_count.items(), key=lambda kv: -kv[1])]
    explains = {}
    for si in species_list:
        for sj in species_list:
            if si == sj:
                continue
            if all_count[si] - mcount >= ratio * all_count[si] and u_count.get(sj, 0) < error_rate * u_count.get(si, 0):
                explains.setdefault(si, set()).add(sj)
    if not explains:
        for x in rows:
            x['is_in_explain_other'] = None
        return dict(explains=None, rows=sorted(rows, key=lambda x: x['index']))
    # :38-64
    for x in rows:
        x['is_in_explain_other'] = x['name'] in explains
    by_read = {}
    for x in rows:
        by_read.setdefault(table['read_id'][x['src']], []).append(x)
    snap = [(x['name'], x) for x in rows if x['is_in_explain_other']]
    if explainer_order == 'alphabetical':
        keyf = lambda t: t[0]  # noqa: E731
    else:
        freq = Counter(nm for nm, _ in snap)
        rank = {nm: i for i, (nm, _) in enumerate(sorted(freq.items(), key=lambda kv: -kv[1]))}
        keyf = lambda t: rank[t[0]]  # noqa: E731
    for nm, x in sorted(snap, key=keyf):
        first_score = table['alignment_score'][x['src']]
        for y in by_read[table['read_id'][x['src']]]:
            if y['name'] in explains[nm] and y['name'] != nm and \
                    table['alignment_score'][y['src']] * AS_threshold <= first_score:
                y['name'] = nm
                y['sequence_id'] = f'{nm}_reassigned'
    return dict(explains={k: sorted(v) for k, v in explains.items()}, rows=sorted(rows, key=lambda x: x['index']))


def best_per_read(table, rows):
    """megapath_nano.py:1287: sort (read_id, alignment_score, tiebreaker), keep the last row per read."""
    best = {}
    for x in rows:
        r = x['src']
        key = (table['alignment_score'][r], table['alignment_score_tiebreaker'][r])
        rid = table['read_id'][r]
        if rid not in best or key >= best[rid][0]:
            best[rid] = (key, x)
    return [v[1] for v in best.values()]


def read_count_by_name(table, rows):
    """megapath_nano.py:3664-3667 with --reassignment: reads per name over the best rows."""
    return dict(Counter(x['name'] for x in best_per_read(table, rows)))


def aligned_bp_by_species(table, rows):
    """megapath_nano.py:1289"""
    out = Counter()
    for x in best_per_read(table, rows):
        r = x['src']
        out[int(table['species_tax_id'][r])] += int(table['sequence_to'][r]) - int(table['sequence_from'][r])
    return dict(out)
