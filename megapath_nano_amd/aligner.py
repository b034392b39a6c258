"""Host-side mirror of /root/reference/bin/lib/aligner.py `Align()` (:93-342) on top of libmpn.so.

Same keyword-only signature, same returned DataFrame (columns and dtypes of aligner.py:291-294 plus the taxonomy join
of :317-332 and `alignment_score_tiebreaker` from Python's `random` seeded with the md5 of the query basenames,
:160-168,:334-335), same `os.sys.exit(...)` messages on bad arguments (:120,:126,:132,:140,:153,:157,:164), same side
file `<paf_path_and_prefix>.paf`.  What differs: no child process.  The `minimap2` argv the reference would build
(:187-199,:219-224) is parsed into mpn_map_opt, the target FASTA files are read here and indexed on the GPU (cached per
target set), reads are mapped in batches through mpn_map_batch_ex (PAF text + integer columns), and no
awk/SAM2PAF/samtools step exists: the PAF comes out in minimap2's native `-c` tag order directly.

With `paf_path_and_prefix` set the reference runs minimap2 with `-a` and keeps `<prefix>.sam` (:183-184,:219-227): so does
this mirror (second text pass of the same batch with `out_sam`).  Not produced: `<prefix>.bam/.bai` (the reference pipes
the SAM through samtools, :246-252).
"""
import hashlib
import os
import random
import shlex

import numpy as np
import pandas

from . import fastx, mapper

align_list_col_name_no_assembly_id = ['read_id', 'read_length', 'read_from', 'read_to', 'strand', 'sequence_id',
                                      'sequence_length', 'sequence_from', 'sequence_to', 'match', 'mapq', 'edit_dist',
                                      'alignment_score']
list_col = ('read_id', 'read_length', 'read_from', 'read_to', 'strand', 'sequence_id', 'sequence_length', 'sequence_from',
            'sequence_to', 'match', 'alignment_block_length', 'mapq', 'edit_dist', 'alignment_score')

_INDEX_CACHE = {}


read_fastx = fastx.read_fastx


def load_target(path, k, w):
    """The target of an aligner call -> mapper.Index.  The path is opened exactly once (it is a FIFO in the reference's
    human/decoy call, aligner.py:143-144): a saved index is loaded (megapath_nano.py:1641-1645), sequences are indexed."""
    kind, stream = fastx.open_once(path)
    try:
        if kind == 'index':
            if not fastx.is_fifo(path):
                stream.close()
                return mapper.Index.load(path)
            import tempfile
            with tempfile.NamedTemporaryFile(suffix='.mpi') as tmp:  # mpn_index_load wants a seekable file
                while True:
                    chunk = stream.read(1 << 24)
                    if not chunk:
                        break
                    tmp.write(chunk)
                tmp.flush()
                return mapper.Index.load(tmp.name)
        return mapper.Index([(n, sq) for n, sq, _ in fastx.iter_fastx(stream)], k=k, w=w)
    finally:
        if not stream.closed:
            stream.close()


def parse_aligner_options(aligner_options, mapping_only):
    """minimap2 CLI subset used by the reference (megapath_nano.py:1124,:1270,:1383,:221-241) -> (MapOpt, k, w)."""
    k, w = 15, 10
    kw = {}
    args = list(aligner_options or [])
    i = 0

    def val():
        nonlocal i
        a = args[i]
        if len(a) > 2 and not a.startswith('--'):
            return a[2:]
        i += 1
        return args[i]

    while i < len(args):
        a = args[i]
        if a.startswith('-x'):
            preset = val()
            if preset not in ('map-ont',):
                raise ValueError(f'preset {preset} is not implemented (map-ont only)')
        elif a.startswith('-N'):
            kw['best_n'] = int(val())
        elif a.startswith('-p'):
            kw['pri_ratio'] = float(val())
        elif a.startswith('-k'):
            k = int(val())
        elif a.startswith('-w'):
            w = int(val())
        elif a.startswith('-A'):
            kw['a'] = int(val())
        elif a.startswith('-B'):
            kw['b'] = int(val())
        elif a.startswith('-O'):
            v = val().split(',')
            kw['q'] = int(v[0])
            kw['q2'] = int(v[1]) if len(v) > 1 else kw.get('q2', 24)
        elif a.startswith('-E'):
            v = val().split(',')
            kw['e'] = int(v[0])
            kw['e2'] = int(v[1]) if len(v) > 1 else kw.get('e2', 1)
        elif a.startswith('-s'):
            kw['min_dp_max'] = int(val())
        elif a.startswith('-z'):
            v = val().split(',')
            kw['zdrop'] = int(v[0])
            kw['zdrop_inv'] = int(v[1]) if len(v) > 1 else int(v[0])
        elif a.startswith('-f'):
            kw['mid_occ_frac'] = float(val())
        elif a.startswith('-t') or a.startswith('-I'):
            val()  # threads / index batch size: owned by the library
        elif a == '--split-prefix':
            i += 1
        elif a in ('-c', '-a'):
            pass
        else:
            raise ValueError(f'aligner option {a} is not understood')
        i += 1
    opt = mapper.default_opt(with_cigar=0 if mapping_only else 1, **kw)
    return opt, k, w


def Align(*, assembly_metadata, global_options, temp_dir_name, log_file, query_filename_list=None,
          query_assembly_list=None, target_filename_list=None, target_assembly_list=None, aligner_options=None,
          paf_path_and_prefix=None, mapping_only=False, module_option='', AMR_output_folder='', align_concat_fa=False,
          batch_bases=200_000_000):
    nano_dir = global_options.get('nano_dir', os.getcwd())
    if not align_concat_fa:
        num_target_specification = 0
        if target_filename_list is not None and target_filename_list['path'].shape[0] > 0:
            num_target_specification += 1
        if target_assembly_list is not None and target_assembly_list['assembly_id'].shape[0] > 0:
            num_target_specification += 1
        if num_target_specification != 1:
            os.sys.exit('Exactly one of target_filename_list and target_assembly_list must be specified')
        if target_assembly_list is not None and len(target_assembly_list['assembly_id']) > 0:
            target_filename_list = assembly_metadata.get_assembly_path(assembly_list=target_assembly_list)
            if target_filename_list is None:
                os.sys.exit('Target assembly_id not found')
            target_filename_list['path'] = target_filename_list['path'].map(
                lambda x: os.path.join(global_options['assembly_folder'], x))
            target_assembly_length = assembly_metadata.get_assembly_length(assembly_list=target_assembly_list)
            if target_assembly_length is None:
                os.sys.exit('Target assembly_length not found')
        else:
            target_assembly_length = target_filename_list.assign(assembly_length=lambda x: 1)
        if target_assembly_length.shape[0] != target_filename_list.shape[0]:
            os.sys.exit('Number of target_assembly_length does not match number of target_filename_list')
        target_paths = list(target_filename_list['path'])
    elif module_option != 'amplicon_filter_module':
        target_paths = [f'{nano_dir}/genomes/refseq/refseq.fna.gz']
    else:
        target_paths = [f'{target_filename_list["path"][0]}']

    num_query_specification = 0
    if query_filename_list is not None and query_filename_list['path'].shape[0] > 0:
        num_query_specification += 1
    if query_assembly_list is not None and query_assembly_list['assembly_id'].shape[0] > 0:
        num_query_specification += 1
    if num_query_specification != 1:
        os.sys.exit('Exactly one of query_filename_list and query_assembly_list must be specified')
    if query_assembly_list is not None and query_assembly_list['assembly_id'].shape[0] > 0:
        query_filename_list = assembly_metadata.get_assembly_path(assembly_list=query_assembly_list)
        if query_filename_list is None or query_filename_list.shape[0] != query_assembly_list['assembly_id'].shape[0]:
            os.sys.exit('Query assembly_id not found')
        query_filename_list['path'] = query_filename_list['path'].map(
            lambda x: os.path.join(global_options['assembly_folder'], x))

    random_hash_string = ''
    for query in query_filename_list['path']:
        if not os.path.isfile(query):
            os.sys.exit('Query file ' + query + ' not exists')
        random_hash_string = random_hash_string + os.path.split(query)[1]
    random.seed(hashlib.md5(random_hash_string.encode()).hexdigest())                            # :167-168

    opt, k, w = parse_aligner_options(aligner_options, mapping_only)
    idx_key = (tuple(target_paths), k, w)
    idx = _INDEX_CACHE.get(idx_key)
    if idx is None:
        if len(target_paths) == 1:
            idx = load_target(target_paths[0], k, w)    # sequences, or a prebuilt index (megapath_nano.py:1641-1645)
        else:
            genomes = []
            for tp in target_paths:
                genomes.extend(read_fastx(tp))
            idx = mapper.Index(genomes, k=k, w=w)
            del genomes
        _INDEX_CACHE[idx_key] = idx
    seq_names = np.array(idx.names, dtype=object)
    seq_lens = idx.lens

    output_paf = not (paf_path_and_prefix is None or paf_path_and_prefix == '')
    paf_file = open(f'{paf_path_and_prefix}.paf', 'w') if output_paf else None
    sam_file = open(f'{paf_path_and_prefix}.sam', 'w') if output_paf and not mapping_only else None
    if sam_file is not None:
        sam_file.write(idx.sam_header())
        sam_opt = mapper.MapOpt.from_buffer_copy(bytes(opt))
        sam_opt.out_sam = 1
    frames = []
    try:
        for query in query_filename_list['path']:
            reads = read_fastx(query)
            lo = 0
            while lo < len(reads):
                hi, acc = lo, 0
                while hi < len(reads) and (hi == lo or acc + len(reads[hi][1]) <= batch_bases):
                    acc += len(reads[hi][1])
                    hi += 1
                packed = mapper.PackedReads([r[0] for r in reads[lo:hi]], [r[1] for r in reads[lo:hi]])
                paf, c = mapper.map_batch_ex(idx, opt, packed, want_paf=output_paf, want_cols=True)
                if paf_file is not None:
                    paf_file.write(paf)
                if sam_file is not None:
                    sam_file.write(mapper.map_batch_ex(idx, sam_opt, packed, want_paf=True, want_cols=False)[0])
                names = np.array(packed.names, dtype=object)
                frames.append(pandas.DataFrame({
                    'read_id': names[c['read_idx']], 'read_length': packed.lens[c['read_idx']].astype(np.int64),
                    'read_from': c['qs'].astype(np.int64), 'read_to': c['qe'].astype(np.int64),
                    'strand': np.where(c['rev'] != 0, '-', '+'), 'sequence_id': seq_names[c['rid']],
                    'sequence_length': seq_lens[c['rid']].astype(np.int64), 'sequence_from': c['rs'].astype(np.int64),
                    'sequence_to': c['re'].astype(np.int64), 'match': c['mlen'].astype(np.int64),
                    'alignment_block_length': c['blen'].astype(np.int64), 'mapq': c['mapq'].astype(np.int64),
                    'edit_dist': c['nm'].astype(np.int64), 'alignment_score': c['as_'].astype(np.int64)}))
                lo = hi
    finally:
        if paf_file is not None:
            paf_file.close()
        if sam_file is not None:
            sam_file.close()
    if frames:
        prefilter_align_list = pandas.concat(frames, ignore_index=True)[list(list_col)]
    else:
        prefilter_align_list = pandas.DataFrame({c: pandas.Series(dtype=(str if c in ('read_id', 'strand', 'sequence_id')
                                                                         else np.int64)) for c in list_col})
    min_alignment_score = global_options['min_alignment_score']
    align_list = prefilter_align_list.query('sequence_length > 0 and alignment_score >= @min_alignment_score')  # :312
    if target_assembly_list is not None and len(target_assembly_list['assembly_id']) > 0:
        sequence_assembly_tax_id = assembly_metadata.get_sequence_tax_id(assembly_list=target_assembly_list).set_index(
            ['sequence_id'])[['assembly_id', 'tax_id', 'species_tax_id', 'genus_tax_id']]
        num_align = align_list.shape[0]
        align_list = align_list.merge(right=sequence_assembly_tax_id, how='inner', left_on='sequence_id', right_index=True,
                                      suffixes=['', '_y'], validate='m:1')
        if align_list.shape[0] != num_align:
            print('Some sequence id cannot be matched', file=os.sys.stderr)
    align_list = align_list.assign(alignment_score_tiebreaker=lambda x: 0)
    align_list['alignment_score_tiebreaker'] = align_list['alignment_score_tiebreaker'].apply(lambda x: random.random())
    return align_list
