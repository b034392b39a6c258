"""Host-side mirror of the reference's `Align()` (/root/reference/bin/lib/aligner.py:93-342) on top of libmpn.so.

Contract kept: the keyword-only signature, the returned DataFrame (the 14 alignment columns of :291-294, the taxonomy
columns of the join at :317-332, `alignment_score_tiebreaker` drawn from Python's `random` seeded with the md5 of the
query basenames, :160-168,:334-335), the `os.sys.exit(...)` messages on bad arguments (:120,:126,:132,:140,:153,:157,
:164) and the side files `<paf_path_and_prefix>.sam/.paf/.bam/.bam.bai` (:183-184,:245-261).

What differs: no child processes.  The minimap2 command line the reference would build (:187-199,:219-224) is parsed
into mpn_map_opt; the target is read ONCE (it may be a FIFO, :143-144) and cut into index parts the way minimap2's `-I`
does; every part is indexed on the GPU and every batch of reads is mapped against it; with `--split-prefix` (which the
reference always passes, megapath_nano.py:1124,:1270) the per-part hits are merged like minimap2 merges its dumps.
PAF and SAM come from the same mapping pass; awk / SAM2PAF / samtools are replaced by integer columns, and the BAM is
written by megapath_nano_amd.bam.
"""
import hashlib
import os
import random

import numpy as np
import pandas

from . import fastx, mapper
from .pipeline import random_block

align_list_col_name_no_assembly_id = ['read_id', 'read_length', 'read_from', 'read_to', 'strand', 'sequence_id',
                                      'sequence_length', 'sequence_from', 'sequence_to', 'match', 'mapq', 'edit_dist',
                                      'alignment_score']
list_col = ('read_id', 'read_length', 'read_from', 'read_to', 'strand', 'sequence_id', 'sequence_length', 'sequence_from',
            'sequence_to', 'match', 'alignment_block_length', 'mapq', 'edit_dist', 'alignment_score')
_TEXT_COLS = ('read_id', 'strand', 'sequence_id')

# minimap2 reads the target in mini-batches of at least this many bases (mm_idx_reader); the environment override exists so
# that tests can cut small target sets into parts
IDX_MINI_BATCH = int(os.environ.get('MPN_IDX_MINI_BATCH', 50_000_000))
IDX_BATCH_DEFAULT = 4_000_000_000  # minimap2 -I default

read_fastx = fastx.read_fastx
_INDEX_CACHE = {}


# ---- minimap2 command line -------------------------------------------------------------------------------------------
def parse_num(text):
    """minimap2's mm_parse_num: a number with an optional K/M/G suffix."""
    text = text.strip()
    mult = {'k': 1e3, 'm': 1e6, 'g': 1e9}.get(text[-1:].lower())
    return int(float(text[:-1]) * mult + .499) if mult else int(float(text) + .499)


class AlignerOptions:
    """The subset of minimap2's command line the reference uses (megapath_nano.py:1124,:1270,:1383,:221-241)."""

    _INT = {'-N': 'best_n', '-k': None, '-w': None, '-A': 'a', '-B': 'b', '-s': 'min_dp_max'}

    def __init__(self, args, mapping_only):
        self.k, self.w = 15, 10
        self.batch_bases = IDX_BATCH_DEFAULT
        self.split = False
        fields = {}
        args = list(args or [])
        i = 0
        while i < len(args):
            a = args[i]
            if a in ('-c', '-a'):
                i += 1
                continue
            if a == '--split-prefix':
                self.split = True
                i += 2
                continue
            flag = a[:2]
            if not a.startswith('-') or a.startswith('--') or flag not in ('-x', '-N', '-p', '-k', '-w', '-A', '-B', '-O', '-E', '-s',
                                                                           '-z', '-f', '-t', '-I'):
                raise ValueError(f'aligner option {a} is not understood')
            if len(a) > 2:
                value = a[2:]
                i += 1
            else:
                if i + 1 >= len(args):
                    raise ValueError(f'aligner option {a} needs a value')
                value = args[i + 1]
                i += 2
            if flag == '-x':
                if value != 'map-ont':
                    raise ValueError(f'preset {value} is not implemented (map-ont only)')
            elif flag == '-k':
                self.k = int(value)
            elif flag == '-w':
                self.w = int(value)
            elif flag in self._INT:
                fields[self._INT[flag]] = int(value)
            elif flag == '-p':
                fields['pri_ratio'] = float(value)
            elif flag == '-f':
                fields['mid_occ_frac'] = float(value)
            elif flag in ('-O', '-E', '-z'):
                first, _, second = value.partition(',')
                lo, hi = {'-O': ('q', 'q2'), '-E': ('e', 'e2'), '-z': ('zdrop', 'zdrop_inv')}[flag]
                fields[lo] = int(first)
                if second:
                    fields[hi] = int(second)
                elif flag == '-z':
                    fields[hi] = int(first)
            elif flag == '-I':
                self.batch_bases = parse_num(value)
            # -t: the library owns its threads
        self.opt = mapper.default_opt(with_cigar=0 if mapping_only else 1, **fields)


def parse_aligner_options(aligner_options, mapping_only):
    """-> (MapOpt, k, w)"""
    o = AlignerOptions(aligner_options, mapping_only)
    return o.opt, o.k, o.w


# ---- targets: one pass over the files, cut into index parts like minimap2 -I ----------------------------------------------
def iter_target_records(paths):
    for path in paths:
        kind, stream = fastx.open_once(path)
        try:
            if kind == 'index':
                raise ValueError(f'{path}: a saved index cannot be mixed with sequence targets')
            for name, seq, _ in fastx.iter_fastx(stream):
                yield name, seq
        finally:
            stream.close()


def iter_target_parts(records, batch_bases):
    """minimap2's index reader (index.c: mm_idx_gen): the target is taken in mini-batches of at least IDX_MINI_BATCH bases
    (whole sequences); a part is closed after the first mini-batch that brings it OVER batch_bases.  Yields lists of
    (name, sequence)."""
    part, part_bases, mini = [], 0, 0
    mini_batch = min(IDX_MINI_BATCH, max(1, batch_bases))   # mm_idx_gen reads min(mini_batch_size, batch_size) at a time
    for name, seq in records:
        part.append((name, seq))
        mini += len(seq)
        if mini >= mini_batch:
            part_bases += mini
            mini = 0
            if part_bases > batch_bases:
                yield part
                part, part_bases = [], 0
    if part:
        yield part


def load_target(path, k, w):
    """One target path -> mapper.Index: a saved index is loaded (megapath_nano.py:1641-1645), sequences are indexed.
    The path is opened exactly once (it is a FIFO in the reference's human/decoy call, aligner.py:143-144)."""
    kind, stream = fastx.open_once(path)
    try:
        if kind != 'index':
            return mapper.Index([(n, sq) for n, sq, _ in fastx.iter_fastx(stream)], k=k, w=w)
        if not fastx.is_fifo(path):
            stream.close()
            return mapper.Index.load(path)
        import tempfile
        with tempfile.NamedTemporaryFile(suffix='.mpi') as tmp:  # mpn_index_load wants a seekable file
            for chunk in iter(lambda: stream.read(1 << 24), b''):
                tmp.write(chunk)
            tmp.flush()
            return mapper.Index.load(tmp.name)
    finally:
        if not stream.closed:
            stream.close()


def _is_saved_index(paths):
    if len(paths) != 1 or fastx.is_fifo(paths[0]):
        return False
    try:
        with open(paths[0], 'rb') as f:
            return f.read(8) == fastx.INDEX_MAGIC
    except OSError:
        return False


# ---- reads ---------------------------------------------------------------------------------------------------------------
def iter_read_batches(query_paths, batch_bases=200_000_000, device=None):
    """PackedReads of at most batch_bases each, qualities included, in file order."""
    names, seqs, quals, acc = [], [], [], 0

    def flush():
        return mapper.PackedReads(names, seqs, device=device, quals=quals)
    for path in query_paths:
        kind, stream = fastx.open_once(path)
        try:
            if kind == 'index':
                raise ValueError(f'{path}: a saved index is not a read file')
            for name, seq, qual in fastx.iter_fastx(stream):
                if names and acc + len(seq) > batch_bases:
                    yield flush()
                    names, seqs, quals, acc = [], [], [], 0
                names.append(name)
                seqs.append(seq)
                quals.append(qual)
                acc += len(seq)
        finally:
            stream.close()
    if names:
        yield flush()


# ---- the engine shared by Align() and bin/mpn-aligner ----------------------------------------------------------------------
class MappedBatch:
    """What one batch of reads produced: text(s), integer columns, and the target table the column `rid` indexes."""

    def __init__(self, packed, paf, sam, cols, target_names, target_lens):
        self.packed, self.paf, self.sam, self.cols = packed, paf, sam, cols
        self.target_names, self.target_lens = target_names, target_lens


class _ReadTable:
    """What stays of a batch of reads once its text has been handed over: names and lengths (the sequences leave memory)."""

    def __init__(self, names, lens):
        self.names, self.lens, self.n = names, lens, len(names)


class _Results(list):
    """The MappedBatch list of map_files.  With a sink every batch is handed over as soon as it is final and its text is
    dropped (a run of configs[2]'s size has tens of GB of SAM text); the integer columns stay."""

    def __init__(self, on_batch=None, on_header=None):
        super().__init__()
        self.on_batch, self.on_header, self.header_sent = on_batch, on_header, False

    def header(self, text):
        if self.on_header and not self.header_sent and text is not None:
            self.on_header(text)
            self.header_sent = True

    def append(self, b):
        if self.on_batch:
            self.on_batch(b)
            b.paf = b.sam = None
            b.packed = _ReadTable(b.packed.names, b.packed.lens)   # (the table of the batch's rows needs names and lengths only)
        super().append(b)


def map_files(target_paths, query_paths, options, want_paf=True, want_sam=False, want_cols=True, read_batch_bases=200_000_000,
              save_index=None, cache_key=None, on_batch=None, on_header=None):
    """Map the reads of query_paths against the targets.  -> (list of MappedBatch, SAM header text or None).
    on_header(text) / on_batch(MappedBatch): called with the SAM header before the first batch and with every batch as
    soon as its text is final; the text is then not kept.

    Without `--split-prefix` and with more than one index part minimap2 reports every part on its own; the batches are
    then returned part by part, in that order."""
    opt = options.opt
    out_opt = mapper.MapOpt.from_buffer_copy(bytes(opt))
    out_opt.out_sam = (2 if want_paf else 1) if want_sam else 0
    text_wanted = want_paf or want_sam
    results, header = _Results(on_batch, on_header), None

    def map_single(idx, b, names, lens):
        text, sam, cols = mapper.map_batch_full(idx, out_opt, b, want_paf=text_wanted, want_cols=want_cols)
        if want_sam and not want_paf:
            text, sam = None, text
        results.append(MappedBatch(b, text, sam, cols, names, lens))

    def stream_one(idx):
        """The whole target set is this ONE index (the usual case on an accelerator whose memory holds tens of Gbp of targets):
        read batches are read, mapped, handed over and dropped one at a time -- a run's FASTQ never sits in memory as a whole."""
        nonlocal header
        names, lens = np.array(idx.names, dtype=object), idx.lens
        n_batches = 0
        for b in iter_read_batches(query_paths, read_batch_bases):
            n_batches += 1
            if options.split:       # --split-prefix: minimap2 takes its merge path even for one part
                h = mapper.Hits(b)
                try:
                    h.add_part(idx, opt)
                    if n_batches == 1 and want_sam:
                        header = h.sam_header()
                        results.header(header)
                    _finish_hits([h], out_opt, want_paf, want_sam, want_cols, results)
                finally:
                    h.close()
            else:
                if n_batches == 1 and want_sam:
                    header = idx.sam_header()
                    results.header(header)
                map_single(idx, b, names, lens)
        if n_batches == 0 and want_sam:
            header = idx.sam_header()
            results.header(header)

    def all_parts(parts):
        """Several index parts that are built (or loaded) one after another and dropped again: every read batch meets every part,
        so the batches are kept (their hits accumulate per batch: mapper.Hits) while the parts stream by."""
        nonlocal header
        batches = list(iter_read_batches(query_paths, read_batch_bases))
        hits = [mapper.Hits(b) for b in batches] if options.split else None
        headers, n_parts = [], 0
        try:
            for idx in parts:
                n_parts += 1
                if hits is not None:
                    for h in hits:
                        h.add_part(idx, opt)
                else:   # without --split-prefix minimap2 reports every part on its own
                    if want_sam:
                        header = idx.sam_header()
                        results.header(header)
                        headers.append(header)
                    names, lens = np.array(idx.names, dtype=object), idx.lens
                    for b in batches:
                        map_single(idx, b, names, lens)
                idx.close()
            if hits is not None:
                if want_sam:
                    header = hits[0].sam_header() if hits else ''
                    results.header(header)
                _finish_hits(hits, out_opt, want_paf, want_sam, want_cols, results)
            elif headers:
                header = headers[0]
        finally:
            for h in hits or []:
                h.close()

    cached = _INDEX_CACHE.get(cache_key) if cache_key else None
    if cached is not None:
        stream_one(cached)
        return results, header
    if _is_saved_index(target_paths):
        # a saved index may hold several parts (minimap2 -d dumps every part of a -I split into the one file): they are loaded
        # one at a time, like the parts of a FASTA target
        first, nxt = mapper.Index.load_at(target_paths[0], 0)
        if nxt < 0:
            stream_one(first)
            if cache_key:
                _INDEX_CACHE[cache_key] = first
            else:
                first.close()
        else:
            def rest():
                off = nxt
                yield first
                while off >= 0:
                    part, off = mapper.Index.load_at(target_paths[0], off)
                    yield part
            all_parts(rest())
        return results, header

    raw_parts = iter_target_parts(iter_target_records(target_paths), options.batch_bases)
    n_built = 0

    def build(part):
        nonlocal n_built
        idx = mapper.Index(part, k=options.k, w=options.w)
        n_built += 1
        if save_index:
            # minimap2 -d FILE dumps every part into the one file (bin/megapath_nano.py:1641-1645): so does this
            idx.save(save_index, append=n_built > 1)
        return idx

    first = next(raw_parts, None)
    if first is None:
        all_parts(iter(()))
        return results, header
    idx = build(first)
    del first
    second = next(raw_parts, None)    # (the target stream is read on while the first index is resident: only its raw records are held)
    if second is None:
        stream_one(idx)
        if cache_key:
            _INDEX_CACHE[cache_key] = idx
        else:
            idx.close()
        return results, header

    def built():
        nonlocal second
        yield idx
        part = second
        second = None
        while part is not None:
            nxt_idx = build(part)
            del part
            yield nxt_idx
            part = next(raw_parts, None)
    all_parts(built())
    return results, header


def _finish_hits(hits, out_opt, want_paf, want_sam, want_cols, results):
    for h in hits:
        text, sam, cols = h.finish(out_opt, want_paf=want_paf or want_sam, want_cols=want_cols)
        if want_sam and not want_paf:
            text, sam = None, text
        names, lens = h.targets()
        results.append(MappedBatch(h.packed, text, sam, cols, np.array(names, dtype=object), lens))


# ---- Align() ---------------------------------------------------------------------------------------------------------------
def _count_given(*tables_and_columns):
    return sum(1 for table, column in tables_and_columns if table is not None and table[column].shape[0] > 0)


def _paths_of_assemblies(assembly_metadata, assembly_list, folder, missing_message, need_all=False):
    table = assembly_metadata.get_assembly_path(assembly_list=assembly_list)
    if table is None or (need_all and table.shape[0] != assembly_list['assembly_id'].shape[0]):
        os.sys.exit(missing_message)
    table['path'] = [os.path.join(folder, p) for p in table['path']]
    return table


def _resolve_targets(assembly_metadata, global_options, target_filename_list, target_assembly_list, align_concat_fa, module_option):
    """-> list of target paths (exit messages of aligner.py:120,126,132,140)."""
    if align_concat_fa:
        if module_option == 'amplicon_filter_module':
            return [str(target_filename_list['path'][0])]
        nano_dir = global_options.get('nano_dir', os.getcwd())
        return [f'{nano_dir}/genomes/refseq/refseq.fna.gz']
    if _count_given((target_filename_list, 'path'), (target_assembly_list, 'assembly_id')) != 1:
        os.sys.exit('Exactly one of target_filename_list and target_assembly_list must be specified')
    by_assembly = target_assembly_list is not None and len(target_assembly_list['assembly_id']) > 0
    if by_assembly:
        target_filename_list = _paths_of_assemblies(assembly_metadata, target_assembly_list, global_options['assembly_folder'],
                                                    'Target assembly_id not found')
        lengths = assembly_metadata.get_assembly_length(assembly_list=target_assembly_list)
        if lengths is None:
            os.sys.exit('Target assembly_length not found')
        n_lengths = lengths.shape[0]
    else:
        n_lengths = target_filename_list.shape[0]
    if n_lengths != target_filename_list.shape[0]:
        os.sys.exit('Number of target_assembly_length does not match number of target_filename_list')
    return [str(p) for p in target_filename_list['path']]


def _resolve_queries(assembly_metadata, global_options, query_filename_list, query_assembly_list):
    """-> list of query paths (exit messages of aligner.py:153,157,164)."""
    if _count_given((query_filename_list, 'path'), (query_assembly_list, 'assembly_id')) != 1:
        os.sys.exit('Exactly one of query_filename_list and query_assembly_list must be specified')
    if query_assembly_list is not None and query_assembly_list['assembly_id'].shape[0] > 0:
        query_filename_list = _paths_of_assemblies(assembly_metadata, query_assembly_list, global_options['assembly_folder'],
                                                   'Query assembly_id not found', need_all=True)
    paths = [str(p) for p in query_filename_list['path']]
    for p in paths:
        if not os.path.isfile(p):
            os.sys.exit('Query file ' + p + ' not exists')
    return paths


def _run_amr(global_options, bam_filename, amr_output_folder, log_file):
    """aligner.py:250-256 starts the AMR module (megapath_nano_amr.py, outside this path: SURVEY section 8 scope) on the BAM.
    A caller may hand over its own hook (global_options['amr_hook'](bam, folder)); otherwise the reference's script is run
    when the deployment has it (global_options['nano_bin_dir']), and its absence is reported instead of hidden."""
    hook = global_options.get('amr_hook')
    if callable(hook):
        hook(bam_filename, amr_output_folder)
        return
    import subprocess
    script = os.path.join(global_options.get('nano_bin_dir', ''), 'megapath_nano_amr.py')
    if os.path.isfile(script):
        subprocess.Popen(['python', script, '--query_bam', bam_filename, '--output_folder', str(amr_output_folder), '--threads',
                          str(global_options.get('AMRThreadOption', 1))], stderr=log_file if hasattr(log_file, 'fileno') else None).wait()
    else:
        print(f'AMR module not started: {script or "megapath_nano_amr.py"} not found (give global_options[\'nano_bin_dir\'] or amr_hook)',
              file=os.sys.stderr)


def _frame_of(batch):
    c = batch.cols
    i64 = lambda a: a.astype(np.int64)  # noqa: E731
    names = np.array(batch.packed.names, dtype=object)
    return pandas.DataFrame({
        'read_id': names[c['read_idx']], 'read_length': i64(batch.packed.lens[c['read_idx']]), 'read_from': i64(c['qs']),
        'read_to': i64(c['qe']), 'strand': np.where(c['rev'] != 0, '-', '+'), 'sequence_id': batch.target_names[c['rid']],
        'sequence_length': i64(batch.target_lens[c['rid']]), 'sequence_from': i64(c['rs']), 'sequence_to': i64(c['re']),
        'match': i64(c['mlen']), 'alignment_block_length': i64(c['blen']), 'mapq': i64(c['mapq']), 'edit_dist': i64(c['nm']),
        'alignment_score': i64(c['as_'])})


def _join_taxonomy(align_list, sequence_tax):
    """Inner join on sequence_id with a many-to-one check (aligner.py:317-332), row order of align_list kept."""
    keys = pandas.Index(sequence_tax['sequence_id'])
    if keys.has_duplicates:
        raise pandas.errors.MergeError('Merge keys are not unique in right dataset; not a many-to-one merge')
    where = keys.get_indexer(align_list['sequence_id'])
    hit = where >= 0
    if not hit.all():
        print('Some sequence id cannot be matched', file=os.sys.stderr)
    joined = align_list[hit].copy()
    for col in ('assembly_id', 'tax_id', 'species_tax_id', 'genus_tax_id'):
        joined[col] = sequence_tax[col].to_numpy()[where[hit]]
    return joined


def Align(*, assembly_metadata, global_options, temp_dir_name, log_file, query_filename_list=None,
          query_assembly_list=None, target_filename_list=None, target_assembly_list=None, aligner_options=None,
          paf_path_and_prefix=None, mapping_only=False, module_option='', AMR_output_folder='', align_concat_fa=False,
          batch_bases=200_000_000):
    target_paths = _resolve_targets(assembly_metadata, global_options, target_filename_list, target_assembly_list,
                                    align_concat_fa, module_option)
    query_paths = _resolve_queries(assembly_metadata, global_options, query_filename_list, query_assembly_list)
    # the tiebreaker stream: Python's generator seeded with the md5 of the query basenames (aligner.py:160-168)
    random.seed(hashlib.md5(''.join(os.path.split(q)[1] for q in query_paths).encode()).hexdigest())

    output_paf = not (paf_path_and_prefix is None or paf_path_and_prefix == '')
    # the reference adds -a whenever a PAF prefix is given (aligner.py:190-191), and minimap2's -a implies base-level
    # alignment: with a prefix the CIGARs are computed even for mapping_only
    options = AlignerOptions(aligner_options, mapping_only and not output_paf)
    want_sam = output_paf
    regular = all(os.path.isfile(p) for p in target_paths)
    cache_key = (tuple(target_paths), options.k, options.w, options.batch_bases) if regular else None
    if output_paf:
        # PAF and SAM go to disk batch by batch: nothing but the integer columns of a batch is kept in memory
        with open(f'{paf_path_and_prefix}.paf', 'w') as paf_f, open(f'{paf_path_and_prefix}.sam', 'w') as sam_f:
            batches, header = map_files(target_paths, query_paths, options, want_paf=True, want_sam=True, want_cols=True,
                                        read_batch_bases=batch_bases, cache_key=cache_key, on_header=sam_f.write,
                                        on_batch=lambda b: (paf_f.write(b.paf), sam_f.write(b.sam)))
        print('Finished species alignment step.')                                                        # aligner.py:244
        from . import abundance, bam
        # samtools view -F<flags> -b | samtools sort; samtools index (aligner.py:245-252): 1796 = unmapped | secondary | QC fail |
        # duplicate; the amplicon filter keeps everything that is mapped
        exclude = 4 if module_option == 'amplicon_filter_module' else 1796
        bam.sam_to_sorted_bam(f'{paf_path_and_prefix}.sam', f'{paf_path_and_prefix}.bam', exclude_flags=exclude,
                              sort_keys=abundance.device_sort_order)   # the coordinate sort runs on the GPU (mpn_sort_order)
        if module_option in ('taxon_and_AMR_module', 'AMR_module_only'):                                # aligner.py:250-256
            _run_amr(global_options, f'{paf_path_and_prefix}.bam', AMR_output_folder, log_file)
        if module_option in ('AMR_module_only', 'amplicon_filter_module'):                               # aligner.py:257-259
            os.sys.exit()
    else:
        batches, header = map_files(target_paths, query_paths, options, want_paf=False, want_sam=False, want_cols=True,
                                    read_batch_bases=batch_bases, cache_key=cache_key)

    frames = [_frame_of(b) for b in batches if len(b.cols['read_idx'])]
    if frames:
        table = pandas.concat(frames, ignore_index=True)[list(list_col)]
    else:
        table = pandas.DataFrame({c: pandas.Series(dtype=(str if c in _TEXT_COLS else np.int64)) for c in list_col})
    keep = (table['sequence_length'].to_numpy() > 0) & (table['alignment_score'].to_numpy() >= global_options['min_alignment_score'])
    align_list = table[keep]                                                                          # aligner.py:311-312
    if target_assembly_list is not None and len(target_assembly_list['assembly_id']) > 0:
        align_list = _join_taxonomy(align_list, assembly_metadata.get_sequence_tax_id(assembly_list=target_assembly_list))
    # one random.random() per surviving row, in row order (aligner.py:334-335)
    return align_list.assign(alignment_score_tiebreaker=random_block(random, align_list.shape[0]))
