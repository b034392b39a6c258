"""ctypes wrapper over include/mpn_map.h: index build, stage entry points and the batch mapper."""
import ctypes as ct
import os

import numpy as np

from . import _ffi


class MapOpt(ct.Structure):
    _fields_ = [('k', ct.c_int32), ('w', ct.c_int32), ('mid_occ_frac', ct.c_float), ('mid_occ', ct.c_int32),
                ('max_gap', ct.c_int32), ('bw', ct.c_int32), ('max_chain_skip', ct.c_int32), ('max_chain_iter', ct.c_int32),
                ('min_cnt', ct.c_int32), ('min_chain_score', ct.c_int32), ('mask_level', ct.c_float),
                ('pri_ratio', ct.c_float), ('best_n', ct.c_int32), ('max_join_long', ct.c_int32),
                ('max_join_short', ct.c_int32), ('min_join_flank_sc', ct.c_int32), ('min_join_flank_ratio', ct.c_float),
                ('a', ct.c_int32), ('b', ct.c_int32), ('q', ct.c_int32), ('e', ct.c_int32), ('q2', ct.c_int32),
                ('e2', ct.c_int32), ('sc_ambi', ct.c_int32), ('zdrop', ct.c_int32), ('zdrop_inv', ct.c_int32),
                ('end_bonus', ct.c_int32), ('min_dp_max', ct.c_int32), ('min_ksw_len', ct.c_int32),
                ('max_clip_ratio', ct.c_float), ('max_sw_mat', ct.c_int64), ('with_cigar', ct.c_int32),
                ('seed', ct.c_uint32), ('host_threads', ct.c_int32), ('out_sam', ct.c_int32)]


COL_NAMES = ('read_idx', 'qs', 'qe', 'rev', 'rid', 'rs', 're', 'mlen', 'blen', 'mapq', 'nm', 'as_', 'primary')


class AlnCols(ct.Structure):
    _fields_ = [('cap', ct.c_int64), ('n_rows', ct.c_int64)] + [(n, ct.c_void_p) for n in COL_NAMES]


_bound = False


def _bind():
    global _bound
    lib = _ffi.lib()
    if not _bound:
        P = ct.c_void_p
        lib.mpn_map_opt_init.argtypes = [ct.POINTER(MapOpt)]
        lib.mpn_map_opt_init.restype = None
        lib.mpn_index_build.argtypes = [ct.c_int32, ct.POINTER(ct.c_char_p), ct.POINTER(ct.c_char_p), P, ct.c_int32, ct.c_int32]
        lib.mpn_index_build.restype = P
        lib.mpn_index_build_device.argtypes = [ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, ct.c_int32, ct.c_int32]
        lib.mpn_index_build_device.restype = P
        lib.mpn_index_destroy.argtypes = [P]
        lib.mpn_index_destroy.restype = None
        lib.mpn_index_n_minimizers.argtypes = [P]
        lib.mpn_index_n_minimizers.restype = ct.c_int64
        lib.mpn_index_n_keys.argtypes = [P]
        lib.mpn_index_n_keys.restype = ct.c_int64
        lib.mpn_index_mid_occ.argtypes = [P, ct.c_float]
        lib.mpn_index_mid_occ.restype = ct.c_int32
        lib.mpn_index_fetch_seq.argtypes = [P, ct.c_int32, ct.c_int64, ct.c_int64, ct.c_char_p]
        lib.mpn_index_fetch_seq.restype = ct.c_int64
        lib.mpn_index_export.argtypes = [P, P, P, P]
        lib.mpn_index_export.restype = ct.c_int
        lib.mpn_sam_header.argtypes = [P, ct.c_char_p, ct.c_char_p, ct.c_int64]
        lib.mpn_sam_header.restype = ct.c_int64
        lib.mpn_index_save.argtypes = [P, ct.c_char_p]
        lib.mpn_index_save.restype = ct.c_int
        lib.mpn_index_load.argtypes = [ct.c_char_p]
        lib.mpn_index_load.restype = P
        lib.mpn_index_save_append.argtypes = [P, ct.c_char_p]
        lib.mpn_index_save_append.restype = ct.c_int
        lib.mpn_index_load_at.argtypes = [ct.c_char_p, ct.c_int64, ct.POINTER(ct.c_int64)]
        lib.mpn_index_load_at.restype = P
        for fn in ('mpn_index_n_seq', 'mpn_index_k', 'mpn_index_w'):
            getattr(lib, fn).argtypes = [P]
            getattr(lib, fn).restype = ct.c_int32
        lib.mpn_index_seq_len.argtypes = [P, ct.c_int32]
        lib.mpn_index_seq_len.restype = ct.c_int32
        lib.mpn_index_seq_name.argtypes = [P, ct.c_int32, ct.c_char_p, ct.c_int32]
        lib.mpn_index_seq_name.restype = ct.c_int32
        lib.mpn_sketch_batch.argtypes = [ct.c_int32, P, P, P, ct.c_int32, ct.c_int32, P, P, ct.c_int64]
        lib.mpn_sketch_batch.restype = ct.c_int64
        lib.mpn_seed_chain_batch.argtypes = [P, ct.POINTER(MapOpt), ct.c_int32, P, P, P, P, P, P, P, ct.c_int64, P, P, ct.c_int64]
        lib.mpn_seed_chain_batch.restype = ct.c_int
        if True:
            lib.mpn_map_batch.argtypes = [P, ct.POINTER(MapOpt), ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, P, ct.c_int64]
            lib.mpn_map_batch.restype = ct.c_int64
        lib.mpn_map_batch_ex.argtypes = [P, ct.POINTER(MapOpt), ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, P, P, P, P,
                                         ct.c_int64, ct.POINTER(AlnCols)]
        lib.mpn_map_batch_ex.restype = ct.c_int64
        lib.mpn_map_batch_q.argtypes = [P, ct.POINTER(MapOpt), ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, P, P, P, P, P,
                                        ct.c_int64, ct.POINTER(AlnCols)]
        lib.mpn_map_batch_q.restype = ct.c_int64
        lib.mpn_hits_create.argtypes = [ct.c_int32]
        lib.mpn_hits_create.restype = P
        lib.mpn_hits_destroy.argtypes = [P]
        lib.mpn_hits_destroy.restype = None
        for fn in ('mpn_hits_n_seq', 'mpn_hits_n_parts'):
            getattr(lib, fn).argtypes = [P]
            getattr(lib, fn).restype = ct.c_int32
        lib.mpn_hits_seq_len.argtypes = [P, ct.c_int32]
        lib.mpn_hits_seq_len.restype = ct.c_int32
        lib.mpn_hits_seq_name.argtypes = [P, ct.c_int32, ct.c_char_p, ct.c_int32]
        lib.mpn_hits_seq_name.restype = ct.c_int32
        lib.mpn_hits_sam_header.argtypes = [P, ct.c_char_p, ct.c_char_p, ct.c_int64]
        lib.mpn_hits_sam_header.restype = ct.c_int64
        lib.mpn_map_batch_part.argtypes = [P, ct.POINTER(MapOpt), ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, P, P, P, P]
        lib.mpn_map_batch_part.restype = ct.c_int
        lib.mpn_map_batch_parts.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int32, ct.POINTER(MapOpt), ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, P, P, P, P]
        lib.mpn_map_batch_parts.restype = ct.c_int
        lib.mpn_hits_set_text.argtypes = [P, ct.c_int32]
        lib.mpn_hits_set_text.restype = None
        lib.mpn_hits_finish.argtypes = [P, ct.POINTER(MapOpt), ct.c_int32, ct.POINTER(ct.c_char_p), P, P, P, P, P, ct.c_int64,
                                        ct.POINTER(AlnCols)]
        lib.mpn_hits_finish.restype = ct.c_int64
        lib.mpn_map_fetch_sam.argtypes = [ct.c_char_p, ct.c_int64]
        lib.mpn_map_fetch_sam.restype = ct.c_int64
        lib.mpn_map_fetch_cols.argtypes = [ct.POINTER(AlnCols)]
        lib.mpn_map_fetch_cols.restype = ct.c_int64
        lib.mpn_map_fetch_text.argtypes = [ct.c_char_p, ct.c_int64]
        lib.mpn_map_fetch_text.restype = ct.c_int64
        lib.mpn_ext_dp_batch.argtypes = [ct.POINTER(MapOpt), ct.c_int32, P, P, P, P, P, P, P, P, P, P, ct.c_int32, P, P, ct.c_int64, P]
        lib.mpn_ext_dp_batch.restype = ct.c_int
        lib.mpn_map_last_stats.argtypes = [P]
        lib.mpn_map_last_stats.restype = None
        lib.mpn_map_last_stats_ex.argtypes = [P, ct.c_int32]
        lib.mpn_map_last_stats_ex.restype = ct.c_int32
        _bound = True
    return lib


def default_opt(**kw):
    o = MapOpt()
    _bind().mpn_map_opt_init(ct.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def pack_seqs(seqs):
    """list of bytes / uint8 arrays -> (concatenated uint8 buffer, offsets int64, lengths int32)"""
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    off = np.zeros(len(seqs), dtype=np.int64)
    if len(seqs) > 1:
        off[1:] = np.cumsum(lens[:-1].astype(np.int64))
    total = int(lens.astype(np.int64).sum())
    buf = np.zeros(total + 16, dtype=np.uint8)
    for s, o in zip(seqs, off):
        a = np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.asarray(s, dtype=np.uint8)
        buf[o:o + len(a)] = a
    return buf, off, lens


class Index:
    """Target sequences + minimizer index resident in HBM (mpn_index_build), or loaded from a file written by save()."""

    MAGIC = b'MPNIDX01'

    def __init__(self, genomes, k=15, w=10):
        lib = _bind()
        n = len(genomes)
        self.names = [g[0] for g in genomes]
        seqs_b = [bytes(g[1]) if not isinstance(g[1], bytes) else g[1] for g in genomes]  # alive for the call only
        self.lens = np.array([len(s) for s in seqs_b], dtype=np.int32)
        names = (ct.c_char_p * n)(*[x.encode() for x in self.names])
        seqs = (ct.c_char_p * n)(*seqs_b)
        self.k, self.w = k, w
        self.h = lib.mpn_index_build(n, names, seqs, self.lens.ctypes.data, k, w)
        if not self.h:
            raise _ffi.MpnError('mpn_index_build failed: ' + _ffi.hint(_ffi.last_error()))

    @classmethod
    def from_device(cls, names, d_seqs_ptr, lens, k=15, w=10):
        """Index of targets that are resident in HBM as concatenated ASCII (device pointer; target i at sum(lens[:i]))."""
        lib = _bind()
        self = cls.__new__(cls)
        n = len(names)
        self.names = list(names)
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum(self.lens.astype(np.int64))
        cn = (ct.c_char_p * n)(*[x.encode() for x in self.names])
        self.k, self.w = k, w
        self.h = lib.mpn_index_build_device(n, cn, d_seqs_ptr, off.ctypes.data, self.lens.ctypes.data, k, w)
        if not self.h:
            raise _ffi.MpnError('mpn_index_build_device failed: ' + _ffi.hint(_ffi.last_error()))
        return self

    def sam_header(self, cmdline=None):
        cap = 64 * len(self.names) + sum(len(n) for n in self.names) + 4096 + (len(cmdline) if cmdline else 0)
        buf = ct.create_string_buffer(cap)
        r = _bind().mpn_sam_header(self.h, cmdline.encode() if cmdline else None, buf, cap)
        if r < 0:
            raise _ffi.MpnError(f'mpn_sam_header rc={r}')
        return buf.raw[:r].decode()

    def save(self, path, append=False):
        """Persistent form (minimap2 `-d FILE`): load() gives back an index that maps identically.  append=True adds this
        index as a further PART of the target set the file holds (minimap2 dumps all parts of a -I split into the one file)."""
        lib = _bind()
        _ffi.check((lib.mpn_index_save_append if append else lib.mpn_index_save)(self.h, os.fsencode(path)), 'mpn_index_save')

    @classmethod
    def load(cls, path):
        """The (first) index part of a saved file."""
        idx, _ = cls.load_at(path, 0)
        return idx

    @classmethod
    def iter_parts(cls, path):
        """The index parts of a saved file, one at a time (the caller closes each before asking for the next)."""
        off = 0
        while off >= 0:
            idx, off = cls.load_at(path, off)
            yield idx

    @classmethod
    def load_at(cls, path, offset):
        """-> (the part that starts at byte `offset`, offset of the next part or -1)"""
        lib = _bind()
        nxt = ct.c_int64(-1)
        h = lib.mpn_index_load_at(os.fsencode(path), int(offset), ct.byref(nxt))
        if not h:
            raise _ffi.MpnError('mpn_index_load failed: ' + _ffi.last_error())
        self = cls._from_handle(h)
        return self, int(nxt.value)

    @classmethod
    def _from_handle(cls, h):
        lib = _bind()
        self = cls.__new__(cls)
        self.h = h
        n = lib.mpn_index_n_seq(h)
        buf = ct.create_string_buffer(1 << 16)
        self.names, lens = [], []
        for i in range(n):
            lib.mpn_index_seq_name(h, i, buf, len(buf))
            self.names.append(buf.value.decode())
            lens.append(lib.mpn_index_seq_len(h, i))
        self.lens = np.array(lens, dtype=np.int32)
        self.k, self.w = lib.mpn_index_k(h), lib.mpn_index_w(h)
        return self

    @classmethod
    def is_index_file(cls, path):
        try:
            with open(path, 'rb') as f:
                return f.read(8) == cls.MAGIC
        except OSError:
            return False

    @property
    def n_minimizers(self):
        return _bind().mpn_index_n_minimizers(self.h)

    @property
    def n_keys(self):
        return _bind().mpn_index_n_keys(self.h)

    def mid_occ(self, f=2e-4):
        return _bind().mpn_index_mid_occ(self.h, f)

    def fetch_seq(self, i, start, length):
        """bases [start, start+length) of target i as bytes, decoded from the packed targets in HBM"""
        buf = ct.create_string_buffer(int(length) + 1)
        r = _bind().mpn_index_fetch_seq(self.h, int(i), int(start), int(length), buf)
        if r < 0:
            raise _ffi.MpnError(f'mpn_index_fetch_seq rc={r}: {_ffi.last_error()}')
        return buf.raw[:r]

    def export(self):
        keys = np.zeros(self.n_keys, dtype=np.uint64)
        key_off = np.zeros(self.n_keys + 1, dtype=np.int64)
        pos = np.zeros(self.n_minimizers, dtype=np.uint64)
        _ffi.check(_bind().mpn_index_export(self.h, keys.ctypes.data, key_off.ctypes.data, pos.ctypes.data), 'mpn_index_export')
        return keys, key_off, pos

    def close(self):
        if self.h:
            _bind().mpn_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sketch_batch(seqs, k=15, w=10):
    """-> list of (n_i, 2) uint64 arrays, one per sequence"""
    lib = _bind()
    buf, off, lens = pack_seqs(seqs)
    n = len(seqs)
    mz_off = np.zeros(n + 1, dtype=np.int64)
    cap = int(lens.astype(np.int64).sum()) + 16
    mz = np.zeros((cap, 2), dtype=np.uint64)
    r = lib.mpn_sketch_batch(n, buf.ctypes.data, off.ctypes.data, lens.ctypes.data, k, w, mz_off.ctypes.data, mz.ctypes.data, cap)
    if r < 0:
        raise _ffi.MpnError(f'mpn_sketch_batch rc={r}: {_ffi.last_error()}')
    return [mz[mz_off[i]:mz_off[i + 1]].copy() for i in range(n)]


def seed_chain_batch(idx, opt, seqs):
    """-> list of dict(n_anchor, rep_len, u (uint64[n_chain]), b ((n,2) uint64)) per read"""
    lib = _bind()
    buf, off, lens = pack_seqs(seqs)
    n = len(seqs)
    n_anchor = np.zeros(n, dtype=np.int64)
    rep_len = np.zeros(n, dtype=np.int32)
    chain_off = np.zeros(n + 1, dtype=np.int64)
    anchor_off = np.zeros(n + 1, dtype=np.int64)
    u_cap, b_cap = 1 << 16, 1 << 20
    while True:
        u = np.zeros(u_cap, dtype=np.uint64)
        b = np.zeros((b_cap, 2), dtype=np.uint64)
        rc = lib.mpn_seed_chain_batch(idx.h, ct.byref(opt), n, buf.ctypes.data, off.ctypes.data, lens.ctypes.data,
                                      n_anchor.ctypes.data, rep_len.ctypes.data, chain_off.ctypes.data, u.ctypes.data, u_cap,
                                      anchor_off.ctypes.data, b.ctypes.data, b_cap)
        if rc == -3:
            u_cap, b_cap = max(u_cap, int(chain_off[n]) + 1), max(b_cap, int(anchor_off[n]) + 1)
            continue
        _ffi.check(rc, 'mpn_seed_chain_batch')
        break
    return [dict(n_anchor=int(n_anchor[i]), rep_len=int(rep_len[i]), u=u[chain_off[i]:chain_off[i + 1]].copy(),
                 b=b[anchor_off[i]:anchor_off[i + 1]].copy()) for i in range(n)]


def map_batch(idx, opt, names, seqs):
    """-> PAF text of the whole batch (reads in input order)"""
    packed = PackedReads(names, seqs)
    return map_batch_ex(idx, opt, packed, want_paf=True, want_cols=False)[0]


class PackedReads:
    """Reads packed for mpn_map_batch_ex; optionally also resident in HBM (torch uint8/int64/int32 tensors)."""

    def __init__(self, names, seqs, device=None, quals=None):
        """quals: per read a quality string of the read's length, or None (then the SAM QUAL column is '*')"""
        self.n = len(seqs)
        self.names = list(names)
        self.buf, self.off, self.lens = pack_seqs(seqs)
        self.qbuf = None
        if quals is not None and any(q is not None for q in quals):
            # reads without qualities in a batch that has some: the '*' cannot be mixed per read, they get '!' (Phred 0)
            self.qbuf = pack_seqs([q if q is not None else b'!' * int(l) for q, l in zip(quals, self.lens)])[0]
        self._finish(device)

    @classmethod
    def from_arrays(cls, names, buf, off, lens, dev=None):
        """buf: concatenated ASCII (uint8, padded to a 4-byte multiple past the last base), off int64, lens int32 (host
        numpy); dev: the same three as torch tensors already resident in HBM, or None."""
        self = cls.__new__(cls)
        self.n = len(lens)
        self.names = list(names)
        self.buf = np.ascontiguousarray(buf, dtype=np.uint8)
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        self.qbuf = None
        self._finish(None)
        self.dev = dev
        return self

    def _finish(self, device):
        self.cnames = (ct.c_char_p * self.n)(*[x.encode() for x in self.names])
        self.bases = int(self.lens.astype(np.int64).sum())
        self.dev = None
        if device is not None:
            import torch
            self.dev = (torch.from_numpy(self.buf).to(device), torch.from_numpy(self.off).to(device),
                        torch.from_numpy(self.lens).to(device))
            torch.cuda.synchronize(device)

    def seq(self, i):
        return self.buf[self.off[i]:self.off[i] + self.lens[i]]


_rows_per_read = 2.0  # running estimate used to size the column arrays (a short guess costs a copy, not a second mapping)


def _emit(call, opt, packed, want_paf, want_cols):
    """Shared buffer handling of mpn_map_batch_q / mpn_hits_finish: call(text_buf, text_cap, cols_ref) -> rc.
    -> (text or None, SAM text or None (opt.out_sam == 2), column dict or None)"""
    global _rows_per_read
    lib = _bind()
    n = packed.n
    per_base = 3 if opt.out_sam == 1 else 0.5
    paf_cap = int(packed.bases * per_base) + 512 * n + 4096 if want_paf else 0
    rows_cap = max(64, int(n * _rows_per_read * 1.25) + 16)

    def make_cols(cap):
        cols, arrs = AlnCols(), {}
        cols.cap = cap
        for c in COL_NAMES:
            arrs[c] = np.empty(cap, dtype=np.int32)
            setattr(cols, c, arrs[c].ctypes.data)
        return cols, arrs

    out = ct.create_string_buffer(paf_cap) if want_paf else None
    cols, arrs = make_cols(rows_cap) if want_cols else (None, None)
    r = call(out, paf_cap, ct.byref(cols) if want_cols else None)
    if r == -3:  # the library kept what did not fit: fetch it, do not map again
        if want_cols and cols.n_rows > rows_cap:
            cols, arrs = make_cols(int(cols.n_rows))
            if lib.mpn_map_fetch_cols(ct.byref(cols)) < 0:
                raise _ffi.MpnError('mpn_map_fetch_cols: ' + _ffi.last_error())
        need = lib.mpn_map_fetch_text(None, 0) if want_paf else -1
        if need > 0:
            out = ct.create_string_buffer(need)
            r = lib.mpn_map_fetch_text(out, need)
            if r < 0:
                raise _ffi.MpnError('mpn_map_fetch_text: ' + _ffi.last_error())
        elif want_paf:
            r = len(out.value)
    elif r < 0:
        raise _ffi.MpnError(f'mapping call rc={r}: {_ffi.last_error()}')
    text = out.raw[:r].decode() if want_paf else None
    sam = None
    if opt.out_sam == 2:
        need = lib.mpn_map_fetch_sam(None, 0)
        if need < 0:
            raise _ffi.MpnError('mpn_map_fetch_sam: ' + _ffi.last_error())
        sbuf = ct.create_string_buffer(need)
        k = lib.mpn_map_fetch_sam(sbuf, need)
        sam = sbuf.raw[:k].decode()
    if want_cols:
        nr = int(cols.n_rows)
        _rows_per_read = max(_rows_per_read, nr / max(n, 1))
        arrs = {k: v[:nr] for k, v in arrs.items()}
    return text, sam, (arrs if want_cols else None)


def _dev_ptrs(packed, use_device):
    return [0, 0, 0] if (packed.dev is None or not use_device) else [t.data_ptr() for t in packed.dev]


def map_batch_full(idx, opt, packed, want_paf=False, want_cols=True, use_device=True):
    """-> (text (PAF; SAM if opt.out_sam == 1) or None, SAM text if opt.out_sam == 2 else None, column dict or None)"""
    lib = _bind()
    d = _dev_ptrs(packed, use_device)
    q = packed.qbuf.ctypes.data if getattr(packed, 'qbuf', None) is not None else None

    def call(out, cap, cols):
        return lib.mpn_map_batch_q(idx.h, ct.byref(opt), packed.n, packed.cnames, packed.buf.ctypes.data, q, packed.off.ctypes.data,
                                   packed.lens.ctypes.data, d[0], d[1], d[2], out, cap, cols)
    return _emit(call, opt, packed, want_paf, want_cols)


def map_batch_ex(idx, opt, packed, want_paf=False, want_cols=True, use_device=True):
    """-> (paf text or None, dict of int32 column arrays or None)"""
    text, _, cols = map_batch_full(idx, opt, packed, want_paf, want_cols, use_device)
    return text, cols


class Hits:
    """Hits of one batch of reads accumulated over the parts of a split index (minimap2 -I), merged by finish()
    the way minimap2 --split-prefix merges its per-part dumps."""

    def __init__(self, packed, want_text=True):
        self.packed = packed
        self.h = _bind().mpn_hits_create(packed.n)
        if not self.h:
            raise _ffi.MpnError('mpn_hits_create: ' + _ffi.last_error())
        if not want_text:   # columns only at finish(): the CIGARs of the parts' hits never leave the GPU
            _bind().mpn_hits_set_text(self.h, 0)

    def add_part(self, idx, opt, use_device=True):
        p = self.packed
        d = _dev_ptrs(p, use_device)
        _ffi.check(_bind().mpn_map_batch_part(idx.h, ct.byref(opt), p.n, p.cnames, p.buf.ctypes.data, p.off.ctypes.data, p.lens.ctypes.data,
                                              d[0], d[1], d[2], self.h), 'mpn_map_batch_part')

    def add_parts(self, parts, opt, use_device=True):
        """Several RESIDENT index parts in one call: reads uploaded once, (sub-batch, part) pairs in one pipeline."""
        p = self.packed
        d = _dev_ptrs(p, use_device)
        arr = (ct.c_void_p * len(parts))(*[i.h for i in parts])
        _ffi.check(_bind().mpn_map_batch_parts(arr, len(parts), ct.byref(opt), p.n, p.cnames, p.buf.ctypes.data, p.off.ctypes.data,
                                               p.lens.ctypes.data, d[0], d[1], d[2], self.h), 'mpn_map_batch_parts')

    def finish(self, opt, want_paf=False, want_cols=True):
        """-> (text, SAM text or None, columns); column `rid` indexes targets()"""
        lib = _bind()
        p = self.packed
        q = p.qbuf.ctypes.data if getattr(p, 'qbuf', None) is not None else None

        def call(out, cap, cols):
            return lib.mpn_hits_finish(self.h, ct.byref(opt), p.n, p.cnames, p.buf.ctypes.data, q, p.off.ctypes.data, p.lens.ctypes.data,
                                       out, cap, cols)
        return _emit(call, opt, p, want_paf, want_cols)

    def targets(self):
        lib = _bind()
        n = lib.mpn_hits_n_seq(self.h)
        buf = ct.create_string_buffer(1 << 16)
        names, lens = [], np.zeros(n, dtype=np.int32)
        for i in range(n):
            lib.mpn_hits_seq_name(self.h, i, buf, len(buf))
            names.append(buf.value.decode())
            lens[i] = lib.mpn_hits_seq_len(self.h, i)
        return names, lens

    def sam_header(self, cmdline=None):
        lib = _bind()
        names, _ = self.targets()
        cap = 64 * len(names) + sum(len(x) for x in names) + 4096 + (len(cmdline) if cmdline else 0)
        buf = ct.create_string_buffer(cap)
        r = lib.mpn_hits_sam_header(self.h, cmdline.encode() if cmdline else None, buf, cap)
        if r < 0:
            raise _ffi.MpnError(f'mpn_hits_sam_header rc={r}')
        return buf.raw[:r].decode()

    def close(self):
        if self.h:
            _bind().mpn_hits_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ext_dp_batch(opt, queries, targets, w, zdrop, end_bonus, flag, force_kernel=0):
    """DP stage on (query, target) pairs of 0..4 codes -> list of dicts (max, zdropped, max_q, max_t, mqe, mqe_t, score, reach_end, n_cigar, cigar)."""
    lib = _bind()
    n = len(queries)
    qbuf, qoff, qlen = pack_seqs([np.asarray(q, dtype=np.uint8) for q in queries])
    tbuf, toff, tlen = pack_seqs([np.asarray(t, dtype=np.uint8) for t in targets])
    arr = lambda v: np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.int32), (n,)))  # noqa: E731
    w, zdrop, end_bonus, flag = arr(w), arr(zdrop), arr(end_bonus), arr(flag)
    out = np.zeros((n, 9), dtype=np.int32)
    cap = int(qlen.astype(np.int64).sum() + tlen.astype(np.int64).sum()) + 4 * n
    pool = np.zeros(cap, dtype=np.uint32)
    coff = np.zeros(n, dtype=np.int64)
    rc = lib.mpn_ext_dp_batch(ct.byref(opt), n, qbuf.ctypes.data, qoff.ctypes.data, qlen.ctypes.data, tbuf.ctypes.data,
                              toff.ctypes.data, tlen.ctypes.data, w.ctypes.data, zdrop.ctypes.data, end_bonus.ctypes.data,
                              flag.ctypes.data, force_kernel, out.ctypes.data, pool.ctypes.data, cap, coff.ctypes.data)
    _ffi.check(rc, 'mpn_ext_dp_batch')
    keys = ('max', 'zdropped', 'max_q', 'max_t', 'mqe', 'mqe_t', 'score', 'reach_end', 'n_cigar')
    res = []
    for i in range(n):
        d = {k: int(out[i, j]) for j, k in enumerate(keys)}
        d['cigar'] = [int(x) for x in pool[coff[i]:coff[i] + d['n_cigar']]]
        res.append(d)
    return res


STAT_NAMES = {0: 'bases', 1: 'minimizers', 2: 'anchors', 3: 'chains', 4: 'dp_jobs', 5: 'dp_cells', 6: 'alignments',
              7: 'dp_rounds', 8: 'second_pass_jobs', 10: 'ev_sketch_ns', 11: 'ev_seed_ns', 12: 'ev_sort_ns',
              13: 'ev_chain_dp_ns', 14: 'ev_chain_bt_ns', 15: 'ev_ext_dp_ns', 25: 'ev_ext_bt_ns', 26: 'ev_ext_ztest_ns',
              16: 'wall_h2d_ns', 17: 'wall_seed_chain_ns', 18: 'wall_d2h_chains_ns', 19: 'wall_host_hits_ns',
              20: 'wall_host_plan_ns', 21: 'wall_ext_stage_ns', 22: 'wall_host_stitch_ns', 23: 'wall_host_final_ns',
              24: 'wall_total_ns', 27: 'wall_ext_host_prep_ns', 28: 'wall_ext_enqueue_ns', 29: 'wall_ext_gpu_wait_ns',
              30: 'wall_ext_finish_ns', 9: 'ev_ext_strip_ns', 31: 'strip_cells', 32: 'sub_batches',
              33: 'k_sketch_count_ns', 34: 'k_sketch_fill_ns', 35: 'k_seed_lookup_ns', 36: 'k_seed_fill_ns',
              37: 'k_chain_dp_ns', 38: 'k_strip16_ns', 39: 'k_strip32_ns', 40: 'k_strip64_ns', 41: 'strip16_cells',
              42: 'strip32_cells', 43: 'strip64_cells', 44: 'sort_records_moved', 45: 'anchors_kept', 46: 'k_compact_ns', 47: 'k_sort_msd_ns', 48: 'k_sort_chunk_ns', 49: 'k_sort_radix_ns',
              50: 'k_seed_filter_ns', 51: 'anchors_emitted', 52: 'k_finish_ns', 53: 'cigar_ops', 54: 'k_stitch_ns', 55: 'k_plan_ns', 56: 'k_layout_ns', 57: 'k_xstrip_ns', 58: 'xstrip_cells', 59: 'anchors_squeezed', 60: 'workers_shed', 61: 'workers', 62: 'k_hit_select_ns', 63: 'reads_hits_on_host'}


def last_stats():
    s = np.zeros(64, dtype=np.int64)
    _bind().mpn_map_last_stats_ex(s.ctypes.data, 64)
    return {name: int(s[i]) for i, name in STAT_NAMES.items()}
