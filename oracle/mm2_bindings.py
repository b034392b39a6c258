"""ctypes bindings for oracle/libmm2_oracle.so (TEST INFRASTRUCTURE ONLY; parity unpinned, see mm2_oracle.h)."""
import ctypes as ct
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class MM128(ct.Structure):
    _fields_ = [('x', ct.c_uint64), ('y', ct.c_uint64)]


class Opt(ct.Structure):
    _fields_ = [('mid_occ_frac', ct.c_float), ('mid_occ', ct.c_int32), ('max_gap', ct.c_int32), ('bw', ct.c_int32),
                ('max_chain_skip', ct.c_int32), ('max_chain_iter', ct.c_int32), ('min_cnt', ct.c_int32),
                ('min_chain_score', ct.c_int32), ('mask_level', ct.c_float), ('pri_ratio', ct.c_float),
                ('best_n', ct.c_int32), ('max_join_long', ct.c_int32), ('max_join_short', ct.c_int32),
                ('min_join_flank_sc', ct.c_int32), ('min_join_flank_ratio', ct.c_float), ('a', ct.c_int32),
                ('b', ct.c_int32), ('q', ct.c_int32), ('e', ct.c_int32), ('q2', ct.c_int32), ('e2', ct.c_int32),
                ('sc_ambi', ct.c_int32), ('zdrop', ct.c_int32), ('zdrop_inv', ct.c_int32), ('end_bonus', ct.c_int32),
                ('min_dp_max', ct.c_int32), ('min_ksw_len', ct.c_int32), ('max_clip_ratio', ct.c_float),
                ('max_sw_mat', ct.c_int64), ('with_cigar', ct.c_int32), ('seed', ct.c_uint32)]


class Reg(ct.Structure):
    _fields_ = [(n, ct.c_int32) for n in ('id', 'cnt', 'rid', 'score', 'qs', 'qe', 'rs', 're', 'parent', 'subsc', 'as_',
                                          'mlen', 'blen', 'n_sub', 'score0')] + \
               [(n, ct.c_uint32) for n in ('mapq', 'split', 'rev', 'inv', 'sam_pri', 'split_inv', 'hash')] + \
               [(n, ct.c_int32) for n in ('has_p', 'dp_score', 'dp_max', 'dp_max2', 'n_ambi', 'n_cigar')] + \
               [('cigar', ct.POINTER(ct.c_uint32))]


class Ez(ct.Structure):
    _fields_ = [(n, ct.c_int32) for n in ('max', 'zdropped', 'max_q', 'max_t', 'mqe', 'mqe_t', 'mte', 'mte_q', 'score',
                                          'reach_end', 'n_cigar')] + [('cigar', ct.POINTER(ct.c_uint32))]


def lib():
    global _lib
    if _lib is None:
        L = ct.CDLL(os.path.join(HERE, 'libmm2_oracle.so'))
        L.mmo_opt_init.argtypes = [ct.POINTER(Opt)]
        L.mmo_idx_build.argtypes = [ct.c_int32, ct.POINTER(ct.c_char_p), ct.POINTER(ct.c_char_p), ct.c_void_p,
                                    ct.c_int, ct.c_int]
        L.mmo_idx_build.restype = ct.c_void_p
        L.mmo_idx_destroy.argtypes = [ct.c_void_p]
        L.mmo_idx_cal_max_occ.argtypes = [ct.c_void_p, ct.c_float]
        L.mmo_idx_cal_max_occ.restype = ct.c_int32
        L.mmo_free.argtypes = [ct.c_void_p]
        L.mmo_sketch.argtypes = [ct.c_char_p, ct.c_int32, ct.c_int, ct.c_int, ct.c_uint32, ct.POINTER(ct.c_void_p)]
        L.mmo_sketch.restype = ct.c_int64
        L.mmo_collect_anchors.argtypes = [ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int64, ct.c_int32,
                                          ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_int32)]
        L.mmo_collect_anchors.restype = ct.c_int64
        L.mmo_chain.argtypes = [ct.POINTER(Opt), ct.c_int64, ct.c_void_p, ct.POINTER(ct.c_int32),
                                ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p)]
        L.mmo_chain.restype = ct.c_int64
        L.mmo_map_read.argtypes = [ct.c_void_p, ct.POINTER(Opt), ct.c_char_p, ct.c_char_p, ct.c_int32,
                                   ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]
        L.mmo_map_read.restype = ct.POINTER(Reg)
        L.mmo_free_regs.argtypes = [ct.POINTER(Reg), ct.c_int32]
        L.mmo_write_paf.argtypes = [ct.c_void_p, ct.POINTER(Opt), ct.c_char_p, ct.c_int32, ct.POINTER(Reg), ct.c_int32,
                                    ct.c_int32, ct.c_char_p, ct.c_int64]
        L.mmo_write_paf.restype = ct.c_int64
        L.mmo_write_sam.argtypes = [ct.c_void_p, ct.POINTER(Opt), ct.c_char_p, ct.c_int32, ct.c_char_p, ct.POINTER(Reg), ct.c_int32,
                                    ct.c_int32, ct.c_char_p, ct.c_int64]
        L.mmo_write_sam.restype = ct.c_int64
        L.mmo_write_sam_q.argtypes = [ct.c_void_p, ct.POINTER(Opt), ct.c_char_p, ct.c_int32, ct.c_char_p, ct.c_char_p, ct.POINTER(Reg),
                                      ct.c_int32, ct.c_int32, ct.c_char_p, ct.c_int64]
        L.mmo_write_sam_q.restype = ct.c_int64
        L.mmo_idx_concat_names.argtypes = [ct.c_int32, ct.POINTER(ct.c_void_p)]
        L.mmo_idx_concat_names.restype = ct.c_void_p
        L.mmo_map_read_split.argtypes = [ct.c_int32, ct.POINTER(ct.c_void_p), ct.POINTER(Opt), ct.c_char_p, ct.c_char_p, ct.c_int32,
                                         ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int32)]
        L.mmo_map_read_split.restype = ct.POINTER(Reg)
        L.mmo_extd2.argtypes = [ct.c_int, ct.c_void_p, ct.c_int, ct.c_void_p, ct.c_int8, ct.c_int8, ct.c_int8, ct.c_int8,
                                ct.c_int8, ct.c_int8, ct.c_int8, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(Ez)]
        _lib = L
    return _lib


def default_opt(**kw):
    o = Opt()
    lib().mmo_opt_init(ct.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Index:
    def __init__(self, genomes, k=15, w=10):
        """genomes: list of (name, uint8 ASCII numpy array or bytes)"""
        L = lib()
        n = len(genomes)
        self.names = [g[0] for g in genomes]
        self._seqs = [bytes(g[1]) if not isinstance(g[1], bytes) else g[1] for g in genomes]
        names = (ct.c_char_p * n)(*[x.encode() for x in self.names])
        seqs = (ct.c_char_p * n)(*self._seqs)
        lens = np.array([len(s) for s in self._seqs], dtype=np.int32)
        self.lens = lens
        self.h = L.mmo_idx_build(n, names, seqs, lens.ctypes.data, k, w)
        self.k, self.w = k, w

    def mid_occ(self, f=2e-4):
        return lib().mmo_idx_cal_max_occ(self.h, f)

    def close(self):
        if self.h:
            lib().mmo_idx_destroy(self.h)
            self.h = None


def _take128(ptr, n):
    if n == 0 or not ptr:
        return np.zeros((0, 2), dtype=np.uint64)
    arr = np.ctypeslib.as_array(ct.cast(ptr, ct.POINTER(ct.c_uint64)), shape=(n, 2)).copy()
    lib().mmo_free(ptr)
    return arr


def sketch(seq, w=10, k=15, rid=0):
    seq = bytes(seq)
    out = ct.c_void_p()
    n = lib().mmo_sketch(seq, len(seq), w, k, rid, ct.byref(out))
    return _take128(out, n)


def collect_anchors(idx, max_occ, mv, qlen):
    mv = np.ascontiguousarray(mv, dtype=np.uint64)
    out = ct.c_void_p()
    rep = ct.c_int32()
    n = lib().mmo_collect_anchors(idx.h, max_occ, mv.ctypes.data, len(mv), qlen, ct.byref(out), ct.byref(rep))
    return _take128(out, n), rep.value


def chain(opt, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n_u = ct.c_int32()
    u = ct.c_void_p()
    b = ct.c_void_p()
    n_b = lib().mmo_chain(ct.byref(opt), len(a), a.ctypes.data, ct.byref(n_u), ct.byref(u), ct.byref(b))
    if n_u.value == 0:
        return np.zeros(0, dtype=np.uint64), np.zeros((0, 2), dtype=np.uint64)
    uu = np.ctypeslib.as_array(ct.cast(u, ct.POINTER(ct.c_uint64)), shape=(n_u.value,)).copy()
    lib().mmo_free(u)
    return uu, _take128(b, n_b)


def map_read_sam(idx, opt, name, seq, qual=None):
    """-> SAM records of one read (minimap2 -a), unmapped reads included; qual: the read's quality string or None"""
    L = lib()
    seq = bytes(seq)
    n = ct.c_int32()
    rep = ct.c_int32()
    regs = L.mmo_map_read(idx.h, ct.byref(opt), name.encode(), seq, len(seq), ct.byref(n), ct.byref(rep))
    cap = (4096 + 2 * len(seq)) * max(n.value, 1) + sum(regs[i].n_cigar for i in range(n.value)) * 12 + 4096
    buf = ct.create_string_buffer(cap)
    nb = L.mmo_write_sam_q(idx.h, ct.byref(opt), name.encode(), len(seq), seq, bytes(qual) if qual is not None else None, regs, n.value,
                           rep.value, buf, cap)
    assert nb >= 0
    if n.value > 0:
        L.mmo_free_regs(regs, n.value)
    return buf.raw[:nb].decode()


class SplitIndex:
    """Index parts (minimap2 -I) + the concatenated name table the writers need."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.arr = (ct.c_void_p * len(self.parts))(*[p.h for p in self.parts])
        self.names_h = lib().mmo_idx_concat_names(len(self.parts), self.arr)

    def map_read(self, opt, name, seq, sam=False, qual=None):
        """-> PAF lines (or SAM records) of one read mapped against every part and merged (--split-prefix)"""
        L = lib()
        seq = bytes(seq)
        n = ct.c_int32()
        rep = ct.c_int32()
        regs = L.mmo_map_read_split(len(self.parts), self.arr, ct.byref(opt), name.encode(), seq, len(seq), ct.byref(n), ct.byref(rep))
        cap = (4096 + 2 * len(seq)) * max(n.value, 1) + sum(regs[i].n_cigar for i in range(n.value)) * 12 + 4096
        buf = ct.create_string_buffer(cap)
        if sam:
            nb = L.mmo_write_sam_q(self.names_h, ct.byref(opt), name.encode(), len(seq), seq, bytes(qual) if qual is not None else None,
                                   regs, n.value, rep.value, buf, cap)
        else:
            nb = L.mmo_write_paf(self.names_h, ct.byref(opt), name.encode(), len(seq), regs, n.value, rep.value, buf, cap) if n.value else 0
        assert nb >= 0
        if n.value > 0:
            L.mmo_free_regs(regs, n.value)
        return buf.raw[:nb].decode()

    def close(self):
        if self.names_h:
            lib().mmo_idx_destroy(self.names_h)
            self.names_h = None


def map_read(idx, opt, name, seq):
    """-> (list of reg dicts, rep_len, paf text)"""
    L = lib()
    seq = bytes(seq)
    n = ct.c_int32()
    rep = ct.c_int32()
    regs = L.mmo_map_read(idx.h, ct.byref(opt), name.encode(), seq, len(seq), ct.byref(n), ct.byref(rep))
    out = []
    paf = ''
    if n.value > 0:
        cap = 4096 * n.value + sum(regs[i].n_cigar for i in range(n.value)) * 12 + 4096
        buf = ct.create_string_buffer(cap)
        nb = L.mmo_write_paf(idx.h, ct.byref(opt), name.encode(), len(seq), regs, n.value, rep.value, buf, cap)
        assert nb >= 0
        paf = buf.raw[:nb].decode()
        for i in range(n.value):
            r = regs[i]
            d = {f: getattr(r, f) for f, _ in Reg._fields_ if f != 'cigar'}
            d['cigar'] = [int(r.cigar[k]) for k in range(r.n_cigar)]
            out.append(d)
        L.mmo_free_regs(regs, n.value)
    return out, rep.value, paf


def extd2(query, target, sc_mch=2, sc_mis=-4, sc_n=-1, q=4, e=2, q2=24, e2=1, w=751, zdrop=400, end_bonus=-1, flag=0):
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    ez = Ez()
    lib().mmo_extd2(len(query), query.ctypes.data, len(target), target.ctypes.data, sc_mch, sc_mis, sc_n, q, e, q2, e2,
                    w, zdrop, end_bonus, flag, ct.byref(ez))
    d = {f: getattr(ez, f) for f, _ in Ez._fields_ if f != 'cigar'}
    d['cigar'] = [int(ez.cigar[k]) for k in range(ez.n_cigar)]
    if ez.cigar:
        lib().mmo_free(ez.cigar)
    return d
