"""Thin ctypes wrapper over mpn_ssw_align_batch (include/mpn_ssw.h, part 2)."""
import ctypes as ct

import numpy as np

from . import _ffi

_bound = False


def _bind():
    global _bound
    lib = _ffi.lib()
    if not _bound:
        P = ct.c_void_p
        lib.mpn_ssw_align_batch.argtypes = [ct.c_int32, P, P, P, P, P, P, P, ct.c_int32, ct.c_int8, ct.c_uint8,
                                            ct.c_uint8, ct.c_uint8, ct.c_uint16, ct.c_int32, P, P, P, P, P, P, P, P,
                                            P, ct.c_int64, P, P, P]
        lib.mpn_ssw_align_batch.restype = ct.c_int
        _bound = True
    return lib


def ssw_align_batch(reads, refs, mat, n, score_size, gap_open, gap_extend, flag, filters, filterd, mask_len,
                    shared_ref=False):
    """reads/refs: lists of int8 numpy arrays (codes 0..n-1).  Returns one dict per pair."""
    lib = _bind()
    npairs = len(reads)
    if npairs == 0:
        return []
    read_len = np.array([len(r) for r in reads], dtype=np.int32)
    read_off = np.zeros(npairs, dtype=np.int64)
    read_off[1:] = np.cumsum(read_len[:-1])
    read_buf = np.ascontiguousarray(np.concatenate(reads), dtype=np.int8)
    ref_len = np.array([len(r) for r in refs], dtype=np.int32)
    ref_off = np.zeros(npairs, dtype=np.int64)
    if shared_ref:
        ref_buf = np.ascontiguousarray(refs[0], dtype=np.int8)
    else:
        ref_off[1:] = np.cumsum(ref_len[:-1])
        ref_buf = np.ascontiguousarray(np.concatenate(refs), dtype=np.int8)
    if read_buf.size == 0:
        read_buf = np.zeros(1, dtype=np.int8)
    if ref_buf.size == 0:
        ref_buf = np.zeros(1, dtype=np.int8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    mask = np.ascontiguousarray(mask_len, dtype=np.int32)
    s1 = np.zeros(npairs, dtype=np.uint16)
    s2 = np.zeros(npairs, dtype=np.uint16)
    i32 = [np.zeros(npairs, dtype=np.int32) for _ in range(5)]
    cap = int(read_len.astype(np.int64).sum() + ref_len.astype(np.int64).sum() + 8 * npairs)
    pool = np.zeros(cap, dtype=np.uint32)
    coff = np.zeros(npairs, dtype=np.int64)
    clen = np.zeros(npairs, dtype=np.int32)
    status = np.zeros(npairs, dtype=np.int32)

    def p(a):
        return a.ctypes.data

    rc = lib.mpn_ssw_align_batch(npairs, p(read_buf), p(read_off), p(read_len), p(ref_buf), p(ref_off), p(ref_len),
                                 p(mat), n, score_size, gap_open, gap_extend, flag, filters, filterd, p(mask),
                                 p(s1), p(s2), p(i32[0]), p(i32[1]), p(i32[2]), p(i32[3]), p(i32[4]),
                                 p(pool), cap, p(coff), p(clen), p(status))
    _ffi.check(rc, 'mpn_ssw_align_batch')
    out = []
    for i in range(npairs):
        out.append(dict(score1=int(s1[i]), score2=int(s2[i]), ref_begin1=int(i32[0][i]), ref_end1=int(i32[1][i]),
                        read_begin1=int(i32[2][i]), read_end1=int(i32[3][i]), ref_end2=int(i32[4][i]),
                        cigar=[int(x) for x in pool[coff[i]:coff[i] + clen[i]]], status=int(status[i])))
    return out
