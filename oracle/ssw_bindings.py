"""ctypes bindings for the SSW parity oracles (TEST INFRASTRUCTURE ONLY).

  ref_align(...)     -> oracle/_ref/libssw.so, the reference's ssw.c compiled in place
                        (struct layout from /root/reference/bin/realignment/realign/ssw.h:47-57,
                        argtypes from bin/realignment/pyssw.py:30-48)
  oracle_align(...)  -> oracle/libssw_oracle.so, this repo's scalar restatement

Both return a tuple (score1, score2, ref_begin1, ref_end1, read_begin1, read_end1, ref_end2,
[cigar...]) or None where the reference returns NULL.
"""
import ctypes as ct
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class SAlign(ct.Structure):
    _fields_ = [('score1', ct.c_uint16), ('score2', ct.c_uint16), ('ref_begin1', ct.c_int32),
                ('ref_end1', ct.c_int32), ('read_begin1', ct.c_int32), ('read_end1', ct.c_int32),
                ('ref_end2', ct.c_int32), ('cigar', ct.POINTER(ct.c_uint32)), ('cigarLen', ct.c_int32)]


class OracleResult(ct.Structure):
    _fields_ = [('score1', ct.c_uint16), ('score2', ct.c_uint16), ('ref_begin1', ct.c_int32),
                ('ref_end1', ct.c_int32), ('read_begin1', ct.c_int32), ('read_end1', ct.c_int32),
                ('ref_end2', ct.c_int32), ('cigar_len', ct.c_int32), ('status', ct.c_int32)]


_ref = None
_orc = None


def have_ref():
    return os.path.exists(os.path.join(HERE, '_ref', 'libssw.so'))


def bind_ssw_abi(lib):
    """Apply the pyssw.py:30-48 prototypes to any library exporting the four ssw symbols."""
    lib.ssw_init.argtypes = [ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int32, ct.c_int8]
    lib.ssw_init.restype = ct.c_void_p
    lib.init_destroy.argtypes = [ct.c_void_p]
    lib.init_destroy.restype = None
    lib.ssw_align.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int32, ct.c_uint8, ct.c_uint8, ct.c_uint8,
                              ct.c_uint16, ct.c_int32, ct.c_int32]
    lib.ssw_align.restype = ct.POINTER(SAlign)
    lib.align_destroy.argtypes = [ct.POINTER(SAlign)]
    lib.align_destroy.restype = None
    return lib


def ssw_abi_align(lib, read, ref, mat, gap_open, gap_extend, flag, filters, filterd, mask, score_size):
    read = np.ascontiguousarray(read, dtype=np.int8)
    ref = np.ascontiguousarray(ref, dtype=np.int8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    prof = lib.ssw_init(read.ctypes.data, len(read), mat.ctypes.data, 5, score_size)
    res = lib.ssw_align(prof, ref.ctypes.data, len(ref), gap_open, gap_extend, flag, filters, filterd, mask)
    if not res:
        lib.init_destroy(prof)
        return None
    c = res.contents
    out = (c.score1, c.score2, c.ref_begin1, c.ref_end1, c.read_begin1, c.read_end1, c.ref_end2,
           [int(c.cigar[k]) for k in range(c.cigarLen)])
    lib.align_destroy(res)
    lib.init_destroy(prof)
    return out


def ref_align(**kw):
    global _ref
    if _ref is None:
        _ref = bind_ssw_abi(ct.CDLL(os.path.join(HERE, '_ref', 'libssw.so')))
    return ssw_abi_align(_ref, **kw)


def oracle_align(read, ref, mat, gap_open, gap_extend, flag, filters, filterd, mask, score_size):
    global _orc
    if _orc is None:
        _orc = ct.CDLL(os.path.join(HERE, 'libssw_oracle.so'))
        _orc.ssw_oracle_align.argtypes = [ct.c_void_p, ct.c_int32, ct.c_void_p, ct.c_int32, ct.c_int8,
                                          ct.c_void_p, ct.c_int32, ct.c_uint8, ct.c_uint8, ct.c_uint8,
                                          ct.c_uint16, ct.c_int32, ct.c_int32, ct.POINTER(OracleResult),
                                          ct.c_void_p, ct.c_int32]
        _orc.ssw_oracle_align.restype = ct.c_int
    read = np.ascontiguousarray(read, dtype=np.int8)
    ref = np.ascontiguousarray(ref, dtype=np.int8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    cap = len(read) + len(ref) + 8
    cig = np.zeros(cap, dtype=np.uint32)
    r = OracleResult()
    st = _orc.ssw_oracle_align(read.ctypes.data, len(read), mat.ctypes.data, 5, score_size, ref.ctypes.data,
                               len(ref), gap_open, gap_extend, flag, filters, filterd, mask, ct.byref(r),
                               cig.ctypes.data, cap)
    if st == 1:
        return None
    if st == 2:
        return 'undefined'
    if st == 3:
        return 'unsupported'
    return (r.score1, r.score2, r.ref_begin1, r.ref_end1, r.read_begin1, r.read_end1, r.ref_end2,
            [int(x) for x in cig[:r.cigar_len]])
