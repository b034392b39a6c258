"""Host-side mirror of /root/reference/bin/realignment/pyssw.py on top of libmpn.so.

Same class names, constructor arguments, return tuples and CIGAR formatting (`M` printed as `=`,
soft clips, pyssw.py:100-118) so reference callers (fast_align_reads2ref.py:36-49) run unchanged.
`SSW.align` goes through the reference-compatible ssw_init/ssw_align symbols; `SSW.align_batch` is
the batched form (one launch for many reads) the reference lacks.
"""
import ctypes as ct

import numpy as np

from . import _ffi


class CAlignRes(ct.Structure):  # pyssw.py:7-16
    _fields_ = [('nScore', ct.c_uint16), ('nScore2', ct.c_uint16), ('nRefBeg', ct.c_int32),
                ('nRefEnd', ct.c_int32), ('nQryBeg', ct.c_int32), ('nQryEnd', ct.c_int32),
                ('nRefEnd2', ct.c_int32), ('sCigar', ct.POINTER(ct.c_uint32)), ('nCigarLen', ct.c_int32)]


class CSsw(object):  # pyssw.py:30-48
    def __init__(self, sLibPath=None):
        self.ssw = ct.CDLL(sLibPath) if sLibPath else _ffi.lib()
        self.ssw_init = self.ssw.ssw_init
        self.ssw_init.argtypes = [ct.POINTER(ct.c_int8), ct.c_int32, ct.POINTER(ct.c_int8), ct.c_int32, ct.c_int8]
        self.ssw_init.restype = ct.c_void_p
        self.init_destroy = self.ssw.init_destroy
        self.init_destroy.argtypes = [ct.c_void_p]
        self.init_destroy.restype = None
        self.ssw_align = self.ssw.ssw_align
        self.ssw_align.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int8), ct.c_int32, ct.c_uint8, ct.c_uint8,
                                   ct.c_uint8, ct.c_uint16, ct.c_int32, ct.c_int32]
        self.ssw_align.restype = ct.POINTER(CAlignRes)
        self.align_destroy = self.ssw.align_destroy
        self.align_destroy.argtypes = [ct.POINTER(CAlignRes)]
        self.align_destroy.restype = None


_CODE = np.full(256, 4, dtype=np.int8)
for _i, _c in enumerate('ACGT'):
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.lower())] = _i


def encode(seq):
    """ASCII -> 0..4 codes (pyssw.py:81-98: unknown letters map to N=4)."""
    return _CODE[np.frombuffer(seq.encode('latin-1'), dtype=np.uint8)].copy()


class SSW(object):
    def __init__(self, match=4, mismatch=6, gap_open=8, gap_extend=2, lib_path=None):
        self.match = match
        self.mismatch = mismatch
        self.gap_open = gap_open
        self.gap_extend = gap_extend
        self.lib_path = lib_path
        self.mat = self.build_matrix()
        self.ssw = CSsw(self.lib_path)

    def build_matrix(self):  # pyssw.py:56-79: the N row/column stays 0
        self.lEle = ['A', 'C', 'G', 'T', 'N']
        self.dEle2Int = {}
        for i, ele in enumerate(self.lEle):
            self.dEle2Int[ele] = i
            self.dEle2Int[ele.lower()] = i
        n = len(self.lEle)
        score = [0] * (n * n)
        for i in range(n - 1):
            for j in range(n - 1):
                score[i * n + j] = self.match if i == j else -self.mismatch
        mat = (len(score) * ct.c_int8)()
        mat[:] = score
        return mat

    def set_reference_sequence(self, reference):
        self.reference = reference
        self.rNum = self.to_int(reference)
        self.reference_len = len(reference)

    def to_int(self, seq):
        codes = encode(seq)
        num = (len(seq) * ct.c_int8)()
        if len(seq):
            ct.memmove(num, codes.ctypes.data, len(seq))
        return num

    def get_cigar(self, cigar, ref_position_start, ref_position_end, query_position_start, query_position_end, query):
        ops = 'MIDNSHP=X'
        out = []
        if query_position_start > 0:
            out.append(f'{query_position_start}S')
        for x in cigar:
            n, m = x >> 4, x & 15
            c = 'M' if m > 8 else ops[m]
            if c == 'M':
                c = '='
            out.append(f'{n}{c}')
        cigar_l = query_position_end - query_position_start + 1
        if cigar_l < len(query):
            out.append(f'{len(query) - cigar_l}S')
        return ''.join(out)

    def align_one(self, qProfile, rNum, nRLen, nOpen, nExt, nFlag, nMaskLen):
        res = self.ssw.ssw_align(qProfile, rNum, ct.c_int32(nRLen), nOpen, nExt, nFlag, 0, 0, nMaskLen)
        if not res:
            raise _ffi.MpnError('ssw_align returned NULL: ' + _ffi.last_error())
        c = res.contents
        out = (c.nScore, c.nScore2, c.nRefBeg, c.nRefEnd, c.nQryBeg, c.nQryEnd, c.nRefEnd2, c.nCigarLen,
               [c.sCigar[idx] for idx in range(c.nCigarLen)])
        self.ssw.align_destroy(res)
        return out

    def align(self, query):  # pyssw.py:137-147
        align_flag = 2
        query_len = len(query)
        qNum = self.to_int(query)
        qProfile = self.ssw.ssw_init(qNum, ct.c_int32(query_len), self.mat, len(self.lEle), 2)
        mask_len = 15 if query_len <= 30 else query_len
        try:
            res = self.align_one(qProfile, self.rNum, self.reference_len, self.gap_open, self.gap_extend,
                                 align_flag, mask_len)
        finally:
            self.ssw.init_destroy(qProfile)
        cigar = self.get_cigar(res[8], res[2], res[3], res[4], res[5], query)
        return res[0], cigar, res[2]

    def align_batch(self, queries):
        """[(score, cigar_str, ref_begin)] for every query against the current reference: one launch."""
        from .ssw_batch import ssw_align_batch
        reads = [encode(q) for q in queries]
        ref = encode(self.reference)
        masks = [15 if len(q) <= 30 else len(q) for q in queries]
        res = ssw_align_batch(reads, [ref] * len(reads), np.frombuffer(bytes(self.mat), dtype=np.int8), 5, 2,
                              self.gap_open, self.gap_extend, 2, 0, 0, masks, shared_ref=True)
        out = []
        for q, r in zip(queries, res):
            if r['status'] != 0:
                raise _ffi.MpnError(f'pair rejected with status {r["status"]}')
            out.append((r['score1'], self.get_cigar(r['cigar'], r['ref_begin1'], r['ref_end1'], r['read_begin1'],
                                                    r['read_end1'], q), r['ref_begin1']))
        return out
