"""Host-side mirror of the reference's SSW Python wrapper, on top of libmpn.so.

Public surface kept from /root/reference/bin/realignment/pyssw.py so that its callers
(fast_align_reads2ref.py:36-49) run unchanged: classes `CAlignRes` (:7-16), `CSsw` (:30-48), `SSW` with
`set_reference_sequence`, `to_int`, `get_cigar`, `align_one`, `align` (:81-147) and the same return tuples; the CIGAR
string prints `M` as `=` and adds soft clips exactly like :100-118.  `SSW.align` drives the four reference-compatible
C symbols one read at a time; `SSW.align_batch` is the batched call (one launch for many reads) the reference lacks.
"""
import ctypes as ct

import numpy as np

from . import _ffi

# s_align (ssw.h:47-57): two u16 scores, five i32 coordinates, cigar pointer, cigar length
_ALIGN_FIELDS = (('nScore', ct.c_uint16), ('nScore2', ct.c_uint16)) + \
    tuple((n, ct.c_int32) for n in ('nRefBeg', 'nRefEnd', 'nQryBeg', 'nQryEnd', 'nRefEnd2')) + \
    (('sCigar', ct.POINTER(ct.c_uint32)), ('nCigarLen', ct.c_int32))
CAlignRes = type('CAlignRes', (ct.Structure,), {'_fields_': list(_ALIGN_FIELDS)})

_I8P = ct.POINTER(ct.c_int8)
_PROTOTYPES = {  # symbol -> (argtypes, restype), as bound at pyssw.py:34-48
    'ssw_init': ([_I8P, ct.c_int32, _I8P, ct.c_int32, ct.c_int8], ct.c_void_p),
    'init_destroy': ([ct.c_void_p], None),
    'ssw_align': ([ct.c_void_p, _I8P, ct.c_int32, ct.c_uint8, ct.c_uint8, ct.c_uint8, ct.c_uint16, ct.c_int32, ct.c_int32],
                  ct.POINTER(CAlignRes)),
    'align_destroy': ([ct.POINTER(CAlignRes)], None),
}

ALPHABET = 'ACGTN'
_CODE = np.full(256, len(ALPHABET) - 1, dtype=np.int8)  # anything unknown is N
for _i, _c in enumerate(ALPHABET[:-1]):
    _CODE[[ord(_c), ord(_c.lower())]] = _i
_OPS = 'MIDNSHP=X'


def encode(seq):
    """ASCII -> 0..4 codes (the mapping of pyssw.py:56-98)."""
    return _CODE[np.frombuffer(seq.encode('latin-1'), dtype=np.uint8)].copy()


class CSsw(object):
    """The four C symbols with the reference's prototypes, from libmpn.so unless a path is given."""

    def __init__(self, sLibPath=None):
        self.ssw = ct.CDLL(sLibPath) if sLibPath else _ffi.lib()
        for name, (argtypes, restype) in _PROTOTYPES.items():
            fn = getattr(self.ssw, name)
            fn.argtypes, fn.restype = argtypes, restype
            setattr(self, name, fn)


class SSW(object):
    def __init__(self, match=4, mismatch=6, gap_open=8, gap_extend=2, lib_path=None):
        self.match, self.mismatch, self.gap_open, self.gap_extend, self.lib_path = match, mismatch, gap_open, gap_extend, lib_path
        self.lEle = list(ALPHABET)
        self.dEle2Int = {c: i for i, ch in enumerate(ALPHABET) for c in (ch, ch.lower())}
        self.mat = self.build_matrix()
        self.ssw = CSsw(lib_path)

    def build_matrix(self):
        n = len(ALPHABET)
        m = np.zeros((n, n), dtype=np.int8)              # the N row and column stay 0 (pyssw.py:70-76)
        m[:n - 1, :n - 1] = -self.mismatch
        np.fill_diagonal(m[:n - 1, :n - 1], self.match)
        self._mat_np = m.reshape(-1).copy()
        return (ct.c_int8 * (n * n))(*[int(x) for x in self._mat_np])

    def to_int(self, seq):
        codes = encode(seq)
        return (ct.c_int8 * len(seq)).from_buffer_copy(codes.tobytes()) if len(seq) else (ct.c_int8 * 0)()

    def set_reference_sequence(self, reference):
        self.reference, self.reference_len = reference, len(reference)
        self.rNum = self.to_int(reference)

    def get_cigar(self, cigar, ref_position_start, ref_position_end, query_position_start, query_position_end, query):
        parts = [f'{query_position_start}S'] if query_position_start > 0 else []
        for word in cigar:
            op = _OPS[word & 15] if (word & 15) < len(_OPS) else 'M'
            parts.append(f"{word >> 4}{'=' if op == 'M' else op}")
        tail = len(query) - (query_position_end - query_position_start + 1)
        if tail > 0:
            parts.append(f'{tail}S')
        return ''.join(parts)

    def align_one(self, qProfile, rNum, nRLen, nOpen, nExt, nFlag, nMaskLen):
        res = self.ssw.ssw_align(qProfile, rNum, ct.c_int32(nRLen), nOpen, nExt, nFlag, 0, 0, nMaskLen)
        if not res:
            raise _ffi.MpnError('ssw_align returned NULL: ' + _ffi.last_error())
        r = res.contents
        out = tuple(getattr(r, n) for n, _ in _ALIGN_FIELDS[:7]) + (r.nCigarLen, [r.sCigar[i] for i in range(r.nCigarLen)])
        self.ssw.align_destroy(res)
        return out

    @staticmethod
    def _mask_len(query_len):
        return 15 if query_len <= 30 else query_len       # pyssw.py:142

    def align(self, query):
        q = self.to_int(query)
        prof = self.ssw.ssw_init(q, ct.c_int32(len(query)), self.mat, len(ALPHABET), 2)
        try:
            r = self.align_one(prof, self.rNum, self.reference_len, self.gap_open, self.gap_extend, 2, self._mask_len(len(query)))
        finally:
            self.ssw.init_destroy(prof)
        return r[0], self.get_cigar(r[8], r[2], r[3], r[4], r[5], query), r[2]

    def align_batch(self, queries):
        """[(score, cigar_str, ref_begin)] for every query against the current reference: one launch."""
        from .ssw_batch import ssw_align_batch
        ref = encode(self.reference)
        res = ssw_align_batch([encode(q) for q in queries], [ref] * len(queries), self._mat_np, len(ALPHABET), 2, self.gap_open,
                              self.gap_extend, 2, 0, 0, [self._mask_len(len(q)) for q in queries], shared_ref=True)
        out = []
        for q, r in zip(queries, res):
            if r['status'] != 0:
                raise _ffi.MpnError(f'pair rejected with status {r["status"]}')
            out.append((r['score1'], self.get_cigar(r['cigar'], r['ref_begin1'], r['ref_end1'], r['read_begin1'], r['read_end1'], q),
                        r['ref_begin1']))
        return out
