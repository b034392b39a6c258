"""ctypes binding of the amplicon realigner's C entry points (TEST INFRASTRUCTURE ONLY).

  ref_realign(...)  -> oracle/_ref/librealigner.so, the reference's own realigner.cpp + ssw_cpp.cpp + ssw.c compiled in
                       place (oracle/Makefile).  Prototypes as the reference binds them,
                       /root/reference/bin/realignment/realign_illumina_reads.py:40-43,596-605,627-629; struct layout from
                       /root/reference/bin/realignment/realign/realigner.h:42-46 (1000 fixed slots).
  abi_realign(lib, ...) applies the same prototypes to any library exporting realign_reads / free_memory (libmpn.so).
"""
import ctypes as ct
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_READS = 1000


class StructPointer(ct.Structure):
    _fields_ = [('position', ct.c_int * MAX_READS), ('cigar_string', ct.c_char_p * MAX_READS)]


_ref = None


def have_ref():
    return os.path.exists(os.path.join(HERE, '_ref', 'librealigner.so'))


def abi_realign(lib, seqs, positions, cigars, reference, haplotypes, ref_start, ref_prefix, ref_suffix):
    """seqs / cigars: lists of str, positions: list of int, haplotypes: list of str -> [(position, cigar), ...]"""
    n = len(seqs)
    assert n <= MAX_READS and len(positions) == n and len(cigars) == n
    seq_list = (ct.c_char_p * n)(*[s.encode() for s in seqs])
    pos_list = (ct.c_int * n)(*positions)
    cig_list = (ct.c_char_p * n)(*[c.encode() for c in cigars])
    lib.realign_reads.restype = ct.POINTER(StructPointer)
    lib.realign_reads.argtypes = [ct.c_char_p * n, ct.c_int * n, ct.c_char_p * n, ct.c_char_p, ct.c_char_p, ct.c_int,
                                  ct.c_int, ct.c_int, ct.c_int]
    p = lib.realign_reads(seq_list, pos_list, cig_list, reference.encode(), ' '.join(haplotypes).encode(), ref_start,
                          ref_prefix, ref_suffix, n)
    if not p:
        raise RuntimeError('realign_reads returned NULL')
    out = [(int(p.contents.position[i]), p.contents.cigar_string[i].decode()) for i in range(n)]
    lib.free_memory.restype = None
    lib.free_memory.argtypes = [ct.POINTER(StructPointer), ct.c_int]
    lib.free_memory(p, n)
    return out


def ref_realign(**kw):
    global _ref
    if _ref is None:
        _ref = ct.CDLL(os.path.join(HERE, '_ref', 'librealigner.so'))
    return abi_realign(_ref, **kw)
