"""Amplicon realigner: host-side mirror of the reference's ctypes use of `realigner`
(/root/reference/bin/realignment/realign_illumina_reads.py:32,40-43,579-629) over libmpn.so.

    realign_reads(...)   one window through the reference's own C entry points (realign_reads / free_memory)
    realign_batch([...]) many windows in one call (mpn_realign_batch): one launch per stage over all of them

Both return, per read, (position, cigar_string) exactly as the reference's `realigner_p.contents` would hold them
(a match run is written 'X'; reads the realigner leaves alone come back with their input position and CIGAR).
"""
import ctypes as ct

from . import _ffi

MAX_REGION_READS = 1000   # realign_illumina_reads.py:38 max_region_reads_num, realigner.h:44-45


class StructPointer(ct.Structure):   # realign_illumina_reads.py:40-43
    _fields_ = [('position', ct.c_int * MAX_REGION_READS), ('cigar_string', ct.c_char_p * MAX_REGION_READS)]


class Window(ct.Structure):          # include/mpn_realign.h mpn_realign_window
    _fields_ = [('n_reads', ct.c_int32), ('seqs', ct.POINTER(ct.c_char_p)), ('positions', ct.POINTER(ct.c_int32)),
                ('cigars', ct.POINTER(ct.c_char_p)), ('reference', ct.c_char_p), ('n_haps', ct.c_int32),
                ('haplotypes', ct.POINTER(ct.c_char_p)), ('ref_start', ct.c_int32), ('ref_prefix', ct.c_int32),
                ('ref_suffix', ct.c_int32)]


def _b(s):
    return s if isinstance(s, bytes) else s.encode()


def realign_reads(seqs, positions, cigars, reference, haplotypes, ref_start, ref_prefix, ref_suffix):
    lib = _ffi.lib()
    n = len(seqs)
    if n > MAX_REGION_READS:
        raise ValueError('at most %d reads per window' % MAX_REGION_READS)
    seq_list = (ct.c_char_p * n)(*[_b(s) for s in seqs])
    pos_list = (ct.c_int * n)(*positions)
    cig_list = (ct.c_char_p * n)(*[_b(c) for c in cigars])
    lib.realign_reads.restype = ct.POINTER(StructPointer)
    lib.realign_reads.argtypes = [ct.c_char_p * n, ct.c_int * n, ct.c_char_p * n, ct.c_char_p, ct.c_char_p, ct.c_int, ct.c_int,
                                  ct.c_int, ct.c_int]
    p = lib.realign_reads(seq_list, pos_list, cig_list, _b(reference), b' '.join(_b(h) for h in haplotypes), ref_start,
                          ref_prefix, ref_suffix, n)
    if not p:
        raise _ffi.MpnError('realign_reads failed: ' + _ffi.last_error())
    out = [(int(p.contents.position[i]), p.contents.cigar_string[i].decode()) for i in range(n)]
    lib.free_memory.restype = None
    lib.free_memory.argtypes = [ct.POINTER(StructPointer), ct.c_int]
    lib.free_memory(p, n)
    return out


def realign_batch(windows):
    """windows: dicts with the keyword arguments of realign_reads -> one list of (position, cigar) per window."""
    lib = _ffi.lib()
    nw = len(windows)
    if nw == 0:
        return []
    arr = (Window * nw)()
    keep = []
    total = 0
    for k, w in enumerate(windows):
        n, nh = len(w['seqs']), len(w['haplotypes'])
        seqs = (ct.c_char_p * max(n, 1))(*[_b(s) for s in w['seqs']])
        cigs = (ct.c_char_p * max(n, 1))(*[_b(c) for c in w['cigars']])
        pos = (ct.c_int32 * max(n, 1))(*w['positions'])
        haps = (ct.c_char_p * max(nh, 1))(*[_b(h) for h in w['haplotypes']])
        keep.append((seqs, cigs, pos, haps))
        arr[k] = Window(n, seqs, pos, cigs, _b(w['reference']), nh, haps, w['ref_start'], w['ref_prefix'], w['ref_suffix'])
        total += n
    out_pos = (ct.c_int32 * max(total, 1))()
    out_cig = (ct.c_char_p * max(total, 1))()
    lib.mpn_realign_batch.restype = ct.c_int
    lib.mpn_realign_batch.argtypes = [ct.c_int32, ct.POINTER(Window), ct.POINTER(ct.c_int32), ct.POINTER(ct.c_char_p)]
    _ffi.check(lib.mpn_realign_batch(nw, arr, out_pos, out_cig), 'mpn_realign_batch')
    res, i = [], 0
    for w in windows:
        n = len(w['seqs'])
        res.append([(int(out_pos[i + j]), out_cig[i + j].decode()) for j in range(n)])
        i += n
    lib.mpn_realign_free_cigars.restype = None
    lib.mpn_realign_free_cigars.argtypes = [ct.POINTER(ct.c_char_p), ct.c_int64]
    lib.mpn_realign_free_cigars(out_cig, total)
    return res
