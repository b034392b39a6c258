"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the per-read arithmetic of the reference's read filter,
/root/reference/bin/tools/nanofastq.c:165-203: the sum of base-call error probabilities of a read and the sum after
cropping, as sequential double-precision additions / subtractions in the reference's order.

Pinned: tests/golden/nanofastq_golden.json holds stdin -> (stdout, stderr) of the reference's own prebuilt binary
(/root/reference/bin/tools/nanofastq, run in the build container by tests/golden/make_nanofastq_golden.py); the
assembled outputs of this oracle reproduce them byte for byte (tests/test_fastq_filter.py).
Only tests/ may import this module.
"""
import math

TABLE = [math.pow(10.0, -i / 10.0) for i in range(128)]          # nanofastq.c:147-149


def qsums(qual, head_crop, tail_crop, min_len):
    """qual: bytes.  -> (total, cropped) Python floats (IEEE doubles)."""
    total = 0.0
    for c in qual:                                                # :168-171
        total += TABLE[c - 33]
    cropped = total
    start, end = head_crop, len(qual) - tail_crop
    if end - start >= min_len:                                    # :182
        for c in qual[:start]:                                    # :188-191
            cropped -= TABLE[c - 33]
        for c in qual[end:]:                                      # :192-195
            cropped -= TABLE[c - 33]
    return total, cropped
