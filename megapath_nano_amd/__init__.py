"""megapath_nano_amd -- MI355X-native alignment + read-reassignment hot path of MegaPath-Nano.

Only what the path needs lives here: csrc/ (HIP kernels + the C-ABI in libmpn.so), _ffi.py (ctypes
loader, no CPU fallback), and host-side mirrors of the reference's stage interfaces:
  pyssw.py        <- /root/reference/bin/realignment/pyssw.py       (SSW class)
  aligner.py      <- /root/reference/bin/lib/aligner.py             (Align)
  reassignment.py <- /root/reference/bin/lib/reassignment.py        (Reassign)
  fastq_filter.py <- /root/reference/bin/tools/nanofastq.c          (read quality / length filter)
"""
import os as _os

# 8 pipeline workers x 2 HIP streams need more than ROCm's default 4 hardware queues (streams sharing a queue serialise);
# the HIP runtime reads this when it initialises, so import this package before the first GPU call (see csrc/mpn_runtime.hip)
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '20')
