"""ctypes loader for libmpn.so.  There is deliberately no fallback: if the HIP library is missing or
fails to load, using the product path raises."""
import ctypes as ct
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmpn.so')

_lib = None


class MpnError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpnError(f'{LIB_PATH} not built: run `python -m megapath_nano_amd.build` '
                           '(or __graft_entry__.build()); there is no CPU fallback')
        _lib = ct.CDLL(LIB_PATH)
        _lib.mpn_last_error.restype = ct.c_char_p
    return _lib


def last_error():
    return lib().mpn_last_error().decode()


def hint(msg):
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  A process that uses both must import torch (and touch the GPU
    with it) BEFORE libmpn.so is loaded, as bench.py does: then both HIP runtimes sit on one HSA runtime.  The other order
    leaves two HSA runtimes in the process and the second one finds no device."""
    import sys
    if 'no ROCm-capable device' in msg and 'torch' in sys.modules:
        msg += ' [libmpn.so was loaded before PyTorch initialised the GPU: import torch and call torch.cuda.init() first]'
    return msg


def check(rc, what):
    if rc != 0:
        raise MpnError(f'{what} failed (rc={rc}): {hint(last_error())}')
