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


def check(rc, what):
    if rc != 0:
        raise MpnError(f'{what} failed (rc={rc}): {last_error()}')
