"""CPU tests: libmpn.so builds, loads, and exports every symbol include/*.h declares (no compute)."""
import ctypes as ct
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        text = open(h).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        text = re.sub(r'//[^\n]*', '', text)
        for m in re.finditer(r'\b([A-Za-z_][A-Za-z0-9_]*)\s*\(', text):
            name = m.group(1)
            if name.startswith(('mpn_', 'ssw_')) or name in ('init_destroy', 'align_destroy', 'realign_reads', 'free_memory'):
                syms.add(name)
    return syms


def test_headers_declare_something():
    syms = declared_symbols()
    assert {'ssw_init', 'ssw_align', 'init_destroy', 'align_destroy', 'mpn_ssw_align_batch', 'mpn_last_error', 'realign_reads',
            'free_memory', 'mpn_realign_batch', 'mpn_realign_free_cigars'} <= syms


def test_library_exports_all_declared_symbols(libmpn):
    for s in sorted(declared_symbols()):
        assert hasattr(libmpn, s), f'libmpn.so does not export {s}'


def test_s_align_layout_matches_reference_struct():
    # ssw.h:47-57 / pyssw.py:7-16: 2xu16, 5xi32, pointer, i32  -> 40 bytes on LP64
    from megapath_nano_amd.pyssw import CAlignRes
    assert ct.sizeof(CAlignRes) == 40
    assert CAlignRes.sCigar.offset == 24 and CAlignRes.nCigarLen.offset == 32


def test_product_does_not_import_oracle():
    """The product path must never import, load or link anything under oracle/ (it is the checker)."""
    import re
    pat = re.compile(r'(^|\n)\s*(from\s+oracle|import\s+oracle)|libssw_oracle|libmm2_oracle|mm2_oracle\.h|reassign_oracle|realign_oracle|_ref/libssw|_ref/librealigner')
    for path in glob.glob(os.path.join(ROOT, 'megapath_nano_amd', '**', '*'), recursive=True):
        if path.endswith(('.py', '.hip', '.h', '.cpp')):
            assert not pat.search(open(path).read()), f'{path} uses the oracle'
    for path in (os.path.join(ROOT, 'bin', 'mpn-aligner'),):
        assert not pat.search(open(path).read())


def test_headers_are_plain_c():
    """The boundary is a C ABI: every header must compile as C99 on its own (and as C++)."""
    import subprocess
    for h in sorted(glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        for lang, std in (('c', 'c99'), ('c++', 'c++11')):
            p = subprocess.run(['gcc', '-x', lang, f'-std={std}', '-Wall', '-Werror', '-fsyntax-only', '-'], input=f'#include "{h}"\n',
                               capture_output=True, text=True)
            assert p.returncode == 0, (h, lang, p.stderr[-500:])
