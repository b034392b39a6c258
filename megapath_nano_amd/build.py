"""Builds megapath_nano_amd/libmpn.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmpn.so')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdr_t = max([os.path.getmtime(h) for h in headers()] + [0])
    objs, todo = [], []
    for src in sources():
        obj = os.path.join(CSRC, os.path.splitext(os.path.basename(src))[0] + '.o')
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append([hipcc, '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-c', src, '-o', obj])
        objs.append(obj)

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if todo:   # the translation units are independent: compile them side by side (hipcc is single-threaded per file)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(len(todo), os.cpu_count() or 1, 8))) as ex:
            list(ex.map(run, todo))
    rebuilt = bool(todo)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
