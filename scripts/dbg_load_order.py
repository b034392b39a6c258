"""debug: libmpn.so loaded before / after torch (run on the GPU box), with and without RTLD_DEEPBIND"""
import ctypes as ct, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else None
if variant is None:
    for v in ('mpn_first', 'torch_first', 'mpn_first_deep', 'torch_first_deep'):
        p = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(v, '->', (p.stdout.strip().splitlines() or ['-'])[-1], '|', (p.stderr.strip().splitlines() or ['-'])[-1][-200:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
from megapath_nano_amd import _ffi
if variant.endswith('deep'):
    _ffi.lib.__globals__['_MODE'] = os.RTLD_NOW | os.RTLD_DEEPBIND
if variant.startswith('torch_first'):
    import torch
    torch.zeros(4, device='cuda')
from megapath_nano_amd import mapper, synth
mapper._bind()
import torch
x = torch.zeros(4, device='cuda')
gen = synth.make_genomes(1, 2, 50000, strain_pairs=0)
idx = mapper.Index(gen)
names, flat, lens = synth.make_genomes_device(5, 3, 100000, 0, torch.device('cuda', 0))
torch.cuda.synchronize()
idx2 = mapper.Index.from_device(names, flat.data_ptr(), lens)
libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libhsa-runtime64' in l or 'libamdhip64' in l})
print('ok', idx.n_minimizers, idx2.n_minimizers, libs)
