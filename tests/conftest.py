import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle_built():
    """Build the CPU oracles (test infrastructure) once per session."""
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')], stdout=subprocess.DEVNULL)
    return True


@pytest.fixture(scope='session')
def libmpn():
    # On a GPU box the tests run on the runtime stack bench.py runs on: PyTorch-ROCm initialises the GPU first, libmpn.so is
    # loaded afterwards (megapath_nano_amd/_ffi.py hint(): the other order leaves two HSA runtimes in the process)
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except ImportError:
        pass
    from megapath_nano_amd import build, _ffi
    build.build()
    return _ffi.lib()
