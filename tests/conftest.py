import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

EMU_LIB = ROOT / 'tests' / 'emu' / 'libgigagan_amd_emu.so'
HIP_LIB = ROOT / 'gigagan_pytorch_amd' / 'libgigagan_amd.so'
REFERENCE = Path('/root/reference')


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """the CPU suite (`-m "not gpu"`) runs the kernels on the host-side emulator: ~75 CPU-minutes serially. When nobody asked for
    a worker count, spread it over the cores with pytest-xdist (installed in the image): ~12 minutes on 8 cores. GPU runs (`-m gpu`)
    stay in one process - one GPU, and the driver watches which libraries THAT process maps. GG_TEST_SERIAL=1 switches this off."""
    if os.environ.get('GG_TEST_SERIAL') or 'not gpu' not in (getattr(config.option, 'markexpr', '') or ''):
        return None
    if os.environ.get('PYTEST_XDIST_WORKER') or hasattr(config, 'workerinput'):     # already a worker: never spawn workers of its own
        return None
    if getattr(config.option, 'numprocesses', 'absent') is None and config.pluginmanager.hasplugin('xdist'):
        config.option.numprocesses = max(1, min(8, os.cpu_count() or 1))
    return None


def pytest_configure(config):
    if os.environ.get('GG_TEST_POISON'):    # torch.empty() returns NaN-filled memory: reads of never-written elements surface
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # stale native artefacts are rebuilt ONCE, by the controlling process, before any xdist worker starts (eight workers each
    # running hipcc into the same .so would race); workers find them fresh
    gpu_only = (getattr(config.option, 'markexpr', '') or '').strip() == 'gpu'       # the GPU box runs the prebuilt library
    if not gpu_only and not (os.environ.get('PYTEST_XDIST_WORKER') or hasattr(config, 'workerinput')):
        _ensure_built()


_built = False


def _ensure_built():
    """(re)build stale native artefacts once per session where a compiler exists (mtime-aware)."""
    global _built
    if _built or os.environ.get('PYTEST_XDIST_WORKER'):
        return
    _built = True
    import shutil
    if shutil.which('hipcc') or not (EMU_LIB.exists() and HIP_LIB.exists()):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(autouse=True)
def _bind_library(request):
    """CPU tests bind the C ABI compiled for the host-side kernel emulator; -m gpu tests bind the gfx950 build."""
    from gigagan_pytorch_amd import _C
    if request.node.get_closest_marker('gpu'):
        if not torch.cuda.is_available():
            pytest.skip('no GPU')
        _C.bind(HIP_LIB)
    else:
        _ensure_built()
        _C.bind(EMU_LIB)
    yield


@pytest.fixture
def reference():
    """the unmodified reference package, importable only where /root/reference exists (not on the GPU box)."""
    if not REFERENCE.exists():
        pytest.skip('/root/reference not present')
    stubs = str(ROOT / 'tests' / 'oracle_stubs')
    for p in (str(REFERENCE), stubs):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gigagan_pytorch
    return gigagan_pytorch


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12)).item()
