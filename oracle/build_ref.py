"""ORACLE — test infrastructure only: the recipe that compiles the reference itself into `oracle/_ref/`.

The reference is Python, so "compiling it from the sources where they lie" means byte-compiling
`/root/reference/gigagan_pytorch/*.py` (read in place, never copied) into SOURCELESS modules
`oracle/_ref/gigagan_pytorch/<module>.pyc`. `oracle/_ref/` is git-ignored (binary artefacts, out of history) but not
gpurun-ignored, so it travels to the GPU box like the built `.so` files: `bench.py`'s `cpu_baseline` leg can then time the
UNMODIFIED reference trainer (`GigaGAN(...)(steps=4)`) on the GPU box's host cores (`kind: "reference"`)
instead of a port scaled by a ratio measured elsewhere. Third-party imports the image lacks (beartype, kornia, ema_pytorch,
open_clip, numerize, torchvision) come from `tests/oracle_stubs` (our own inert stand-ins).

Nothing under `gigagan_pytorch_amd/` imports this; only `__graft_entry__.build()` (build), `bench.py`'s cpu_baseline (time)
and tests use it. Same interpreter version here and on the GPU box (one image), which is what a .pyc needs.

    python oracle/build_ref.py            # -> oracle/_ref/gigagan_pytorch/*.pyc  (no-op without /root/reference)
"""
from __future__ import annotations

import py_compile
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
REFERENCE_PKG = Path('/root/reference/gigagan_pytorch')
OUT = ROOT / 'oracle' / '_ref' / 'gigagan_pytorch'
STUBS = ROOT / 'tests' / 'oracle_stubs'


def build_ref(force: bool = False) -> bool:
    """byte-compile the reference package into oracle/_ref (mtime-aware). Returns whether oracle/_ref is usable."""
    if not REFERENCE_PKG.is_dir():
        return available()
    OUT.mkdir(parents=True, exist_ok=True)
    for src in sorted(REFERENCE_PKG.glob('*.py')):
        dst = OUT / (src.stem + '.pyc')
        if force or not dst.exists() or dst.stat().st_mtime < src.stat().st_mtime:
            # dfile: the path tracebacks name (the reference's own file:line, what the docstrings in this repo cite)
            py_compile.compile(str(src), cfile=str(dst), dfile=str(src), doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    return available()


def available() -> bool:
    return (OUT / '__init__.pyc').exists() and (OUT / 'gigagan_pytorch.pyc').exists()


def import_reference():
    """the reference package as the GPU box can import it: sourceless modules from oracle/_ref + the stub third-party packages.
    (Where /root/reference exists the very same byte code is what `import gigagan_pytorch` from there would execute.)"""
    if not available():
        raise ImportError('oracle/_ref is empty: run `python oracle/build_ref.py` where /root/reference exists')
    for p in (str(STUBS), str(OUT.parent)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gigagan_pytorch
    return gigagan_pytorch


if __name__ == '__main__':
    ok = build_ref(force='--force' in sys.argv)
    print('oracle/_ref:', 'ready' if ok else 'not built (/root/reference absent)')
