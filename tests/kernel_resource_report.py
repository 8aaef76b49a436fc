"""Registers, scratch and LDS of every kernel in the product library as hipcc allocates them (cross-compiles without a GPU):
    python tests/kernel_resource_report.py [--all]
prints the kernels that spill to scratch, sit at the 256-VGPR cap or hold more than 64 KB of LDS (--all: every kernel). This is the
listing that found gg_aconv's spilling 4x4 form and the register headroom for gg_conv3's two-workgroups-per-CU tile in round 6
(test infrastructure)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / 'gigagan_pytorch_amd' / 'csrc'
with tempfile.TemporaryDirectory() as tmp:
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm', '-amdgpu-mfma-vgpr-form', f'-I{ROOT / "include"}', f'-I{CSRC}',
           '-Rpass-analysis=kernel-resource-usage', '-c', str(CSRC / 'gg_api.hip'), '-o', f'{tmp}/a.o']
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
rows = []
for blk in re.split(r'remark: Function Name: ', txt)[1:]:
    def field(key):
        m = re.search(key + r': (\d+)', blk)
        return int(m.group(1)) if m else -1
    rows.append((blk.split()[0], field('VGPRs'), field(r'ScratchSize \[bytes/lane\]'), field(r'Occupancy \[waves/SIMD\]'),
                 field(r'LDS Size \[bytes/block\]')))
names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
show_all = '--all' in sys.argv[1:]
print(f'{len(rows)} kernels; static LDS only (kernels with `extern __shared__` report 0)')
for (_, vgpr, scratch, occ, lds), name in zip(rows, names):
    if show_all or scratch > 0 or vgpr >= 256 or lds > 65536:
        print(f'{name[:110]:110s} VGPR {vgpr:3d}  waves/SIMD by registers {occ}  LDS {lds:6d}  scratch {scratch} B/lane')
