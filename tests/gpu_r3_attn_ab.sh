#!/bin/bash
# same-box A/B of the second-order attention kernels: one wave per SIMD (in-tree build, 278 / 296 VGPRs) vs __launch_bounds__(256, 2)
# (gpurun_ab/libgigagan_amd_2wave.so: 256 VGPRs, 21 / 41 spilled registers)
cd "$(dirname "$0")/.."
run() { python -c "
import sys, runpy; sys.path.insert(0, '.')
from gigagan_pytorch_amd import _C
if '$1' != 'tree': _C.bind('gpurun_ab/libgigagan_amd_2wave.so')
sys.argv = ['gpu_attn_probe.py']
runpy.run_path('tests/gpu_attn_probe.py', run_name='__main__')
" 2>&1 | grep -v amdgpu.ids | grep "bwd2" | sed "s/^/$1 /"; }
run tree; run 2wave; run tree; run 2wave
# gg_sconv_kernel<32>: three waves per SIMD with 5 spilled registers (in-tree) vs two waves without spills (gpurun_ab/libgigagan_amd_sc2.so)
runsc() { python -c "
import sys, runpy; sys.path.insert(0, '.')
from gigagan_pytorch_amd import _C
if '$1' != 'tree': _C.bind('gpurun_ab/libgigagan_amd_sc2.so')
sys.argv = ['gpu_modconv_layers.py', '--only', 'sconv', '--json', 'gpurun_out/ab_sconv_$1.json']
runpy.run_path('tests/gpu_modconv_layers.py', run_name='__main__')
" 2>&1 | grep "best" | sed "s/^/$1 /" | cut -c1-120; }
runsc tree; runsc sc2; runsc tree; runsc sc2
