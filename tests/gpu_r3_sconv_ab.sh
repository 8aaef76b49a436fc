#!/bin/bash
# same-box A/B of gg_sconv: one accumulator chain per strip row (in-tree) vs two (gpurun_ab/libgigagan_amd_sc2.so)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
runsc() { python -c "
import sys, runpy; sys.path.insert(0, '.')
from gigagan_pytorch_amd import _C
if '$1' != 'tree': _C.bind('gpurun_ab/libgigagan_amd_sc2.so')
sys.argv = ['gpu_modconv_layers.py', '--only', 'sconv', '--json', 'gpurun_out/ab_sconv_$1.json']
runpy.run_path('tests/gpu_modconv_layers.py', run_name='__main__')
" 2>&1 | grep "best" | sed "s/^/$1 /" | cut -c1-120; }
runsc tree; runsc sc2; runsc tree; runsc sc2
