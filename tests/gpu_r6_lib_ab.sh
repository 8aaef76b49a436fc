#!/bin/bash
# same-box A/B of two builds of the library on the bench line: gpurun_ab/lib_old.so against gpurun_ab/lib_new.so (old, new, old, new)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for tag in old new old2 new2; do
    cp gpurun_ab/lib_${tag%2}.so gigagan_pytorch_amd/libgigagan_amd.so
    timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline 2> gpurun_out/r6_libab_$tag.err | tail -1 > gpurun_out/r6_libab_$tag.json
    python - "$tag" <<'PY'
import json, sys
try:
    r = json.load(open(f'gpurun_out/r6_libab_{sys.argv[1]}.json'))
    print(f"{sys.argv[1]:6s} img/s {r['value']:.1f} ms {r['ms_per_step']:.2f} conv3 {r['roofline']['achieved']:.0f} TF (avg launch {r['roofline'].get('avg_launch_us')}) all-gemm {r['roofline']['all_gemm_kernels']['tflops']:.0f} TF")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
cp gpurun_ab/lib_new.so gigagan_pytorch_amd/libgigagan_amd.so
