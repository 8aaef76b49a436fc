#!/bin/bash
# same-box A/B of two builds of the library on the bench line: gpurun_ab/lib_old.so against gpurun_ab/lib_new.so (old, new, old, new);
# prints img/s, the dominant kernel and the adaptive-conv forward with its per-layer rows matching $1 (a grep pattern, optional)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for tag in old new old2 new2; do
    cp gpurun_ab/lib_${tag%2}.so gigagan_pytorch_amd/libgigagan_amd.so
    timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline 2> gpurun_out/r6_libab_$tag.err | tail -1 > gpurun_out/r6_libab_$tag.json
    python - "$tag" "${1:-@@}" <<'PY'
import json, sys
try:
    r = json.load(open(f'gpurun_out/r6_libab_{sys.argv[1]}.json'))
    m = r['roofline']['modconv_forward']
    print(f"{sys.argv[1]:6s} img/s {r['value']:.1f} ms {r['ms_per_step']:.2f} conv3 {r['roofline']['achieved']:.0f} TF all-gemm {r['roofline']['all_gemm_kernels']['tflops']:.0f} TF | modconv graph_ms {m['graph_ms']:.4f} frac {m['frac']:.4f}")
    for ly in m['layers']:
        if sys.argv[2] in ly['layer']: print(f"        {ly['layer']:40s} kernel_us {ly['kernel_us']:.1f}")
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
cp gpurun_ab/lib_new.so gigagan_pytorch_amd/libgigagan_amd.so
