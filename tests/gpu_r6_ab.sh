#!/bin/bash
# same-box A/B of round-6 switches on the bench line (img/s, adaptive-conv forward graph time).   bash tests/gpu_r6_ab.sh "VAR=0 VAR2=0" ...
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
run() {
    tag=$1; shift
    env "$@" timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline 2> gpurun_out/r6_ab_$tag.err | tail -1 > gpurun_out/r6_ab_$tag.json
    python - "$tag" "$*" <<'PY'
import json, sys
try:
    r = json.load(open(f'gpurun_out/r6_ab_{sys.argv[1]}.json'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); sys.exit(0)
m = r['roofline']['modconv_forward']
print(f"{sys.argv[1]:10s} [{sys.argv[2]}] img/s {r['value']:.1f} ms {r['ms_per_step']:.2f} modconv graph_ms {m.get('graph_ms')} frac {m.get('frac')} conv3 {r['roofline']['achieved']:.0f} TF all-gemm {r['roofline']['all_gemm_kernels']['tflops']:.0f} TF")
for ly in m.get('layers', []):
    print(f"      {ly['layer']:40s} kernel_us {ly['kernel_us']:.1f}  {ly.get('launches')}")
PY
}
i=0
run base GG_R6_AB=base
for cfg in "$@"; do
    i=$((i + 1))
    run v$i $cfg
done
