#!/bin/bash
# round 4 records on ONE box: smoke, the default bench line (with the CPU baseline leg), config 4 / 5 benches, rocprof kernel stats
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=${1:-rec}
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
echo "=== default bench"; timeout 900 python bench.py 2>&1 | grep '^{' > $O/r04_${T}_bench_default.json
cp $O/bench_gemm_shapes.json $O/r04_${T}_gemm_shapes.json 2>/dev/null
python - "$O/r04_${T}_bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d['roofline']; c = d['cpu_baseline']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d['finite'], '| dominant', r['kernel'], round(r['achieved'], 1), r['unit'], 'frac', round(r['frac'], 3),
      '| all gemm', round(r['all_gemm_kernels']['tflops'], 1), 'TF', round(r['all_gemm_kernels']['ms_per_step'], 2), 'ms/step | modconv frac', round(r['modconv_forward']['frac'], 4))
print('cpu_baseline', {k: c[k] for k in c if k in ('value', 'unit', 'cores', 'kind', 'sample', 'port_vs_reference', 'reference_equivalent')})
print('memset nodes repaired', d['config'].get('graph_memset_nodes_repaired'))
PY
for w in text upsampler; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' > $O/r04_${T}_bench_$w.json
  python -c "
import json; d = json.load(open('$O/r04_${T}_bench_$w.json')); print('$w', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d['finite'], d['nonfinite'], d.get('last_losses'))"
done
bash tests/gpu_r4_prof.sh $T | head -30
