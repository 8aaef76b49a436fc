#!/bin/bash
# round-6 closing records on ONE box: GPU suite, default bench line (with the reference CPU leg), rocprofv3 kernel statistics of the
# bench command, PMC traffic of the dominant kernel, PMC summary of the adaptive-conv forward, configs 4 / 5.
#   bash tests/gpu_r6_final.sh        -> gpurun_out/r6_final_*
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q > $O/r6_final_pytest.log 2>&1
tail -3 $O/r6_final_pytest.log | head -2
python bench.py 2> $O/r6_final_bench.err | tail -1 > $O/r6_final_bench.json
cp $O/bench_gemm_shapes.json $O/r6_final_gemm_shapes.json 2>/dev/null
python - <<'PY'
import json
r = json.load(open('gpurun_out/r6_final_bench.json'))
m = r['roofline']['modconv_forward']
c = r['cpu_baseline']
print('bench', round(r['value'], 1), 'img/s', round(r['ms_per_step'], 2), 'ms finite', r['finite'], '| conv3', round(r['roofline']['achieved']), 'TF frac', round(r['roofline']['frac'], 3),
      '| all gemm', round(r['roofline']['all_gemm_kernels']['tflops']), '| modconv', round(m['graph_ms'], 4), 'ms frac', round(m['frac'], 4),
      '| cpu', c['kind'], round(c['value'], 3), 'img/s on', c['cores'], 'threads; port', c.get('port', {}).get('value'))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fin -o b -- python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-profile-cycle > $O/r6_final_bench_under_rocprof.log 2>&1
cp "$(find /tmp/prof_fin -name '*kernel_stats.csv' | head -1)" $O/r6_final_kernel_stats.csv
bash tests/gpu_pmc_conv3.sh > /dev/null 2>&1; cp $O/pmc_conv3.log $O/r6_final_pmc_conv3.log; cat $O/r6_final_pmc_conv3.log
bash tests/gpu_pmc_modconv.sh > $O/r6_final_pmc_modconv.log 2>&1; cp $O/pmc_modconv.json $O/r6_final_pmc_modconv.json; tail -12 $O/r6_final_pmc_modconv.log | cut -c1-220
for w in text upsampler; do
    python bench.py --workload $w --steps 16 --warmup 8 2> $O/r6_final_bench_$w.err | tail -1 > $O/r6_final_bench_$w.json
    python - $w <<'PY'
import json, sys
r = json.load(open(f'gpurun_out/r6_final_bench_{sys.argv[1]}.json'))
print(sys.argv[1], round(r['value'], 1), 'img/s', round(r['ms_per_step'], 2), 'ms finite', r['finite'], 'dominant', r['roofline']['kernel'] if r.get('roofline') else None)
PY
done
