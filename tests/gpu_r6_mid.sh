#!/bin/bash
# round-6 mid-round records on ONE box: GPU suite, default bench line, rocprofv3 kernel statistics of the bench command.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/r6_mid_pytest.log 2>&1
tail -8 $O/r6_mid_pytest.log
timeout 900 python bench.py 2> $O/r6_mid_bench.err | tail -1 > $O/r6_mid_bench.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/r6_mid_bench.json'))
m = r['roofline']['modconv_forward']
c = r['cpu_baseline']
print('bench', round(r['value'], 1), 'img/s', round(r['ms_per_step'], 2), 'ms finite', r['finite'], '| conv3', round(r['roofline']['achieved']), 'TF frac', round(r['roofline']['frac'], 3),
      '| all gemm', round(r['roofline']['all_gemm_kernels']['tflops']), '| modconv', round(m['graph_ms'], 4), 'ms frac', round(m['frac'], 4),
      '| cpu', c['kind'], round(c['value'], 3), 'img/s on', c['cores'], 'threads')
PY
rm -rf /tmp/prof_mid
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mid -o b -- python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-profile-cycle > $O/r6_mid_bench_under_rocprof.log 2>&1
cp "$(find /tmp/prof_mid -name '*kernel_stats.csv' | head -1)" $O/r6_mid_kernel_stats.csv
head -25 $O/r6_mid_kernel_stats.csv | cut -c1-150
