#!/bin/bash
# closing run of round 3: the -m gpu suite three times in a row (flake watch), rocprofv3 kernel stats of the bench command, default bench
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp LIBC_FATAL_STDERR_=1
O=gpurun_out
for i in 1 2 3; do
  echo "=== pytest $i"; timeout 900 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu.ids > $O/close_pytest_$i.log; grep -n "passed\|failed\|Fatal\|Abort" $O/close_pytest_$i.log | tail -3
done
echo "=== prof"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/close_bench_under_rocprof.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/close_kernel_stats.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/close_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
pt = sum(float(r['TotalDurationNs']) for r in rows if 'at::' in r['Name'] or r['Name'].startswith('Cijk'))
print('kernel time ms', round(tot / 1e6, 1), 'launches', sum(int(r['Calls']) for r in rows), 'pytorch share %', round(pt / tot * 100, 2),
      'pytorch launches', sum(int(r['Calls']) for r in rows if 'at::' in r['Name']))
PY
echo "=== bench"; timeout 500 python bench.py 2>&1 | grep -v amdgpu.ids > $O/close_bench.log; grep '^{' $O/close_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms; dominant', round(d['roofline']['achieved'], 1), 'TF', round(d['roofline']['avg_launch_us'], 1), 'us; all gemm', round(d['roofline']['all_gemm_kernels']['tflops'], 1), 'TF; modconv graph', round(m['graph_ms'], 4), 'frac', round(m['frac'], 4), 'traffic', d['roofline']['traffic'])
print('cpu', d.get('cpu_baseline'))
"
echo "=== done"
