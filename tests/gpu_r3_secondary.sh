#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for w in text upsampler; do
  timeout 900 python bench.py --workload $w --steps 16 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_bench_$w.log
  grep '^{' gpurun_out/r3_bench_$w.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d['finite'], 'graphs', d['config']['hip_graphs'], (d['roofline'] or {}).get('kernel'), round((d['roofline'] or {}).get('achieved', 0), 1))"
  grep -v '^{' gpurun_out/r3_bench_$w.log | tail -3
done
