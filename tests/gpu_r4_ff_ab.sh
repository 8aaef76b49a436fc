#!/bin/bash
# same-box A/B of the GELU-on-epilogue FeedForward (GG_NO_FF_FUSE=1 = separate GELU passes), two alternations; replica bisect at c4 / c5
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for i in 1 2; do
  for v in 1 0; do
    if [ $v = 1 ]; then export GG_NO_FF_FUSE=1; else unset GG_NO_FF_FUSE; fi
    timeout 300 python bench.py --steps 24 --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print('GG_NO_FF_FUSE=' + os.environ.get('GG_NO_FF_FUSE', '0'), round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms')"
  done
done
unset GG_NO_FF_FUSE
timeout 300 python tests/gpu_replica_bisect45.py c4 2>&1 | grep -v amdgpu.ids | tail -20
timeout 300 python tests/gpu_replica_bisect45.py c5 2>&1 | grep -v amdgpu.ids | tail -20
