#!/bin/bash
# round 4 final records: -m gpu suite, default bench, text (config 4) and upsampler (config 5) benches, rocprof kernel stats
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O
bash tests/gpu_r4_check.sh final
cp $O/r04_final_pytest.log $O/r04_pytest_gpu.log
for w in text upsampler; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' > $O/r04_bench_$w.json
  python -c "
import json; d = json.load(open('$O/r04_bench_$w.json')); print('$w', round(d['value'], 1), 'img/s finite', d['finite'], d.get('last_losses'))"
done
bash tests/gpu_r4_prof.sh final
