#!/bin/bash
# closing records of the round-5 tree (after gg_se_mlp / the launch-count pass) on ONE box: GPU suite, default bench line (with the
# reference CPU leg), rocprofv3 kernel statistics of the bench command, the per-step launch census of the graph replays, configs 4 / 5.
#   bash tests/gpu_r5_final2.sh        -> gpurun_out/r5c_final_*
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python -m pytest tests -m gpu -q > $O/r5c_final_pytest.log 2>&1
grep -E "passed|failed" $O/r5c_final_pytest.log | tail -1
python bench.py 2> $O/r5c_final_bench.err | tail -1 > $O/r5c_final_bench.json
cp $O/bench_gemm_shapes.json $O/r5c_final_gemm_shapes.json 2>/dev/null
python - <<'PY'
import json
r = json.load(open('gpurun_out/r5c_final_bench.json'))
m = r['roofline']['modconv_forward']
c = r['cpu_baseline']
print('bench', round(r['value'], 1), 'img/s', round(r['ms_per_step'], 2), 'ms finite', r['finite'], '| conv3', round(r['roofline']['achieved']), 'TF frac', round(r['roofline']['frac'], 3),
      '| all gemm', round(r['roofline']['all_gemm_kernels']['tflops']), '| short_k', round(r['roofline']['short_k']['achieved']), 'GB/s | modconv', round(m['graph_ms'], 4), 'ms frac', round(m['frac'], 4),
      '| cpu', c['kind'], round(c['value'], 3), 'img/s on', c['cores'], 'threads; port', c.get('port', {}).get('value'))
PY
rm -rf /tmp/prof_fin2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fin2 -o b -- python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-profile-cycle > $O/r5c_final_bench_under_rocprof.log 2>&1
cp "$(find /tmp/prof_fin2 -name '*kernel_stats.csv' | head -1)" $O/r5c_final_kernel_stats.csv
bash tests/gpu_replay_launch_census.sh > /dev/null 2>&1; cp $O/replay_launch_census.txt $O/r5c_final_replay_launch_census.txt; head -2 $O/r5c_final_replay_launch_census.txt
for w in text upsampler; do
    python bench.py --workload $w --steps 16 --warmup 8 2> $O/r5c_final_bench_$w.err | tail -1 > $O/r5c_final_bench_$w.json
    python - $w <<'PY'
import json, sys
r = json.load(open(f'gpurun_out/r5c_final_bench_{sys.argv[1]}.json'))
print(sys.argv[1], round(r['value'], 1), 'img/s', round(r['ms_per_step'], 2), 'ms finite', r['finite'], 'dominant', r['roofline']['kernel'] if r.get('roofline') else None)
PY
done
