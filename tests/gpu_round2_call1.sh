#!/bin/bash
# Round 2, GPU call 1: parity at config 2 + the whole -m gpu suite, the finite bench line, the MFMA ceiling probe, the race screen
# of the experimental ring kernels, the A/B of every default-off switch, rocprofv3 kernel stats + SQ counters of the dominant kernel.
#   gpurun --timeout 1500 -- 'bash tests/gpu_round2_call1.sh'
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
step() { echo "=== $1 ($(date +%T))"; }
step pytest; timeout 900 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | grep -v amdgpu.ids > $O/r2c1_pytest.log; tail -30 $O/r2c1_pytest.log
step mfma_probe; timeout 120 tests/probes/bin/mfma_probe 2>&1 | tee $O/r2c1_mfma_probe.log
step bench; timeout 400 python bench.py 2>&1 | grep -v amdgpu.ids > $O/r2c1_bench.log; grep '^{' $O/r2c1_bench.log | cut -c1-600
step ring_probe; timeout 200 python tests/gpu_ring_gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r2c1_ring_probe.log
step ab
for cfg in "" "GG_WGRAD_FUSED=1" "GG_MODCONV_NARROW=1" "GG_MODCONV_PREMOD=1" "GG_GEMM_V3=1"; do
  echo "== bench.py [$cfg]"
  env $cfg timeout 150 python bench.py --no-cpu-baseline --steps 8 --warmup 4 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['roofline']['modconv_forward']
print('[$cfg]', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d.get('finite'), '; dominant', d['roofline']['kernel'][:40], round(d['roofline']['achieved'], 1), 'TF; modconv fwd', round(m['achieved'], 1), 'TF', round(m['kernel_ms'], 3), 'ms kernel', round(m['call_ms'], 3), 'ms calls')
for l in m['layers']: print('   ', l['layer'], round(l['kernel_us'], 1), 'us', round(l['kernel_tflops'], 1), 'TF')
"
done 2>&1 | tee $O/r2c1_ab.log
step rocprof_stats
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 8 --warmup 4 > $GRAFT_REPO_ROOT/$O/r2c1_bench_under_rocprof.log 2>&1 )
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $O/r2c1_kernel_stats.csv \; ; head -25 $O/r2c1_kernel_stats.csv | cut -c1-200
step pmc
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d /tmp/prof_pmc -o r2 -- python $GRAFT_REPO_ROOT/tests/gpu_gemm_bench.py --pmc-shapes > $GRAFT_REPO_ROOT/$O/r2c1_pmc_run.log 2>&1 )
find /tmp/prof_pmc -name '*counter_collection.csv' -exec cp {} $O/r2c1_pmc_sq_counter_collection.csv \; ; wc -l $O/r2c1_pmc_sq_counter_collection.csv
step done
