#!/bin/bash
# where do gg_modcoef_bwd_w / _s spend their wave cycles? (SQ counters, separate pass from any tracing)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/pmc_mc
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_mc -o m -- python $GRAFT_REPO_ROOT/tests/gpu_modcoef_probe.py > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmc_mc/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'].split('(')[0]
    if 'modcoef' in k and r.get('Grid_Size') in ('131072',):      # the O = I = 512 launches (512 workgroups of 256)
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, v in agg.items():
    print(k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
done 2>&1 | tee $O/r04_pmc_modcoef.log
