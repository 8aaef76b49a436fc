#!/bin/bash
# memory-pipeline counters of the streaming adaptive-conv kernels (gg_sconv<16|32|64>) inside a no-grad generator forward: L1 reads /
# hits / L2 requests, wave wait cycles and outstanding vector-memory instructions, texture-unit busy. Own passes, --kernel-trace only.
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
i=0
for grp in "TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TA_TA_BUSY TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && rm -rf /tmp/pmcs_$i && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcs_$i -o p -- python $R/tests/gpu_gforward_profile.py 3 > /tmp/pmcs_$i.log 2>&1 )
  f=$(find /tmp/pmcs_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|gg_sconv" "$f" > gpurun_out/pmc_sconv_$i.csv
  echo "pass $i: $(wc -l < gpurun_out/pmc_sconv_$i.csv 2>/dev/null) rows"; tail -2 /tmp/pmcs_$i.log | cut -c1-200
done
python - <<'PY'
import csv, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    try:
        for r in csv.DictReader(open(f'gpurun_out/pmc_sconv_{i}.csv')):
            key = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Grid_Size'])
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if i == 1:
                agg[key]['us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    except FileNotFoundError:
        pass
out = []
for key, c in sorted(agg.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    rec = dict(kernel=key[0], grid=key[1], **{k: round(v, 1) for k, v in m.items()})
    if m.get('TCP_PERF_SEL_TOTAL_READ'):
        rec['l1_hit_frac'] = round(m.get('TCP_PERF_SEL_TOTAL_HIT_LRU_READ', 0) / m['TCP_PERF_SEL_TOTAL_READ'], 3)
    if m.get('SQ_WAVE_CYCLES'):
        rec['wait_frac_of_wave_cycles'] = round(m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES'], 3)
    out.append(rec)
    print(rec)
json.dump(out, open('gpurun_out/pmc_sconv.json', 'w'), indent=1)
PY
