#!/bin/bash
# HBM-side traffic of EVERY kernel of the training step: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over 4 eager steps
# (one 4-step cycle incl. the gradient-penalty step), aggregated per kernel name with the kernel-trace durations of the same run.
#   bash tests/gpu_pmc_step.sh  -> gpurun_out/pmc_step.json + pmc_step.log
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmc_step_$c && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_step_$c -o p -- python $R/bench.py --steps 4 --warmup 4 --no-graphs --no-cpu-baseline --no-profile-cycle > /tmp/pmc_step_$c.log 2>&1 )
done
python - <<'PY' 2>&1 | tee gpurun_out/pmc_step.log
import csv, glob, json, re
agg = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'/tmp/pmc_step_{c}/**/*counter_collection.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    n = len(rows)
    # the last 4 of 8 steps: the second half of the dispatches (warm-up steps have the same kernels)
    for r in rows[n // 2:]:
        if r.get('Counter_Name') != c:
            continue
        name = re.sub(r'\(.*', '', r['Kernel_Name'])[:90]
        a = agg.setdefault(name, dict(FETCH_SIZE=0.0, WRITE_SIZE=0.0, launches=0, ns=0.0))
        a[c] += float(r['Counter_Value'])
        if c == 'FETCH_SIZE':
            a['launches'] += 1
            a['ns'] += float(r.get('End_Timestamp', 0)) - float(r.get('Start_Timestamp', 0))
out = []
for k, a in agg.items():
    mb = (a['FETCH_SIZE'] * 2 + a['WRITE_SIZE']) * 1024 / 1e6          # gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md), KiB
    out.append(dict(kernel=k, launches=a['launches'], ms=a['ns'] / 1e6, fetch_MB_x2=a['FETCH_SIZE'] * 2 * 1024 / 1e6, write_MB=a['WRITE_SIZE'] * 1024 / 1e6,
                    TBps=(mb / 1e6) / (a['ns'] / 1e9) if a['ns'] else None))
out.sort(key=lambda d: -d['ms'])
json.dump(out, open('gpurun_out/pmc_step.json', 'w'), indent=1)
tot = sum(d['ms'] for d in out)
print(f'{len(out)} kernels, {tot:.1f} ms of kernel time in 4 eager steps under the counter pass')
for d in out[:45]:
    print(f"{d['ms']:8.2f} ms {d['launches']:5d} x  fetch {d['fetch_MB_x2']:9.1f} MB  write {d['write_MB']:9.1f} MB  {d['TBps'] or 0:5.2f} TB/s  {d['kernel']}")
PY
