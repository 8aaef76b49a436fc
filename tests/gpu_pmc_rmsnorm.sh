#!/bin/bash
# HBM-side traffic of gg_rmsnorm_rows_kernel against its algorithmic bytes (separate --pmc passes with --kernel-trace only;
# MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in 32-byte... units as reported, FETCH_SIZE x 2 on gfx950 for wide coalesced reads).
mkdir -p gpurun_out; cd "$(dirname "$0")/.."; export TMPDIR=/tmp; R=$PWD
for grp in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmcr_$grp && timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmcr_$grp -o p -- python $R/tests/gpu_rmsnorm_probe.py > /tmp/pmcr_$grp.log 2>&1 )
done
python - <<'PY'
import csv, glob, collections, json
SHAPES = [(64 * 1024, 256), (64 * 256, 512), (32 * 1024, 256), (32 * 256, 512), (32 * 4096, 128), (64 * 4096, 128), (32 * 16384, 64), (16 * 4096, 256), (32 * 65536, 32)]
res = collections.defaultdict(dict)
for grp in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob(f'/tmp/pmcr_{grp}/**/*counter_collection.csv', recursive=True)
    if not f:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'gg_rmsnorm_rows_kernel' not in k:
            continue
        mode = 'fwd' if 'ILi0E' in k or '<0,' in k else 'bwd'
        per[(mode, int(r['Grid_Size']), k)].append(float(r['Counter_Value']))
    for key, v in per.items():
        res[key][grp] = sum(v) / len(v)
out = []
for (mode, grid, k), v in sorted(res.items(), key=lambda kv: -kv[0][1]):
    out.append(dict(kernel=k.split('(')[0][-40:], mode=mode, grid=grid, fetch_counter=v.get('FETCH_SIZE'), write_counter=v.get('WRITE_SIZE')))
json.dump(out, open('gpurun_out/r04_pmc_rmsnorm.json', 'w'), indent=1)
for o in out[:24]:
    print(o)
PY
