#!/bin/bash
# rocprofv3 kernel stats of the bench command on the current tree -> gpurun_out/r04_<tag>_kernel_stats.csv (+ the bench line under rocprof)
mkdir -p gpurun_out; cd "$(dirname "$0")/.."; export TMPDIR=/tmp
T=${1:-prof}; O=gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-profile-cycle > $GRAFT_REPO_ROOT/$O/r04_${T}_bench_under_rocprof.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $O/r04_${T}_kernel_stats.csv \;
python - "$O/r04_${T}_kernel_stats.csv" <<'PY'
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 28
pt = [r for r in rows if 'at::' in r['Name'] or r['Name'].startswith('Cijk') or 'rocclr' in r['Name']]
print('kernel ms/step', round(tot / 1e6 / steps, 2), 'launches/step', round(sum(int(r['Calls']) for r in rows) / steps), 'pytorch ms/step',
      round(sum(float(r['TotalDurationNs']) for r in pt) / 1e6 / steps, 2), 'pytorch launches/step', round(sum(int(r['Calls']) for r in pt) / steps))
grp = collections.defaultdict(lambda: [0., 0])
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs'])
    k = ('pytorch' if ('at::' in n or 'rocclr' in n) else 'gg_conv3' if 'gg_conv3' in n else 'gg_gemm2' if 'gg_gemm2' in n else 'gg_gemm(4w)' if 'gg_gemm_kernel' in n
         else 'attn_bwd2' if 'attn_bwd2' in n else 'attn' if 'attn' in n else re.sub(r'<.*', '', n.replace('void ', '')).split('(')[0])
    grp[k][0] += t; grp[k][1] += int(r['Calls'])
for k, (t, c) in sorted(grp.items(), key=lambda kv: -kv[1][0])[:26]:
    print(f'{t / 1e6 / steps:7.2f} ms/step {t / tot * 100:5.1f}%  x{c / steps:6.1f}  {k}')
PY
grep '^{' $O/r04_${T}_bench_under_rocprof.log | cut -c1-200
