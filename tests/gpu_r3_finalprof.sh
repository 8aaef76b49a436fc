mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/finalprof_bench.log 2>&1 )
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} gpurun_out/finalprof_kernel_stats.csv \;
python - <<'PY'
import csv, collections, re
rows = list(csv.DictReader(open('gpurun_out/finalprof_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
pt = sum(float(r['TotalDurationNs']) for r in rows if 'at::' in r['Name'] or r['Name'].startswith('Cijk'))
print('kernel time ms', round(tot / 1e6, 1), 'launches', sum(int(r['Calls']) for r in rows), 'pytorch share %', round(pt / tot * 100, 2))
grp = collections.defaultdict(float)
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs'])
    k = ('pytorch' if 'at::' in n else 'gg_conv3' if 'gg_conv3' in n else 'gg_gemm2' if 'gg_gemm2' in n else 'gg_gemm(4w)' if 'gg_gemm_kernel' in n
         else 'attn_bwd2' if 'attn_bwd2' in n else 'attn' if 'attn' in n else re.sub(r'<.*', '', n.replace('void ', '')).split('(')[0])
    grp[k] += t
print({k: round(v / tot * 100, 1) for k, v in sorted(grp.items(), key=lambda kv: -kv[1])[:14]})
PY
grep '^{' gpurun_out/finalprof_bench.log | cut -c1-160
