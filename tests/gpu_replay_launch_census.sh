#!/bin/bash
# launches per steady-state step by kernel: rocprofv3 kernel statistics of the bench command at two step counts, differenced
# (everything before the timed region - warm-up, captures, the profile cycle - cancels).   -> gpurun_out/replay_launch_census.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for n in 8 24; do
    rm -rf /tmp/prof_lc$n
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lc$n -o b -- python bench.py --steps $n --warmup 8 --no-cpu-baseline --no-profile-cycle > $O/replay_lc_$n.log 2>&1
    cp "$(find /tmp/prof_lc$n -name '*kernel_stats.csv' | head -1)" $O/replay_lc_$n.csv
done
python - <<'PY'
import csv, re
def load(n):
    return {r['Name']: (int(r['Calls']), int(r['TotalDurationNs'])) for r in csv.DictReader(open(f'gpurun_out/replay_lc_{n}.csv'))}
a, b = load(8), load(24)
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0))
    if c - c0 > 0:
        rows.append(((c - c0) / 16, (t - t0) / 16e3, k))
tot_c, tot_t = sum(r[0] for r in rows), sum(r[1] for r in rows)
with open('gpurun_out/replay_launch_census.txt', 'w') as f:
    f.write(f'steady-state step: {tot_c:.1f} launches, {tot_t / 1e3:.2f} ms of kernel time (under the profiler)\n')
    at = [r for r in rows if not re.match(r'(void )?gg_', r[2])]
    f.write(f'PyTorch / runtime kernels: {sum(r[0] for r in at):.1f} launches, {sum(r[1] for r in at) / 1e3:.2f} ms\n')
    for c, t, k in sorted(rows, key=lambda r: -r[0]):
        f.write(f'{c:8.2f} /step {t:9.1f} us/step  avg {t / c:7.1f} us  {re.sub("at::native::", "", k)[:170]}\n')
print(open('gpurun_out/replay_launch_census.txt').read()[:6000])
PY
