#!/bin/bash
# HBM-side traffic of the persistent short-K contraction (gg_pgemm, plan tile 15) against the tiled kernel (tile 6) on a 1x1 convolution of
# the FeedForward up-projection's shape (M 262144 = 256 x 32 x 32, K 256, N 1024; alpha-only epilogue): FETCH_SIZE and WRITE_SIZE in
# separate counter passes. Algorithmic bytes: 2 * M * (K + N) + 2 * N * K = 671.6 MB.   -> gpurun_out/pmc_pgemm.log
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
for tile in 15 6; do
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmc_pg && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_pg -o p -- python $R/tests/gpu_kernel_probe.py fwd 256 32 256 1024 1 $tile 5 > /tmp/pmc_pg.log 2>&1 )
  f=$(find /tmp/pmc_pg -name '*counter_collection.csv' | head -1)
  python - "$f" $c $tile <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gg_pgemm' in r.get('Kernel_Name', '') or 'gg_gemm2' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[2]]
name = rows[0]['Kernel_Name'][:60] if rows else None
print('tile', sys.argv[3], name, sys.argv[2], 'launches', len(vals), 'mean KiB', sum(vals) / max(1, len(vals)), 'min', min(vals) if vals else None, 'max', max(vals) if vals else None)
PY
done
done 2>&1 | tee gpurun_out/pmc_pgemm.log
