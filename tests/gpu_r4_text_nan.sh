#!/bin/bash
# which round-4 switch (if any) makes the text-conditional bench's last generator loss NaN? (its losses are ~1e5 .. 1e11 by step 4)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python bench.py --workload text --steps 8 --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', round(d['value'], 1), 'finite', d['finite'], d['last_losses'])"; }
run A=0
run GG_NO_FF_FUSE=1
run GG_SFWD=0
run GG_WGRADS=0
run GG_NO_FF_FUSE=1 GG_SFWD=0 GG_WGRADS=0 GG_WB_NARROW=1
