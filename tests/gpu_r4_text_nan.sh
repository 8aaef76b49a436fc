#!/bin/bash
# text-conditional bench (config 4; its losses are ~1e5 .. 1e12 within a few steps on random data): which buffers go non-finite, and with which switch?
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python bench.py --workload text --no-cpu-baseline --no-profile-cycle $EXTRA 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$* $EXTRA', round(d['value'], 1), 'finite', d['finite'], d['nonfinite'], d['last_losses'])"; }
EXTRA="--steps 8" run A=0
EXTRA="--steps 8" run GG_SFWD=0
run A=0
run GG_SFWD=0
EXTRA="--no-graphs" run A=0
EXTRA="--no-graphs" run GG_SFWD=0
run GG_NO_FF_FUSE=1 GG_SFWD=0 GG_WGRADS=0 GG_WB_NARROW=1
