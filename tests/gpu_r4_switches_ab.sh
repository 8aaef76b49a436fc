#!/bin/bash
# same-box A/B of everything round 4 added that has a switch (the queue, the hinge kernel, the graph repair and the launch trims have none)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-profile-cycle 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*:', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 2), 'ms finite', d['finite'])"; }
run A=default
run GG_NO_MODGRAM=1 GG_RMS_ROWS=0 GG_RESAMPLE_2X2=0 GG_NO_FF_FUSE=1 GG_SFWD=0 GG_WGRADS=0 GG_WB_NARROW=1
run A=default
run GG_NO_MODGRAM=1 GG_RMS_ROWS=0
