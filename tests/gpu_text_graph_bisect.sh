#!/bin/bash
# text-conditional plain D step: second replay vs eager under each debugging switch
cd "$(dirname "$0")/.."; export TRIG_SHORT=1
run() { echo "## $*"; env "$@" timeout 300 python tests/gpu_text_graph_trigger.py ${MODE:-plain} 2>&1 | grep -E "^replay 2|differing"; }
run A=0
run GG_DEBUG_NO_GRAD_SINK=1
run GG_DEBUG_NO_PACK_TABLE=1
run GG_NO_FF_FUSE=1
run GG_SFWD=0 GG_WGRADS=0
run GG_GEMM_V2=0
run GG_CONV3=0 GG_WGRAD9=0 GG_DCONV=0
run GG_NO_PLAN_TABLE=1
