#!/bin/sh
# probe-only build of the library with the 8-wave kernel's k-loop phase switches compiled in (GG2_PROBE); never shipped:
# the product library is built without the macro (Makefile / __graft_entry__.build()).
set -e
cd "$(dirname "$0")/../.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGG2_PROBE -shared -fPIC gigagan_pytorch_amd/csrc/gg_api.hip -o tests/probes/libgg_gemm2_probe.so
