#!/bin/bash
# Builds tools/_trace/libnbd_hip_trace.so: the library with the fused ContinuousConv kernel's in-kernel probes
# compiled in (-DNBD_CC_TRACE: per-workgroup start / end stamps, steps, pairs, CU id, time waves spend waiting on
# the LDS flags, producer phase times; optional -DNBD_CC_ABL=1|2|3 timing-only ablations: filters of one cell /
# feature rows from a 64 KiB table / no MFMA). Run here (hipcc cross-compiles), then on the GPU box:
#   cp tools/_trace/libnbd_hip_trace.so nbody-deep-sim_amd/csrc/libnbd_hip.so && python tools/contconv_trace.py
# (the copy lives only in that box's snapshot). Extra flags: tools/build_contconv_trace.sh -DNBD_CC_ABL=3
set -e
cd "$(dirname "$0")/../nbody-deep-sim_amd/csrc"
make -s
mkdir -p ../../tools/_trace
OUT=${NBD_TRACE_OUT:-libnbd_hip_trace.so}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off ${NBD_NO_TRACE:--DNBD_CC_TRACE} "$@" -c contconv_fused.hip -o /tmp/contconv_fused_trace.o
objs=$(ls *.o | grep -v contconv_fused.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/contconv_fused_trace.o $objs -o ../../tools/_trace/$OUT
echo built tools/_trace/$OUT
