#!/bin/bash
# Builds tools/_trace/OUT.so: the library with ONE source file recompiled with extra flags (in-kernel probes, timing-only
# ablations). hipcc cross-compiles here; on the GPU box run with NBD_LIB_OVERRIDE=tools/_trace/OUT.so.
#   tools/build_probe.sh graph.hip libnbd_knn_trace.so -DNBD_KNN_TRACE
set -e
SRC=$1; OUT=$2; shift 2
cd "$(dirname "$0")/../nbody-deep-sim_amd/csrc"
make -s
mkdir -p ../../tools/_trace
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c $SRC -o /tmp/probe_${SRC%.hip}.o
objs=$(ls *.o | grep -v "^${SRC%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/probe_${SRC%.hip}.o $objs -o ../../tools/_trace/$OUT
echo built tools/_trace/$OUT
