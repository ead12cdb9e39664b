#!/bin/bash
# tools/_trace/libnbd_abl_r03fp32.so: the library with round 3's fused ContinuousConv source (fp32 matrix instruction,
# commit 4a7cd8c) in place of the current one -- the comparator of the same-box A/B (tools/r04_evidence.sh). Not a product path.
set -e
cd "$(dirname "$0")/../nbody-deep-sim_amd/csrc"
make -s
mkdir -p /tmp/r03src ../../tools/_trace
git show 4a7cd8c:nbody-deep-sim_amd/csrc/contconv_fused.hip > /tmp/r03src/contconv_fused_r03.hip
sed -i 's#"../../include/nbd.h"#"'$(cd ../../include && pwd)'/nbd.h"#' /tmp/r03src/contconv_fused_r03.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c /tmp/r03src/contconv_fused_r03.hip -o /tmp/r03src/cc_r03.o
objs=$(ls *.o | grep -v contconv_fused.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/r03src/cc_r03.o $objs -o ../../tools/_trace/libnbd_abl_r03fp32.so
echo built tools/_trace/libnbd_abl_r03fp32.so
