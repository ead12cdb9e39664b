#!/bin/bash
# Timing-only ablation builds of the fused ContinuousConv kernel (results are WRONG by construction; never the product):
#   tools/build_contconv_abl.sh NAME -DNBD_ABL_x ...   -> tools/_trace/libnbd_abl_NAME.so
# The switches live in tools/_trace/cc_ablations.py, which patches a temporary copy of csrc/contconv_fused.hip:
#   NBD_ABL_B  filter cells folded to 16 (the fragment stays L2-resident)      NBD_ABL_G  feature rows folded to 1024
#   NBD_ABL_P  producers publish without gathering (the consumers' own speed)  NBD_ABL_C  consumers hand buffers straight back
#   NBD_ABL_M  no MFMA            add -DNBD_CC_TRACE for the in-kernel stamps (tools/contconv_trace.py)
set -e
name=$1; shift
cd "$(dirname "$0")/../nbody-deep-sim_amd/csrc"
make -s
cp contconv_fused.hip /tmp/contconv_fused_abl.hip
python3 ../../tools/_trace/cc_ablations.py /tmp/contconv_fused_abl.hip
sed -i 's#"../../include/nbd.h"#"'$(cd ../../include && pwd)'/nbd.h"#' /tmp/contconv_fused_abl.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -c /tmp/contconv_fused_abl.hip -o /tmp/contconv_fused_abl.o
objs=$(ls *.o | grep -v contconv_fused.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/contconv_fused_abl.o $objs -o ../../tools/_trace/libnbd_abl_$name.so
echo built tools/_trace/libnbd_abl_$name.so
