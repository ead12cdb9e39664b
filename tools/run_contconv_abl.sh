#!/bin/bash
# Layer times (D = 6, D = 4; ms) of the product library and of every tools/_trace/libnbd_abl_*.so given by name:
#   bash tools/run_contconv_abl.sh P C M ...      (on the GPU box; the product library is put back at the end)
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp nbody-deep-sim_amd/csrc/libnbd_hip.so /tmp/prod.so
for v in prod "$@"; do
  if [ $v = prod ]; then cp /tmp/prod.so nbody-deep-sim_amd/csrc/libnbd_hip.so; else cp tools/_trace/libnbd_abl_$v.so nbody-deep-sim_amd/csrc/libnbd_hip.so; fi
  echo -n "$v: "; timeout -k 10 100 python tools/bench_contconv.py 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['D6']['fused_layer_ms'],4), round(d['D4']['fused_layer_ms'],4))"
done
cp /tmp/prod.so nbody-deep-sim_amd/csrc/libnbd_hip.so
