#!/bin/bash
# Diagnosis batch of the fused ContinuousConv layer (round 4): layer timings, counters (matrix pipe, LDS, L2 hit / miss,
# fabric bytes), then the in-kernel trace with the probe build.   bash tools/r04_cc_diag.sh TAG   (gpurun_out/TAG_*)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04diag}
cd $R && mkdir -p gpurun_out
python tools/bench_contconv.py 20 > gpurun_out/${T}_layers.json 2> gpurun_out/${T}_layers.err || exit 1
cat gpurun_out/${T}_layers.json
tools/pmc_run.sh ${T}_pmc -- python3 $R/tools/bench_contconv.py 4
python tools/summarize_pmc_kernels.py "gpurun_out/${T}_pmc" gpurun_out/${T}_pmc_summary.json contconv_stream_kernel contconv_pairs contconv_stream_finish > /dev/null
python - <<P
import json
d = json.load(open("gpurun_out/${T}_pmc_summary.json"))
for k, v in d.items():
    if "stream_kernel" in k:
        print(k[:60], v["mean_seconds_in_pmc_passes"], json.dumps(v["counters_mean_per_launch"]), json.dumps(v["derived"]))
P
if [ -f tools/_trace/libnbd_hip_trace.so ]; then
  cp nbody-deep-sim_amd/csrc/libnbd_hip.so /tmp/libnbd_hip_product.so
  cp tools/_trace/libnbd_hip_trace.so nbody-deep-sim_amd/csrc/libnbd_hip.so
  python tools/contconv_trace.py > gpurun_out/${T}_trace.json 2> gpurun_out/${T}_trace.err
  cp /tmp/libnbd_hip_product.so nbody-deep-sim_amd/csrc/libnbd_hip.so
  cat gpurun_out/${T}_trace.json
fi
