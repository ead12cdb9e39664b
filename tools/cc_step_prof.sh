#!/bin/bash
# ContinuousConv rollout step (configs[3]): un-profiled step times, then the per-kernel means of the same workload under
# rocprofv3 --kernel-trace --stats.   bash tools/cc_step_prof.sh TAG   (writes gpurun_out/TAG_*)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-cc}
python3 $R/tools/cc_rollout.py 200 > $R/gpurun_out/${T}_rollout.json 2> $R/gpurun_out/${T}_rollout.err || exit 1
cat $R/gpurun_out/${T}_rollout.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o run -- python3 $R/tools/cc_rollout.py 100 > $R/gpurun_out/${T}_prof.log 2>&1 || exit 1
python3 - <<P
import csv,glob
f=glob.glob("$R/gpurun_out/${T}_prof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
P
