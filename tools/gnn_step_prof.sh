#!/bin/bash
# GNN rollout step (configs[2]): un-profiled captured / eager step times, then the per-kernel means of the same
# workload under rocprofv3 --kernel-trace --stats.   bash tools/gnn_step_prof.sh TAG   (writes gpurun_out/TAG_*)
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-gnn}
python3 $R/tools/bench_gnn.py 200 > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err || exit 1
cat $R/gpurun_out/${T}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o run -- python3 $R/tools/bench_gnn.py 50 > $R/gpurun_out/${T}_prof.log 2>&1 || exit 1
python3 - <<P
import csv,glob
f=glob.glob("$R/gpurun_out/${T}_prof/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"])
P
