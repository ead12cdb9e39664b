#!/bin/bash
# Round-3 evidence for the K1 matrix-pipe prototype (tools/k1_mfma.hip): timing + fp64 row error (plain run), then the
# counters of each variant alone in its own rocprofv3 --pmc pass. Output: gpurun_out/r03_k1_mfma*.{json,txt}
R=$GRAFT_REPO_ROOT
$R/tools/k1_mfma > $R/gpurun_out/r03_k1_mfma_run.json
cat $R/gpurun_out/r03_k1_mfma_run.json
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d $R/gpurun_out/r03_k1_mfma_pmc_$v -o run -- $R/tools/k1_mfma $v > $R/gpurun_out/r03_k1_mfma_pmc_$v.log 2>&1 || echo "pmc pass $v failed"
done
cd $R
python3 - <<'PY'
import csv, glob, json, collections
out = {"run": json.loads(open("gpurun_out/r03_k1_mfma_run.json").read().strip().splitlines()[-1])}
for v, name in ((0, "valu"), (1, "mfma_4x4x1")):
    agg = collections.defaultdict(list); dur = []
    for f in glob.glob(f"gpurun_out/r03_k1_mfma_pmc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    c = {k: sum(x) / len(x) for k, x in agg.items()}
    if not c: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out[name + "_pmc"] = {"counters_mean_per_launch": c, "mean_seconds_in_pmc_pass": sum(dur) / len(dur), "gpu_cycles": cyc,
                          "valu_issue_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
                          "mfma_pipe_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)}
json.dump(out, open("gpurun_out/r03_k1_mfma.json", "w"), indent=1)
print(json.dumps({k: (v if k == "run" else {a: v[a] for a in ("valu_issue_busy_frac", "mfma_pipe_busy_frac", "mean_seconds_in_pmc_pass")}) for k, v in out.items()}))
PY
