#!/bin/bash
# L2 / fabric counters of the fused ContinuousConv layer (D = 6 and D = 4 launches apart).  bash tools/r04_cc_pmc.sh TAG [given|morton|random]
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r04pmc}; ORDER=${2:-given}
cd /tmp && export TMPDIR=/tmp
i=0
for s in "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  rocprofv3 --pmc $s --output-format csv -d $R/gpurun_out/${T}_$i -o run -- python3 $R/tools/bench_contconv.py 4 $ORDER > $R/gpurun_out/${T}_$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - <<P
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/${T}_*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "contconv_stream_kernel" in r["Kernel_Name"]]
    # launches alternate: the tool runs D = 6 (warm-up + timed) first, then D = 4; split by dispatch order halves
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    half = ids[len(ids) // 2]
    for r in rows:
        agg["D6" if int(r["Dispatch_Id"]) < half else "D4"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    c = {a: sum(b) / len(b) for a, b in d.items()}
    out = {a: round(b) for a, b in c.items()}
    if "TCC_HIT_sum" in c: out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
    if "FETCH_SIZE" in c: out["fetch_MB_x2"] = round(c["FETCH_SIZE"] * 2 / 1024, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c: out["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
    print(k, out)
P
