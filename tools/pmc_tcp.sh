#!/bin/bash
# Vector-memory-path counters (L1 / TA / TLB) of one command, each set in its own rocprofv3 run (the TA block takes at
# most three of its counters per pass: six in one set aborted rocprofv3 with "error code 38: Request exceeds the
# capabilities of the hardware to collect" in round 2 -- gpurun_out/r02f_tcp_cc_2.log -- and the later passes never ran):
#   tools/pmc_tcp.sh <out prefix under gpurun_out> <kernel substring> -- python3 <script> [args]
prefix=$1; kern=$2; shift; shift; shift
sets=(
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "GRBM_GUI_ACTIVE TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "GRBM_GUI_ACTIVE TA_TOTAL_WAVEFRONTS_sum TCP_GATE_EN1_sum"
 "GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
)
cd /tmp && export TMPDIR=/tmp
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${prefix}_$i" -o run -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${prefix}_$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 - "$GRAFT_REPO_ROOT/gpurun_out" "$prefix" "$kern" <<'PY'
import csv, glob, sys, collections, json
root, prefix, kern = sys.argv[1:4]
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/{prefix}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if kern in k:
            tag = "NS4" if "ILi4E" in k else "k"
            out[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {t: {c: sum(v) / len(v) for c, v in d.items()} for t, d in out.items()}
res["_launches"] = {t: max(len(v) for v in d.values()) for t, d in out.items()}
print(json.dumps(res))
PY
