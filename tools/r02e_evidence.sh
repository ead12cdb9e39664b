#!/bin/bash
# Final-state evidence of round 2 for the surrogate kernels (one gpurun call; output under gpurun_out/r02j_*).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02j_prof_surr -o run -- python3 $R/tools/bench_surrogates.py 10 > $R/gpurun_out/r02j_prof_surr.log 2>&1
echo trace done
cd $R
tools/pmc_run.sh r02j_pmc_cc -- python3 $R/tools/bench_contconv.py 4
echo pmc done
python tools/summarize_trace.py $(find gpurun_out/r02j_prof_surr -name "*kernel_trace.csv" | head -1) gpurun_out/r02j_surr_trace_summary.json > /dev/null
python tools/summarize_pmc_kernels.py gpurun_out/r02j_pmc_cc gpurun_out/r02j_pmc_cc_summary.json contconv_fused contconv_pairs contconv_finish > /dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02j_pmc_cc_summary.json"))
for k, v in d.items():
    print(k, v.get("mean_seconds_in_pmc_passes"), {a: v["derived"].get(a) for a in ("mfma_pipe_busy_frac", "valu_issue_busy_frac", "effective_clock_GHz")})
PY
