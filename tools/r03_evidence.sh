#!/bin/bash
# Round-3 evidence in one gpurun call (output under gpurun_out/r03f_*; the summaries are copied into profiles/ by hand):
#   headline bench line; kernel trace of the same command; ContinuousConv layer timings, counters and rollout trace.
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/r03f_bench_n1.json 2> gpurun_out/r03f_bench_n1.err
echo bench done
python tools/bench_contconv.py 20 > gpurun_out/r03f_contconv_layers.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f_prof_bench -o run -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-surrogates > $R/gpurun_out/r03f_prof_bench.log 2>&1
echo bench trace done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f_prof_cc -o run -- python3 $R/tools/cc_rollout.py 100 > $R/gpurun_out/r03f_prof_cc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f_prof_gnn -o run -- python3 $R/tools/bench_gnn.py 50 > $R/gpurun_out/r03f_prof_gnn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f_prof_train -o run -- python3 $R/tools/bench_train.py 20 > $R/gpurun_out/r03f_prof_train.log 2>&1
echo traces done
cd $R
python tools/bench_train.py 20 > gpurun_out/r03f_train_bench.json 2>/dev/null
python tools/train_phases.py 30 > gpurun_out/r03f_train_phases.json 2>/dev/null
echo train done
tools/pmc_run.sh r03f_pmc_bench -- python3 $R/bench.py --steps 10 --warmup 5 --prewarm-seconds 0.2 --cpu-seconds 0 --no-surrogates
python tools/summarize_pmc_kernels.py "gpurun_out/r03f_pmc_bench" gpurun_out/r03f_pmc_bench_summary.json accel_kernel finish_kernel kick_drift > /dev/null
tools/pmc_run.sh r03f_pmc_cc -- python3 $R/tools/bench_contconv.py 4
tools/pmc_lds.sh r03f_pmclds_cc -- python3 $R/tools/bench_contconv.py 4
python tools/summarize_pmc_kernels.py "gpurun_out/r03f_pmc_cc" gpurun_out/r03f_pmc_cc_summary.json contconv_stream_kernel contconv_pairs contconv_stream_finish > /dev/null
tools/pmc_run.sh r03f_pmc_gnn -- python3 $R/tools/bench_gnn.py 20
python tools/summarize_pmc_kernels.py "gpurun_out/r03f_pmc_gnn" gpurun_out/r03f_pmc_gnn_summary.json gnn_layer64_kernel knn_select_staged_kernel > /dev/null
python tools/bench_gnn.py 300 > gpurun_out/r03f_gnn_bench.json 2>/dev/null
for t in bench cc gnn train; do
  f=$(find gpurun_out/r03f_prof_$t -name "*kernel_trace.csv" | head -1)
  python tools/summarize_trace.py $f gpurun_out/r03f_${t}_trace_summary.json > /dev/null
  cp $(find gpurun_out/r03f_prof_$t -name "*kernel_stats.csv" | head -1) gpurun_out/r03f_${t}_kernel_stats.csv
done
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03f_pmc_cc_summary.json"))
for k, v in d.items():
    print(k[:70], v.get("mean_seconds_in_pmc_passes"), {a: v["derived"].get(a) for a in ("mfma_pipe_busy_frac", "valu_issue_busy_frac", "effective_clock_GHz")})
b = json.loads(open("gpurun_out/r03f_bench_n1.json").read().strip().splitlines()[-1])
print({k: b[k] for k in ("value", "ms_per_step", "repeats")}, b["roofline"]["frac"])
print(json.dumps(b["secondary"]["contconv_n16384"])[:600])
PY
