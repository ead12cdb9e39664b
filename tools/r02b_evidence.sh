#!/bin/bash
# Evidence runs for the kernels changed late in round 2 (fused ContinuousConv producers, GNN workgroup shape): one
# gpurun call, everything lands under gpurun_out/r02f_*.
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f_prof_surr -o run -- python3 $R/tools/bench_surrogates.py 10 > $R/gpurun_out/r02f_prof_surr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f_prof_gnn -o run -- python3 $R/tools/bench_gnn.py 50 > $R/gpurun_out/r02f_prof_gnn.log 2>&1
cd $R
tools/pmc_run.sh r02f_pmc_cc -- python3 $R/tools/bench_contconv.py 4
tools/pmc_run.sh r02f_pmc_gnn -- python3 $R/tools/bench_gnn.py 10
bash tools/pmc_tcp.sh r02f_tcp_cc contconv_fused -- python3 $R/tools/bench_contconv.py 4 > gpurun_out/r02f_tcp_cc_summary.json
for t in gnn surr; do python tools/summarize_trace.py $(find gpurun_out/r02f_prof_$t -name "*kernel_trace.csv" | head -1) gpurun_out/r02f_${t}_trace_summary.json > gpurun_out/r02f_${t}_trace_summary.txt; done
python tools/summarize_pmc_kernels.py gpurun_out/r02f_pmc_cc gpurun_out/r02f_pmc_cc_summary.json contconv_fused contconv_pairs contconv_finish > /dev/null
python tools/summarize_pmc_kernels.py gpurun_out/r02f_pmc_gnn gpurun_out/r02f_pmc_gnn_summary.json gnn_layer_kernel knn_select_kernel > /dev/null
cat gpurun_out/r02f_tcp_cc_summary.json
