#!/bin/bash
# Round-2 evidence runs (one gpurun call): everything lands under gpurun_out/r02e_*.
set -x
cd $GRAFT_REPO_ROOT
python tools/shard_block.py --world 8 --rank 3 > gpurun_out/r02e_shard_w8.json
python tools/shard_block.py --world 4 --rank 1 > gpurun_out/r02e_shard_w4.json
python tools/shard_block.py --world 2 --rank 1 > gpurun_out/r02e_shard_w2.json
python tools/bench_gnn.py 50 > gpurun_out/r02e_gnn.json
python tools/bench_contconv.py 20 > gpurun_out/r02e_contconv.json
python tools/size_sweep.py > gpurun_out/r02e_size_sweep.jsonl 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_shard8 -o run -- python3 $GRAFT_REPO_ROOT/tools/shard_block.py --world 8 --rank 3 --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_shard8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_bench -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-surrogates > $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_gnn -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_gnn.py 50 > $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_gnn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_surr -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_surrogates.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r02e_prof_surr.log 2>&1
cd $GRAFT_REPO_ROOT
tools/pmc_run.sh r02e_pmc_bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --prewarm-seconds 0.2 --cpu-seconds 0 --no-surrogates
tools/pmc_run.sh r02e_pmc_cc -- python3 $GRAFT_REPO_ROOT/tools/bench_contconv.py 4
tools/pmc_run.sh r02e_pmc_gnn -- python3 $GRAFT_REPO_ROOT/tools/bench_gnn.py 10
for t in shard8 bench gnn surr; do python tools/summarize_trace.py $(find gpurun_out/r02e_prof_$t -name "*kernel_trace.csv" | head -1) gpurun_out/r02e_${t}_trace_summary.json > gpurun_out/r02e_${t}_trace_summary.txt; done
python tools/summarize_pmc_kernels.py gpurun_out/r02e_pmc_bench gpurun_out/r02e_pmc_bench_summary.json accel_kernel finish_kernel kick_drift > /dev/null
python tools/summarize_pmc_kernels.py gpurun_out/r02e_pmc_cc gpurun_out/r02e_pmc_cc_summary.json contconv_fused contconv_pairs contconv_finish linear_big contconv_bin > /dev/null
python tools/summarize_pmc_kernels.py gpurun_out/r02e_pmc_gnn gpurun_out/r02e_pmc_gnn_summary.json gnn_layer_kernel knn_select_kernel kick_drift axpy > /dev/null
cat gpurun_out/r02e_shard_w8.json gpurun_out/r02e_gnn.json gpurun_out/r02e_contconv.json
