#!/bin/bash
# Round-4 evidence in one gpurun call (output under gpurun_out/r04f_*; the summaries are copied into profiles/ by hand):
#   headline bench line (with the general-mass block and the surrogate legs); kernel trace of the same command; the fused
#   ContinuousConv layers against round 3's fp32-MFMA kernel ON THE SAME BOX (tools/_trace/libnbd_abl_r03fp32.so, built from
#   git by tools/r04_build_r03_lib.sh); counters of the stream kernel; kernel trace of the ContinuousConv rollout step; the
#   scenes-together rollouts; the one-rank RCCL rehearsal of the captured sharded step.
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/r04f_bench_n1.json 2> gpurun_out/r04f_bench_n1.err
echo bench done
bash tools/run_contconv_abl.sh r03fp32 > gpurun_out/r04f_contconv_ab_same_box.txt 2>&1
python tools/bench_contconv.py 20 > gpurun_out/r04f_contconv_layers.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04f_prof_bench -o run -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-surrogates > $R/gpurun_out/r04f_prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04f_prof_cc -o run -- python3 $R/tools/cc_rollout.py 100 > $R/gpurun_out/r04f_prof_cc.log 2>&1
echo traces done
cd $R
python tools/cc_rollout.py 200 > gpurun_out/r04f_cc_rollout.json 2>/dev/null
tools/pmc_run.sh r04f_pmc_cc -- python3 $R/tools/bench_contconv.py 4
python tools/summarize_pmc_kernels.py "gpurun_out/r04f_pmc_cc" gpurun_out/r04f_pmc_cc_summary.json contconv_stream_kernel contconv_pairs contconv_stream_finish contconv_plan > /dev/null
bash tools/r04_cc_pmc.sh r04f_l2 > gpurun_out/r04f_l2_by_resolution.txt 2>&1
python tools/bench_scenes.py gnn 1000 > gpurun_out/r04f_scenes_gnn.json 2>/dev/null
python tools/bench_scenes.py contconv 200 > gpurun_out/r04f_scenes_contconv.json 2>/dev/null
NBD_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-surrogates 2>/dev/null | grep "^{" > gpurun_out/r04f_bench_rccl_one_rank_rehearsal.json
NBD_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 tools/shard_capture_check.py 8192 20 2>/dev/null | grep "^{" > gpurun_out/r04f_shard_capture_check.json
for t in bench cc; do
  f=$(find gpurun_out/r04f_prof_$t -name "*kernel_trace.csv" | head -1)
  python tools/summarize_trace.py $f gpurun_out/r04f_${t}_trace_summary.json > /dev/null
  cp $(find gpurun_out/r04f_prof_$t -name "*kernel_stats.csv" | head -1) gpurun_out/r04f_${t}_kernel_stats.csv
done
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r04f_bench_n1.json").read().strip().splitlines()[-1])
print({k: b[k] for k in ("value", "ms_per_step", "repeats")}, b["roofline"]["frac"], b["roofline"]["general_mass"]["frac"])
print(json.dumps(b["secondary"]["contconv_n16384"])[:900])
print(open("gpurun_out/r04f_contconv_ab_same_box.txt").read())
print(open("gpurun_out/r04f_cc_rollout.json").read())
PY
