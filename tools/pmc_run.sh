#!/bin/bash
# Counter passes of one command, each set in its own rocprofv3 run (never combined with API tracing):
#   tools/pmc_run.sh <out dir prefix under gpurun_out> -- python3 <script> [args]
# Sets follow MI355X_MICROARCH.md's profiling section; summarise with tools/summarize_pmc_kernels.py.
prefix=$1; shift; shift
sets=(
 "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32"
 "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS"
 "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
)
cd /tmp && export TMPDIR=/tmp
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${prefix}_$i" -o run -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${prefix}_$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
