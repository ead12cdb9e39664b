#!/bin/bash
prefix=$1; shift; shift
sets=(
 "GRBM_GUI_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "GRBM_GUI_ACTIVE SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
 "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
)
cd /tmp && export TMPDIR=/tmp
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${prefix}_$i" -o run -- "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${prefix}_$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
