#!/bin/bash
# rocprofv3 PMC passes of bench.py's hot loop (one counter group per pass; never combined with trace domains other than
# --kernel-trace).  Usage (on the GPU box, from the repository root): tools/pmc_passes.sh <out-dir-under-gpurun_out> [bench args]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ARGS=${@:-"--steps 4 --warmup 1 --restarts 2 --no-cpu-baseline"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"; echo "pass $name done"; }
pass sq_mfma SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 GRBM_GUI_ACTIVE
pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8
pass fetch FETCH_SIZE
pass write WRITE_SIZE
