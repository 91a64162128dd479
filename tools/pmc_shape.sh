#!/bin/bash
# rocprofv3 PMC passes of a few outer iterations of ONE shape (tools/one_shape.py), one counter group per pass, never with
# trace domains other than --kernel-trace.  Usage (GPU box, repository root): tools/pmc_shape.sh <out-dir> N S n_c n_u [T1]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ARGS="$@"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/one_shape.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"; echo "pass $name done"; }
pass sq_insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
