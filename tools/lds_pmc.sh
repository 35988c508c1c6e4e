#!/bin/bash
# LDS counters of one 2048-window pass (bench.py --files 2 --seconds 1025)
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/lds_a
CMD="python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 2 --seconds 1025"
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/lds_a -o p -- $CMD > gpurun_out/lds_a.out 2>&1
ls gpurun_out/lds_a
