#!/bin/bash
# PMC passes over the front end alone (tools/fe_bench.py: 11 launches of 1024 windows); one counter set per run.
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/fe_pmc1 gpurun_out/fe_pmc2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/fe_pmc1 -o p -- python3 tools/fe_bench.py > gpurun_out/fe_pmc1.out 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/fe_pmc2 -o p -- python3 tools/fe_bench.py > gpurun_out/fe_pmc2.out 2>&1
ls gpurun_out/fe_pmc1 gpurun_out/fe_pmc2
