#!/bin/bash
# L2 / texture-path counters of one 2048-window pass (bench.py --files 2 --seconds 1025), one counter set per run.
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/l2_a gpurun_out/l2_b
CMD="python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 2 --seconds 1025"
timeout -s KILL 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --kernel-trace --output-format csv -d gpurun_out/l2_a -o p -- $CMD > gpurun_out/l2_a.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/l2_b -o p -- $CMD > gpurun_out/l2_b.out 2>&1
ls gpurun_out/l2_a gpurun_out/l2_b
