#!/bin/bash
# Developer tool: wino4_fused_out timing ablations (side builds tools/tmp_libs/libapr_w4abl<N>.so, -DW4_ABL=N) at 4096 windows
export APR_DEV_KNOBS=1 APR_WINO_F4F_MIN_CIN=64 APR_WINO_F4F_MAX_CIN=512
for v in base $@; do
  lib=""; [ $v != base ] && lib=tools/tmp_libs/libapr_w4abl$v.so
  echo "== ablation $v"
  APR_AMD_LIB=$lib timeout -k 10 120 python3 tools/batch_breakdown.py 4096 2>&1 | grep -E "wino4_fused_out|wino_input_f4p" || exit 1
done
