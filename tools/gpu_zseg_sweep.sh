#!/bin/bash
# GPU box: iteration time against the interval resolver's segment length and window width (update_ZQ) at config 3
for seg in 2560 3400 5120 16384; do
  echo "== config 3, INSTRUCT_ZQ_SPEC_SEG=$seg"
  INSTRUCT_ZQ_SPEC_SEG=$seg python tools/gpu_spec_diag.py 10000 5000 5 8 2>&1 | grep -E "^7 |k_wk_table_Z|k_wk_walk_Z|k_zexpect|k_zq_at" | cut -c1-200
done
for k in 4.0 4.5; do
  echo "== config 3, INSTRUCT_WALK_K=$k"
  INSTRUCT_WALK_K=$k python tools/gpu_spec_diag.py 10000 5000 5 8 2>&1 | grep -E "^7 |k_wk_table_Z|k_wk_walk_Z|k_wk_table_P" | cut -c1-200
done
