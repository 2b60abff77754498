#!/bin/bash
# GPU box: iteration time against the walk engine's segment length (update_P) at config 3 and config 5
for seg in 4096 8192 12544 25088; do
  echo "== config 3, INSTRUCT_WALK_SEG=$seg"
  INSTRUCT_WALK_SEG=$seg python tools/gpu_spec_diag.py 10000 5000 5 8 2>&1 | grep -E "^7 |k_wk_table_P|k_wk_walk_P" | cut -c1-160
done
for seg in 8192 16384 24576; do
  echo "== config 5, INSTRUCT_WALK_SEG=$seg"
  INSTRUCT_WALK_SEG=$seg python tools/gpu_poly_time.py 10000 20000 10 4 4 2>&1 | grep -E "^iter 3|k_wk_table_P|k_wk_walk_P" | cut -c1-160
done
