#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_mgpu_launcher.py tests/test_gpu_poly.py tests/test_gpu_walk.py -m gpu -x -q > gpurun_out/r3_lp.log 2>&1; echo "launcher+poly+walk rc=$?"; tail -4 gpurun_out/r3_lp.log
python tools/gpu_poly_time.py 10000 20000 10 4 4 > gpurun_out/r3_polytime.log 2>&1; tail -40 gpurun_out/r3_polytime.log
