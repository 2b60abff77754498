#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_walk.py -x -q > gpurun_out/r3_walk.log 2>&1; echo "walk rc=$?"; tail -3 gpurun_out/r3_walk.log
python -m pytest tests/test_gpu_poly.py -x -q > gpurun_out/r3_poly.log 2>&1; echo "poly rc=$?"; tail -3 gpurun_out/r3_poly.log
python tools/gpu_spec_diag.py 10000 5000 5 10 > gpurun_out/r3_diag.log 2>&1; tail -42 gpurun_out/r3_diag.log
