#!/bin/bash
# the whole GPU suite in one process, then the default bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_all.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r3_gpu_all.log
python bench.py --no-cpu > gpurun_out/r3_bench_full.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r3_bench_full.log | cut -c1-6000
