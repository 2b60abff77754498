#!/bin/bash
# round 3, first GPU call: the device update_P against the oracle, then a short bench
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_walk.py -x -q > gpurun_out/r3_walk.log 2>&1
echo "walk tests rc=$?" | tee -a gpurun_out/r3_walk.log
tail -5 gpurun_out/r3_walk.log
INSTRUCT_HOST_TIMING=1 python bench.py --steps 20 --warmup 2 --no-cpu --no-tetra --no-keyed > gpurun_out/r3_bench1.log 2>&1
echo "bench rc=$?"
tail -3 gpurun_out/r3_bench1.log | cut -c1-3000
