#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_walk.py -x -q > gpurun_out/r3_walk.log 2>&1
echo "walk tests rc=$?"
tail -3 gpurun_out/r3_walk.log
bash tools/gpu_prof_stats.sh ${1:-r3b} | grep -v "k_zq\|rocclr\|loglik\|at::\|k_tape\b" | cut -c1-160
tail -1 gpurun_out/${1:-r3b}_prof_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms'])"
