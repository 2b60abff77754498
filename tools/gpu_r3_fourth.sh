#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_walk.py tests/test_gpu_poly.py -x -q > gpurun_out/r3_wp.log 2>&1; echo "walk+poly rc=$?"; tail -3 gpurun_out/r3_wp.log
python bench.py --no-cpu --no-concurrent > gpurun_out/r3_bench2.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r3_bench2.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('c3 replay', d['value'], 'it/s', d['ms_per_step'], 'ms')
print({k: v for k, v in d['kernels_ms'].items()})
p = d['ploidy4']
print('c5 replay', p['value'], 'it/s', p['ms_per_step'], 'ms;  keyed', p['keyed']['value'], p['keyed']['ms_per_step'])
print({k: v for k, v in p['kernels_ms'].items()})
"
