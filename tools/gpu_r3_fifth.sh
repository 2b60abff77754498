#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gpu_all.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r3_gpu_all.log
python bench.py --no-cpu --no-concurrent --no-keyed > gpurun_out/r3_bench3.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/r3_bench3.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('c3 replay', d['value'], 'it/s', d['ms_per_step'], 'ms')
print({k: v for k, v in d['kernels_ms'].items()})
p = d['ploidy4']
print('c5 replay', p['value'], 'it/s', p['ms_per_step'], 'ms')
print({k: v for k, v in p['kernels_ms'].items()})
"
