#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_walk.py -x -q > gpurun_out/r3_walk.log 2>&1; echo "walk rc=$?"; tail -3 gpurun_out/r3_walk.log
python tools/gpu_spec_diag.py 10000 5000 10 6 > gpurun_out/r3_diag_k10.log 2>&1; grep -v "^  k_" gpurun_out/r3_diag_k10.log | tail -7 | cut -c1-330
python bench.py > gpurun_out/r3_bench_full.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r3_bench_full.log > gpurun_out/r3_bench_full.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3_bench_full.json'))
print('c3 replay', d['value'], 'it/s', d['ms_per_step'], 'ms; speedup', d['speedup_vs_cpu'], 'cpu', d['cpu_baseline']['s_per_iter'])
print('roofline', {k: v for k, v in d['roofline'].items() if k in ('kernel','achieved','frac','avg_launch_ms','ms_per_step','copy_peak_measured')}, d['roofline']['phase'], d['roofline'].get('sweep_kernel'))
print('per step', d['kernels_ms_per_step'])
print('keyed', d['keyed']['value'], d['keyed']['ms_per_step'])
print('concurrent', d['concurrent_chains'])
p = d['ploidy4']
print('c5 replay', p['value'], 'it/s', p['ms_per_step'], 'ms;  keyed', p['keyed']['value'], p['keyed']['ms_per_step'], 'speedup', p.get('speedup_vs_cpu'))
print('c5 per step', p['kernels_ms_per_step'])
PY
