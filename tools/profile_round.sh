#!/bin/bash
# GPU box: per-kernel time of the bench command and HBM traffic of every kernel (separate --pmc passes).
set -e
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_stats -o k -- python3 $REPO/bench.py --steps 20 --warmup 2 --no-cpu --no-concurrent > $REPO/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/pmc_fetch -o k -- python3 $REPO/tools/gpu_prof_driver.py 2 > $REPO/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/gpurun_out/pmc_write -o k -- python3 $REPO/tools/gpu_prof_driver.py 2 > $REPO/gpurun_out/pmc_write.log 2>&1
cd $REPO
tail -2 gpurun_out/prof_stats.log | cut -c1-400
