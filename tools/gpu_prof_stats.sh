#!/bin/bash
# rocprofv3 per-kernel statistics of a short replay-schedule bench run (config 3): gpurun_out/<tag>_kernel_stats.csv
set -o pipefail
TAG=${1:-r3}
OUT=$PWD/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-tetra --no-keyed --no-concurrent > $OUT/${TAG}_prof_bench.log 2>&1
echo "rc=$?"
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
head -40 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
