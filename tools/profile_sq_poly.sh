#!/bin/bash
# GPU box: SQ / cache counters of the ploidy-4 replay kernels (separate --pmc passes)
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum"; do
	tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
	rocprofv3 --pmc $grp --output-format csv -d $REPO/gpurun_out/pmc_sqp/$tag -o k -- python3 $REPO/tools/gpu_prof_poly.py 4000 2 > $REPO/gpurun_out/pmc_sqp_$tag.log 2>&1 || echo "group failed: $grp"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sqp/*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(k4?_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names))
for k in sorted(acc):
    print(k + "," + ",".join("%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "" for c in names))
PY
