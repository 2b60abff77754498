#!/bin/bash
# GPU box: SQ / cache counters of ONE kernel (regex $1) of the ploidy-4 replay iteration (separate --pmc passes)
KRE=${1:-k4_zexpect}
REPO=$PWD
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
	tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
	rocprofv3 --pmc $grp --kernel-include-regex "$KRE" --output-format csv -d $REPO/gpurun_out/pmc_one/$tag -o k -- python3 $REPO/tools/gpu_prof_poly.py 4000 2 > $REPO/gpurun_out/pmc_one_$tag.log 2>&1 || echo "group failed: $grp"
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_one/*/k_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.match(r"(?:void )?(k4?_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("   %-36s %.4g  (%d launches)" % (c, sum(acc[k][c]) / len(acc[k][c]), len(acc[k][c])))
PY
