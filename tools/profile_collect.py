"""Turns the rocprofv3 outputs of tools/profile_round.sh into profiles/<round>_*.{csv,json} and profiles/traffic.json.

HBM bytes per launch = FETCH_SIZE (KB, separate --pmc pass) x calibration + WRITE_SIZE (KB, separate pass):
MI355X_MICROARCH.md says gfx950's FETCH_SIZE reports half the bytes of wide streaming reads and asks to calibrate
narrower access patterns on a known byte count: k_count reads exactly geno + z = 2 N L P bytes with the 8-byte
per lane loads all the sweeps here use, which gives the factor applied to every kernel."""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"

SHORT = [("k_wk_table<0>", "k_wk_table_P"), ("k_wk_table<1>", "k_wk_table_Z"), ("k_zq_at", "k_zq_at"), ("k_zq_probe", "k_zq_probe"), ("k4_zq_probe", "k4_zq_probe"),
         ("k_zexpect", "k_zexpect"), ("k4_zexpect_fin", "k4_zexpect_fin"), ("k4_zexpect", "k4_zexpect"),
         ("k_zq_pipe", "k_zq_pipe"), ("k_zq_spec", "k_zq_spec"), ("k_zq_coop", "k_zq_coop"), ("k4_zq_coop", "k4_zq_coop"), ("k_zq<256", "k_zq_keyed"), ("k_zq<512", "k_zq_chain"),
         ("k_loglik<256, true>", "k_loglik_pair"), ("k_loglik<256, false>", "k_loglik_lkh"),
         ("k_loglik_tab<256, true>", "k_loglik_pair"), ("k_loglik_tab<256, false>", "k_loglik_lkh"),
         ("k_loglik_int<256, true>", "k_loglik_pair"), ("k_loglik_int<256, false>", "k_loglik_lkh"), ("k4_zq<256", "k4_zq_keyed")]


def short(name):
    for pat, s in SHORT:
        if pat in name:
            return s
    m = re.match(r"(?:void )?(k4?_\w+)", name)
    return m.group(1) if m else None


TOTALS = {}


def counter(dirname, cname):
    path = os.path.join(out, dirname, "k_counter_collection.csv")
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == cname:
                s = short(r["Kernel_Name"])
                if s:
                    acc[s].append(float(r["Counter_Value"]))
    # k_zq_at / k4_zq<at> are also launched behind every blind probe round and return at once while the trajectory still holds uncertain
    # bytes: those launches move nothing and are not sweeps
    for k in ("k_zq_at", "k4_zq_keyed"):
        if k in acc and cname in ("FETCH_SIZE", "WRITE_SIZE"):
            top = max(acc[k])
            acc[k] = [x for x in acc[k] if x > 0.02 * top] or acc[k]
    TOTALS[cname] = {k: (sum(v), len(v)) for k, v in acc.items()}
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
known = 2 * 10000 * 5000 * 2
factor = known / (fetch["k_count"] * 1024)
traffic = {k: int(fetch.get(k, 0) * 1024 * factor + write.get(k, 0) * 1024) for k in sorted(set(fetch) | set(write))}
# the replay schedule's update_ZQ is a phase of launches (k_tapef, the block resolution -- k_zq_blocks: one launch for all blocks, or
# k_zq_block: one per block --, k_zq_at): its bytes per SWEEP = all those launches' bytes / number of sweeps (= k_zq_at launches)
if "k_zq_at" in TOTALS["FETCH_SIZE"]:
    # round 3: the phase = expected counts, accept-bit tables, probes, the sweep (the walks' maps are a few hundred KB: left out, their kernels
    # also serve update_P); whatever sweeps fell through to the block resolver add k_tapef + k_zq_blocks / k_zq_block
    nsweep = TOTALS["FETCH_SIZE"]["k_zq_at"][1]
    tot = 0.0
    for k in ("k_zexpect", "k_wk_table_Z", "k_zq_probe", "k_zs_band", "k_zs_offs", "k_tapef", "k_zq_blocks", "k_zq_block", "k_zq_at"):
        tot += TOTALS["FETCH_SIZE"].get(k, (0, 0))[0] * 1024 * factor + TOTALS["WRITE_SIZE"].get(k, (0, 0))[0] * 1024
    traffic["update_ZQ_replay"] = int(tot / nsweep)
    for k in ("k_zq_blocks", "k_zq_block"):
        if k in TOTALS["FETCH_SIZE"]:
            traffic[k + "_launches_per_sweep"] = round(TOTALS["FETCH_SIZE"][k][1] / nsweep, 1)
detail = {"workload": "config 3 (N=10000 L=5000 K=5 diploid) and config 5 (N=10000 L=20000 K=10 ploidy 4), tools/gpu_prof_driver.py",
          "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE calibrated on k_count (reads exactly 200000000 bytes)",
          "raw_KB_per_launch": {k: {"FETCH_SIZE_KB": fetch.get(k), "WRITE_SIZE_KB": write.get(k)} for k in traffic},
          "fetch_calibration_factor": factor, "hbm_bytes_per_launch": traffic,
          "calibration_note": "the factor comes from k_count's access pattern (8-byte loads per lane of the diploid byte arrays); the k4_* kernels read 4-byte words and "
                              "float tables, so their figures are calibrated on a different pattern than their own: read them as estimates"}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", f"{rnd}_traffic_detail.json"), "w") as f:
    json.dump(detail, f, indent=1)
with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
stats = os.path.join(out, "prof_stats", "k_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{rnd}_bench_kernel_stats.csv"))
print(json.dumps(traffic, indent=1))
