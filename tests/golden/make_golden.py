#!/usr/bin/env python3
"""Regenerates tests/golden/*.golden and *.txt inputs (development container only).

Needs `make -C oracle ref` (builds oracle/_ref/{ref_dump,ref_unit,InStruct_ref} from the real
reference sources under /root/reference).  The golden trajectory files hold, after every sweep
of every iteration, FNV-64 hashes of z / allele counts / generation / freq / qq, the RNG seed
triple and hex-float scalars as produced by the reference's own update_P / update_S_POP /
update_G / update_ZQ / update_alpha / cal_lkh (see oracle/ref_dump.c, oracle/dump_fmt.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from instruct_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
CLI_REF = os.path.join(os.path.relpath(REF, HERE), "InStruct_ref")  # the result files echo the command line: keep it relative

# name: (N, L, K, n_alleles, missing, u, b, t, c, e, y, r, j, seeds, mode, pf, detail, commit_text)
CASES = {
    "c1":       (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 2, 1, 10, True),
    "c1_miss":  (50, 100, 3, 2, 0.05, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 2, 0, 10, True),
    "c1_a3":    (50, 100, 3, 3, 0.02, 200, 100, 10, 1, 1, 1, 5, 5, (14, 5, 1973), 2, 1, 10, True),
    "c1_e0":    (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 0, 1, 5, 5, (13, 4, 1972), 2, 0, 10, False),
    "c1_y0":    (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 0, 5, 5, (13, 4, 1972), 2, 0, 10, False),
    "c1_mode1": (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 1, 0, 10, False),
    # mode 4 (-v 4): population inbreeding coefficients (mcmc_POP_inbreedcoff, mcmc.c:242-295)
    "m4_c1": (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 4, 0, 10, False),
    "m4_c1_e0": (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 0, 1, 5, 5, (13, 4, 1972), 4, 0, 10, False),
    # mode 3 (-v 3): one selfing rate per individual, uniform prior (mcmc_INDV_selfing, mcmc.c:297-385)
    "m3_c1": (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 3, 0, 10, False),
    "m3_c1_miss": (50, 100, 3, 2, 0.05, 200, 100, 10, 2, 1, 0, 5, 5, (22, 6, 1981), 3, 0, 10, False),
    # mode 5 (-v 5): one inbreeding coefficient per individual, uniform prior (mcmc_INDV_inbreedcoff, mcmc.c:386-470)
    "m5_c1": (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 5, 0, 10, False),
    "m5_c1_miss": (50, 100, 3, 2, 0.05, 200, 100, 10, 2, 1, 1, 5, 5, (23, 7, 1982), 5, 0, 10, False),
    # mode 0 (-v 0): no admixture, whole individuals assigned (mcmc_POP_no_admixture, mcmc.c:90-132)
    "m0_c1": (50, 100, 3, 2, 0.00, 200, 100, 10, 1, 1, 1, 5, 5, (13, 4, 1972), 0, 0, 0, False),
    "m0_c1_miss": (50, 100, 3, 2, 0.05, 200, 100, 10, 2, 1, 1, 5, 5, (24, 8, 1983), 0, 0, 0, False),
    "m0_c1_a3": (50, 100, 3, 3, 0.02, 200, 100, 10, 1, 1, 1, 5, 5, (25, 9, 1984), 0, 0, 0, False),
    "m4_c1_miss": (50, 100, 3, 2, 0.05, 200, 100, 10, 2, 1, 1, 5, 5, (21, 5, 1980), 4, 0, 10, False),
    "c1_c2":    (50, 100, 3, 2, 0.00, 120, 60, 10, 2, 1, 1, 6, 5, (21, 7, 1999), 2, 0, 0, False),
    "c2s":      (200, 300, 5, 2, 0.01, 40, 20, 5, 1, 1, 1, 4, 4, (13, 4, 1972), 2, 0, 0, False),
    "c2s_a4":   (120, 150, 4, 4, 0.03, 40, 20, 5, 1, 1, 1, 4, 4, (15, 6, 1974), 2, 0, 0, False),
}


# ploidy 4 (autotetraploid): name: (N, L, K, n_alleles, missing, u, b, t, e, r, j, seeds)
POLY_CASES = {
    "t1":    (60, 40, 3, 4, 0.05, 60, 30, 5, 1, 4, 4, (13, 4, 1972)),
    "t2_e0": (30, 30, 2, 3, 0.03, 60, 30, 5, 0, 4, 4, (14, 5, 1973)),
    "t3_a2": (40, 50, 3, 2, 0.00, 60, 30, 5, 1, 4, 4, (15, 6, 1974)),
}


# allotetraploid (-ap 0; update_P_allo, calc_exfreq_allo, allo_genfreq, choose_*_allo): same tuple layout
ALLO_CASES = {
    "ta1":    (50, 36, 3, 4, 0.05, 60, 30, 5, 1, 4, 4, (13, 4, 1972)),
    "ta2_e0": (30, 30, 2, 3, 0.03, 60, 30, 5, 0, 4, 4, (14, 5, 1973)),
    "ta3_a2": (40, 50, 3, 2, 0.00, 60, 30, 5, 1, 4, 4, (15, 6, 1974)),
}


def allo_data_for(name):
    N, L, K, A, miss = ALLO_CASES[name][:5]
    return synth.raw_alleles(N, L, K, 4, A, miss, 20260301 + sorted(ALLO_CASES).index(name))


def poly_data_for(name):
    N, L, K, A, miss = POLY_CASES[name][:5]
    return synth.raw_alleles(N, L, K, 4, A, miss, 20260201 + sorted(POLY_CASES).index(name))


def data_for(name):
    N, L, K, A, miss = CASES[name][:5]
    base = {"c1_e0": "c1", "c1_y0": "c1", "c1_mode1": "c1", "c1_c2": "c1", "m4_c1": "c1", "m4_c1_e0": "c1", "m4_c1_miss": "c1_miss", "m3_c1": "c1", "m3_c1_miss": "c1_miss", "m5_c1": "c1", "m5_c1_miss": "c1_miss", "m0_c1": "c1", "m0_c1_miss": "c1_miss", "m0_c1_a3": "c1_a3"}.get(name, name)
    seed = 20260101 + sorted(CASES).index(base)
    return synth.raw_alleles(N, L, K, 2, A, miss, seed)


def main():
    subprocess.check_call([os.path.join(REF, "ref_unit"), os.path.join(HERE, "unit_random.golden")])
    for name, cfg in CASES.items():
        N, L, K, A, miss, u, b, t, c, e, y, r, j, seeds, mode, pf, detail, commit_text = cfg
        raw = data_for(name)
        txt = os.path.join(HERE, name + ".txt") if commit_text else os.path.join("/tmp", name + ".txt")
        synth.write_text_diploid(txt, raw)
        out = os.path.join(HERE, name + ".golden")
        args = [os.path.join(REF, "ref_dump"), txt, out] + [str(x) for x in
                (K, N, L, u, b, t, c, e, y, r, j, seeds[0], seeds[1], seeds[2], mode, pf, detail)]
        with open(os.devnull, "w") as devnull:
            subprocess.check_call(args, stdout=devnull)
        print(name, os.path.getsize(out), "bytes")
    for name, cfg in POLY_CASES.items():
        N, L, K, A, miss, u, b, t, e, r, j, seeds = cfg
        txt = os.path.join(HERE, name + ".txt")
        synth.write_text_polyploid(txt, poly_data_for(name))
        out = os.path.join(HERE, name + ".golden")
        args = [os.path.join(REF, "ref_dump_poly"), txt, out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)]
        with open(os.devnull, "w") as devnull:
            subprocess.check_call(args, stdout=devnull)
        print(name, os.path.getsize(out), "bytes")
    for name, cfg in ALLO_CASES.items():
        N, L, K, A, miss, u, b, t, e, r, j, seeds = cfg
        txt = os.path.join(HERE, name + ".txt")
        synth.write_text_polyploid(txt, allo_data_for(name))
        out = os.path.join(HERE, name + ".golden")
        args = [os.path.join(REF, "ref_dump_poly"), txt, out] + [str(x) for x in (K, N, L, u, b, t, e, r, j) + tuple(seeds)] + ["0"]
        with open(os.devnull, "w") as devnull:
            subprocess.check_call(args, stdout=devnull)
        print(name, os.path.getsize(out), "bytes")
    cmd = [CLI_REF, "-d", "ta1.txt", "-o", "ta1_cli_output.txt"] + ALLO_CLI
    if os.path.exists(os.path.join(HERE, "ta1_cli_output.txt")):
        os.unlink(os.path.join(HERE, "ta1_cli_output.txt"))
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    # end-to-end reference CLI output for the drop-in test (result file at %.3f)
    # (this one was committed with absolute paths on its command-line echo; kept so that regenerating reproduces the bytes)
    cmd = [os.path.join(REF, "InStruct_ref"), "-d", os.path.join(HERE, "c1.txt"), "-o", os.path.join(HERE, "c1_cli_output.txt"), "-K", "3", "-L", "100", "-N", "50", "-p", "2",
           "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "2", "-g", "1", "-r", "5", "-j", "5",
           "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "1"]
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    cmd = [CLI_REF, "-d", "c1.txt", "-o", "c1_kscan_output.txt"] + KSCAN_CLI
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    cmd = [CLI_REF, "-d", "c1.txt", "-o", "c1_mode4_cli_output.txt"] + MODE4_CLI
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    cmd = [CLI_REF, "-d", "c1.txt", "-o", "c1_mode3_cli_output.txt"] + MODE3_CLI
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    cmd = [CLI_REF, "-d", "c1.txt", "-o", "c1_mode5_cli_output.txt"] + MODE5_CLI
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    cmd = [CLI_REF, "-d", "c1.txt", "-o", "c1_mode0_cli_output.txt"] + MODE0_CLI
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    for r in range(2):
        outp = "mgpu_rank%d_cli_output.txt" % r
        if os.path.exists(os.path.join(HERE, outp)):
            os.unlink(os.path.join(HERE, outp))
        cmd = [CLI_REF, "-d", "c1.txt", "-o", outp, "-cf", "mgpu_rank%d_cf.txt" % r] + mgpu_rank_cli(r)
        with open(os.devnull, "w") as devnull:
            subprocess.check_call(cmd, stdout=devnull, cwd=HERE)
    synth.write_text_diploid(os.path.join(HERE, "ec1.txt"), ec1_data())
    for name, (data, cli) in STDOUT_CLI.items():
        if os.path.exists(os.path.join(HERE, name + "_cli_output.txt")):
            os.unlink(os.path.join(HERE, name + "_cli_output.txt"))
        cmd = [CLI_REF, "-d", data, "-o", name + "_cli_output.txt"] + cli
        with open(os.path.join(HERE, name + "_cli_stdout.txt"), "wb") as so:
            subprocess.check_call(cmd, stdout=so, cwd=HERE)
    # same for the ploidy 4 driver (mcmc_POP_tetra_selfing)
    cmd = [CLI_REF, "-d", "t1.txt", "-o", "t1_cli_output.txt"] + TETRA_CLI
    with open(os.devnull, "w") as devnull:
        subprocess.check_call(cmd, stdout=devnull, cwd=HERE)


# -pi 1 (print_info, mcmc.c:1267-1316: progress text every update/100 iterations) and the empty-cluster restart
# (check_empty_cluster mcmc.c:1944-1974; InStruct.c:185-190: the chain is discarded and re-run, the stream continuing):
# name: (data file, CLI).  Both the result file and the program's STDOUT of the pure reference binary are kept.
# ec1: N=8 L=300, two true clusters analysed with K=4: chain 2 trips the check at its 5th stored step and is re-run.
STDOUT_CLI = {
    "ec1": ("ec1.txt", ["-K", "4", "-L", "300", "-N", "8", "-p", "2", "-u", "400", "-b", "200", "-t", "10", "-c", "2", "-v", "2", "-g", "1",
                        "-r", "5", "-j", "5", "-lb", "0", "-a", "0", "-s", "14", "4", "1972", "-pi", "1"]),
    "pi_mode4": ("c1.txt", ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "200", "-b", "100", "-t", "10", "-c", "1", "-v", "4", "-g", "1",
                            "-r", "5", "-j", "5", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "1", "-e", "0"]),
    "pi_mode3": ("c1.txt", ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "100", "-b", "50", "-t", "10", "-c", "1", "-v", "3", "-f", "0",
                            "-g", "1", "-r", "5", "-j", "5", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "1"]),
    "pi_mode5": ("c1.txt", ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "100", "-b", "50", "-t", "10", "-c", "1", "-v", "5", "-f", "0",
                            "-g", "1", "-r", "5", "-j", "5", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "1"]),
    "pi_tetra": ("t1.txt", ["-K", "3", "-L", "40", "-N", "60", "-p", "4", "-ap", "1", "-af", "1", "-u", "100", "-b", "50", "-t", "5", "-c", "1",
                            "-v", "2", "-g", "1", "-r", "4", "-j", "4", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "1", "-e", "0"]),
}


# multi-GPU launcher (instruct_amd/host/instruct_mgpu.c): chain r of a sharded run = a separate `-c 1 -s s1+r s2+r s3+r` run.
# The pure reference binary's result file and -cf dump (log-likelihood samples at %f) for ranks 0 and 1:
MGPU_BASE = ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "120", "-b", "60", "-t", "10", "-v", "2", "-r", "6", "-j", "5",
             "-lb", "0", "-a", "0", "-pi", "0"]
MGPU_SEEDS = (21, 7, 1999)


def mgpu_rank_cli(r):
    return MGPU_BASE + ["-c", "1", "-g", "1", "-s"] + [str(x + r) for x in MGPU_SEEDS]


def ec1_data():
    return synth.raw_alleles(8, 300, 2, 2, 2, 0.0, 777)


# K scan (-ik 1 -kv 2 3: InStruct.c:536-601 runs all chains for every K and keeps the K with the smallest DIC)
KSCAN_CLI = ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "60", "-b", "30", "-t", "5", "-c", "2", "-v", "2", "-g", "1", "-r", "4", "-j", "4",
             "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0", "-ik", "1", "-kv", "2", "3"]

MODE4_CLI = ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "4", "-g", "1", "-r", "5", "-j", "5",
             "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "1", "-e", "0"]

MODE3_CLI = ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "3", "-f", "0", "-g", "1", "-r", "5",
             "-j", "5", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0"]

MODE5_CLI = ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "5", "-f", "0", "-g", "1", "-r", "5",
             "-j", "5", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0"]

MODE0_CLI = ["-K", "3", "-L", "100", "-N", "50", "-p", "2", "-u", "200", "-b", "100", "-t", "10", "-c", "2", "-v", "0", "-g", "1", "-r", "5",
             "-j", "5", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "1"]

ALLO_CLI = ["-K", "3", "-L", "36", "-N", "50", "-p", "4", "-ap", "0", "-af", "1", "-u", "60", "-b", "30", "-t", "5", "-c", "2",
            "-v", "2", "-g", "1", "-r", "4", "-j", "4", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0"]

TETRA_CLI = ["-K", "3", "-L", "40", "-N", "60", "-p", "4", "-ap", "1", "-af", "1", "-u", "60", "-b", "30", "-t", "5", "-c", "2",
             "-v", "2", "-g", "1", "-r", "4", "-j", "4", "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0"]

if __name__ == "__main__":
    main()
