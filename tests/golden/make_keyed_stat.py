#!/usr/bin/env python3
"""Regenerates tests/golden/keyed_stat_c2s.json (development container only; needs `make -C oracle ref`).

Posterior summaries of the REAL reference chain (oracle/ref_stat.c drives the reference's own six sweeps) on the c2s
data set (N=200 L=300 K=5, 1 % missing) for several seed triples: per run the posterior means of Q (N x K), of the
selfing rates, of alpha and of the log-likelihood over iterations BURN..ITERS.  The keyed RNG schedule draws the same
distributions from other stream positions; tests/test_keyed_stat.py checks that its posterior means agree with these
within Monte-Carlo error."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import make_golden  # noqa: E402
from instruct_amd import synth  # noqa: E402

CASE, ITERS, BURN = "c2s", 2000, 1000
SEEDS = [(13, 4, 1972), (14, 5, 1973), (15, 6, 1974), (16, 7, 1975), (17, 8, 1976), (18, 9, 1977)]


def main():
    N, L, K = make_golden.CASES[CASE][:3]
    geno, allelenum, minall = synth.code_diploid(make_golden.data_for(CASE))
    runs = []
    with tempfile.TemporaryDirectory() as d:
        gp = os.path.join(d, "geno.u8")
        np.where(geno < 0, 255, geno).astype(np.uint8).tofile(gp)  # 0xFF = missing (oracle/ref_stat.c)
        procs = []
        for r, s in enumerate(SEEDS):  # (a run takes the reference about five minutes: all at once)
            out = os.path.join(d, "out%d.txt" % r)
            procs.append((s, out, subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", "ref_stat"), gp, str(N), str(L), str(K), str(ITERS), str(BURN)] + [str(x) for x in s] + [out])))
        for s, out, p in procs:
            assert p.wait() == 0
            lines = open(out).read().splitlines()
            runs.append({"seeds": list(s), "alpha": float(lines[0].split()[1]), "totallkh": float(lines[1].split()[1]),
                         "self": [float(x) for x in lines[2].split()[1:]],
                         "q": [[round(float(x), 6) for x in l.split()] for l in lines[3:3 + N]]})
            print(s, runs[-1]["alpha"], runs[-1]["totallkh"], np.round(runs[-1]["self"], 3), flush=True)
    with open(os.path.join(HERE, "keyed_stat_%s.json" % CASE), "w") as f:
        json.dump({"case": CASE, "iters": ITERS, "burn": BURN, "N": N, "L": L, "K": K, "runs": runs}, f, separators=(",", ":"))
        f.write("\n")


if __name__ == "__main__":
    main()
