"""End to end through the reference's CLI at BASELINE config 3 size (N=10000 L=5000 K=5, -v 2): the reference driver around this
repository's streaming reader + MI355X sampler (oracle/_ref/InStruct_full), a short chain, wall time and peak resident memory
of the process (the reader exists because the reference's own reader needs ~100 bytes per token: ~10 GB here).
Optionally (argument "ref") the pure reference binary's READER is measured too, by letting it run until the chain starts."""
import os, resource, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np
from instruct_amd import synth

N, L, K = 10000, 5000, 5
txt = "/tmp/c3.txt"
t = time.time()
raw = synth.raw_alleles(N, L, K, 2, 2, 0.0, 20260101)
with open(txt, "w") as f:   # -af 0: two lines per individual, L tokens per line
    for i in range(N):
        for k in range(2):
            f.write(" ".join(map(str, raw[i, :, k].tolist())))
            f.write("\n")
print("text file %.1f MB written in %.1f s" % (os.path.getsize(txt) / 1e6, time.time() - t), flush=True)
args = ["-d", txt, "-K", str(K), "-L", str(L), "-N", str(N), "-p", "2", "-u", "20", "-b", "10", "-t", "2", "-c", "1", "-v", "2", "-g", "1", "-r", "4", "-j", "4",
        "-lb", "0", "-a", "0", "-s", "13", "4", "1972", "-pi", "0", "-pf", "0", "-mm", "1e13"]


def run(exe, timeout=None):
    out = "/tmp/c3_%s.out" % exe
    if os.path.exists(out):
        os.unlink(out)
    t0 = time.time()
    before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    try:
        p = subprocess.run([os.path.join("oracle", "_ref", exe), "-o", out] + args, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT, timeout=timeout)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        rc = "timeout"
    dt = time.time() - t0
    peak = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
    print("%-16s rc %s  wall %.1f s  peak RSS of the children so far %.2f GB (before this run %.2f GB)" % (exe, rc, dt, peak / 1e6, before / 1e6), flush=True)
    return out


out = run("InStruct_full")
txto = open(out, "rb").read()
print("result file %d bytes, chain blocks %d, finished: %s" % (len(txto), txto.count(b"Chain#"), b"Posterior Mean" in txto))
if len(sys.argv) > 1 and sys.argv[1] == "ref":
    run("InStruct_ref", timeout=600)   # its reader alone takes minutes and ~10 GB; the CPU chain behind it would take ~3 min per iteration
