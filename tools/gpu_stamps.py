"""Diagnostic: per-phase cycle stamps of the replay ZQ chain kernel (ISG_STAMPS build)."""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
diag = os.path.join(ROOT, "gpurun_out", "libdiag.so")
extra = [a for a in sys.argv[3:] if a.startswith("-D")]
pre = os.path.join(ROOT, "tools", "_diag", "libdiag.so")   # built in the container with the same flags (travels with the snapshot)
pre = os.environ.get("ISG_DIAG_LIB", pre)
if os.path.exists(pre) and not extra:
    diag = pre
else:
  subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DISG_STAMPS"] + extra + [
                       "-o", diag, os.path.join(ROOT, "instruct_amd/csrc/isg_hip.hip")])
from instruct_amd import capi, synth
capi.LIB_PATH = diag
N, L, K = int(sys.argv[1]), int(sys.argv[2]), 5
tetra = "tetra" in sys.argv
if tetra:
    K = 10
    raw = synth.raw_alleles(min(N, 1000), L, K, 4, 4, 0.05, 20260105)
    obs, alleleid, allelenum = synth.code_tetraploid_fast(raw)
    if N > 1000:
        obs, alleleid = np.tile(obs, (N // 1000, 1, 1)), np.tile(alleleid, (N // 1000, 1))
    h = capi.HipPolyChain(obs, alleleid, allelenum, K)
    h.setseeds(13, 4, 1972)
    h.chain_init(np.array([np.float32(h.ran1()) for _ in range(K)], dtype=np.float32))
else:
    geno, an, mi = synth.make_diploid(N, L, K)
    h = capi.HipChain(geno, an, mi, K)
    h.setseeds(13, 4, 1972)
    h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
h.iteration(); h.iteration()
buf = np.zeros((4096, 8), dtype=np.uint64)
h.lib.isg_diag_stamps(buf.ctypes.data_as(C.c_void_p))
print("same-XCD hand-off: %d  (XCC %d, %d workgroups)" % (int(buf[4095, 4]) & 1, (int(buf[4095, 4]) >> 8) & 0xff, int(buf[4095, 4]) >> 16))
s = buf[100:min(N, 4000)].astype(np.int64)
if tetra:
    order = [0, 1, 2, 3, 4, 5]
    names = ["top->draws+counts", "->reduced+published", "->gathered", "->attempts", "->walk"]
    ss = s[:, order]
    d = np.diff(ss, axis=1)
elif os.environ.get("INSTRUCT_ZQ_COOP", "1") != "0":
    order = [0, 6, 7, 1, 2, 3, 4, 5]
    names = ["top->loads issued", "->buckets done", "->counts+store", "->published", "->gathered", "->attempts", "->walk"]
    if os.environ.get("INSTRUCT_ZQ_SPEC", "1") != "0":
        order = [0, 1, 2, 6, 3, 5]
        names = ["top->z picked+stored", "->published", "->candidates drawn", "->gathered", "->Dirichlet (per wave)"]
        if os.environ.get("INSTRUCT_ZQ_PIPE", "1") != "0":
            # draw waves: 0 top, 1 picked, 2 published, 6 candidates; control wave: 4 top, 3 gathered, 5 Dirichlet done
            order = [0, 1, 2, 6, 7]
            names = ["draw: top->z picked", "->stored", "->candidates drawn", "->their counts published"]
            dc = np.diff(s[:, [4, 3, 5]], axis=1)
            for n, col in zip(["ctrl: top->gathered", "->Dirichlet"], dc.T):
                q = np.percentile(col, [5, 25, 50, 75, 95, 99])
                print(f"{n:22s} {col.mean():9.0f} ticks   pct 5/25/50/75/95/99: " + " ".join("%6.0f" % x for x in q))
            print("ctrl top - draw top (same individual):", (s[:, 4] - s[:, 0]).mean(), " draw candidates done -> next top:", (s[1:, 0] - s[:-1, 6]).mean(),
                  " ctrl Dirichlet done -> next ctrl top:", (s[1:, 4] - s[:-1, 5]).mean())
    if "-DISG_EXP_XWAIT" in extra:
        order = [0, 2, 3, 6, 7, 1, 4, 5]
        names = ["top->x issued", "->x arrived", "->prefetch issued", "->buckets", "->counts+store", "->(publish+gather+attempts)", "->walk"]
    ss = s[:, order]
    d = np.diff(ss, axis=1)
else:
    names = ["start->draws_done", "->hist_sync1", "->hist_done", "->attempts_done", "->walk_done", "->end_sync"]
    d = np.diff(s[:, :7], axis=1)
for k, (n, v) in enumerate(zip(names, d.mean(0))):
    q = np.percentile(d[:, k], [5, 25, 50, 75, 95, 99])
    print(f"{n:22s} {v:9.0f} ticks   pct 5/25/50/75/95/99: " + " ".join("%6.0f" % x for x in q))
print("per individual (start i+1 - start i):", np.diff(s[:, 0]).mean(), "ticks")
