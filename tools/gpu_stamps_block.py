"""Diagnostic: cycle stamps inside k_zq_block (ISG_STAMPS build) for one unit, per launch."""
import os, sys, subprocess, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pre = os.environ.get("ISG_DIAG_LIB", os.path.join(ROOT, "tools", "_diag", "libdiag.so"))
from instruct_amd import capi, synth
capi.LIB_PATH = pre
N, L, K = 10000, 5000, 5
geno, an, mi = synth.make_diploid(N, L, K)
h = capi.HipChain(geno, an, mi, K)
h.setseeds(13, 4, 1972)
h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
h.iteration(); h.iteration()
print(h.zq_resolve_stats())
buf = np.zeros((4096, 8), dtype=np.uint64)
h.lib.isg_diag_stamps(buf.ctypes.data_as(C.c_void_p))
s = buf[20:300].astype(np.int64)
names = ["start->tables", "->walk done", "->header loads", "->loci loop", "->reduce barrier", "->dirichlet", "->publish"]
d = np.diff(s, axis=1)
for n, col in zip(names, d.T):
    q = np.percentile(col, [5, 50, 95])
    print(f"{n:22s} mean {col.mean():9.0f} ticks   pct 5/50/95: " + " ".join("%7.0f" % x for x in q))
print("total", (s[:, 7] - s[:, 0]).mean(), "launch to launch", np.diff(s[:, 0]).mean())

u = buf[2048:2048 + 247].astype(np.int64)   # launch 50: all units, all stage stamps (stamp 1 unused)
st = u[:, [0, 2, 3, 4, 5, 6, 7]]
has = u[:, 1] > u[:, 4]
print("units with a noted draw handled by thread 0: %d of %d; loop end -> rows loaded: median %d ; rows loaded -> barrier after: median %d" % (
    has.sum(), len(u), np.median((u[:, 1] - u[:, 4])[has]) if has.any() else -1, np.median((u[:, 5] - u[:, 1])[has]) if has.any() else -1))
dd = np.diff(st, axis=1)
tot = st[:, -1] - st[:, 0]
order = np.argsort(tot)
print("launch 50, per unit (cycles): total pct 5/50/95/max", [int(x) for x in np.percentile(tot, [5, 50, 95, 100])])
for n, col in zip(["start+walk", "header", "loop", "events+barriers", "dirichlet", "publish"], dd.T):
    print("  %-16s median %7d  p95 %7d  max %7d   in the 5 slowest units: %s" % (n, np.median(col), np.percentile(col, 95), col.max(), [int(x) for x in col[order[-5:]]]))
