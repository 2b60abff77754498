"""Diagnostic: cycle stamps inside k_zq_blocks (ISG_STAMPS build): one unit over many blocks, all units of one block."""
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
rst = h.zq_resolve_stats()
print(rst)
buf = np.zeros((4096, 8), dtype=np.uint64)
h.lib.isg_diag_stamps(buf.ctypes.data_as(C.c_void_p))
nb = min(300, rst['blocks'] - 2)
s = buf[20:nb].astype(np.int64)
names = ["start->granules in", "->walk done", "->header loads", "->loci loop", "->noted draws", "->dirichlet", "->publish"]
d = np.diff(s, axis=1)
for n, col in zip(names, d.T):
    q = np.percentile(col, [5, 50, 95])
    print(f"{n:22s} mean {col.mean():9.0f} ticks   pct 5/50/95: " + " ".join("%7.0f" % x for x in q))
print("total", (s[:, 7] - s[:, 0]).mean(), "block to block", np.diff(s[:, 0]).mean())

u = buf[2048:2048 + rst['units']].astype(np.int64)   # block 50: all units, all stage stamps
dd = np.diff(u, axis=1)
tot = u[:, -1] - u[:, 0]
order = np.argsort(tot)
print("block 50, per unit (cycles): total pct 5/50/95/max", [int(x) for x in np.percentile(tot, [5, 50, 95, 100])])
for n, col in zip(["wait for granules", "walk", "header", "loop", "noted draws", "dirichlet", "publish"], dd.T):
    print("  %-18s median %7d  p95 %7d  max %7d   in the 5 slowest units: %s" % (n, np.median(col), np.percentile(col, 95), col.max(), [int(x) for x in col[order[-5:]]]))
work = u[:, 7] - u[:, 1]
print("  work (granules in -> publish): median %d p95 %d max %d" % (np.median(work), np.percentile(work, 95), work.max()))

b2 = np.zeros((2048, 8), dtype=np.uint64)
if hasattr(h.lib, "isg_diag_stamps2"):
    h.lib.isg_diag_stamps2(b2.ctypes.data_as(C.c_void_p))
    w = b2[20:nb, :5].astype(np.int64)
    g_in = buf[20:nb, 1].astype(np.int64)
    done = buf[20:nb, 2].astype(np.int64)
    cols = np.column_stack([g_in, w, done])
    dw = np.diff(cols, axis=1)
    for n, col in zip(["granules in -> table in LDS", "-> (d0, E) read", "-> rows in registers", "-> walk", "-> results in LDS", "-> barrier"], dw.T):
        print("  walk stage %-28s median %6d  p5 %6d p95 %6d" % (n, np.median(col), np.percentile(col, 5), np.percentile(col, 95)))
    if b2[20:nb, 5].any():
        warm = buf[20:nb, 4].astype(np.int64) - b2[20:nb, 5].astype(np.int64)
        cold = b2[20:nb, 5].astype(np.int64) - buf[20:nb, 3].astype(np.int64)
        print("  ISG_LOOP_TWICE: simple cold loop median %d, pipelined loop on warm caches median %d p5 %d p95 %d" % (np.median(cold), np.median(warm), np.percentile(warm, 5), np.percentile(warm, 95)))

# where the slow units of block 50 are: by XCD (blockIdx % 8) and by position in the XCD's share
lp = (u[:, 4] - u[:, 3])
print("loop cycles of block 50 by XCD (blockIdx % 8):", [int(np.median(lp[x::8])) for x in range(8)])
idx = np.arange(len(lp)) // 8
print("loop cycles by position in the XCD's share (eighths):", [int(np.median(lp[(idx * 8 // (idx.max() + 1)) == k])) for k in range(8)])
