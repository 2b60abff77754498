"""GPU: wall time of each phase call of the replay iteration at config 3 (host side included; each phase followed by a
small getter that waits for the stream), to see where the iteration's time outside the kernels goes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from instruct_amd import capi, synth
N, L, K = 10000, 5000, 5
geno, an, mi = synth.make_diploid(N, L, K)
h = capi.HipChain(geno, an, mi, K)
h.setseeds(13, 4, 1972)
h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
h.run(3)
phases = ["update_P", "update_S_POP", "update_G", "update_ZQ", "update_alpha", "cal_lkh"]
acc = {p: 0.0 for p in phases}
it = 10
t00 = time.perf_counter()
for _ in range(it):
    for p in phases:
        t0 = time.perf_counter()
        getattr(h, p)()
        h.alpha()   # a tiny download: waits for the phase
        acc[p] += time.perf_counter() - t0
tot = time.perf_counter() - t00
for p in phases:
    print("%-14s %7.3f ms" % (p, acc[p] / it * 1e3))
print("sum %.3f ms per iteration (with the extra waits)" % (tot / it * 1e3))
t0 = time.perf_counter(); h.run(it); h.totallkh(); print("isg_run: %.3f ms per iteration" % ((time.perf_counter() - t0) / it * 1e3))
