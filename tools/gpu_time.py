"""Scratch: time iterations at config-2/3 shapes with per-kernel HIP-event profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from instruct_amd import synth, capi

def go(N, L, K, sched, iters):
    t0 = time.time()
    geno, an, mi = synth.make_diploid(N, L, K)
    t1 = time.time()
    h = capi.HipChain(geno, an, mi, K, rng_sched=sched)
    h.setseeds(13, 4, 1972)
    initd = np.array([h.ran1() for _ in range(K)], dtype=np.float32)
    t2 = time.time()
    h.chain_init(initd)
    t3 = time.time()
    h.iteration()
    t4 = time.time()
    h.run(iters)
    t5 = time.time()
    print(f"N={N} L={geno.shape[1]} K={K} sched={sched}: gen {t1-t0:.1f}s create {t2-t1:.2f}s init {t3-t2:.3f}s first {t4-t3:.3f}s  {iters} iters {(t5-t4)/iters*1e3:.2f} ms/iter  lkh={h.totallkh():.3f}", flush=True)
    h.profile(True)
    h.run(3)
    for k, (ms, n) in h.profile_results().items():
        print(f"   {k:16s} {ms/n:9.3f} ms x{n}")
    h.profile(False)
    # host phases
    for name in ("update_P", "update_S_POP", "update_G", "update_ZQ", "update_alpha", "cal_lkh"):
        t = time.time(); getattr(h, name)(); print(f"   host+dev {name:14s} {(time.time()-t)*1e3:8.2f} ms")
    h.close()

if __name__ == "__main__":
    go(2000, 1000, 5, 0, 5)
    go(2000, 1000, 5, 1, 10)
    go(10000, 5000, 5, 1, 5)
    go(10000, 5000, 5, 0, 3)
