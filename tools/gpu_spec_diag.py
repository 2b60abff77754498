#!/usr/bin/env python3
"""GPU box: per-iteration diagnostics of the interval resolver (replay update_ZQ) and the device update_P at a given size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instruct_amd import capi, synth

N, L, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (10000, 5000, 5)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 12
geno, an, mi = synth.make_diploid(N, L, K)
h = capi.HipChain(geno, an, mi, K, rng_sched=capi.SCHED_REPLAY)
h.setseeds(13, 4, 1972)
h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
for it in range(iters):
    t0 = time.perf_counter()
    h.iteration()
    h.totallkh()
    dt = time.perf_counter() - t0
    print(it, f"{dt*1e3:.2f} ms", "spec", h.zq_spec_stats(), "pdev", h.p_device_stats(), "fallbacks", h.zq_fallbacks(), flush=True)
h.profile(True)
h.run(10)
h.totallkh()
h.profile(False)
for k, (ms, n) in sorted(h.profile_results().items()):
    print(f"  {k:20s} {ms/n:8.4f} ms x {n/10:.1f} per iteration")
