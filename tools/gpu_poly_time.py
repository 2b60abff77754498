"""Scratch: wall time and per-kernel time of the ploidy-4 iteration at a given size (GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from instruct_amd import capi, synth

N, L, K, A = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
sched = int(sys.argv[6]) if len(sys.argv) > 6 else 0
t = time.time()
obs, alleleid, allelenum = synth.make_tetraploid_fast(N, L, K, A, 0.05, 20260105)   # N distinct individuals (C generator)
print("data %.1fs" % (time.time() - t), obs.shape, flush=True)
t = time.time()
ch = capi.HipPolyChain(obs, alleleid, allelenum, K, rng_sched=sched)
print("ctx %.1fs" % (time.time() - t), flush=True)
ch.setseeds(13, 4, 1972)
initd = np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32)
ch.profile(True)
t = time.time()
ch.chain_init(initd)
print("init %.2fs" % (time.time() - t), flush=True)
ch.profile(True)
for it in range(iters):
    t = time.time()
    ch.iteration()
    lk = ch.totallkh()
    print("iter %d %.3fs totallkh %.6e S %s" % (it, time.time() - t, lk, np.round(ch.self_rates(), 3)), flush=True)
for k, (ms, n) in sorted(ch.profile_results().items(), key=lambda kv: -kv[1][0]):
    print("%-16s %10.3f ms total %6d launches %10.3f ms each" % (k, ms, n, ms / n))
print("resolve", ch.zq_resolve_stats(), "fallbacks", ch.zq_fallbacks())
print("interval resolver", ch.zq_spec_stats(), "update_P device", ch.p_device_stats())
