"""Scratch: iteration time of the diploid modes besides 1/2 at config-3 size (N=10000 L=5000 K=5), both schedules."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from instruct_amd import capi, synth
geno, an, mi = synth.make_diploid(10000, 5000, 5)
for mode in (0, 3, 4, 5):
    for sched in (0, 1):
        h = capi.HipChain(geno, an, mi, 5, mode=mode, rng_sched=sched)
        h.setseeds(13, 4, 1972)
        h.chain_init(np.array([h.ran1() for _ in range(5)], dtype=np.float32))
        h.run(2)
        h.profile(True)
        t = time.perf_counter(); h.run(10); lk = h.totallkh(); dt = time.perf_counter() - t
        print("mode", mode, "sched", sched, "%.2f ms/iter" % (dt * 100), {k: round(ms / n, 3) for k, (ms, n) in h.profile_results().items() if ms / n > 0.05})
        h.close()
