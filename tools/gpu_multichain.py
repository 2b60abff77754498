"""Scratch: C independent replay chains on ONE GPU, one host thread and one HIP stream each."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from instruct_amd import capi, synth, multichain

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
geno, an, mi = synth.make_diploid(10000, 5000, 5)
chains = []
for r in range(C):
    h = capi.HipChain(geno, an, mi, 5, rng_sched=capi.SCHED_REPLAY)
    h.setseeds(*multichain.rank_seeds((13, 4, 1972), r))
    h.chain_init(np.array([h.ran1() for _ in range(5)], dtype=np.float32))
    h.run(2)
    chains.append(h)
def work(h):
    h.run(steps)
    h.totallkh()
for n in (1, 2, 4, C):
    th = [threading.Thread(target=work, args=(h,)) for h in chains[:n]]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(n, "chains:", round(n * steps / dt, 2), "chain-iterations/s", round(dt / steps * 1e3, 2), "ms per step", flush=True)
