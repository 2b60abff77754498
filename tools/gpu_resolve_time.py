"""GPU: replay update_ZQ at config 3 -- resolver statistics and per-kernel times (tuning aid).
usage: python tools/gpu_resolve_time.py [N L K iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from instruct_amd import capi, synth
if os.environ.get('ISG_LIB'): capi.LIB_PATH = os.environ['ISG_LIB']

N, L, K, iters = (int(x) for x in (sys.argv[1:5] + ["10000", "5000", "5", "10"][len(sys.argv) - 1:]))
geno, an, mi = synth.make_diploid(N, L, K)
h = capi.HipChain(geno, an, mi, K)
h.setseeds(13, 4, 1972)
h.chain_init(np.array([h.ran1() for _ in range(K)], dtype=np.float32))
h.run(2)
print("stats", h.zq_resolve_stats(), "fallbacks", h.zq_fallbacks(), flush=True)
h.profile_reset(); h.profile(True)
t0 = time.perf_counter(); h.run(iters); lk = h.totallkh(); dt = time.perf_counter() - t0
h.profile(False)
print("ms/iter %.3f" % (dt / iters * 1e3), "totallkh", lk)
for k, (ms, n) in sorted(h.profile_results().items()):
    print("  %-16s %8.4f ms x %d" % (k, ms / n, n))
print("stats", h.zq_resolve_stats(), "fallbacks", h.zq_fallbacks())
print("seeds", h.seeds())
