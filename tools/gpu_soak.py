"""One-off soak (GPU box): a long replay chain against the canonical oracle, state compared every `every` iterations.
usage: python tools/gpu_soak.py [N L K iters every mode]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from instruct_amd import capi, synth
N, L, K, iters, every, mode = (int(x) for x in (sys.argv[1:7] + ["400", "900", "5", "2000", "100", "2"][len(sys.argv) - 1:]))
orc.build()
geno, an, mi = synth.code_diploid(synth.raw_alleles(N, L, K, 2, 3, 0.03, 41))
h = capi.HipChain(geno, an, mi, K, mode=mode)
o = orc.OrcChain(geno, an, mi, K, mode=mode, math=orc.MATH_ISG, accum=orc.ACC_EXACT, sched=0)
h.setseeds(13, 4, 1972); o.setseeds(13, 4, 1972)
initd = np.array([h.ran1() for _ in range(K)], dtype=np.float32); [o.ran1() for _ in range(K)]
h.chain_init(initd); o.chain_init(initd)
t0 = time.time()
for blk in range(iters // every):
    h.run(every)
    for _ in range(every):
        o.iteration()
    for name in ("z", "qq", "qqnum", "alpha", "self_rates", "freq", "indvlkh", "totallkh", "seeds") + (("generation",) if mode == 2 else ()):
        a, b = getattr(h, name)(), getattr(o, name)()
        ok = np.array_equal(a, np.asarray(b)) if isinstance(a, np.ndarray) else a == b
        assert ok, (blk, name)
    print("iteration %d identical; fallbacks %d; interval resolver %s; update_P %s; block resolver %s; %.0f s" %
          ((blk + 1) * every, h.zq_fallbacks(), h.zq_spec_stats(), h.p_device_stats(), h.zq_resolve_stats(), time.time() - t0), flush=True)
print("SOAK OK", iters, "iterations, fallbacks", h.zq_fallbacks())
