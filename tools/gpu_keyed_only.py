import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from instruct_amd import synth, capi
sched = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
geno, an, mi = synth.make_diploid(10000, 5000, 5)
h = capi.HipChain(geno, an, mi, 5, rng_sched=sched)
h.setseeds(13, 4, 1972)
h.chain_init(np.array([h.ran1() for _ in range(5)], dtype=np.float32))
h.run(iters)
print("done", h.totallkh())
