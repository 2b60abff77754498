"""Profiling driver (run under rocprofv3 on the GPU box): a few iterations of every schedule / ploidy so that each
kernel of the hot path shows up with its launch shape at the benchmark sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from instruct_amd import capi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
geno, an, mi = synth.make_diploid(10000, 5000, 5)
for sched in (capi.SCHED_REPLAY, capi.SCHED_KEYED):
    h = capi.HipChain(geno, an, mi, 5, rng_sched=sched)
    h.setseeds(13, 4, 1972)
    h.chain_init(np.array([h.ran1() for _ in range(5)], dtype=np.float32))
    h.run(iters)
    print("diploid sched", sched, h.totallkh(), flush=True)
    h.close()
if len(sys.argv) <= 2 or sys.argv[2] != "no-tetra":  # both schedules
    K = 10
    obs, alleleid, allelenum = synth.make_tetraploid_fast(10000, 20000, K, 4, 0.05, 20260105)
    for sched in (capi.SCHED_REPLAY, capi.SCHED_KEYED):
        ch = capi.HipPolyChain(obs, alleleid, allelenum, K, rng_sched=sched)
        ch.setseeds(13, 4, 1972)
        ch.chain_init(np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32))
        ch.run(iters)
        print("tetra sched", sched, ch.totallkh(), flush=True)
        ch.close()
