"""Profiling driver (under rocprofv3): a few replay iterations of the ploidy 4 chain at config 5's L and K (N given)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from instruct_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = 10
raw = synth.raw_alleles(1000, 20000, K, 4, 4, 0.05, 20260105)
obs, alleleid, allelenum = synth.code_tetraploid_fast(raw)
obs, alleleid = np.tile(obs, (N // 1000, 1, 1)), np.tile(alleleid, (N // 1000, 1))
ch = capi.HipPolyChain(obs, alleleid, allelenum, K)
ch.setseeds(13, 4, 1972)
ch.chain_init(np.array([np.float32(ch.ran1()) for _ in range(K)], dtype=np.float32))
ch.run(iters)
print("tetra", ch.totallkh(), ch.zq_spec_stats(), ch.p_device_stats(), flush=True)
ch.close()
