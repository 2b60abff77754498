"""CPU: the N>1 path (one chain per rank, gather of the log-likelihood samples, Gelman-Rubin on rank 0)
exercised with world_size 2 over gloo."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from instruct_amd import multichain
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
assert multichain.rank_seeds((13, 4, 1972), rank) == (13 + rank, 4 + rank, 1972 + rank)
ckrep = 6
mine = np.arange(ckrep, dtype=np.float64) * (rank + 1) - 100.0 * rank
allv = multichain.gather_convg(mine)
assert allv.shape == (ws * ckrep,)
for r in range(ws):
    assert np.array_equal(allv[r * ckrep:(r + 1) * ckrep], np.arange(ckrep) * (r + 1) - 100.0 * r)
gr = multichain.gelman_rubin_all_ranks(mine)
tmax = multichain.max_over_ranks(1.0 + rank)
assert tmax == float(ws)
total = multichain.sum_over_ranks(3)
assert total == 3 * ws
dist.barrier()
if rank == 0:
    print("GR", repr(gr))
dist.destroy_process_group()
'''


def test_gather_and_gelman_rubin_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + ROOT)
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)], env=env,
                                  stderr=subprocess.STDOUT, timeout=300).decode()
    line = [l for l in out.splitlines() if l.startswith("GR")][0]
    gr = float(line.split()[1])
    # same number as the oracle's restatement of check_converg.c:100-153 on the gathered vector
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    vec = np.concatenate([np.arange(6) * (r + 1) - 100.0 * r for r in range(2)])
    assert gr == orc.gelman_rubin(vec, 2, 6)
