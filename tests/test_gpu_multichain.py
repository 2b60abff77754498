"""GPU: the RCCL leg of the multi-chain path on ONE device (world size 1): librccl loads, a communicator is created,
the all-gather of the log-likelihood samples (CONVG.convg_ld, mcmc.c:223-224) runs on device buffers and the
Gelman-Rubin value computed from the gathered vector equals isg_gelman_rubin on the same samples
(check_converg.c:100-153).  The 8-GPU run itself is the driver's; this proves the device path before it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from instruct_amd import capi, multichain
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
ckrep = 20
rng = np.random.default_rng(5)
mine = -3700.0 + rng.standard_normal(ckrep)
allv = multichain.gather_convg(mine)            # device tensors through RCCL
assert allv.dtype == np.float64 and np.array_equal(allv, mine)
gr = multichain.gelman_rubin_all_ranks(mine)
want = capi.gelman_rubin(mine, 1, ckrep)
assert gr == want or (gr != gr and want != want), (gr, want)
# the same exchange carrying two chains' samples (what rank 0 evaluates after the gather on 2 GPUs)
two = np.concatenate([mine, mine[::-1] * 1.0001])
assert multichain.gather_convg(two).tobytes() == two.tobytes()
assert multichain.max_over_ranks(1.25) == 1.25 and multichain.sum_over_ranks(3.0) == 3.0
dist.barrier()
dist.destroy_process_group()
print("RCCL_WS1_OK", repr(capi.gelman_rubin(two, 2, ckrep)))
'''


def test_rccl_gather_and_gelman_rubin_world_size_1(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "RCCL_WS1_OK" in text, text[-3000:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import orc
    rng = np.random.default_rng(5)
    mine = -3700.0 + rng.standard_normal(20)
    two = np.concatenate([mine, mine[::-1] * 1.0001])
    gr = float(text.split("RCCL_WS1_OK")[1].split()[0])
    assert gr == orc.gelman_rubin(two, 2, 20)
