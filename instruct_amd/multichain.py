"""Independent chains sharded one per GPU (reference: chains run back to back in one process,
InStruct.c:182-193).  The only cross-chain exchange is the per-chain vector of stored
log-likelihoods (CONVG.convg_ld, mcmc.c:223-224) that the Gelman-Rubin check consumes
(check_converg.c:44-91): one all-gather over RCCL/xGMI per run (`nccl` backend = RCCL on ROCm;
`gloo` in the CPU tests).  torch.distributed is plumbing only.
"""
from __future__ import annotations

import numpy as np


def rank_seeds(base, rank):
    """Seeds of the chain on rank r: (s1 + r, s2 + r, s3 + r)."""
    return (base[0] + rank, base[1] + rank, base[2] + rank)


def _dev(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_convg(mine: np.ndarray) -> np.ndarray:
    """all-gather of ckrep doubles per rank -> [world * ckrep] in rank order (chain r at r*ckrep)."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.ascontiguousarray(mine, dtype=np.float64)).to(_dev(dist))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out).cpu().numpy()


def gelman_rubin_all_ranks(mine: np.ndarray) -> float:
    """GR statistic of the gathered samples with the reference's formula (and its indexing quirk)."""
    import torch.distributed as dist
    from . import capi
    allv = gather_convg(mine)
    return capi.gelman_rubin(allv, dist.get_world_size(), len(mine))


def max_over_ranks(x: float) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float) -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
