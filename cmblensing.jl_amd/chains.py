"""Multi-GPU layer: independent posterior chains / batch slots partitioned across ranks, results gathered.

The reference runs chains under `pmap` over Distributed.jl workers, one worker per GPU (src/sampling.jl:266,292;
src/util_parallel.jl:73-102), moving whole state dicts through host memory.  Here: one process per GPU, chains
`{c : c mod world == rank}`, per-chain seeds `base + chain id`, and one all_gather (RCCL over xGMI on GPUs, gloo in
the CPU tests) of per-chain scalars / maps.  No collective sits on the data path.
"""
import numpy as np
import torch


def partition_chains(nchains, world, rank):
    """chain ids owned by `rank` (round-robin, like `assign_GPU_workers` handing each worker a unique GPU)"""
    return [c for c in range(nchains) if c % world == rank]


def chain_seed(base, chain):
    return int(base) + int(chain)


def gather_chain_values(local_ids, local_vals, nchains, dist=None, device="cpu"):
    """Gather per-chain rows (any trailing shape) from all ranks into chain order.  `dist` is torch.distributed
    (initialised) or None for a single process.  Ranks may own different numbers of chains."""
    vals = torch.as_tensor(np.asarray(local_vals), dtype=torch.float64, device=device)
    if vals.dim() == 1:
        vals = vals[:, None]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = torch.empty((nchains,) + tuple(vals.shape[1:]), dtype=torch.float64, device=device)
        out[torch.as_tensor(local_ids, dtype=torch.long, device=device)] = vals
        return out.cpu().numpy()
    world = dist.get_world_size()
    per = (nchains + world - 1) // world                      # pad so every rank contributes the same shape
    pad_v = torch.zeros((per,) + tuple(vals.shape[1:]), dtype=torch.float64, device=device)
    pad_i = torch.full((per,), -1, dtype=torch.long, device=device)
    pad_v[: len(local_ids)] = vals
    pad_i[: len(local_ids)] = torch.as_tensor(local_ids, dtype=torch.long, device=device)
    all_v = [torch.empty_like(pad_v) for _ in range(world)]
    all_i = [torch.empty_like(pad_i) for _ in range(world)]
    dist.all_gather(all_v, pad_v)
    dist.all_gather(all_i, pad_i)
    out = torch.empty((nchains,) + tuple(vals.shape[1:]), dtype=torch.float64, device=device)
    for v, i in zip(all_v, all_i):
        m = i >= 0
        out[i[m]] = v[m]
    return out.cpu().numpy()


def allreduce_sum(t, dist=None):
    """Sum a (complex or real) tensor over ranks -- the mean-field average of MAP_marg over simulations that live on different
    GPUs (`mean(pmap(...))`, src/maximization.jl:304-311).  RCCL all_reduce in place on the device for the nccl backend; the gloo
    backend (CPU tests) stages through host memory."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    cplx = t.is_complex()
    buf = (torch.view_as_real(t) if cplx else t).contiguous()
    if dist.get_backend() == "nccl":
        dist.all_reduce(buf)
    else:
        host = buf.cpu()
        dist.all_reduce(host)
        buf = host.to(t.device)
    return torch.view_as_complex(buf) if cplx else buf
