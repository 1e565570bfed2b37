"""Sharding of a score batch over the GPUs of one node (one process per GPU, torch.distributed with the
"nccl" backend = RCCL over xGMI; "gloo" on CPU for tests).

Independent units = candidates: every rank holds the whole table, scores the candidates
i with i % world == rank on its own GPU, and one all_gather of <= n^2 doubles per batch gives every rank
the full result in fixed rank order, so that all ranks take the same deterministic find_max decision.
No other collective is on the data path.
"""
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return None
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def shard_indices(n, rank, world):
    return list(range(rank, n, world))


def sharded_batch(score, model, var, ntype, off, par, kind, min_per_rank=1):
    """Score a batch; with torch.distributed initialised the candidates are sharded over the ranks."""
    dist = _dist()
    n = len(var)
    if dist is None or n < 2 * min_per_rank:
        return score._batch_raw(model, var, ntype, off, par, kind)
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_indices(n, rank, world)
    v = [var[i] for i in mine]
    t = [ntype[i] for i in mine]
    o, p = [0], []
    for i in mine:
        p.extend(par[off[i]: off[i + 1]])
        o.append(len(p))
    local = score._batch_raw(model, v, t, o, p, kind)
    per = (n + world - 1) // world
    buf = np.zeros(per)
    buf[: len(mine)] = local
    backend = dist.get_backend()
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty(world * per, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send)
    allv = recv.cpu().numpy().reshape(world, per)
    out = np.zeros(n)
    for r in range(world):
        idx = shard_indices(n, r, world)
        out[idx] = allv[r, : len(idx)]
    return out
