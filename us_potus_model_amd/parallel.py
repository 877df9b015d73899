"""Chains over GPUs: one process per GPU, no communication while sampling, one all-gather.

The reference runs each chain as an independent CmdStan process (`parallel_chains`,
scripts/model/final_2016.R:536); nothing is exchanged between chains.  Here rank r of a
`torch.distributed` job owns a contiguous block of chains; the RNG stream of a chain is keyed by
its GLOBAL id (chain_id_offset), so the draws do not depend on the number of GPUs.  The only
collective is the all-gather that pools the draws-of-interest for R-hat / ESS (RCCL over xGMI
when the backend is "nccl", gloo in the CPU tests).
"""
from __future__ import annotations

import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def chain_block(total_chains: int, rank: int, world: int):
    """Contiguous split; returns (chain_id_offset, n_local). Ranks beyond total_chains get 0."""
    base, rem = divmod(total_chains, world)
    n = base + (1 if rank < rem else 0)
    off = rank * base + min(rank, rem)
    return off, n


def init_process_group(backend: str | None = None):
    """Join the job's process group (no-op for a single process).  HSA_ENABLE_IPC_MODE_LEGACY=0 is set if unset: the host driver of the
    MI355X boxes only supports dmabuf IPC, and RCCL's intra-node transport (hipIpcGetMemHandle) fails without it.  ROCr reads the variable
    when the runtime initialises, so call this -- or set the variable -- before the process's first HIP call (bench.py does both)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def all_gather_draws(local: np.ndarray, total_chains: int, device=None) -> np.ndarray:
    """Pool [n_local_chains, n_draws, k] arrays from every rank into [total_chains, n_draws, k].

    Ranks may own different chain counts: shards are padded to the maximum before the
    (equal-size) all_gather and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist
    rank, world, _ = env_rank_world()
    if world == 1 or not dist.is_initialized():
        return local
    n_max = -(-total_chains // world)
    shape = (n_max,) + tuple(local.shape[1:])
    buf = torch.zeros(shape, dtype=torch.float64, device=device)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.as_tensor(local, dtype=torch.float64, device=device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    parts = []
    for r in range(world):
        _, n = chain_block(total_chains, r, world)
        parts.append(out[r][:n].cpu().numpy())
    return np.concatenate(parts, axis=0)


def all_gather_chains(local, device=None):
    """Device-side pooling of draws-of-interest: `local` is a torch tensor [n_draws, C, k] on this rank's GPU (the same
    C on every rank); returns [n_draws, world * C, k], chains in rank order, on every rank.  With a device the
    collective is dist.all_gather_into_tensor on device memory (RCCL over xGMI); device=None is the gloo development
    mode (CPU tensors)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return local
    world = dist.get_world_size()           # (a one-rank group still goes through the collective: tests/test_gpu_boundary.py)
    if device is None:
        x = local.cpu().contiguous()
        out = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(out, x)
        return torch.cat(out, dim=1)
    x = local.contiguous()
    out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x)
    return out.permute(1, 0, 2, 3).reshape(x.shape[0], world * x.shape[1], x.shape[2])


def column_block(n_columns: int, rank: int, world: int):
    """The columns whose convergence diagnostics rank `rank` computes: a contiguous 1/world share [a, b) of the pooled block's columns.
    Every rank holds the pooled chains after the all-gather; sorting every column on every rank would make the per-GPU work grow with
    the number of GPUs (world x the draws per column), so the columns are dealt out and only the results travel (all_gather_columns)."""
    off, n = chain_block(n_columns, rank, world)
    return off, off + n


def all_gather_columns(local, n_columns: int, device=None):
    """Per-column results (1-D float64 numpy array of this rank's column_block) from every rank -> [n_columns] on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(local, dtype=np.float64)
    world = dist.get_world_size()
    w_max = -(-n_columns // world)
    buf = torch.zeros(w_max, dtype=torch.float64, device=device)
    loc = np.asarray(local, dtype=np.float64)
    if loc.size:
        buf[: loc.size] = torch.as_tensor(loc, dtype=torch.float64, device=device)
    if device is None:
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        rows = [o.numpy() for o in out]
    else:
        out = torch.empty((world, w_max), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, buf)
        rows = list(out.cpu().numpy())
    parts = []
    for r in range(world):
        a, b = column_block(n_columns, r, world)
        parts.append(rows[r][: b - a])
    return np.concatenate(parts)


def barrier():
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


# ------------------------------------------------------------------------------------------------ pooled dense metric across ranks
# potus_opts.pooled_metric = 2: at a window end every rank holds (count, mean, M2 = sum of centred outer products) of ITS chains' window draws.  The pooled
# covariance of all chains of the job follows Chan et al.'s pairwise update,
#     N = sum_r n_r,   mean = sum_r n_r mean_r / N,   M2 = sum_r [ M2_r + n_r (mean_r - mean)(mean_r - mean)' ],
# i.e. two small all-reduces and ONE all-reduce of D x D doubles (13.9 GB at the configs[4] shape: RCCL over xGMI, in slices so that no rank needs a second
# copy) -- the only all-reduce of the whole path, and a declared deviation from Stan, whose chains never talk (SURVEY 8e "Anything else?").
# sampler.run_pooled drives it; the gloo world-size-2 test (tests/test_host.py) holds the algebra to numpy.
def pool_window_mean(count: float, count_times_mean, device=None):
    """(N, pooled mean) from this rank's draw count and count x mean (a 1-D float64 torch tensor); device=None: the collectives run on CPU tensors (gloo)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(count), count_times_mean / float(count)
    n = torch.tensor([float(count)], dtype=torch.float64, device=device)
    s = count_times_mean.to(device) if device is not None else count_times_mean.cpu()
    s = s.contiguous().clone()
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(n.item()), (s / n.item()).to(count_times_mean.device)


def pool_window_m2(m2, device=None, slice_bytes=1 << 30):
    """In place: m2 (this rank's M2, already shifted to the pooled mean; a 2-D float64 torch tensor) becomes the sum over the ranks, slice by slice."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return m2
    rows = max(1, slice_bytes // max(1, m2.shape[1] * 8))
    for r0 in range(0, m2.shape[0], rows):
        blk = m2[r0:r0 + rows]
        if device is None and blk.is_cuda:                 # development mode: the collective on CPU tensors
            t = blk.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            blk.copy_(t)
        else:
            dist.all_reduce(blk, op=dist.ReduceOp.SUM)     # (a row slice of a contiguous matrix is contiguous)
    return m2


def pool_window_moments(count: float, mean, m2, device=None):
    """The whole update on tensors the caller owns (what sampler.run_pooled does on the library's buffers): returns (N, pooled mean); m2 is updated in place to
    the pooled M2.  mean [D], m2 [D, >= D] (columns beyond D are padding)."""
    n_tot, gmean = pool_window_mean(count, float(count) * mean, device)
    d = mean - gmean
    m2[:, :mean.numel()].addr_(d, d, alpha=float(count))
    pool_window_m2(m2, device)
    return n_tot, gmean
