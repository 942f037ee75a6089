"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in
the CPU tests).

The hot path shards by independent streams -- rank r owns streams r*B .. r*B+B-1 and nothing is exchanged
(weak scaling).  The only collective is for the optional sharding of ONE RANSAC problem's hypotheses over
ranks: each rank scores a contiguous slice of the sample table, the int32 inlier
counts are all-gathered, and every rank computes the same global winner (max count, lowest global index).
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n items for `rank` (the first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def stream_ids(streams_per_rank: int, rank: int) -> range:
    return range(rank * streams_per_rank, (rank + 1) * streams_per_rank)


def allgather_counts(local_counts: np.ndarray, total: int, dist=None, device=None, comm=None) -> np.ndarray:
    """All-gather ragged int32 slices (shard_range order) into the full per-hypothesis count vector.

    comm: an api.Comm (mslam_comm_*: RCCL at the C ABI, one rank per GPU) -- the native path; otherwise a torch.distributed
    process group (gloo in the CPU tests)."""
    local_counts = np.ascontiguousarray(local_counts, np.int32)
    if comm is not None:
        world = comm.nranks
        width = -(-total // world)  # pad every slice to the same length: one fixed-size collective
        buf = np.full(width, -1, np.int32)
        buf[: local_counts.shape[0]] = local_counts
        out = comm.allgather_i32(buf).reshape(world, width)
        return np.concatenate([out[r, : shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)])
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local_counts.shape[0] == total
        return local_counts
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = -(-total // world)  # pad every slice to the same length: one fixed-size collective
    buf = torch.full((width,), -1, dtype=torch.int32, device=device)
    buf[: local_counts.shape[0]] = torch.from_numpy(local_counts).to(buf.device)
    out = torch.empty((world, width), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(out, buf) if hasattr(dist, "all_gather_into_tensor") and buf.is_cuda else \
        dist.all_gather(list(out.unbind(0)), buf)
    out = out.cpu().numpy()
    parts = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        parts.append(out[r, : hi - lo])
    return np.concatenate(parts)


def global_winner(counts: np.ndarray) -> tuple[int, int]:
    """max count, lowest index on ties (same rule as mslam_argmax_counts / ransac_select_k without confidence)."""
    best = int(np.argmax(counts))  # numpy returns the first maximum
    return best, int(counts[best])


def max_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sharded_pnp_ransac(ctx, X, x, K, dist_coeffs, samples, params, dist=None, device=None, rank=None, world=None, comm=None):
    """ONE PnP RANSAC problem with its hypotheses sharded over ranks (BASELINE config C4).

    Every rank scores its contiguous slice of the sample table on its own GPU (`ctx.pnp_ransac`); with a process group the
    int32 inlier counts are all-gathered (the only collective: <= 16 KiB).  `finish_sharded_pnp` then picks the same
    global winner on every rank (max count, lowest global index -- `params.confidence` must be 0: the adaptive bound is
    defined over the ordered full sequence) and refines it locally, so all ranks end with identical results and nothing
    else is exchanged.  Passing `rank` / `world` without a process group rehearses one rank's share in a single process.
    Returns (lo, hi, local_counts, all_counts or None).
    """
    assert params.confidence == 0.0, "sharded scoring uses the plain argmax rule"
    samples = np.ascontiguousarray(samples, np.int32)
    H = samples.shape[0]
    if comm is not None:   # native RCCL communicator: rank / size come from it
        lo, hi = shard_range(H, comm.rank, comm.nranks)
        local = ctx.pnp_ransac(X, x, K, dist_coeffs, samples[lo:hi], params)["counts"] if hi > lo else np.zeros(0, np.int32)
        return lo, hi, local, allgather_counts(local, H, comm=comm)
    grouped = dist is not None and dist.is_initialized() and dist.get_world_size() > 1 and rank is None
    if grouped:
        rank, world = dist.get_rank(), dist.get_world_size()
    elif rank is None:
        rank, world = 0, 1
    lo, hi = shard_range(H, rank, world)
    local = ctx.pnp_ransac(X, x, K, dist_coeffs, samples[lo:hi], params)["counts"] if hi > lo else np.zeros(0, np.int32)
    if grouped:
        return lo, hi, local, allgather_counts(local, H, dist, device)
    return lo, hi, local, (local if world == 1 else None)


def finish_sharded_pnp(ctx, X, x, K, dist_coeffs, samples, params, counts):
    """Global winner over the gathered counts + its refinement (identical on every rank)."""
    best, best_count = global_winner(np.asarray(counts))
    out = ctx.pnp_ransac(X, x, K, dist_coeffs, np.ascontiguousarray(samples[best:best + 1], np.int32), params)
    out.update(best_h=best if out["rc"] == 0 else -1, counts=np.asarray(counts, np.int32), winner_count=best_count)
    return out
