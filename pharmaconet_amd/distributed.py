"""Multi-GPU screening: shard the library by ligand, score shards independently, exchange top-k.

The reference parallelises with `multiprocessing.Pool(cpus).map` over ligand files and a final
Python sort (`screening.py:66-70`). Here each rank (one process per GPU) owns a contiguous range of
ligand indices; the only exchange is one all-gather of every rank's k best `(score, index)` pairs
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), after which every rank holds
the same global ranking. Per-ligand scores stay rank-local.
"""

from __future__ import annotations

import numpy as np

__all__ = ["shard_range", "merge_topk", "allgather_topk", "screen_sharded"]


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard `[first, first + count)` of rank `rank`: sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def merge_topk(scores: np.ndarray, indices: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """k best of candidate `(score, index)` pairs: descending score, ties by ascending index - the order
    the stable `result.sort(key=score, reverse=True)` of screening.py:70 gives a library in index order.
    Padding entries (index < 0) are dropped."""
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    indices = np.asarray(indices, dtype=np.int64).reshape(-1)
    keep = indices >= 0
    scores, indices = scores[keep], indices[keep]
    order = np.lexsort((indices, -scores.astype(np.float64)))[:k]
    return scores[order], indices[order]


def allgather_topk(local_scores, local_indices, k: int, group=None) -> tuple[np.ndarray, np.ndarray]:
    """All-gather every rank's top-k tensors (same length on every rank, padded with index -1) and merge.

    Tensors stay on the device they live on (HBM for nccl / RCCL, host for gloo)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = local_scores.numel()
    gathered_s = torch.empty(world * n, dtype=local_scores.dtype, device=local_scores.device)
    gathered_i = torch.empty(world * n, dtype=local_indices.dtype, device=local_indices.device)
    dist.all_gather_into_tensor(gathered_s, local_scores.contiguous(), group=group)
    dist.all_gather_into_tensor(gathered_i, local_indices.contiguous(), group=group)
    return merge_topk(gathered_s.cpu().numpy(), gathered_i.cpu().numpy(), k)


def screen_sharded(model, shard_library, k: int, shard_first: int, weights=None, group=None):
    """Score this rank's shard (a `DeviceLibrary` or packed library holding ligands
    `shard_first .. shard_first + len(shard)` of the global library) and return
    `(ScreeningResult of the shard, global top-k scores, global top-k indices)`; identical on every rank."""
    import torch.distributed as dist

    from .engine import screen

    result = screen(model, shard_library, weights=weights, topk=k, index_base=shard_first)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        top_s, top_i = allgather_topk(result.topk_scores, result.topk_indices, k, group=group)
    else:
        top_s, top_i = merge_topk(result.topk_scores.cpu().numpy(), result.topk_indices.cpu().numpy(), k)
    return result, top_s, top_i
