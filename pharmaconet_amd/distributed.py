"""Multi-GPU screening: shard the library by ligand, score shards independently, exchange top-k.

The reference parallelises with `multiprocessing.Pool(cpus).map` over ligand files and a final
Python sort (`screening.py:66-70`). Here each rank (one process per GPU) owns a contiguous range of
ligand indices; the only exchange is one all-gather of every rank's k best `(score, index)` pairs
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests), after which every rank holds
the same global ranking. Per-ligand scores stay rank-local.
"""

from __future__ import annotations

import numpy as np

__all__ = ["shard_range", "merge_topk", "allgather_topk", "allgather_topk_device", "screen_sharded", "TopkExchange"]


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


def allgather_topk_device(local_scores, local_indices, k: int, group=None):
    """The exchange with the transport swapped: the ranks' lists travel over whatever backend the process group has (gloo on a
    box without one GPU per rank), the merge is the one `pmx_topk_allgather` runs behind its `ncclAllGather` - `pmx_topk` over the
    gathered `(score, global index)` lists on this rank's device. Returns device tensors, identical on every rank."""
    import torch
    import torch.distributed as dist

    from .engine import topk

    world = dist.get_world_size(group)
    dev = local_scores.device
    host = dist.get_backend(group) != "nccl"
    ls = local_scores.contiguous().cpu() if host else local_scores.contiguous()
    li = local_indices.contiguous().cpu() if host else local_indices.contiguous()
    gs = torch.empty(world * ls.numel(), dtype=ls.dtype, device=ls.device)
    gi = torch.empty(world * li.numel(), dtype=li.dtype, device=li.device)
    dist.all_gather_into_tensor(gs, ls, group=group)
    dist.all_gather_into_tensor(gi, li, group=group)
    return topk(gs.to(dev), int(k), indices=gi.to(dev))


class TopkExchange:
    """The top-k exchange through libpmx's C ABI (`pmx_comm_*`, `pmx_topk_allgather`): RCCL all-gather of the per-rank
    lists and the merge, both on the device. The communicator id is made on rank 0 and handed to the other ranks through
    the `torch.distributed` store (control plane only); the data never leaves HBM until the caller reads the result."""

    def __init__(self, device, group=None):
        import ctypes

        import torch
        import torch.distributed as dist

        from . import _ffi

        self._lib = _ffi.load()
        self.device = torch.device(device)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ident = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
        if self.rank == 0:
            _ffi.check(self._lib.pmx_comm_unique_id(ident))
        payload = [ident.raw]
        if self.world > 1:  # (src is a global rank: the group's rank 0 need not be the world's)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(payload, src=src, group=group)
        self._handle = ctypes.c_void_p()
        _ffi.check(self._lib.pmx_comm_create(payload[0], self.rank, self.world, self.device.index or 0, ctypes.byref(self._handle)))
        # what RCCL itself reports: a communicator that does not span the job's ranks must not pass for one that does
        r, n, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        _ffi.check(self._lib.pmx_comm_info(self._handle, ctypes.byref(r), ctypes.byref(n), ctypes.byref(d)))
        self.rccl_rank, self.rccl_ranks, self.rccl_device = r.value, n.value, d.value
        if self.rccl_ranks != self.world or self.rccl_rank != self.rank:
            raise _ffi.PmxError(f"RCCL communicator has {self.rccl_ranks} rank(s), this one is {self.rccl_rank}: expected rank {self.rank} of {self.world}")

    def allgather(self, local_scores, local_indices, k: int):
        """`local_scores` float32 [k], `local_indices` int64 [k] (global indices, -1 padding) on this rank's GPU ->
        the global top-k as device tensors, identical on every rank."""
        import ctypes

        import torch

        from . import _ffi

        out_s = torch.empty(k, dtype=torch.float32, device=self.device)
        out_i = torch.empty(k, dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _ffi.check(self._lib.pmx_topk_allgather(self._handle, local_scores.contiguous().data_ptr(), local_indices.contiguous().data_ptr(), int(k),
                                                out_s.data_ptr(), out_i.data_ptr(), ctypes.c_void_p(stream)))
        return out_s, out_i

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.pmx_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
