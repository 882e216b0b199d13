"""Multi-GPU orchestration on CPU: contiguous shards, the all-gather of per-rank top-k (gloo, world_size 2)
and the merge that reproduces the stable descending sort of screening.py:70."""

import os
import socket

import numpy as np
import pytest

from pharmaconet_amd.distributed import merge_topk, shard_range


def test_shards_are_contiguous_and_cover():
    for n, w in [(10, 3), (7, 8), (1_000_000, 8), (0, 2)]:
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        for (f0, c0), (f1, _) in zip(parts, parts[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def reference_ranking(scores, k):
    """screening.py:70: result.sort(key=score, reverse=True) - stable, so ties keep library order."""
    order = sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)
    return order[:k]


def test_merge_topk_matches_stable_descending_sort():
    rng = np.random.default_rng(0)
    scores = rng.integers(0, 20, size=500).astype(np.float32)  # many ties
    k = 40
    parts = []
    for r in range(4):
        first, count = shard_range(len(scores), r, 4)
        local = reference_ranking(scores[first : first + count], k)
        parts.append((scores[first : first + count][local], np.array(local) + first))
    s = np.concatenate([p[0] for p in parts])
    i = np.concatenate([p[1] for p in parts])
    top_s, top_i = merge_topk(s, i, k)
    assert top_i.tolist() == reference_ranking(scores, k)
    np.testing.assert_array_equal(top_s, scores[top_i])
    # padding entries (index -1) are ignored
    top_s2, top_i2 = merge_topk(np.concatenate([s, [np.float32(-np.inf)] * 3]), np.concatenate([i, [-1] * 3]), k)
    assert top_i2.tolist() == top_i.tolist()


def _worker(rank, world, port, scores, k, out):
    import torch
    import torch.distributed as dist

    from pharmaconet_amd.distributed import allgather_topk

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = shard_range(len(scores), rank, world)
        local = scores[first : first + count]
        order = np.lexsort((np.arange(count), -local.astype(np.float64)))[:k]
        ls = np.full(k, -np.inf, np.float32)
        li = np.full(k, -1, np.int64)
        ls[: len(order)] = local[order]
        li[: len(order)] = order + first
        top_s, top_i = allgather_topk(torch.from_numpy(ls), torch.from_numpy(li), k)
        out.put((rank, top_s.tolist(), top_i.tolist()))
    finally:
        dist.destroy_process_group()


def test_allgather_topk_world_size_2_gloo():
    import torch.multiprocessing as mp

    rng = np.random.default_rng(1)
    scores = np.round(rng.gamma(2.0, 50.0, size=301)).astype(np.float32)
    scores[::7] = 0.0
    k = 25
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scores, k, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = reference_ranking(scores, k)
    for rank, top_s, top_i in results:
        assert top_i == want, f"rank {rank}"
        assert top_s == scores[want].tolist()
