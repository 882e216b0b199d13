"""The multi-rank path rehearsed on one GPU.

The reference's only parallelism is `pool.map` over ligand files and one sort (`screening.py:66-70`). Its counterpart here
- contiguous shards, per-rank top-k, one all-gather, one merge - runs on 8 GPUs under the driver only, so what can run on a
1-GPU box has to run there:

* the merge `pmx_topk_allgather` does after its `ncclAllGather` (pmx_topk over the gathered `(score, global index)` lists)
  fed with several ranks' lists directly - the kernels that rank on the 8-GPU node, without the collective;
* `bench.py --gpus 2` started the way the driver starts `--gpus 1`: no rendezvous in the environment, bench.py starts its two
  ranks itself (here both on device 0 over gloo: RCCL refuses two ranks on one device), and the line it prints says
  `n_gpus: 2`; the merged top-k equals a one-process stable sort of both shards' scores.
"""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _ranking(scores, index, k):
    """screening.py:70 over (score, global index) pairs: descending score, ties by ascending index, NaN after every real score."""
    key = np.where(np.isnan(scores), -np.inf, scores.astype(np.float64))
    order = np.lexsort((index, -key, np.isnan(scores).astype(np.int64)))[:k]
    return order


@pytest.mark.parametrize("world,k,n_per_rank", [(2, 500, 20_000), (8, 1000, 50_000), (3, 64, 40), (2, 64, 20)])
def test_device_merge_of_several_ranks_lists(world, k, n_per_rank):
    """What every rank does after the all-gather, with the gathered buffer built by hand: per-rank lists (each from pmx_topk over
    the rank's shard, global indices, padded with index -1 where a shard has fewer than k ligands) concatenated in rank order."""
    import torch

    from pharmaconet_amd.engine import topk

    rng = np.random.default_rng(77 + world)
    dev = torch.device("cuda", 0)
    shards, lists_s, lists_i = [], [], []
    for r in range(world):
        scores = rng.integers(0, 30, size=n_per_rank).astype(np.float32)  # many ties, within and across ranks
        scores[rng.integers(0, n_per_rank, size=3)] = np.nan                # unsupported ligands
        if r == world - 1:
            scores = scores[: max(1, n_per_rank // 3)]                      # a short last shard (fewer than k ligands when n_per_rank is small)
        shards.append(scores)
        ls, li = topk(torch.from_numpy(scores).to(dev), k, base_index=r * n_per_rank)
        lists_s.append(ls)
        lists_i.append(li)
    gathered_s, gathered_i = torch.cat(lists_s), torch.cat(lists_i)
    assert gathered_s.numel() == world * k
    out_s, out_i = topk(gathered_s, k, indices=gathered_i)  # == pmx_topk_allgather's merge (csrc/pmx_topk.hip)
    torch.cuda.synchronize()
    all_s = np.concatenate(shards)
    all_i = np.concatenate([r * n_per_rank + np.arange(len(s)) for r, s in enumerate(shards)])
    order = _ranking(all_s, all_i, k)
    n_real = min(k, len(all_s))
    got_i = out_i.cpu().numpy()[:n_real]
    got_s = out_s.cpu().numpy()[:n_real]
    assert got_i.tolist() == all_i[order].tolist()
    np.testing.assert_array_equal(got_s, all_s[order])
    if n_real < k:  # padding after the real entries
        assert np.all(out_i.cpu().numpy()[n_real:] == -1)


def _run_bench_ranks(tmp_path, world, ligands, k, env_extra):
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    cmd = [sys.executable, str(REPO / "bench.py"), "--gpus", str(world), "--ligands", str(ligands), "--steps", "1", "--warmup", "1", "--topk", str(k),
           "--no-cpu-baseline", "--no-serial-leg", "--dump-dir", str(tmp_path)]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert run.returncode == 0, run.stderr[-4000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world
    assert line["parity_sample"]["above_1e-5"] == 0 and line["parity_sample"]["zero_nonzero_mismatches"] == 0
    shards = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    n = len(shards[0]["scores"])
    for r, s in enumerate(shards):
        assert int(s["index_base"]) == r * n and len(s["scores"]) == n
    assert line["config"]["ligands_per_gpu"] == n
    assert abs(line["value"] - world * n * 8 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    all_s = np.concatenate([s["scores"] for s in shards])
    order = _ranking(all_s, np.arange(world * n), k)
    for s in shards:  # every rank holds the same merged ranking
        assert s["top_indices"].tolist() == order.tolist()
        np.testing.assert_array_equal(s["top_scores"], all_s[order])
    assert not np.array_equal(shards[0]["scores"], shards[1]["scores"])  # (the shards are different ligands)
    return line


@pytest.mark.parametrize("world,ligands", [(2, 20000), (8, 8192)])
def test_bench_starts_its_own_ranks(tmp_path, world, ligands):
    """`python bench.py --gpus N ...` with no WORLD_SIZE: N ranks (all on device 0 here, lists exchanged over gloo), one line, n_gpus N,
    merged top-k == one-process sort. World 8 is the shape of the driver's last scaling point (small buffers: eight workspaces on one device)."""
    extra = dict(PMX_BENCH_DEVICE="0", PMX_BENCH_BACKEND="gloo")
    if world > 2:
        extra.update(PMX_ARENA_MB="2048", PMX_BIG_TOTAL_MB="512", PMX_TASKQ_MB="128")
    line = _run_bench_ranks(tmp_path, world, ligands, 200, extra)
    assert line["config"]["exchange"]["ranks"] == world


def test_bench_two_ranks_over_rccl(tmp_path):
    """The real thing where the box has two GPUs: one rank per GPU, the per-rank top-k lists all-gathered by libpmx's own RCCL communicator
    (`pmx_topk_allgather`), the count of ranks taken from RCCL itself (`pmx_comm_info`). Skipped on a 1-GPU box (RCCL refuses two ranks on one device)."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU over RCCL)")
    line = _run_bench_ranks(tmp_path, 2, 20000, 200, {})
    ex = line["config"]["exchange"]
    assert ex["rccl_ranks"] == 2 == line["n_gpus"] and "ncclAllGather" in ex["collective"]
