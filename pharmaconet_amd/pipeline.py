"""`screening.py:63-70` for libraries that arrive in pieces: typed features -> packed records -> scores -> top-k as ONE device pipeline.

The reference maps `scoring_file` over the files of a library and sorts once (`screening.py:66-70`). Here a library comes as a sequence of
feature batches (`library.flatten_features` / `ligand.perceive_batch` layout - what perception hands to `LigandGraph`), and per batch

    host arrays --copy stream--> HBM --packer stream--> records (pmx_pack_features_device, adopted in place) --scoring stream--> pmx_score + pmx_topk

with batch i + 1 copied and packed while batch i is scored; the batches' k best are merged at the end with the ranking rule of
`screening.py:70` (`pmx_topk` over the candidates' global indices). No host core does more than enqueue - the host packer
(`pmx_pack_features`, 2.7 CPU-seconds per 10^6 molecules) is only taken for a batch that holds a molecule beyond the device builder's
fixed scratch. [MI355X] 67 x 10^6 ligand-conformers/s from pinned host arrays (bench.py: `end_to_end.device_packed_*`, which calls this).

Two orderings matter on this hardware and are kept here: the HOST waits for a batch's copy and enqueues the next
one only then (queued event markers hold back kernels of other streams that share a hardware queue), and nothing in the loop frees
device memory (`hipFree` waits for every stream)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable

import numpy as np

from . import engine

__all__ = ["PipelineResult", "screen_feature_batches", "pin_features"]

_STREAMS: dict = {}  # device index -> (copy streams, packer stream, scoring stream)


@dataclass
class PipelineResult:
    topk_scores: "object"   # torch.float32 [k] on the device, best first
    topk_indices: "object"  # torch.int64 [k]: position in the concatenation of all batches
    num_ligands: int = 0
    num_conformers: int = 0
    num_unsupported: int = 0       # molecules the packers turned into header-only records (scored NaN, ranked last)
    batches_packed_on_host: int = 0
    library_bytes: int = 0         # packed records made (all of them resident until the call returns)
    batch_sizes: list = field(default_factory=list)
    scores: "object | None" = None  # with keep_scores: torch.float32 [num_ligands] on the device, in the order the molecules came (NaN: not scored)
    status: "object | None" = None  # with keep_scores: torch.int32 [num_ligands], PMX_LIGAND_* per molecule

    def ranking(self) -> list[tuple[int, float]]:
        idx, sc = self.topk_indices.cpu().numpy(), self.topk_scores.cpu().numpy()
        return [(int(i), float(s)) for i, s in zip(idx, sc) if i >= 0]


def pin_features(flat) -> dict:
    """The arrays of a feature batch as pinned host tensors (uint64 offsets travel as int64 bits): what `screen_feature_batches` copies from
    without staging. Pinning costs a copy; a producer that fills pinned tensors in the first place skips it."""
    import torch

    out = {}
    for k in engine.FEATURE_FIELDS:
        a = flat[k]
        if isinstance(a, torch.Tensor):
            out[k] = a if a.is_pinned() else a.pin_memory()
            continue
        a = np.ascontiguousarray(a)
        out[k] = torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).pin_memory()
    return out


def _host_numpy(flat) -> dict:
    import torch

    out = {}
    for k in engine.FEATURE_FIELDS:
        a = flat[k]
        if isinstance(a, torch.Tensor):
            a = a.cpu().numpy()
        a = np.ascontiguousarray(a)
        out[k] = a.view(np.uint64) if k.endswith("_off") and a.dtype == np.int64 else a
    return out


def screen_feature_batches(model, batches: Iterable[dict], topk: int, weights: dict[str, float] | None = None, device=None, host_threads: int = 16,
                           keep_scores: bool = False) -> PipelineResult:
    """Score every molecule of `batches` (an iterable of feature batches: dicts of NumPy arrays or - faster - of pinned host tensors, see
    `pin_features`) against `model` and return the `topk` best of all of them. Batch sizes are the caller's: a small first batch starts the GPU
    early, large later ones amortise the scoring call's fixed tail (bench.py uses 1 : 3 : 4). The packed records of all batches stay in device memory until the
    call returns (a 10^6-ligand library of 8 conformers: 1.7 GB, twice that reserved); a library beyond the device's memory is screened in several calls.
    `keep_scores`: every molecule's score and status come back as well (what a CSV of the whole library, `screening.py:70-75`, is written from)."""
    import torch

    tdev = torch.device("cuda", engine._device_index(device))
    with torch.cuda.device(tdev):
        return _run(model, iter(batches), int(topk), weights, tdev, int(host_threads), bool(keep_scores))


def _run(model, it, topk, weights, tdev, host_threads, keep_scores):
    import torch

    from .library import pack_features_native

    # The pipeline's streams are made once per device: libpmx keeps its work buffers (tens of GB) per scoring stream, and torch hands out another
    # stream of its pool at every request - a fresh scoring stream per call would make a fresh workspace per call.
    if tdev.index not in _STREAMS:
        _STREAMS[tdev.index] = ([torch.cuda.Stream(tdev), torch.cuda.Stream(tdev)], torch.cuda.Stream(tdev), torch.cuda.Stream(tdev))
    copy_streams, pack_stream, score_stream = _STREAMS[tdev.index]
    main = torch.cuda.current_stream(tdev)
    for st in (*copy_streams, pack_stream, score_stream):
        st.wait_stream(main)  # (what the caller enqueued - a model upload, say - comes first)

    def fetch():
        try:
            flat = next(it)
        except StopIteration:
            return None
        if not all(isinstance(flat[k], torch.Tensor) for k in engine.FEATURE_FIELDS):
            flat = pin_features(flat)
        return flat

    def enqueue_copy(flat, slot):
        with torch.cuda.stream(copy_streams[slot]):
            return {k: flat[k].to(tdev, non_blocking=True) for k in engine.FEATURE_FIELDS}

    res = PipelineResult(None, None)
    tops, in_flight = [], []  # in_flight: the batches' libraries and score tensors, alive until the scoring stream is through
    cur = fetch()
    cur_dev = enqueue_copy(cur, 0) if cur is not None else None
    cur_bound = engine.pack_bound(cur) if cur is not None else 0  # (host arithmetic over the batch's offsets: done while the copy runs)
    i = 0
    while cur is not None:
        copy_streams[i % 2].synchronize()  # the host does the waiting (see the module's header)
        nxt = fetch()
        nxt_dev = enqueue_copy(nxt, (i + 1) % 2) if nxt is not None else None
        n = int(cur["atom_off"].numel()) - 1
        with torch.cuda.stream(pack_stream):
            offsets, data, status = engine.pack_features_device(cur_dev, tdev, bound=cur_bound)
            bad = status != 0
            n_bad = int(bad.sum())  # (waits for the packer stream; the sizes' total has been read by then anyway)
            if n_bad and bool((status == 3).any()):  # beyond the device builder's scratch: the host packer takes the batch
                host_lib, host_status = pack_features_native(_host_numpy(cur), threads=host_threads)
                dlib = engine.DeviceLibrary(host_lib, tdev)
                n_bad = int((host_status != 0).sum())
                res.batches_packed_on_host += 1
            else:
                dlib = engine.DeviceLibrary.from_device_buffers(offsets, data, tdev, adopt=True)
        del cur_dev  # (its kernels are through: the packer stream has been waited for)
        with torch.cuda.stream(score_stream):
            out = engine.screen(model, dlib, weights=weights, topk=topk, index_base=res.num_ligands)
        tops.append((out.topk_scores, out.topk_indices))
        in_flight.append((dlib, out))  # (kept to the end: knowing earlier that a batch is through would take an event in the scoring stream's queue)
        res.library_bytes += dlib.num_bytes
        res.num_ligands += n
        res.num_conformers += dlib.total_conformers
        res.num_unsupported += n_bad
        res.batch_sizes.append(n)
        cur, cur_dev = nxt, nxt_dev
        cur_bound = engine.pack_bound(cur) if cur is not None else 0  # (this batch is enqueued: the host has time now)
        i += 1
    main.wait_stream(score_stream)
    if tops:
        res.topk_scores, res.topk_indices = engine.topk(torch.cat([t[0] for t in tops]), topk, indices=torch.cat([t[1] for t in tops]))
    else:
        res.topk_scores = torch.full((topk,), float("-inf"), dtype=torch.float32, device=tdev)
        res.topk_indices = torch.full((topk,), -1, dtype=torch.int64, device=tdev)
    if keep_scores:
        res.scores = torch.cat([o.scores for _, o in in_flight]) if in_flight else torch.empty(0, dtype=torch.float32, device=tdev)
        res.status = torch.cat([o.status for _, o in in_flight]) if in_flight else torch.empty(0, dtype=torch.int32, device=tdev)
    main.synchronize()
    for dlib, _ in in_flight:
        dlib.close()
    return res
