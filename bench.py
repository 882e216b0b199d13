#!/usr/bin/env python3
"""Benchmark of the screening hot path: ligand-conformers scored per second.

    python bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path (`PharmacophoreModel.screen`: score tables -> tree search -> scores,
then top-k) over this rank's resident library. Workloads (BASELINE.json `configs`):

* default, N = 1: configs[1] - the 6OIM-like pharmacophore model against 1M synthetic ligands (<= 32
  pharmacophore points, 8 conformers each);
* N > 1: configs[2] - every rank holds its own 12.5M-ligand shard of the synthetic library (100M ligands on 8
  GPUs, weak scaling) and each step ends with the all-gather of per-rank top-k over RCCL. `--gpus N` with no
  `WORLD_SIZE` in the environment starts the N ranks itself (re-executes under `torch.distributed.run`, one
  rank per GPU, rendezvous on 127.0.0.1); under the driver's own `torch.distributed.run` it is one of the ranks;
* `--pockets 16 --ligands 1253376`: the per-GPU shard of configs[3] (16 pockets x one shared library);
* `--model stress64`: configs[4] - the 64-node model against 64-conformer ligands (100 352 of them by default).

Rank 0 prints ONE JSON line. `value` is measured with the library already resident in HBM.
`roofline.achieved` prices the dominant kernel (the one with the largest HIP-event time per chunk) at the
ALGORITHMIC bytes of the path (packed ligand bytes + 8-byte offset in, 4-byte score + 4-byte status out, per
ligand it processes) over its HIP-event duration; this path is not HBM-bound, the fraction is
reported as asked. `parity_sample` compares 512 ligands of the timed library with the CPU oracle (outside the
timed region), so the line certifies what it measured.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec
PROFILE_TAG = "r6"  # the committed rocprofv3 --pmc summaries (profiles/<tag>_*.json) the static blocks of the line are read from

# name -> (model file under tests/golden, conformers, ligands on one GPU, topologies, fraction of ligands drawn on the model's nodes,
#          (seed of the topologies | None = tools.synthetic.BASE_SEED, offset of the perturbation stream's seed))
WORKLOADS = {
    "6oim": ("model_6oim_like.pm", 8, 1_000_000, 4096, 0.1, (None, 0)),
    "stress64": ("model_stress64.pm", 64, 100_352, 512, 0.2, (6464, 1)),  # (the library of tools/stress_shape.py, rounds 3-4)
}


def csrc_digest():
    """Content hash of the device sources (csrc/*.hip, csrc/*.h) + compile flags libpmx.so was BUILT from - the stamp pharmaconet_amd/build.py
    writes next to the library - so that a committed profile summary is quoted only for the kernels it was taken on. An A/B library
    (PMX_LIBPMX) has no stamp of its own: "variant", which matches no committed profile."""
    from pharmaconet_amd.build import read_stamp

    if os.environ.get("PMX_LIBPMX"):
        return "variant:" + Path(os.environ["PMX_LIBPMX"]).name
    stamp = read_stamp()
    return stamp["csrc_sha16"] if stamp else "unstamped"


def committed_profile(name):
    """profiles/<tag>_<name>.json if it was taken on THIS build of csrc/ (tools/collect_profiles.py stamps `csrc_sha16`), else None + why."""
    path = REPO / "profiles" / f"{PROFILE_TAG}_{name}.json"
    try:
        d = json.loads(path.read_text())
    except Exception:
        return None, f"profiles/{PROFILE_TAG}_{name}.json not found: not quoted"
    if d.get("csrc_sha16") != csrc_digest():
        return None, f"profiles/{PROFILE_TAG}_{name}.json was taken on another build of csrc/ ({d.get('csrc_sha16')} != {csrc_digest()}): not quoted"
    return d, f"profiles/{PROFILE_TAG}_{name}.json"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def start_ranks(args):
    """`--gpus N` with no rendezvous in the environment: start the N ranks (one per GPU) under torch.distributed.run and pass
    their output through; rank 0 of the children prints the JSON line."""
    import socket

    if os.environ.get("PMX_BENCH_BACKEND", "nccl") == "nccl":
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            log(f"bench.py: --gpus {args.gpus} but {have} GPU(s) visible (one rank per GPU over RCCL)")
            sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    log("bench.py: starting", args.gpus, "ranks:", " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def build_survey_library(model, n_ligands, n_conf, rank, device, active_fraction=0.1):
    """`--library survey`: SURVEY.md 8d-2's generator (tools/survey_library.py) - every ligand an independent draw from its own
    counter-based stream, n ~ clip(N(20, 6), 4, 32), conformer sigma 0.5 A, 10 % on the model's nodes (sigma 0.7 A) - made on the device;
    rank r holds ligands [r N, (r + 1) N) of the one library."""
    import torch

    from pharmaconet_amd.constants import TYPE_ID
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.survey_library import survey_library

    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    t0 = time.time()
    offsets, data, stats = survey_library(centers, types, n_ligands, n_conf, device, first=rank * n_ligands, active_fraction=active_fraction)
    lib = DeviceLibrary.from_device_buffers(offsets, data, device)
    torch.cuda.synchronize()
    log(f"[rank {rank}] survey library: {n_ligands} independent ligands, {lib.num_bytes / 1e9:.2f} GB, {stats}, built in {time.time() - t0:.1f}s")
    return lib, offsets, data, stats


def build_library(model, n_ligands, n_conf, base_count, rank, device, active_fraction=0.1, seed=None):
    import torch

    from pharmaconet_amd.constants import TYPE_ID
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import BASE_SEED, expand_library_on_device, synthetic_library

    seed, expand_offset = seed if seed is not None else (None, 0)
    seed = BASE_SEED if seed is None else seed
    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    t0 = time.time()
    base_count = min(base_count, n_ligands)
    # Every rank's shard holds the same topologies (ligand index = copy * topologies + topology, shards are
    # contiguous ranges of copies) with its own perturbation stream, so the ranks' work is statistically equal.
    molecules = []
    base = synthetic_library(
        base_count, first=0, num_conformers=n_conf, model_nodes=(centers, types),
        active_fraction=active_fraction, seed=seed, max_nodes=32, conformer_noise=0.0, molecules_out=molecules,
    )
    replicas = (n_ligands + base_count - 1) // base_count
    offsets, data = expand_library_on_device(base, replicas, device, seed=seed + expand_offset + 1000 * rank)
    n_total = replicas * base_count
    lib = DeviceLibrary.from_device_buffers(offsets, data, device)
    torch.cuda.synchronize()
    log(f"[rank {rank}] library: {n_total} ligands ({base_count} topologies x {replicas}), "
        f"{lib.num_bytes / 1e9:.2f} GB, max nodes {lib.max_nodes}, built in {time.time() - t0:.1f}s")
    return lib, offsets, data, molecules


def end_to_end(molecules, lib, data, ms_per_step, n_conf_total):
    """What the stages in front of the resident-library pass cost: the native packer (pmx_pack_features, every host core) on
    the library's molecule topologies, and the copy of the packed library over PCIe. Stages are summed, not overlapped."""
    import torch

    from pharmaconet_amd.library import flatten_features, pack_features_native

    # (the packer is memory-bound beyond some 64 threads on the 256-core host: 14.0e6 ligands/s at 64, 11.3e6 at 256 [MI355X box])
    cores = min(os.cpu_count() or 1, 64)
    reps = max(1, 262144 // max(len(molecules), 1))
    flat = flatten_features(molecules * reps)
    pack_features_native(flat, threads=cores)  # warm
    pack_s = 1e9
    for _ in range(3):  # best of three (the host is shared with whatever else the box runs)
        t0 = time.perf_counter()
        packed, _ = pack_features_native(flat, threads=cores)
        pack_s = min(pack_s, time.perf_counter() - t0)
    pack_rate = len(packed) / pack_s
    host = torch.empty(data.numel(), dtype=torch.uint8).pin_memory()
    host.copy_(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev = host.to(data.device, non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t0) * 1e3
    del dev, host
    n_lig = len(lib)
    total_s = n_lig / pack_rate + h2d_ms / 1e3 + ms_per_step / 1e3
    return {
        "pack_ligands_per_s": pack_rate,
        "pack_threads": cores,
        "pack_s_for_this_library": n_lig / pack_rate,
        "h2d_ms": h2d_ms,
        "h2d_GBps": data.numel() / 1e9 / (h2d_ms / 1e3),
        "gpu_pass_ms": ms_per_step,
        "ligand_conformers_per_s": n_conf_total / total_s,
        "note": "typed features -> packed records (native packer, byte-identical to the reference's LigandGraph on the fixtures) "
                "+ pinned host -> HBM copy + one resident pass, summed without overlap; ligand perception (OpenBabel in the "
                "reference) is not part of it and is not pinned here",
    }


def tile_features(flat, reps, rng, node_sigma=0.35, conf_sigma=0.30):
    """`reps` copies of a flat feature batch (flatten_features) as ONE batch, every copy with its own geometry: each atom displaced by
    N(0, node_sigma) in all conformers and every conformer coordinate by a further N(0, conf_sigma) - the host-side twin of
    tools.synthetic.expand_library_on_device, so that what the pipeline packs and scores is the bench library's kind of ligand."""
    n_mol = len(flat["atom_off"]) - 1
    out = {}
    for off_key, arrays in (("atom_off", ("atomic_num",)), ("nbr_off", ("nbr",)), ("feat_off", ("feat_type", "feat_flags")),
                            ("feat_atom_off", ("feat_atoms",)), ("feat_center_off", ("feat_centers",)), ("pos_off", ())):
        off = flat[off_key].astype(np.uint64)
        total = off[-1]
        out[off_key] = np.concatenate([(off[:-1][None, :] + (np.arange(reps, dtype=np.uint64) * total)[:, None]).reshape(-1), [reps * total]]).astype(np.uint64)
        for a in arrays:
            out[a] = np.tile(flat[a], reps)
    out["n_conf"] = np.tile(flat["n_conf"], reps)
    # positions [atoms][C][3] per molecule: per-atom shift + per-coordinate jitter
    base = flat["positions"]
    n_atoms = np.diff(flat["atom_off"].astype(np.int64))
    conf = flat["n_conf"].astype(np.int64)
    atom_of_float = np.repeat(np.arange(int(n_atoms.sum())), np.repeat(conf * 3, n_atoms))  # float -> atom (of the base batch)
    axis = np.tile(np.arange(3), base.size // 3)
    pos = np.empty(reps * base.size, dtype=np.float32)
    for r in range(reps):
        shift = rng.normal(scale=node_sigma, size=(int(n_atoms.sum()), 3)).astype(np.float32)
        pos[r * base.size : (r + 1) * base.size] = base + shift[atom_of_float, axis] + rng.normal(scale=conf_sigma, size=base.size).astype(np.float32)
    out["positions"] = pos
    assert len(out["atom_off"]) == n_mol * reps + 1
    return out


def end_to_end_overlapped(molecules, pocket, n_lig_target, topk_k, device, shares=(1, 3, 4)):
    """The path of `screening.py:63-70` as ONE pipeline: host packer threads -> pinned double buffer -> H2D on a copy stream -> library
    upload + `pmx_score` + `pmx_topk` on the compute stream, chunk by chunk, the packer running ahead of the GPU. Timed from the first
    byte packed to the merged top-k on the host. (Perception is in front of this and needs the chemistry toolkit.)"""
    import ctypes
    import threading

    import torch

    from pharmaconet_amd import _ffi, engine
    from pharmaconet_amd.engine import DeviceLibrary
    from pharmaconet_amd.library import flatten_features

    cores, host = host_cores()
    # (the box's quota - 16 cores' worth per 100 ms period - does not stop a burst of more threads from running side by side: [MI355X box] 262 144 molecules in 48 / 33 / 18.5 /
    # 19.8 ms on 16 / 32 / 64 / 128 threads; the pipeline packs in bursts and idles in between)
    threads = int(os.environ.get("PMX_BENCH_PACK_THREADS", max(1, min(host["hardware_threads"], 64))))
    # Chunks of growing size - a small first one so that the GPU starts a few milliseconds after the packer, large later ones because every chunk of a scoring call
    # ends in the same 6 ms of task-round tail - in the proportions `shares`. ONE tiled feature batch of the largest chunk's size is made;
    # a chunk packs a prefix of it (the flat layout's offsets are prefix-compatible), so every chunk is real packer work and real, distinct-geometry ligands.
    if os.environ.get("PMX_BENCH_E2E_SHARES"):
        shares = tuple(int(x) for x in os.environ["PMX_BENCH_E2E_SHARES"].split(","))
    total_reps = max(len(shares), n_lig_target // len(molecules))
    chunk_reps = [max(1, total_reps * sh // sum(shares)) for sh in shares]
    n_chunks = len(chunk_reps)
    flat = tile_features(flatten_features(molecules), max(chunk_reps), np.random.default_rng(12345))
    lib = _ffi.load_packer()
    n = max(chunk_reps) * len(molecules)
    chunk_n = [r_ * len(molecules) for r_ in chunk_reps]
    fields = ("atom_off", "atomic_num", "nbr_off", "nbr", "feat_off", "feat_type", "feat_flags", "feat_atom_off", "feat_atoms",
              "feat_center_off", "feat_centers", "n_conf", "pos_off", "positions")
    batch = _ffi.FeatureBatch(n, *(flat[k].ctypes.data for k in fields))
    batches = [_ffi.FeatureBatch(cn, *(flat[k].ctypes.data for k in fields)) for cn in chunk_n]
    need = ctypes.c_uint64(0)
    off_probe = np.zeros(n + 1, dtype=np.uint64)
    assert lib.pmx_pack_features(ctypes.byref(batch), threads, off_probe.ctypes.data, None, 0, ctypes.byref(need), None) == 0
    cap = int(need.value)
    pinned = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
    pinned_off = [torch.empty(n + 1, dtype=torch.int64).pin_memory() for _ in range(2)]
    dev_data = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(2)]
    dev_off = [torch.empty(n + 1, dtype=torch.int64, device=device) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device)
    compute = torch.cuda.current_stream(device)

    def run(timed):
        packed = [threading.Event() for _ in range(n_chunks)]
        free = [threading.Event() for _ in range(n_chunks)]  # free[i]: pinned buffer of chunk i may be overwritten (its H2D is done)
        nbytes = [0] * n_chunks
        err = []

        def producer():
            try:
                for i in range(n_chunks):
                    if i >= 2:
                        free[i - 2].wait()
                    got = ctypes.c_uint64(0)
                    rc = lib.pmx_pack_features(ctypes.byref(batches[i]), threads, pinned_off[i % 2].data_ptr(), pinned[i % 2].data_ptr(), cap, ctypes.byref(got), None)
                    if rc != 0:
                        raise RuntimeError(lib.pmx_last_error().decode())
                    nbytes[i] = int(got.value)
                    packed[i].set()
            except Exception as e:  # pragma: no cover
                err.append(e)
                for ev in packed:
                    ev.set()

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = threading.Thread(target=producer)
        th.start()
        tops, libs, h2d_done, scored = [], [], [], []
        for i in range(n_chunks):
            packed[i].wait()
            if err:
                raise err[0]
            b = i % 2
            with torch.cuda.stream(copy_stream):
                if i >= 2:
                    copy_stream.wait_event(scored[i - 2])  # the device buffer's last reader
                dev_data[b][: nbytes[i]].copy_(pinned[b][: nbytes[i]], non_blocking=True)
                dev_off[b][: chunk_n[i] + 1].copy_(pinned_off[b][: chunk_n[i] + 1], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            h2d_done.append(ev)
            compute.wait_event(ev)
            ev.synchronize()  # (the pinned buffer is free for the packer; pmx_library_upload reads the offsets on the host side of its checks)
            free[i].set()
            dlib = DeviceLibrary.from_device_buffers(dev_off[b][: chunk_n[i] + 1], dev_data[b][: nbytes[i]], device)
            res = engine.screen(pocket, dlib, topk=topk_k, index_base=sum(chunk_n[:i]))
            sev = torch.cuda.Event()
            sev.record(compute)
            scored.append(sev)
            tops.append((res.topk_scores, res.topk_indices))
            libs.append(dlib)
        th.join()
        top = engine.topk(torch.cat([t[0] for t in tops]), topk_k, indices=torch.cat([t[1] for t in tops]))
        best = top[0].cpu()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        total_conf = sum(l.total_conformers for l in libs)
        for l in libs:
            l.close()
        return dt, total_conf, best

    run(False)  # warm: pinned pages touched, workspaces sized for the chunk
    dt, total_conf, best = min((run(True) for _ in range(3)), key=lambda r: r[0])
    return {
        "overlapped_ligand_conformers_per_s": total_conf / dt,
        "overlapped_s": dt,
        "ligands": sum(chunk_n),
        "chunks": chunk_n,
        "pack_threads": threads,
        "host": host,
        "best_score_of_the_run": float(best[0]),
        "note": "packer threads -> pinned double buffer -> H2D on a copy stream -> pmx_library_upload + pmx_score + pmx_topk on the compute stream, per chunk, "
                "the packer running ahead of the GPU, chunks of growing size; first byte packed to merged top-k on the host; best of three. The packer needs 2.7 CPU-seconds per "
                "10^6 ligands (0.375 x 10^6 ligands/s per thread): on the 16 cores' worth of time the box grants that is 0.17 s sustained whatever the thread count - the floor of this leg",
    }


def end_to_end_device_packed(molecules, pocket, n_lig_target, topk_k, device, shares=(1, 3, 4)):
    """The same pipeline with the packer ON THE DEVICE - the product's `pharmaconet_amd.pipeline.screen_feature_batches`: the typed features and conformer coordinates
    go over PCIe as they are (pinned flat arrays -> HBM on copy streams, chunk i + 1 while chunk i is packed and scored); per chunk a packer stream runs graph builder + record
    writer (`pmx_pack_features_device`), the library adopts their output in place, and a scoring stream runs `pmx_score` + `pmx_topk`. No host core does more than enqueue.
    Timed from the first copy enqueued to the merged top-k on the host."""
    import torch

    from pharmaconet_amd import engine, pipeline
    from pharmaconet_amd.engine import FEATURE_FIELDS
    from pharmaconet_amd.library import flatten_features

    if os.environ.get("PMX_BENCH_E2E_SHARES"):
        shares = tuple(int(x) for x in os.environ["PMX_BENCH_E2E_SHARES"].split(","))
    total_reps = max(len(shares), n_lig_target // len(molecules))
    chunk_reps = [max(1, total_reps * sh // sum(shares)) for sh in shares]
    flat = tile_features(flatten_features(molecules), max(chunk_reps), np.random.default_rng(12345))
    chunk_n = [r_ * len(molecules) for r_ in chunk_reps]

    # what a chunk of cn molecules takes of each flat array (prefixes: the flat layout's offsets are prefix-compatible)
    def prefix_lengths(cn):
        ln = {"atom_off": cn + 1, "feat_off": cn + 1, "pos_off": cn + 1, "n_conf": cn}
        n_atoms, n_feat = int(flat["atom_off"][cn]), int(flat["feat_off"][cn])
        ln.update(atomic_num=n_atoms, nbr_off=n_atoms + 1, feat_type=n_feat, feat_flags=n_feat, feat_atom_off=n_feat + 1, feat_center_off=n_feat + 1)
        ln.update(nbr=int(flat["nbr_off"][n_atoms]), feat_atoms=int(flat["feat_atom_off"][n_feat]), feat_centers=int(flat["feat_center_off"][n_feat]),
                  positions=int(flat["pos_off"][cn]))
        return ln

    pinned = pipeline.pin_features(flat)
    lengths = [prefix_lengths(cn) for cn in chunk_n]
    raw_bytes = [sum(ln[k] * pinned[k].element_size() for k in FEATURE_FIELDS) for ln in lengths]
    batches = [{k: pinned[k][: ln[k]] for k in FEATURE_FIELDS} for ln in lengths]

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipeline.screen_feature_batches(pocket, batches, topk_k, device=device)
        best = res.topk_scores.cpu()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, res, best

    run()  # warm: torch's pools hold the buffers, the scoring stream its workspace
    dt, res, best = min((run() for _ in range(3)), key=lambda r: r[0])
    # the packer's kernels alone, on the largest chunk, resident
    dev_in = engine.features_to_device(batches[-1], device)
    bound = engine.pack_bound(batches[-1])
    engine.pack_features_device(dev_in, device, bound=bound)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.pack_features_device(dev_in, device, bound=bound)
    torch.cuda.synchronize()
    pack_ms = (time.perf_counter() - t0) * 1e3
    return {
        "device_packed_ligand_conformers_per_s": res.num_conformers / dt,
        "device_packed_s": dt,
        "device_packed_chunks": res.batch_sizes,
        "device_packed_feature_bytes_over_pcie": sum(raw_bytes),
        "device_packed_library_bytes": res.library_bytes,
        "device_packed_molecules_not_packed": res.num_unsupported,
        "device_packed_batches_packed_on_host": res.batches_packed_on_host,
        "device_packer_molecules_per_s": chunk_n[-1] / (pack_ms / 1e3),
        "device_packed_best_score_of_the_run": float(best[0]),
        "device_packed_note": "pharmaconet_amd.pipeline.screen_feature_batches: typed features + conformer coordinates (pinned) -> HBM on copy streams -> pmx_pack_features_device (graph builder + record "
                              "writer, records byte-identical to the host packer's: tests/test_gpu_pack_device.py) on a packer stream + pmx_library_upload (adopting the packer's buffers in place) + "
                              "pmx_score + pmx_topk on a scoring stream, per chunk, chunk i + 1 packed while chunk i is scored; first copy enqueued to merged top-k on the host; best of three",
    }


def host_sample(offsets, data, index):
    """The records `index` (ascending ligand numbers) of the device library as a host `PackedLibrary`."""
    from pharmaconet_amd.library import PackedLibrary

    off = offsets.cpu().numpy().astype(np.int64)
    recs = [bytes(data[int(off[i]) : int(off[i + 1])].cpu().numpy()) for i in index]
    return PackedLibrary.from_records(recs)


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup's CPU quota (cpu.max = "quota period": the GPU
    boxes show 256 hardware threads and grant 16 cores' worth of time - os.cpu_count() is not what OpenMP gets)."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = os.cpu_count() or 1
    info = {"hardware_threads": os.cpu_count() or 1, "affinity": affinity, "cgroup_cpu_max": None}
    cores = affinity
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(text)
                if text[0] != "max":
                    cores = min(cores, max(1, int(int(text[0]) / int(text[1]) + 0.5)))
            else:
                quota = int(text[0])
                period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                info["cgroup_cpu_max"] = f"{quota} {period}"
                if quota > 0:
                    cores = min(cores, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    try:
        for ln in Path("/proc/cpuinfo").read_text().splitlines():
            if ln.startswith("model name"):
                info["cpu_model"] = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    info["usable_cores"] = cores
    return cores, info


def reference_rate():
    """The reference's own NumPy path (`GraphMatcher.run()` imported from /root/reference, one process) timed in the BUILD container on the
    golden ligands - it cannot run on the GPU box (tools/reference_rate.py writes the file). A quoted, static number."""
    try:
        return json.loads((REPO / "profiles" / "reference_numpy_rate.json").read_text())
    except Exception:
        return None


def cpu_baseline(pockets, offsets, data, n_conf, budget_s=12.0):
    """Time the CPU oracle (oracle/, a port of the reference's algorithm pinned to its outputs) on a bounded sample of the same
    library: with every core this process may use (`host_cores`), and with one (`screening.py --cpus N` /
    `--cpus 1`). The all-core sample holds at least 256 ligands per thread so that no thread idles behind a heavy ligand for long
    (the top 1 % of the ligands are a third of the tree nodes)."""
    from oracle import oracle as orc
    from pharmaconet_amd.constants import weights_vector
    from pharmaconet_amd.library import PackedLibrary

    cores, host = host_cores()
    n_all = offsets.numel() - 1

    def sample(n):
        off = offsets[: n + 1].cpu().numpy().astype(np.uint64)
        return PackedLibrary(off, data[: int(off[-1])].cpu().numpy())

    def timed(lib, threads):
        t0 = time.perf_counter()
        for pocket in pockets:
            orc.oracle_score(pocket.flat, lib, w, num_threads=threads)
        return time.perf_counter() - t0

    w = weights_vector(None)
    probe = sample(min(2048 if n_conf <= 16 else 256, n_all))
    t0 = time.perf_counter()
    _, probe_stats = orc.oracle_score(pockets[0].flat, probe, w, num_threads=cores, with_stats=True)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6) / len(pockets)  # ligands/s with every core, all pockets
    work = {  # what the reference's algorithm does per ligand on this library (oracle counters on the probe, first pocket)
        "gaussian_terms_per_ligand_conformer": float(probe_stats["n_terms"].mean()),
        "tree_nodes_per_ligand_without_bound_test": float(probe_stats["n_tree"].mean()),
    }
    per_thread = 256 if n_conf <= 16 else 32
    n = int(min(n_all, max(len(probe), rate * budget_s, per_thread * cores)))
    lib = sample(n)
    dt = timed(lib, cores)
    # one core: sized from a short probe of its own (threads share caches and clocks: one core alone is faster than 1 / cores of all)
    probe1 = sample(min(64, n))
    rate1 = len(probe1) / max(timed(probe1, 1), 1e-6)
    n1 = int(min(n, max(len(probe1), rate1 * 0.6 * budget_s)))
    lib1 = sample(n1)
    dt1 = timed(lib1, 1)
    all_rate, one_rate = n * n_conf * len(pockets) / dt, n1 * n_conf * len(pockets) / dt1
    out = {
        "value": all_rate,
        "unit": "ligand-conformers/s" if len(pockets) == 1 else "pocket-ligand-conformers/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} ligands of the same library ({n // max(cores, 1)} per thread)" + (f" against the {len(pockets)} pockets" if len(pockets) > 1 else "")
                  + f", OpenMP over ligands (dynamic schedule), {cores} threads, {dt:.1f}s",
        "host": host,
        "one_core": {"value": one_rate, "cores": 1, "sample": f"first {n1} ligands, {dt1:.1f}s"},
        # thread-seconds spent per unit of one-core work: 1.0 = the threads scale perfectly
        "parallel_efficiency": all_rate / max(one_rate * cores, 1e-9),
        "reference_numpy_path": reference_rate(),
    }
    return out, work


def parity_sample(pocket, scores, offsets, data, n=512):
    """512 evenly spaced ligands of the timed library, the GPU's scores of the last timed step against the oracle's."""
    from oracle import oracle as orc
    from pharmaconet_amd.constants import weights_vector

    n_lig = offsets.numel() - 1
    index = np.unique(np.linspace(0, n_lig - 1, num=min(n, n_lig)).astype(np.int64))
    lib = host_sample(offsets, data, index)
    import torch

    got = scores[torch.from_numpy(index).to(scores.device)].cpu().numpy().astype(np.float64)
    ref = orc.oracle_score(pocket.flat, lib, weights_vector(None), num_threads=host_cores()[0])
    nz = ref > 0
    err = np.abs(got[nz] - ref[nz]) / ref[nz]
    return {
        "ligands": int(len(index)),
        "against": "oracle/ (CPU restatement of GraphMatcher.run(), pinned to the reference's outputs), outside the timed region",
        "nonzero_scores": int(nz.sum()),
        "max_rel_err": float(err.max()) if nz.any() else 0.0,
        "median_rel_err": float(np.median(err)) if nz.any() else 0.0,
        "above_1e-5": int((err > 1e-5).sum()),
        "zero_nonzero_mismatches": int(((got > 0) != nz).sum()),
    }


def issue_block(n_lig, pass_ms):
    """Scalar / vector wave-instructions per ligand (committed SQ counters of THIS build) against the CU's issue rates at this run's pass time."""
    prof, src = committed_profile("pmc_sq_summary")
    if prof is None:
        return {"from_profile": None, "note": src}
    per = prof["ligands"] * prof.get("passes", 2)
    salu = sum(v["SQ_INSTS_SALU"] for v in prof["counters"].values()) / per
    valu = sum(v["SQ_INSTS_VALU"] for v in prof["counters"].values()) / per
    cus, clock_hz = 256, 2.4e9
    scalar_ms = salu * n_lig / cus / clock_hz * 1e3          # one scalar wave-instruction per cycle and CU
    vector_ms = valu * n_lig * 2 / (4 * cus) / clock_hz * 1e3  # a wave64 VALU instruction holds one of 4 SIMD-32s for 2 cycles
    out = {"scalar_insts_per_ligand": salu, "vector_insts_per_ligand": valu, "scalar_unit_busy": scalar_ms / pass_ms,
           "vector_units_busy": vector_ms / pass_ms, "pass_ms": pass_ms, "priced_at_clock_hz": clock_hz}
    # the same against the clock the profile measured under this load, with the branches (they are issued from the same scalar stream)
    if all("SQ_INSTS_BRANCH" in v for v in prof["counters"].values()) and prof.get("clock", {}).get("clock_hz"):
        branches = sum(v["SQ_INSTS_BRANCH"] for v in prof["counters"].values()) / per
        measured = float(prof["clock"]["clock_hz"])
        out.update({"branches_per_ligand": branches, "measured_clock_hz": measured,
                    "scalar_unit_busy_at_measured_clock": salu * n_lig / cus / measured * 1e3 / pass_ms,
                    "scalar_and_branch_issue_at_measured_clock": (salu + branches) * n_lig / cus / measured * 1e3 / pass_ms})
    return {**out,
            "from_profile": src,
            "note": "instruction counts are NOT measured in this run: they come from the committed rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU pass of the same build of csrc/ (checked by content hash), priced at this run's pass time"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", choices=sorted(WORKLOADS), default="6oim", help="6oim: BASELINE configs[1] / [2]; stress64: configs[4] (64-node model, 64 conformers)")
    ap.add_argument("--ligands", type=int, default=0, help="ligands per GPU (default: 1M on one GPU = BASELINE configs[1]; 12.5M per GPU on several = configs[2]; 100 352 for --model stress64)")
    ap.add_argument("--conformers", type=int, default=0)
    ap.add_argument("--topologies", type=int, default=0, help="distinct synthetic molecules per GPU")
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--library", choices=("expanded", "survey"), default="expanded",
                    help="expanded: molecule topologies x jittered copies (tools/synthetic.py; the default line); survey: SURVEY.md 8d-2's generator, every ligand an independent draw (tools/survey_library.py)")
    ap.add_argument("--pockets", type=int, default=1, help="score this many distinct pockets (1..16) against the shared library (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial-leg", action="store_true", help="skip the end-to-end (pack + copy) side measurement")
    ap.add_argument("--no-survey-leg", action="store_true", help="skip the side measurement on SURVEY 8d-2's own library (the default line carries it as `survey_library`)")
    ap.add_argument("--no-parity-sample", action="store_true")
    ap.add_argument("--dump-dir", default=None, help="every rank writes its shard's scores and rank 0 the merged top-k here (tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        start_ranks(args)  # does not return

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    model_file, wl_conf, wl_ligands, wl_topo, wl_active, wl_seed = WORKLOADS[args.model]
    if args.ligands <= 0:
        args.ligands = wl_ligands if (world == 1 or args.model != "6oim") else 12_500_000
    args.conformers = args.conformers or wl_conf
    args.topologies = args.topologies or wl_topo
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    # PMX_BENCH_DEVICE / PMX_BENCH_BACKEND exist to rehearse the multi-rank path on a 1-GPU box (all ranks on one
    # device, gloo); the driver's runs use one GPU per rank and RCCL ("nccl").
    dev_index = int(os.environ.get("PMX_BENCH_DEVICE", local_rank))
    backend = os.environ.get("PMX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # Control plane (barrier, the maximum of the ranks' times, the hand-over of the RCCL id) on gloo: the one RCCL user of the process
    # is libpmx's own communicator, which carries the data path (pmx_topk_allgather). PMX_BENCH_CONTROL=nccl puts torch's process group
    # on RCCL as well.
    control = os.environ.get("PMX_BENCH_CONTROL", "gloo")
    dist = None
    if world > 1:
        import torch.distributed as dist

        if control == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    import __graft_entry__ as entry

    entry.build()
    from pharmaconet_amd import PharmacophoreModel
    from pharmaconet_amd import engine
    from pharmaconet_amd.distributed import TopkExchange, allgather_topk_device

    model = PharmacophoreModel.load(REPO / "tests" / "golden" / model_file)
    pockets = [model]
    if args.pockets > 1:
        pockets = [PharmacophoreModel.load(REPO / "tests" / "golden" / "pockets16" / f"model_{k:02d}.pm") for k in range(min(args.pockets, 16))]
    # (TopkExchange raises unless RCCL itself reports WORLD_SIZE ranks, pmx_comm_info: a line that says n_gpus = N is an N-rank exchange)
    exchange = TopkExchange(device) if (world > 1 and backend == "nccl") else None
    if exchange is not None and exchange.rccl_ranks != world:
        raise SystemExit(f"bench.py: RCCL communicator spans {exchange.rccl_ranks} rank(s), WORLD_SIZE is {world}")
    survey_stats = None
    if args.library == "survey":
        lib, offsets, data, survey_stats = build_survey_library(model, args.ligands, args.conformers, rank, device, wl_active)
        molecules = None  # (packed records are generated directly: there are no molecules for the packer leg)
    else:
        lib, offsets, data, molecules = build_library(model, args.ligands, args.conformers, args.topologies, rank, device, wl_active, wl_seed)
    n_lig = len(lib)
    n_conf_total = lib.total_conformers
    index_base = rank * n_lig

    def step():
        for pocket in pockets:  # pocket-outer / ligand-inner: the library stays resident, every pocket makes one pass
            res = engine.screen(pocket, lib, topk=args.topk, index_base=index_base)
            if world > 1:
                if exchange is not None:  # RCCL all-gather + merge on the device through libpmx's C ABI; the ranking stays on the device
                    top = exchange.allgather(res.topk_scores, res.topk_indices, args.topk)
                else:
                    top = allgather_topk_device(res.topk_scores, res.topk_indices, args.topk)  # rehearsal transport, the device merge of the RCCL path
            else:
                top = (res.topk_scores, res.topk_indices)
        return res, top

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # the timed steps run the product as a user runs it: profiling off, nothing read back but the final ranking
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, top = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if args.dump_dir:
        os.makedirs(args.dump_dir, exist_ok=True)
        np.savez(os.path.join(args.dump_dir, f"rank{rank}.npz"), scores=res.scores.cpu().numpy(), index_base=index_base,
                 top_scores=np.asarray(top[0].cpu() if hasattr(top[0], "cpu") else top[0]), top_indices=np.asarray(top[1].cpu() if hasattr(top[1], "cpu") else top[1]))
    # One more, untimed pass with HIP events around the phases (pmx_set_profiling: event records only) and the device
    # counters read back: the kernel durations behind the roofline block.
    prof = None
    if rank == 0:
        engine.set_profiling(True)
        try:
            engine.screen(pockets[-1], lib, topk=args.topk, index_base=index_base)
            torch.cuda.synchronize()
            prof = engine.last_score_stats()
            log(f"[rank 0] profiled pass: {prof}")
        finally:
            engine.set_profiling(False)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device if (world > 1 and control == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / max(args.steps, 1)
        value = world * n_conf_total * len(pockets) * args.steps / elapsed
        # algorithmic bytes: records + 8 B offset in, 4 B score + 4 B status out, per ligand
        alg_bytes_per_ligand = lib.num_bytes / n_lig + 8 + 4 + 4
        kernels = {
            "ligand_kernel (score tables in per-wavefront slices + tree search within the pass budget; 3 launches: slices, large slices, arena)": prof["ms_ligand"],
            "task_kernel (queued subtrees of over-budget trees; all rounds) + finalize": prof["ms_tasks"],
        }
        dominant = max(kernels, key=kernels.get)
        dom_ms = kernels[dominant]
        ligands_per_launch = prof["ligands_last"]
        achieved = (alg_bytes_per_ligand * ligands_per_launch) / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM traffic per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately with
        # rocprofv3 --pmc and corrected as MI355X_MICROARCH.md prescribes), scaled to this launch's ligands - quoted only
        # when the summary was taken on this build of csrc/ and on this workload
        bench_shape = args.model == "6oim" and args.conformers == 8 and len(pockets) == 1 and args.library == "expanded"
        traffic, traffic_src = None, "profiled on the 6OIM-like model at 8 conformers only: not quoted for this workload"
        if bench_shape:
            pmc, traffic_src = committed_profile("hbm_traffic")
            if pmc is not None:
                key = "ligand_kernel" if dominant.startswith("ligand_kernel") else "task_kernel"
                traffic = pmc["kernels"][key]["hbm_bytes_per_ligand"] * ligands_per_launch
        model_words = {"6oim": "6OIM-like model (37 nodes, 11 clusters)", "stress64": "stress model (64 nodes, 54 clusters)"}[args.model]
        out = {
            "metric": f"ligand-conformers scored/sec ({len(pockets)} pocket{'s' if len(pockets) > 1 else ''})",
            "value": value,
            "unit": "ligand-conformers/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (model_words if len(pockets) == 1 else f"{len(pockets)} fixture pockets (pockets16)")
                            + (f" vs {n_lig} synthetic ligands per GPU = {min(args.topologies, n_lig)} molecule topologies x "
                               f"{-(-n_lig // min(args.topologies, n_lig))} copies with every node displaced (sigma 0.35 A) and every conformer "
                               f"coordinate jittered (sigma 0.30 A); mean {lib.num_bytes / n_lig / (12.0 * args.conformers):.1f} pharmacophore nodes "
                               f"(<= 32), {args.conformers} conformers each, {100 * wl_active:.0f} % drawn on the model's nodes; top-{args.topk}"
                               if survey_stats is None else
                               f" vs {n_lig} synthetic ligands per GPU, SURVEY.md 8d-2's generator (tools/survey_library.py): every ligand an independent draw "
                               f"from its own counter-based stream, n ~ clip(N(20, 6), 4, 32) nodes (mean {survey_stats['mean_nodes']:.1f}, "
                               f"{survey_stats['mean_clusters']:.1f} clusters), {args.conformers} conformers = base + N(0, 0.5 A), "
                               f"{100 * survey_stats['active_share']:.0f} % on the model's nodes (sigma 0.7 A, random rigid motion), the rest random walks; top-{args.topk}"),
                "library": args.library,
                "library_stats": survey_stats,
                "pockets": len(pockets),
                "ligands_per_gpu": n_lig,
                "conformers_per_ligand": args.conformers,
                "parallelism": f"ligand-sharded x{world}" if world > 1 else "single GPU",
                "exchange": None if world == 1 else ({"collective": "ncclAllGather of per-rank top-k (pmx_topk_allgather, merge on the device)", "rccl_ranks": exchange.rccl_ranks, "control_plane": control}
                                                     if exchange is not None else {"collective": f"{backend} all_gather of per-rank top-k (rehearsal transport) + the device merge of pmx_topk_allgather (pmx_topk over the gathered lists)", "ranks": world}),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dominant,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_from_profile": traffic_src,
                "traffic_note": "NOT measured in this run: HBM bytes per ligand of that kernel from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same build of csrc/ (content hash checked; separate runs, read side doubled per MI355X_MICROARCH.md), scaled to this launch's ligands; null if no summary of this build and workload is committed",
                "algorithmic_bytes_per_ligand": alg_bytes_per_ligand,
                "ligands_per_launch": ligands_per_launch,
                "kernel_ms_per_launch": kernels,
                "profiled_pass_ms": prof["ms_total"],
                "note": "HIP-event times of one extra, untimed pass with pmx_set_profiling(1) (event records on the call's stream, no synchronisation); "
                        "the timed steps run with profiling off. The path is not HBM-bound (SURVEY.md section 0, DESIGN.md section 4): what binds is "
                        "instruction issue and memory latency, see `issue` (DESIGN.md section 4).",
            },
            # The resource that is busiest on this path: the CU's scalar unit, one wave-instruction per cycle.
            "issue": issue_block(n_lig, prof["ms_total"]) if bench_shape else None,
            # what the kernels do per ligand
            "work": {
                "tree_frames_per_ligand": prof["n_frames"] / max(n_lig, 1),
                "walker_passes_per_ligand": prof["n_passes"] / max(n_lig, 1),
                "table_items_per_ligand_conformer": prof["n_items"] * (64 // max(1, 1 << (args.conformers - 1).bit_length())) / max(n_lig, 1),  # (wave-iterations x slots of a wavefront)
                "queued_subtrees_per_ligand": prof["n_tasks"] / max(n_lig, 1),
                "path_bound_tests_per_ligand": prof["n_path_bounds"] / max(n_lig, 1),
                "children_dropped_by_path_bound_per_ligand": prof["n_path_drops"] / max(n_lig, 1),
                "probe_passes_per_ligand": prof["n_probe_passes"] / max(n_lig, 1),
                "walks_over_budget_per_ligand": prof["n_heavy"] / max(n_lig, 1),
                "ligands_with_tables_beyond_a_slice": prof["n_slice_overflow"],
                "longest_walk_passes": prof["max_passes"],
                # (above 1: split trees found the table arena full and were walked by one wavefront each - exact, slow: PMX_SUPER / PMX_ARENA_MB)
                "arena_asked_over_capacity": prof["arena_bytes"] / max(prof.get("arena_capacity", 0), 1),
                "wave_time_share": {k: prof["ticks_" + k] / max(prof["ticks_alive"], 1) for k in ("scan", "tables", "bounds", "walk")},
            },
            "csrc_sha16": csrc_digest(),
        }
        if not args.no_parity_sample:
            try:
                out["parity_sample"] = parity_sample(pockets[-1], res.scores, offsets, data)
            except Exception as e:  # never lose the bench line over the check - but say so
                out["parity_sample"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], work = cpu_baseline(pockets, offsets, data, args.conformers)
            # Gaussian terms as the reference evaluates them (counted by the oracle on the same ligands) per second: the
            # table phase replaces each group of |A||B| terms by one tabulated pair function, so this is an equivalent rate
            terms_per_conf = work["gaussian_terms_per_ligand_conformer"]
            out["work"].update(work)
            out["work"]["reference_equivalent_gaussian_terms_per_s"] = terms_per_conf * value
            out["work"]["reference_terms_per_table_item"] = terms_per_conf / max(out["work"]["table_items_per_ligand_conformer"], 1e-9)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_serial_leg and molecules is not None:
            try:
                out["end_to_end"] = end_to_end(molecules, lib, data, ms_per_step, n_conf_total * len(pockets))
                if len(pockets) == 1:
                    out["end_to_end"].update(end_to_end_overlapped(molecules, pockets[0], n_lig, args.topk, device))
                    out["end_to_end"].update(end_to_end_device_packed(molecules, pockets[0], n_lig, args.topk, device))
            except Exception as e:  # never lose the bench line over the side measurement
                out["end_to_end"] = {**(out.get("end_to_end") or {}), "error": repr(e)}
        # The same pass on SURVEY.md 8d-2's OWN generator (`--library survey` makes it the line's workload): a side block of the default line, so that
        # one driver run carries both libraries. Untimed for `value`; its own warm-up, two timed passes, its own parity sample.
        if world == 1 and bench_shape and not args.no_survey_leg and args.ligands >= 1_000_000:
            try:
                slib, soff, sdata, sstats = build_survey_library(model, 1_000_000, args.conformers, 0, device, wl_active)
                engine.screen(model, slib, topk=args.topk)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    sres = engine.screen(model, slib, topk=args.topk)
                torch.cuda.synchronize()
                sdt = (time.perf_counter() - t0) / 2
                out["survey_library"] = {
                    "value": slib.total_conformers / sdt, "unit": "ligand-conformers/s", "ms_per_step": sdt * 1e3, "ligands": len(slib),
                    "library_stats": sstats, "command": "python bench.py --library survey",
                    "parity_sample": None if args.no_parity_sample else parity_sample(model, sres.scores, soff, sdata),
                    "note": "tools/survey_library.py: every ligand an independent draw (n ~ clip(N(20, 6), 4, 32), conformer sigma 0.5 A, 10 % on the model's nodes at 0.7 A); "
                            "5.5 x the tree search per ligand of the default library",
                }
                slib.close()
                del soff, sdata
            except Exception as e:  # never lose the bench line over the side measurement
                out["survey_library"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
