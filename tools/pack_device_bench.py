#!/usr/bin/env python3
"""The device packer alone: `pmx_pack_features_device` on a tiled batch of the bench generator's molecules, already resident.

    python tools/pack_device_bench.py [--molecules 500000] [--reps 5] [--check]

Prints molecules/s (wall clock around the call, which ends with the record writer enqueued; synchronised). `--check` compares the library
with the host packer's, byte for byte. Under `rocprofv3 --kernel-trace --stats` the graph builder and the record writer show up apart."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=500_000)
    ap.add_argument("--topologies", type=int, default=4096)
    ap.add_argument("--conformers", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch

    import bench
    from pharmaconet_amd import engine
    from pharmaconet_amd.library import flatten_features, pack_features_native
    from tools.synthetic import synthetic_library

    mols = []
    synthetic_library(args.topologies, num_conformers=args.conformers, seed=20240611, molecules_out=mols)
    flat = bench.tile_features(flatten_features(mols), max(1, args.molecules // len(mols)), np.random.default_rng(1))
    n = len(flat["atom_off"]) - 1
    dev = engine.features_to_device(flat)
    bound = engine.pack_bound(flat)
    out = (torch.empty(n + 1, dtype=torch.int64, device="cuda"), torch.empty(bound, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda"))
    engine.pack_features_device(dev, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(args.reps):
        t0 = time.perf_counter()
        offsets, data, status = engine.pack_features_device(dev, out=out)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    feats = int(flat["feat_off"][-1])
    print(f"{n} molecules ({feats / n:.1f} features, {int(flat['atom_off'][-1]) / n:.1f} atoms each), {data.numel() / 1e9:.3f} GB packed: "
          f"{best * 1e3:.2f} ms, {n / best / 1e6:.2f}e6 molecules/s; not packed: {int((status != 0).sum())}")
    if args.check:
        want, ws = pack_features_native(flat, threads=32)
        ok = np.array_equal(offsets.cpu().numpy().astype(np.uint64), want.offsets) and np.array_equal(data.cpu().numpy(), want.data) and np.array_equal(status.cpu().numpy(), ws)
        print("identical to the host packer" if ok else "DIFFERS from the host packer")
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
