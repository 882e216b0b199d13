#!/usr/bin/env python3
"""Differential soak of the two packers: `pmx_pack_features_device` (csrc/pmx_pack_device.hip) against `pmx_pack_features` (csrc/pmx_pack.cpp)
on random feature batches - chemically meaningless on purpose: random bond graphs, random element per atom, random feature lists (type, int
key or tuple key of 1..18 atoms with repeats, random centre lists), repeated keys, molecules of 0..80 features and 1..300 atoms, so that the wave
builder, the general builder and the status-3 exit all get their share.

    python tools/fuzz_pack_device.py [--molecules 20000] [--seed 1] [--rounds 1]

Exit code 0 if statuses, offsets and every byte agree (status 3 of the device = whatever the host says, and is not compared byte-wise)."""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

ELEMENTS = np.array([1, 6, 6, 6, 6, 7, 8, 9, 16, 17])


def random_batch(rng, n_mol):
    atom_off, feat_off, pos_off = [0], [0], [0]
    z, nbr_off, nbr = [], [0], []
    ftype, fflags, fa_off, fa, fc_off, fc = [], [], [0], [], [0], []
    n_conf, pos = [], []
    for _ in range(n_mol):
        shape = rng.random()
        na = int(rng.integers(1, 30)) if shape < 0.9 else int(rng.integers(100, 300)) if shape < 0.95 else int(rng.integers(30, 128))
        zz = rng.choice(ELEMENTS, size=na)
        # a random tree plus a few ring-closing bonds, neighbour lists in insertion order
        adj = [[] for _ in range(na)]
        for a in range(1, na):
            b = int(rng.integers(0, a))
            adj[a].append(b)
            adj[b].append(a)
        for _ in range(int(rng.integers(0, 4))):
            a, b = (int(x) for x in rng.integers(0, na, size=2))
            if a != b and b not in adj[a]:
                adj[a].append(b)
                adj[b].append(a)
        z.extend(int(x) for x in zz)
        for row in adj:
            nbr.extend(row)
            nbr_off.append(len(nbr))
        atom_off.append(atom_off[-1] + na)
        nf = int(rng.integers(0, 24)) if rng.random() < 0.9 else int(rng.integers(24, 80))
        pool = []  # keys seen, for repeats
        for _ in range(nf):
            t = int(rng.integers(0, 7))
            r = rng.random()
            if pool and r < 0.25:
                atoms, a_tuple = pool[int(rng.integers(0, len(pool)))]
            elif r < 0.75:
                atoms, a_tuple = [int(rng.integers(0, na))], bool(rng.random() < 0.15)
            else:
                k = int(rng.integers(2, 8)) if rng.random() < 0.93 else int(rng.integers(8, 19))
                atoms, a_tuple = [int(x) for x in rng.integers(0, na, size=k)], True
            pool.append((atoms, a_tuple))
            if rng.random() < 0.7:
                centers, c_tuple = list(atoms), a_tuple
            else:
                centers, c_tuple = [int(x) for x in rng.integers(0, na, size=int(rng.integers(1, 7)))], True
            if not c_tuple:
                centers = centers[:1]
            ftype.append(t)
            fflags.append((1 if a_tuple else 0) | (2 if c_tuple else 0))
            fa.extend(atoms)
            fa_off.append(len(fa))
            fc.extend(centers)
            fc_off.append(len(fc))
        feat_off.append(len(ftype))
        c = int(rng.integers(1, 10)) if rng.random() < 0.97 else int(rng.integers(60, 70))
        n_conf.append(c)
        p = rng.normal(scale=4.0, size=(na, c, 3)).astype(np.float32)
        pos.append(p.reshape(-1))
        pos_off.append(pos_off[-1] + p.size)
    return dict(
        atom_off=np.array(atom_off, np.uint64), atomic_num=np.array(z, np.uint8), nbr_off=np.array(nbr_off, np.uint64), nbr=np.array(nbr, np.int32),
        feat_off=np.array(feat_off, np.uint64), feat_type=np.array(ftype, np.uint8), feat_flags=np.array(fflags, np.uint8),
        feat_atom_off=np.array(fa_off, np.uint64), feat_atoms=np.array(fa, np.int32), feat_center_off=np.array(fc_off, np.uint64),
        feat_centers=np.array(fc, np.int32), n_conf=np.array(n_conf, np.int32), pos_off=np.array(pos_off, np.uint64), positions=np.concatenate(pos),
    )


def compare(flat):
    """(number of molecules, number with device status 3, list of disagreements)."""
    from pharmaconet_amd.engine import pack_features_device
    from pharmaconet_amd.library import pack_features_native

    want, ws = pack_features_native(flat, threads=8)
    offsets, data, status = pack_features_device(flat)
    offsets, data, status = offsets.cpu().numpy().astype(np.uint64), data.cpu().numpy(), status.cpu().numpy()
    bad = []
    n = len(ws)
    for i in range(n):
        if status[i] == 3:
            if int(offsets[i + 1] - offsets[i]) != 16:
                bad.append((i, "status 3 without a header-only record"))
            continue
        got = data[int(offsets[i]) : int(offsets[i + 1])].tobytes()
        if status[i] != ws[i]:
            bad.append((i, f"status {status[i]} != host {ws[i]}"))
        elif got != want.record(i):
            bad.append((i, "bytes differ"))
    return n, int((status == 3).sum()), int((ws != 0).sum()), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=1)
    args = ap.parse_args()
    worst = 0
    for r in range(args.rounds):
        flat = random_batch(np.random.default_rng([args.seed, r]), args.molecules)
        n, n3, nh, bad = compare(flat)
        nf = np.diff(flat["feat_off"].astype(np.int64))
        print(f"round {r}: {n} molecules (features: mean {nf.mean():.1f}, max {nf.max()}; {int((nf > 64).sum())} beyond 64), device status 3: {n3}, host status != 0: {nh}, disagreements: {len(bad)}")
        for i, why in bad[:10]:
            print("   molecule", i, why)
        worst = max(worst, len(bad))
    sys.exit(1 if worst else 0)


if __name__ == "__main__":
    main()
