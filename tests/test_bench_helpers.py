"""Host-side helpers of bench.py (no GPU): the cores a process is really granted, and the tiling of a feature batch for the end-to-end pipeline leg."""

import os

import numpy as np


def test_host_cores_respects_affinity_and_cgroup_quota():
    import bench

    cores, info = bench.host_cores()
    assert 1 <= cores <= (os.cpu_count() or 1)
    assert info["usable_cores"] == cores and info["affinity"] >= cores
    if info["cgroup_cpu_max"] and info["cgroup_cpu_max"].split()[0] not in ("max", "-1"):
        quota, period = (int(x) for x in info["cgroup_cpu_max"].split())
        if quota > 0:
            assert cores <= max(1, int(quota / period + 0.5))


def test_tile_features_keeps_the_topologies_and_moves_the_geometry():
    import bench
    from pharmaconet_amd.library import flatten_features, pack_features_native
    from tools.synthetic import synthetic_library

    mols = []
    base = synthetic_library(48, num_conformers=8, seed=5, conformer_noise=0.0, molecules_out=mols)
    flat = flatten_features(mols)
    tiled = bench.tile_features(flat, 4, np.random.default_rng(1))
    lib, status = pack_features_native(tiled, threads=2)
    assert len(lib) == 4 * len(base) and status.sum() == 0
    hb = base.headers()
    for r in range(4):
        np.testing.assert_array_equal(lib.headers()[r * len(base) : (r + 1) * len(base)], hb)
    n, _, k = base.header(3)
    head = 8 + n + k
    assert lib.record(3)[:head] == base.record(3)[:head] == lib.record(3 + len(base))[:head]  # same types and clusters ...
    assert lib.record(3) != lib.record(3 + len(base)) != base.record(3)  # ... other coordinates in every copy


def test_reference_rate_file_is_what_bench_quotes():
    import bench

    ref = bench.reference_rate()
    assert ref is not None and ref["kind"] == "reference" and ref["cores"] == 1 and ref["value"] > 0
    for row in ref["sets"].values():
        assert row["ligand_conformers"] == 8 * row["ligands"]
