"""Packed ligand library: the `LigandGraph` restatement (library.cluster_ligand) against records extracted
from the reference's real `LigandGraph` (tests/golden/*_mols.npz -> *.pmxlib), format round trips, edge cases."""

import json

import numpy as np
import pytest

from conftest import GOLDEN
from pharmaconet_amd import PackedLibrary
from pharmaconet_amd.library import ClusteredLigand, LigandFeatures, as_packed_library, pack_clustered_ligand, pack_ligand


def golden_molecules(name):
    d = np.load(GOLDEN / f"{name}_mols.npz")
    topo = json.loads(str(d["topology"]))
    shapes, pos = d["shapes"], d["positions"]
    off = 0
    for i, t in enumerate(topo):
        n = int(np.prod(shapes[i]))
        p = pos[off : off + n].reshape(shapes[i])
        off += n
        feats = [
            (f[0], f[1] if isinstance(f[1], int) else tuple(f[1]), f[2] if isinstance(f[2], int) else tuple(f[2]))
            for f in t["features"]
        ]
        yield LigandFeatures(t["z"], t["nbrs"], feats, p)


@pytest.mark.parametrize("name", ["set_6oim_c8", "set_6oim_c1", "set_6oim_c64", "set_c21_c8", "set_s64_c8"])
def test_packer_matches_reference_ligandgraph(name):
    """Byte-identical records: node merging, dependence, functional groups, hydrophobic flood, cluster
    order inside clusters, priority sort (ligand.py:134-259, graph_match.py:43-60)."""
    lib = PackedLibrary.load(GOLDEN / f"{name}.pmxlib")
    count = 0
    for i, mol in enumerate(golden_molecules(name)):
        assert pack_ligand(mol) == lib.record(i), f"{name}[{i}]"
        count += 1
    assert count == len(lib)


def test_record_layout():
    lib = PackedLibrary.load(GOLDEN / "set_6oim_c5.pmxlib")
    assert np.all(lib.offsets % 16 == 0)
    hdr = lib.headers()
    for i in range(len(lib)):
        u = lib.unpack(i)
        assert (u["n_nodes"], u["n_conf"], u["n_clusters"]) == tuple(hdr[i])
        assert u["n_conf"] == 5
        ends = u["cluster_end"]
        assert np.all(np.diff(np.concatenate([[0], ends])) > 0) and (len(ends) == 0 or ends[-1] == u["n_nodes"])
        assert np.all((u["typemask"] > 0) & (u["typemask"] < 128))
        assert u["xyz"].shape == (u["n_nodes"], 3, 5)


def test_save_load_slice(tmp_path):
    lib = PackedLibrary.load(GOLDEN / "set_6oim_c1.pmxlib")
    lib.save(tmp_path / "a.pmxlib")
    again = PackedLibrary.load(tmp_path / "a.pmxlib")
    np.testing.assert_array_equal(again.offsets, lib.offsets)
    np.testing.assert_array_equal(again.data, lib.data)
    part = lib.slice(3, 5)
    assert len(part) == 5 and part.record(0) == lib.record(3) and part.record(4) == lib.record(7)
    assert as_packed_library([lib.record(0), lib.record(1)]).record(1) == lib.record(1)
    (tmp_path / "bad").write_bytes(b"nope" * 10)
    with pytest.raises(ValueError):
        PackedLibrary.load(tmp_path / "bad")


def test_zero_feature_ligand_packs_to_empty_record():
    mol = LigandFeatures([6, 8], [[1], [0]], [], np.zeros((2, 3, 3), np.float32))
    rec = pack_ligand(mol)
    assert len(rec) == 16 and rec[:8] == bytes([0, 0, 3, 0, 0, 0, 0, 0])


def test_priority_sort_is_stable_and_follows_priority_fn():
    # clusters: Hydrophobic(2 nodes), Aromatic(1), HBond(2), Cation(1), Anion(1), Halogen(1), Hydrophobic(1)
    pos = np.zeros((9, 2, 3), np.float32)
    pos[:, :, 0] = np.arange(9)[:, None]
    cl = ClusteredLigand(
        typemask=np.array([1, 1, 2, 16, 32, 4, 8, 64, 1], np.uint8),
        positions=pos,
        clusters=[[0, 1], [2], [3, 4], [5], [6], [7], [8]],
        cluster_types=["Hydrophobic", "Aromatic", "HBond", "Cation", "Anion", "Halogen", "Hydrophobic"],
        cluster_key_atom=[10, 3, 5, 7, 2, 9, 1],
    )
    u = PackedLibrary.from_records([pack_clustered_ligand(cl)]).unpack(0)
    # group 0 (Aromatic, Cation, Anion; all size 1 -> by subtype), then group 1 by size desc, subtype, atom
    order = [2, 5, 6, 3, 4, 0, 1, 7, 8]
    np.testing.assert_array_equal(u["xyz"][:, 0, 0], np.array(order, np.float32))
    np.testing.assert_array_equal(u["cluster_end"], [1, 2, 3, 5, 7, 8, 9])


def test_limits_are_enforced():
    pos = np.zeros((65, 1, 3), np.float32)
    cl = ClusteredLigand(np.ones(65, np.uint8), pos, [[i] for i in range(65)], ["Hydrophobic"] * 65, list(range(65)))
    with pytest.raises(ValueError):
        pack_clustered_ligand(cl)
    cl = ClusteredLigand(np.ones(1, np.uint8), np.zeros((1, 65, 3), np.float32), [[0]], ["Hydrophobic"], [0])
    with pytest.raises(ValueError):
        pack_clustered_ligand(cl)


@pytest.mark.parametrize("name", ["set_6oim_c8", "set_6oim_c1", "set_6oim_c64", "set_c21_c8", "set_s64_c8"])
def test_native_packer_matches_reference_ligandgraph(name):
    """`pmx_pack_features` (csrc/pmx_pack.cpp, the C ABI's packer) on the same molecules: the whole library byte for byte,
    whatever the number of threads."""
    from pharmaconet_amd.library import pack_features_native

    lib = PackedLibrary.load(GOLDEN / f"{name}.pmxlib")
    mols = list(golden_molecules(name))
    for threads in (1, 4):
        got, status = pack_features_native(mols, threads=threads)
        assert np.all(status == 0)
        np.testing.assert_array_equal(got.offsets, lib.offsets)
        assert got.data.tobytes() == lib.data.tobytes()


def test_native_packer_marks_oversized_molecules():
    from pharmaconet_amd.library import UNSUPPORTED_RECORD, LigandFeatures, pack_features_native

    n = 70
    big = LigandFeatures([17] * n + [6], [[n]] * n + [list(range(n))], [("Halogen", i, i) for i in range(n)], np.zeros((n + 1, 2, 3), np.float32))
    small = LigandFeatures([6, 17], [[1], [0]], [("Halogen", 1, 1)], np.ones((2, 4, 3), np.float32))
    lib, status = pack_features_native([small, big, small], threads=2)
    assert status.tolist() == [0, 1, 0]
    assert lib.record(1) == UNSUPPORTED_RECORD and lib.record(0) == lib.record(2) == pack_ligand(small)


def test_native_packer_sizing_protocol():
    """include/pmx.h: a call without a buffer packs nothing and returns an upper bound; a buffer that turns out too small
    makes the call fail with the exact need in *data_bytes; with room, *data_bytes is what was written."""
    import ctypes

    from pharmaconet_amd import _ffi
    from pharmaconet_amd.library import flatten_features

    mols = list(golden_molecules("set_6oim_c8"))
    want = PackedLibrary.load(GOLDEN / "set_6oim_c8.pmxlib")
    flat = flatten_features(mols)
    lib = _ffi.load_packer()
    n = len(mols)
    batch = _ffi.FeatureBatch(n, *(flat[k].ctypes.data for k in (
        "atom_off", "atomic_num", "nbr_off", "nbr", "feat_off", "feat_type", "feat_flags", "feat_atom_off", "feat_atoms",
        "feat_center_off", "feat_centers", "n_conf", "pos_off", "positions")))
    offsets = np.zeros(n + 1, dtype=np.uint64)
    nbytes = ctypes.c_uint64(0)
    assert lib.pmx_pack_features(ctypes.byref(batch), 2, offsets.ctypes.data, None, 0, ctypes.byref(nbytes), None) == 0
    bound = int(nbytes.value)
    assert bound >= want.data.size
    small = np.zeros(want.data.size - 16, dtype=np.uint8)
    assert lib.pmx_pack_features(ctypes.byref(batch), 2, offsets.ctypes.data, small.ctypes.data, small.size, ctypes.byref(nbytes), None) != 0
    assert int(nbytes.value) == want.data.size  # the exact need
    data = np.zeros(bound, dtype=np.uint8)
    assert lib.pmx_pack_features(ctypes.byref(batch), 2, offsets.ctypes.data, data.ctypes.data, data.size, ctypes.byref(nbytes), None) == 0
    assert int(nbytes.value) == want.data.size
    assert data[: want.data.size].tobytes() == want.data.tobytes()
    np.testing.assert_array_equal(offsets, want.offsets)


def test_native_packer_rejects_malformed_input_per_molecule():
    """pmx_pack_features trusts nothing in the raw arrays: a type id above 6, an atom index outside the molecule, a feature
    without atoms or too few positions make THAT molecule a header-only record with status 2; the rest of the batch is
    packed as usual. Offsets that run backwards fail the call."""
    import ctypes

    from pharmaconet_amd import _ffi
    from pharmaconet_amd.library import flatten_features, pack_ligand

    mols = list(golden_molecules("set_6oim_c8"))[:6]
    good = [bytes(pack_ligand(m)) for m in mols]
    lib = _ffi.load_packer()

    def run(flat):
        n = len(mols)
        batch = _ffi.FeatureBatch(n, *(flat[k].ctypes.data for k in (
            "atom_off", "atomic_num", "nbr_off", "nbr", "feat_off", "feat_type", "feat_flags", "feat_atom_off", "feat_atoms",
            "feat_center_off", "feat_centers", "n_conf", "pos_off", "positions")))
        offsets = np.zeros(n + 1, dtype=np.uint64)
        status = np.zeros(n, dtype=np.int32)
        nbytes = ctypes.c_uint64(0)
        rc = lib.pmx_pack_features(ctypes.byref(batch), 2, offsets.ctypes.data, None, 0, ctypes.byref(nbytes), None)
        if rc != 0:
            return rc, None, None, None
        data = np.zeros(int(nbytes.value), dtype=np.uint8)
        rc = lib.pmx_pack_features(ctypes.byref(batch), 2, offsets.ctypes.data, data.ctypes.data, data.size, ctypes.byref(nbytes), status.ctypes.data)
        return rc, offsets, data, status

    def corrupt(fn):
        flat = {k: np.array(v, copy=True) for k, v in flatten_features(mols).items()}
        fn(flat)
        return flat

    f0 = int(flatten_features(mols)["feat_off"][2])  # first feature of molecule 2
    cases = {
        "type id": lambda f: f["feat_type"].__setitem__(f0, 9),
        "atom index": lambda f: f["feat_atoms"].__setitem__(int(f["feat_atom_off"][f0]), 10_000),
        "negative centre": lambda f: f["feat_centers"].__setitem__(int(f["feat_center_off"][f0]), -1),
        "neighbour index": lambda f: f["nbr"].__setitem__(int(f["nbr_off"][int(f["atom_off"][2])]), 777),
        "no conformers": lambda f: f["n_conf"].__setitem__(2, 0),
    }
    for name, fn in cases.items():
        rc, offsets, data, status = run(corrupt(fn))
        assert rc == 0, name
        assert status.tolist() == [0, 0, 2, 0, 0, 0], name
        for i in (0, 1, 3, 4, 5):
            assert data[int(offsets[i]):int(offsets[i + 1])].tobytes() == good[i], name
        assert int(offsets[3] - offsets[2]) == 16
    rc, _, _, _ = run(corrupt(lambda f: f["feat_off"].__setitem__(3, int(f["feat_off"][2]) - 1)))
    assert rc != 0 and b"backwards" in lib.pmx_last_error()
