"""The library packer on the device (`pmx_pack_features_device`, csrc/pmx_pack_device.hip) against the host packer and the records extracted
from the reference's own `LigandGraph` (tests/golden/*.pmxlib): byte for byte, status for status."""

import ctypes

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from test_library import golden_molecules

pytestmark = pytest.mark.gpu


def device_pack(flat):
    from pharmaconet_amd.engine import pack_features_device

    offsets, data, status = pack_features_device(flat)
    return offsets.cpu().numpy().astype(np.uint64), data.cpu().numpy(), status.cpu().numpy()


@pytest.mark.parametrize("name", ["set_6oim_c8", "set_6oim_c1", "set_6oim_c64", "set_c21_c8", "set_s64_c8"])
def test_device_packer_matches_reference_ligandgraph(name):
    """The 708 fixture molecules: node merging, dependence, functional groups, hydrophobic flood, cluster order, priority sort
    (ligand.py:134-259, graph_match.py:43-60) and the float32 tuple centres (ligand.py:293-301)."""
    from pharmaconet_amd import PackedLibrary
    from pharmaconet_amd.library import flatten_features

    want = PackedLibrary.load(GOLDEN / f"{name}.pmxlib")
    offsets, data, status = device_pack(flatten_features(list(golden_molecules(name))))
    assert np.all(status == 0)
    np.testing.assert_array_equal(offsets, want.offsets)
    assert data.tobytes() == want.data.tobytes()


def test_device_packer_on_a_tiled_synthetic_batch():
    """A hundred thousand molecules of the bench generator's kind (distinct geometry per copy): the whole library equal to the host packer's."""
    import bench
    from pharmaconet_amd.library import flatten_features, pack_features_native
    from tools.synthetic import synthetic_library

    mols = []
    synthetic_library(512, num_conformers=8, seed=5, molecules_out=mols)
    flat = bench.tile_features(flatten_features(mols), 200, np.random.default_rng(7))
    want, want_status = pack_features_native(flat, threads=16)
    offsets, data, status = device_pack(flat)
    np.testing.assert_array_equal(status, want_status)
    np.testing.assert_array_equal(offsets, want.offsets)
    assert data.size == want.data.size and np.array_equal(data, want.data)


def test_device_packer_statuses():
    """1: outside the format's limits (65 nodes; 65 conformers) - 2: malformed (what pmx_pack_features checks per molecule) or a feature graph the
    reference's builder raises on - 3: outside the device builder's scratch. The neighbours of such a molecule are packed as usual."""
    from pharmaconet_amd.library import UNSUPPORTED_RECORD, LigandFeatures, flatten_features, pack_features_native, pack_ligand

    n = 70
    many_nodes = LigandFeatures([17] * n + [6], [[n]] * n + [list(range(n))], [("Halogen", i, i) for i in range(n)], np.zeros((n + 1, 2, 3), np.float32))
    small = LigandFeatures([6, 17], [[1], [0]], [("Halogen", 1, 1)], np.ones((2, 4, 3), np.float32))
    many_conf = LigandFeatures([6, 17], [[1], [0]], [("Halogen", 1, 1)], np.ones((2, 65, 3), np.float32))
    na = 300
    many_atoms = LigandFeatures([6] * na, [[(i + 1) % na, (i - 1) % na] for i in range(na)], [("Halogen", 1, 1)], np.ones((na, 2, 3), np.float32))
    long_feature = LigandFeatures([6] * 20, [[(i + 1) % 20, (i - 1) % 20] for i in range(20)], [("Aromatic", tuple(range(17)), tuple(range(17)))], np.ones((20, 2, 3), np.float32))
    mols = [small, many_nodes, small, many_conf, many_atoms, long_feature, small]
    flat = flatten_features(mols)
    offsets, data, status = device_pack(flat)
    assert status.tolist() == [0, 1, 0, 1, 3, 3, 0]
    host, host_status = pack_features_native(mols[:4] + [small], threads=2)
    assert host_status.tolist() == [0, 1, 0, 1, 0]
    rec = lambda i: data[int(offsets[i]) : int(offsets[i + 1])].tobytes()
    assert rec(0) == rec(2) == rec(6) == pack_ligand(small)
    for i in (1, 3, 4, 5):
        assert rec(i) == UNSUPPORTED_RECORD
    # malformed input, molecule by molecule (tests/test_library.py's cases)
    gm = list(golden_molecules("set_6oim_c8"))[:6]
    good = [bytes(pack_ligand(m)) for m in gm]
    base = flatten_features(gm)
    f0 = int(base["feat_off"][2])
    cases = {
        "type id": lambda f: f["feat_type"].__setitem__(f0, 9),
        "atom index": lambda f: f["feat_atoms"].__setitem__(int(f["feat_atom_off"][f0]), 10_000),
        "negative centre": lambda f: f["feat_centers"].__setitem__(int(f["feat_center_off"][f0]), -1),
        "neighbour index": lambda f: f["nbr"].__setitem__(int(f["nbr_off"][int(f["atom_off"][2])]), 777),
        "no conformers": lambda f: f["n_conf"].__setitem__(2, 0),
        "feature without atoms": lambda f: f["feat_atom_off"].__setitem__(f0 + 1, int(f["feat_atom_off"][f0])),
    }
    for name, fn in cases.items():
        flat = {k: np.array(v, copy=True) for k, v in base.items()}
        fn(flat)
        _, hs = pack_features_native(flat, threads=2)
        assert hs.tolist() == [0, 0, 2, 0, 0, 0], name
        offsets, data, status = device_pack(flat)
        assert status.tolist() == [0, 0, 2, 0, 0, 0], name
        for i in (0, 1, 3, 4, 5):
            assert data[int(offsets[i]) : int(offsets[i + 1])].tobytes() == good[i], name
        assert int(offsets[3] - offsets[2]) == 16 and not data[int(offsets[2]) : int(offsets[3])].any()


def test_device_packer_on_merged_keys_and_dependences():
    """Hand-made feature lists on the corners of __add_nodes: an int key and a 1-tuple key are different nodes, equal tuple keys merge their types,
    an H-bond atom inside an ion group joins the ion's cluster (ligand.py:134-156,303-329), repeated features, unsorted and repeated key atoms."""
    from pharmaconet_amd.library import LigandFeatures, flatten_features, pack_features_native, pack_ligand

    pos = np.arange(5 * 3 * 3, dtype=np.float32).reshape(5, 3, 3) * 0.37
    z = [8, 6, 8, 6, 7]
    nbrs = [[1], [0, 2, 3], [1], [1, 4], [3]]
    lists = [
        [("HBond_acceptor", 0, 0), ("HBond_acceptor", (0,), (0,)), ("Anion", (2, 0, 1), (0, 2)), ("HBond_acceptor", 2, 2), ("Cation", 4, 4), ("HBond_donor", 4, 4)],
        [("Anion", (0, 2, 1), (2, 0)), ("Anion", (0, 2, 1), (2, 0)), ("HBond_acceptor", 0, 0), ("Hydrophobic", 3, 3), ("Hydrophobic", 1, 1), ("Hydrophobic", 3, 3)],
        [("Hydrophobic", 1, 1), ("Aromatic", (3, 1, 1, 0), (0, 1, 3)), ("Hydrophobic", 3, 3), ("Halogen", 2, 2), ("Hydrophobic", 4, 4)],
        [],
    ]
    mols = [LigandFeatures(z, nbrs, feats, pos) for feats in lists]
    want, hs = pack_features_native(mols, threads=1)
    assert [want.record(i) for i in range(len(mols))] == [bytes(pack_ligand(m)) for m in mols]  # (the step-by-step restatement of LigandGraph)
    offsets, data, status = device_pack(flatten_features(mols))
    np.testing.assert_array_equal(status, hs)
    np.testing.assert_array_equal(offsets, want.offsets)
    assert data.tobytes() == want.data.tobytes()


def test_device_packer_sizing_protocol_and_empty_batch():
    """include/pmx.h: data_out = NULL sizes exactly; a buffer that is too small fails the call with the need in *data_bytes; an empty batch is an empty library."""
    import torch

    from pharmaconet_amd import PackedLibrary, _ffi
    from pharmaconet_amd.engine import FEATURE_FIELDS, features_to_device
    from pharmaconet_amd.library import flatten_features

    want = PackedLibrary.load(GOLDEN / "set_6oim_c8.pmxlib")
    flat = features_to_device(flatten_features(list(golden_molecules("set_6oim_c8"))))
    n = len(want)
    lib = _ffi.load()
    batch = _ffi.FeatureBatch(n, *(flat[k].data_ptr() for k in FEATURE_FIELDS))
    offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    nbytes = ctypes.c_uint64(0)
    assert lib.pmx_pack_features_device(ctypes.byref(batch), 0, None, offsets.data_ptr(), None, 0, ctypes.byref(nbytes), None) == 0
    assert int(nbytes.value) == want.data.size
    small = torch.zeros(want.data.size - 16, dtype=torch.uint8, device="cuda")
    assert lib.pmx_pack_features_device(ctypes.byref(batch), 0, None, offsets.data_ptr(), small.data_ptr(), small.numel(), ctypes.byref(nbytes), None) != 0
    assert int(nbytes.value) == want.data.size and b"too small" in lib.pmx_last_error()
    empty = _ffi.FeatureBatch(0, *(flat[k].data_ptr() for k in FEATURE_FIELDS))
    offsets.fill_(7)
    assert lib.pmx_pack_features_device(ctypes.byref(empty), 0, None, offsets.data_ptr(), small.data_ptr(), small.numel(), ctypes.byref(nbytes), None) == 0
    torch.cuda.synchronize()
    assert int(nbytes.value) == 0 and int(offsets[0]) == 0


def test_library_from_features_scores_like_the_uploaded_library():
    """Features -> `DeviceLibrary.from_features` -> screen: the scores of the library uploaded from the host, bit for bit."""
    from pharmaconet_amd.engine import DeviceLibrary
    from pharmaconet_amd.library import flatten_features

    model, lib, weights, d = load_golden("set_6oim_c8")
    dlib = DeviceLibrary.from_features(flatten_features(list(golden_molecules("set_6oim_c8"))))
    assert len(dlib) == len(lib) and dlib.num_bytes == lib.data.size
    got = model.screen(dlib, weights=weights).scores.cpu().numpy()
    want = model.screen(lib, weights=weights).scores.cpu().numpy()
    np.testing.assert_array_equal(got, want)
    dlib.close()


def test_adopted_device_buffers_are_used_in_place():
    """pmx_library_view.on_device = 2: no copy - the library scores from the caller's buffers and leaves them alone when it is destroyed."""
    import torch

    from pharmaconet_amd.engine import DeviceLibrary

    model, lib, weights, d = load_golden("set_c21_c8")
    offsets = torch.from_numpy(lib.offsets.astype(np.int64)).cuda()
    data = torch.from_numpy(np.ascontiguousarray(lib.data)).cuda()
    want = model.screen(lib, weights=weights).scores.cpu().numpy()
    dlib = DeviceLibrary.from_device_buffers(offsets, data, adopt=True)
    np.testing.assert_array_equal(model.screen(dlib, weights=weights).scores.cpu().numpy(), want)
    # in place: another record's bytes under the same offsets change the scores
    first = slice(int(lib.offsets[0]), int(lib.offsets[1]))
    keep = data[first].clone()
    data[first.start + 8 + 64 : first.stop] += 1  # (coordinates of ligand 0, beyond its header / type bytes)
    changed = model.screen(dlib, weights=weights).scores.cpu().numpy()
    assert np.array_equal(changed[1:], want[1:])
    data[first] = keep
    np.testing.assert_array_equal(model.screen(dlib, weights=weights).scores.cpu().numpy(), want)
    dlib.close()
    assert np.array_equal(data.cpu().numpy(), lib.data)  # still the caller's, still there


def test_screening_packs_a_directory_on_the_device(tmp_path, monkeypatch):
    """`screening.load_library(..., on_device=True)` (what `main` calls): files read through the (stand-in) toolkit, perception rules on the batch, records made by
    the device packer - the library the host path makes, resident."""
    import gzip
    import json
    import sys

    import fake_openbabel
    from pharmaconet_amd import screening

    fake_openbabel.install()  # (`import openbabel` resolves to the stand-in for this test only)

    def uninstall():
        for name in ("openbabel", "openbabel.pybel", "openbabel.pybel.ob"):
            sys.modules.pop(name, None)

    with gzip.open(GOLDEN / "perception.json.gz", "rt") as f:
        golden = json.load(f)
    picks = [0, 3, 5, 8, 13, 21, 34]
    for i in picks:
        (tmp_path / f"mol{i:03d}.sdf").write_text(json.dumps(golden["molecules"][i]))

    class _Pool:  # (the stand-in toolkit lives in this process only)
        def __init__(self, n):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def map(self, fn, items):
            return [fn(x) for x in items]

    monkeypatch.setattr(screening.multiprocessing, "Pool", _Pool)
    try:
        names, host = screening.load_library(tmp_path, cpus=2)
        names_dev, dlib = screening.load_library(tmp_path, cpus=2, on_device=True)
    finally:
        uninstall()
    assert names == names_dev and len(dlib) == len(host) and dlib.num_bytes == host.data.size
    offsets, data = dlib._adopted
    assert np.array_equal(offsets.cpu().numpy().astype(np.uint64), host.offsets) and np.array_equal(data.cpu().numpy(), host.data)
    dlib.close()


def test_device_packer_against_the_host_packer_on_random_feature_batches():
    """Differential soak (tools/fuzz_pack_device.py): random bond graphs, random feature lists with repeated keys, tuple keys of up to 18 atoms, up to 80 features
    and 300 atoms per molecule - the wave builder, the general builder and the status-3 exit all take part; every status and every byte equal to the host packer's."""
    from tools import fuzz_pack_device as fz

    for seed in (3, 4):
        n, n3, n_host_bad, bad = fz.compare(fz.random_batch(np.random.default_rng([seed, 0]), 4000))
        assert n == 4000 and not bad, bad[:5]
        assert n3 > 50 and n_host_bad > 10  # (the draw reaches the exits it is meant to reach)


def test_library_from_an_empty_feature_batch():
    """No molecules: an empty resident library that screens to nothing."""
    from pharmaconet_amd.engine import DeviceLibrary
    from pharmaconet_amd.library import flatten_features

    model, lib, weights, d = load_golden("set_6oim_c8")
    dlib = DeviceLibrary.from_features(flatten_features([]))
    assert len(dlib) == 0 and dlib.num_bytes == 0
    res = model.screen(dlib, topk=5)
    assert res.scores.numel() == 0
    dlib.close()


def test_device_packer_calls_on_two_streams_do_not_share_descriptors():
    """The packer's work buffers are shared by all calls: a call on a second stream, made while the first call's record writer may still run, has to leave that one's records alone."""
    import torch

    import bench
    from pharmaconet_amd.engine import features_to_device, pack_bound, pack_features_device
    from pharmaconet_amd.library import flatten_features, pack_features_native
    from tools.synthetic import synthetic_library

    mols = []
    synthetic_library(256, num_conformers=8, seed=5, molecules_out=mols)
    big = bench.tile_features(flatten_features(mols), 400, np.random.default_rng(3))
    small = flatten_features(list(golden_molecules("set_c21_c8")))
    want_big, _ = pack_features_native(big, threads=16)
    want_small, _ = pack_features_native(small, threads=2)
    dev_big, dev_small = features_to_device(big), features_to_device(small)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(streams[0]):
            o1, d1, s1 = pack_features_device(dev_big, bound=pack_bound(big))
        with torch.cuda.stream(streams[1]):
            o2, d2, s2 = pack_features_device(dev_small, bound=pack_bound(small))
        torch.cuda.synchronize()
        assert np.array_equal(d1.cpu().numpy(), want_big.data) and np.array_equal(o1.cpu().numpy().astype(np.uint64), want_big.offsets)
        assert np.array_equal(d2.cpu().numpy(), want_small.data) and np.array_equal(o2.cpu().numpy().astype(np.uint64), want_small.offsets)
