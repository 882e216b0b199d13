"""`PharmacophoreModel.create` / `model_builder.build_model_state` against the reference's own `create`
(SURVEY.md section 8, row f3). Fixtures: tests/golden/make_golden_create.py ran the reference's
`PharmacophoreModel.create` (`pharmacophore_model.py:108-149`, `utils/density_map.py`) on synthetic hotspot
density maps (several components per map, touching blobs, specks under the 8-voxel floor) and saved its state."""

from pathlib import Path

import numpy as np
import pytest

from pharmaconet_amd import PharmacophoreModel
from pharmaconet_amd.model_builder import build_model_state, voxel_components

GOLDEN = Path(__file__).parent / "golden"
CASES = ["mixed24", "charged40", "crowded48"]


def load_inputs(name):
    z = np.load(GOLDEN / f"create_{name}_inputs.npz")
    infos = []
    for h in range(int(z["n"])):
        m = np.zeros((64, 64, 64), dtype=np.float32)
        idx = z[f"vox{h}"].astype(np.int64)
        m[tuple(idx.T)] = z[f"val{h}"]
        infos.append(dict(nci_type=str(z["kinds"][h]), hotspot_position=z["positions"][h], hotspot_score=float(z["scores"][h]),
                          point_map=m))
    return z["center"], infos


def _assert_state_equals_reference(name, device):
    center, infos = load_inputs(name)
    want = PharmacophoreModel.load(GOLDEN / f"create_{name}.pm").__getstate__()
    got = build_model_state(want["pdbblock"], center, infos, device=device)
    assert got["pdbblock"] == want["pdbblock"]
    # nodes: numbering (component order), float32 centres, radii, edge maps and overlap lists - exact
    assert len(got["nodes"]) == len(want["nodes"])
    for g, w in zip(got["nodes"], want["nodes"]):
        assert g == w, (g["index"], g, w)
    # edges: index, end points, types, float mean / std - exact
    assert got["edges"] == want["edges"]
    assert got["node_dict"] == want["node_dict"]
    # clusters: kind order, member tuples (CPython set order), float32-mean centre and size - exact;
    # node_types is a tuple built from a set of strings in the reference (no defined order): compare as sets
    assert list(got["node_cluster_dict"]) == list(want["node_cluster_dict"])
    for kind in want["node_cluster_dict"]:
        gl, wl = got["node_cluster_dict"][kind], want["node_cluster_dict"][kind]
        assert len(gl) == len(wl), kind
        for g, w in zip(gl, wl):
            assert set(g.pop("node_types")) == set(w.pop("node_types"))
            assert g == w, (kind, g, w)


@pytest.mark.parametrize("name", CASES)
def test_state_equals_reference(name):
    _assert_state_equals_reference(name, None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_state_equals_reference_with_the_searches_on_the_gpu(name):
    """the voxel searches of `DensityMapGraph` on the device (`csrc/pmx_density.hip`: component labels +
    breadth-first member order per component), the seeds from CPython's own `set.pop()`: the same state as the reference's."""
    _assert_state_equals_reference(name, 0)


@pytest.mark.gpu
def test_device_member_lists_equal_the_host_search():
    """Every component of every map, voxel for voxel in discovery order, device against the host search (`voxel_components`, itself
    held to the reference by test_state_equals_reference): random blobs that touch through faces, edges and corners, specks, a
    solid 20^3 cube (8 000 voxels: wide frontiers), a thin diagonal line, voxels on the grid's faces, an empty map."""
    from pharmaconet_amd.model_builder import voxel_components_device

    rng = np.random.default_rng(20250523)
    masks = []
    for k in range(6):
        m = np.zeros((64, 64, 64), dtype=np.float32)
        for _ in range(12 + 6 * k):
            c = rng.integers(2, 62, size=3)
            r = rng.uniform(1.0, 5.5)
            g = np.stack(np.meshgrid(*[np.arange(64)] * 3, indexing="ij"), -1)
            inside = ((g - c) ** 2).sum(-1) <= r * r
            m[inside] = rng.uniform(0.05, 1.0, size=int(inside.sum())).astype(np.float32)
        m[rng.random(m.shape) < 0.0005] = 0.3  # specks
        masks.append(m)
    cube = np.zeros((64, 64, 64), dtype=np.float32)
    cube[5:25, 30:50, 40:60] = rng.uniform(0.1, 1.0, size=(20, 20, 20)).astype(np.float32)
    for i in range(40):
        cube[20 + i, 5 + i // 2, 63 - i] = 0.5  # a line that moves through corners, ending on a face
    cube[0, 0, 0] = cube[63, 63, 63] = cube[0, 63, 0] = 0.2
    masks.append(cube)
    masks.append(np.zeros((64, 64, 64), dtype=np.float32))
    got = voxel_components_device(masks, 0)
    n_comp = 0
    for m, comps in zip(masks, got):
        want = list(voxel_components(m))
        assert len(comps) == len(want)
        for (gm, gv), (wm, wv) in zip(comps, want):
            assert gm.tolist() == [list(p) for p in wm]
            assert gv.tolist() == wv
        n_comp += len(want)
    assert n_comp > 100 and max(len(w[0]) for w in voxel_components(cube)) >= 8000


def test_create_gives_a_scorable_model():
    center, infos = load_inputs("mixed24")
    made = PharmacophoreModel.create("X", center, infos)
    ref = PharmacophoreModel.load(GOLDEN / "create_mixed24.pm")
    a, b = made.flat, ref.flat
    assert a.num_nodes == b.num_nodes and a.num_clusters == b.num_clusters
    np.testing.assert_array_equal(a.edge_mean, b.edge_mean)
    np.testing.assert_array_equal(a.edge_std, b.edge_std)
    np.testing.assert_array_equal(a.cluster_nodes, b.cluster_nodes)
    np.testing.assert_array_equal(a.cluster_center, b.cluster_center)
    np.testing.assert_array_equal(a.cluster_size, b.cluster_size)


def test_components_are_26_connected_and_small_ones_are_dropped():
    m = np.zeros((64, 64, 64), dtype=np.float32)
    m[10:13, 10:13, 10:13] = 0.7          # 27 voxels
    m[13, 13, 13] = 0.6                    # touches the cube only through a corner: same component
    m[30, 30, 30] = m[31, 31, 30] = 0.9    # 2 voxels: a component, but below the floor of 8
    comps = sorted((len(v), sorted(v)[0]) for v, _ in voxel_components(m))
    assert comps == [(2, (30, 30, 30)), (28, (10, 10, 10))]
    st = build_model_state(None, (0.0, 0.0, 0.0), [dict(nci_type="Hydrophobic", hotspot_position=np.zeros(3, np.float32),
                                                      hotspot_score=1.0, point_map=m)])
    assert len(st["nodes"]) == 1 and len(st["edges"]) == 1
    assert st["nodes"][0]["overlapped_nodes"] == [0, 0]  # the self loop registers the node twice (density_map.py:247-249)
    assert [len(v) for v in st["node_cluster_dict"].values()] == [0, 0, 0, 0, 1, 0]


@pytest.mark.gpu
def test_device_search_checks_itself_against_the_interpreter():
    """`device_search_agrees`: the one-time comparison of the device search with the reference's own loop on a synthetic map passes
    on this interpreter (and is what `build_model_state` consults before it trusts the device with a model's state)."""
    from pharmaconet_amd import model_builder

    model_builder._DEVICE_SEARCH_OK.clear()
    assert model_builder.device_search_agrees(0) is True
    assert model_builder._DEVICE_SEARCH_OK == {0: True}
