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


@pytest.mark.parametrize("name", CASES)
def test_state_equals_reference(name):
    center, infos = load_inputs(name)
    want = PharmacophoreModel.load(GOLDEN / f"create_{name}.pm").__getstate__()
    got = build_model_state(want["pdbblock"], center, infos)
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
