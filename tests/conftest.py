import json
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


GOLDEN_SETS = (
    "set_6oim_c8",
    "set_6oim_c1",
    "set_6oim_c5",
    "set_6oim_c64",
    "set_6oim_c8_weights",
    "set_c21_c8",
    "set_s64_c8",
    "set_s64_c64",
    "set_l110_c8",  # 110-node / 86-cluster model (tests/golden/make_golden_large.py)
)


def load_golden(name):
    """(model, packed library, weights dict | None, npz of the reference's outputs)."""
    from pharmaconet_amd import PackedLibrary, PharmacophoreModel

    d = np.load(GOLDEN / f"{name}.npz")
    model = PharmacophoreModel.load(GOLDEN / f"{str(d['model'])}.pm")
    lib = PackedLibrary.load(GOLDEN / f"{name}.pmxlib")
    weights = json.loads(str(d["weights"]))
    return model, lib, weights, d


def rel_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o
