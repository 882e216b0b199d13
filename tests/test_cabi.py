"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pmx.h declares.
No compute calls here (those are the -m gpu tests)."""

import ctypes
import re

import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def libpmx():
    import __graft_entry__ as entry

    entry.build()
    from pharmaconet_amd import _ffi

    return _ffi.load()


def declared_symbols():
    text = (REPO / "include" / "pmx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmx_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(libpmx):
    names = declared_symbols()
    assert {"pmx_model_create", "pmx_library_upload", "pmx_score", "pmx_score_multi", "pmx_topk", "pmx_last_error"} <= set(names)
    for name in names:
        assert hasattr(libpmx, name), f"{name} declared in include/pmx.h but not exported by libpmx.so"


def test_binding_covers_the_header(libpmx):
    from pharmaconet_amd import _ffi

    assert sorted(_ffi.SIGNATURES) == declared_symbols()
    assert libpmx.pmx_version() >= 100


def test_struct_sizes_match_the_header():
    from pharmaconet_amd import _ffi

    assert ctypes.sizeof(_ffi.ModelDesc) == 8 + 7 * 8
    assert ctypes.sizeof(_ffi.LibraryView) == 32
    assert ctypes.sizeof(_ffi.LibraryInfo) == 24 + 16
    assert ctypes.sizeof(_ffi.ScoreStats) == 3 * 8 + 19 * 8 + 8 + 8 * 8 + 16 + 8 + 8  # + n_exact_values, dbg[8], n_path_bounds, n_path_drops, n_dead_entries, arena_capacity


def test_invalid_arguments_return_status_not_crash(libpmx):
    from pharmaconet_amd import _ffi

    assert libpmx.pmx_model_create(None, 0, None) == 1
    assert b"null" in libpmx.pmx_last_error()
    with pytest.raises(_ffi.PmxError):
        _ffi.check(libpmx.pmx_library_upload(None, 0, None))
    assert libpmx.pmx_density_create(None, 0, 0, 0, None) == 1 and b"pmx_density_create" in libpmx.pmx_last_error()
    assert libpmx.pmx_density_labels(None, None) == 1
    assert libpmx.pmx_density_order(None, 1, None, None, None, None) == 1
    assert libpmx.pmx_density_destroy(None) == 0


def test_model_build_without_gpu_uses_the_host_search():
    """`PharmacophoreModel.create(device="auto")` with no GPU visible: the host search, same state (tests/test_model_builder.py holds it to
    the reference's); asking for a device that is not there fails loudly."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    from pharmaconet_amd import PharmacophoreModel, _ffi
    from pharmaconet_amd.model_builder import build_model_state

    m = np.zeros((64, 64, 64), dtype=np.float32)
    m[10:14, 10:14, 10:14] = 0.5
    info = [dict(nci_type="Hydrophobic", hotspot_position=np.zeros(3, np.float32), hotspot_score=1.0, point_map=m)]
    model = PharmacophoreModel.create(None, (0.0, 0.0, 0.0), info)
    assert model.num_nodes == 1
    with pytest.raises(_ffi.PmxError):
        build_model_state(None, (0.0, 0.0, 0.0), info, device=0)


def test_scoring_without_gpu_fails_loudly():
    """No CPU fallback: the product path refuses to score when no GPU is visible."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from conftest import load_golden
    from pharmaconet_amd import _ffi

    model, lib, weights, _ = load_golden("set_6oim_c5")
    with pytest.raises(_ffi.PmxError):
        model.screen(lib)
    with pytest.raises(_ffi.PmxError):
        model._scoring(lib.record(0))
