"""GPU tests beyond the golden sets: both of the reference's kernel variants, zero weights, the alternative engines, the
structural limits, corrupt libraries, the table arena, the multi-pocket batch (BASELINE.json configs[3]) and the stress
configuration at full size (configs[4])."""

import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, GOLDEN_SETS, REPO, load_golden, rel_err

pytestmark = pytest.mark.gpu

RTOL = 2e-6


def _model_nodes(model):
    from pharmaconet_amd.constants import TYPE_ID

    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]], dtype=np.float64)
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    return centers, types


def _gpu(model, lib, weights=None, **kw):
    from pharmaconet_amd.engine import screen

    res = screen(model, lib, weights=weights, **kw)
    return res.scores.cpu().numpy().astype(np.float64), res.status.cpu().numpy()


def test_within_1e5_of_both_reference_variants(oracle):
    """north_star: scores within 1e-5 relative of the reference's CPU path - NumPy kernels (pinned) and Numba kernels
    (restated, match_utils_numba.py:54-86). Flips (a ligand beyond 1e-6 of either) are counted and must be absent."""
    from pharmaconet_amd.constants import weights_vector
    from tools.synthetic import synthetic_library

    model, _, _, _ = load_golden("set_6oim_c8")
    lib = synthetic_library(600, num_conformers=8, model_nodes=_model_nodes(model), active_fraction=0.3, seed=90210)
    got, status = _gpu(model, lib)
    assert np.all(status == 0)
    for variant in ("numpy", "numba"):
        ref = oracle.oracle_score(model.flat, lib, weights_vector(None), num_threads=os.cpu_count() or 8, variant=variant)
        zero = ref == 0
        assert np.all(got[zero] == 0.0)
        err = rel_err(got[~zero], ref[~zero])
        flips = int((err > 1e-6).sum())
        print(f"GPU vs {variant} variant: max rel err {err.max():.2e}, {flips} of {err.size} ligands beyond 1e-6")
        assert err.max() < 1e-5
        assert flips == 0


def test_zero_type_weight_matches_oracle(oracle):
    """`--hydrophobic 0` (screening.py:33): node pairs whose weights sum to 0 give 0 * (1 / 0) = NaN in the reference
    (match_utils.py:50-52,69), which invalidates the pair entry / poisons the self entry. GPU == oracle."""
    from pharmaconet_amd.constants import weights_vector
    from tools.synthetic import synthetic_library

    model, _, _, _ = load_golden("set_6oim_c8")
    lib = synthetic_library(300, num_conformers=8, model_nodes=_model_nodes(model), active_fraction=0.4, seed=31337)
    for weights in ({"Hydrophobic": 0.0}, {"Aromatic": 0.0, "Halogen": 0.0}):
        ref = oracle.oracle_score(model.flat, lib, weights_vector(weights), num_threads=os.cpu_count() or 8)
        got, status = _gpu(model, lib, weights)
        assert np.all(status == 0) and np.all(np.isfinite(got))
        zero = ref == 0
        assert np.all(got[zero] == 0.0)
        assert rel_err(got[~zero], ref[~zero]).max() < RTOL + 6e-8
        default, _ = _gpu(model, lib)
        assert np.abs(default - got).max() > 0.5  # the weight matters on this library


@pytest.mark.parametrize("env", [{"PMX_TREE_FLAGS": "8"}, {"PMX_TREE_FLAGS": "4"}, {"PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0"},
                                 {"PMX_SLICE_KB": "4"}, {"PMX_SLICE_KB": "4", "PMX_ARENA_MB": "16", "PMX_BUDGET": "64"},
                                 {"PMX_TREE_FLAGS": "128"}, {"PMX_TREE_FLAGS": "1024"}, {"PMX_PATH_KB": "2"}, {"PMX_TREE_FLAGS": "32768"},
                                 {"PMX_TREE_FLAGS": "65536"}, {"PMX_DEAD_MIN_ENTRIES": "1"}, {"PMX_TREE_FLAGS": "131072"}, {"PMX_PAIR_TAILS": "1"},
                                 {"PMX_TASK_DECAY_FROM": "1", "PMX_TASK_BUDGET_MIN": "8", "PMX_BUDGET": "32"}],
                         ids=["exact-terms", "no-bound-test", "tiny-budget", "tiny-slices", "tiny-slices-and-arena", "no-candidate-filter",
                              "no-path-bound", "tiny-path-buffer", "no-chain-lengths", "no-dead-entry-test", "dead-entry-test-everywhere", "no-wide-path-test",
                              "pair-items-honour-rough-cells", "decaying-task-budget"])
@pytest.mark.parametrize("name", GOLDEN_SETS)
def test_engine_settings_match_reference_golden(name, env, monkeypatch):
    """The golden sets under settings that force the rarely taken paths of the engine: Gaussian terms evaluated one by
    one instead of the tabulated pair functions (PMX_TREE_FLAGS=8), every subtree walked (4), almost every tree split
    into queued subtrees, tables that do not fit the per-wavefront slices (arena pass), and an arena too small to hold
    them all at once (carry pass)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model, lib, weights, d = load_golden(name)
    got, status = _gpu(model, lib, weights)
    assert np.all(status == 0)
    ref = d["score"]
    zero = ref == 0
    assert np.all(got[zero] == 0.0)
    assert rel_err(got[~zero], ref[~zero]).max() < RTOL + 6e-8


@pytest.mark.parametrize("conformers", [2, 3, 4, 12, 16, 20, 32, 33, 48])
def test_every_lane_shape_matches_the_oracle(conformers, oracle, monkeypatch):
    """The kernels are instantiated per lane count G = 2^ceil(log2(conformers)) (1 ... 64); the golden sets cover 1, 8 and 64. Every
    other shape - and the two models: the 6OIM-like one and the 64-node one, whose trees are large enough for the bound tests,
    the path-aware tests (path_bound() up to 16 lanes, path_bound_wide() at 32 / 64) and the task rounds to matter - against the
    oracle, default and with the trees split almost at once."""
    from pharmaconet_amd.constants import weights_vector
    from tools.synthetic import synthetic_library

    for name, n in (("set_6oim_c8", 160), ("set_s64_c8", 48)):
        model, _, _, _ = load_golden(name)
        lib = synthetic_library(n, num_conformers=conformers, model_nodes=_model_nodes(model), active_fraction=0.5, seed=5000 + conformers)
        ref = oracle.oracle_score(model.flat, lib, weights_vector(None), num_threads=os.cpu_count() or 8)
        zero = ref == 0
        for env in ({}, {"PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0"}):
            with monkeypatch.context() as mp:
                for k, v in env.items():
                    mp.setenv(k, v)
                got, status = _gpu(model, lib)
            assert np.all(status == 0)
            assert np.all(got[zero] == 0.0)
            assert rel_err(got[~zero], ref[~zero]).max() < RTOL, (name, env)
        assert np.count_nonzero(ref) > n // 2


def test_structural_limits_are_explicit():
    """include/pmx.h: at most 256 model nodes / 128 clusters and 64 ligand nodes / clusters / conformers. A model beyond
    them is refused with a message; a ligand beyond them is reported per ligand (status 1, score NaN) and ranks last."""
    from pharmaconet_amd import PharmacophoreModel
    from pharmaconet_amd.library import UNSUPPORTED_RECORD

    model, lib, _, _ = load_golden("set_6oim_c5")
    st = json.loads(json.dumps(model.__getstate__()))
    node = dict(st["nodes"][0])
    for i in range(len(st["nodes"]), 257):
        extra = dict(node, index=i)
        st["nodes"].append(extra)
    with pytest.raises(ValueError, match="at most 256"):
        big = PharmacophoreModel.__new__(PharmacophoreModel)
        big.__setstate__(st)
        _ = big.flat
    recs = [lib.record(0), UNSUPPORTED_RECORD, lib.record(1)]
    res = model.screen(recs, topk=3)
    status = res.status.cpu().numpy()
    scores = res.scores.cpu().numpy()
    assert status.tolist() == [0, 1, 0] and np.isnan(scores[1]) and np.isfinite(scores[[0, 2]]).all()
    assert [i for i, _ in res.ranking()][-1] == 1


def test_models_beyond_64_nodes_and_clusters(oracle):
    """Rounds 1-3 refused models of more than 64 nodes or clusters (64-bit node and candidate sets); the reference has no limit
    (pharmacophore_model.py:191-204). Node subsets are lists on the device now and a ligand cluster's candidates two words: the
    110-node / 86-cluster fixture model (minted by tests/golden/make_golden_large.py with the reference's own scores, also one of
    GOLDEN_SETS) against a fresh library and the oracle, default and term-by-term table phase. What stays limited is the number
    of candidates of ONE ligand cluster (64, a tree level's width): a model in which every cluster shares a type makes ligands
    that meet it unsupported - reported per ligand, the others scored as the oracle scores them."""
    import copy

    from pharmaconet_amd import PharmacophoreModel
    from pharmaconet_amd.constants import weights_vector
    from tools.synthetic import synthetic_library

    model = PharmacophoreModel.load(GOLDEN / "model_large110.pm")
    assert model.flat.num_nodes > 64 and model.flat.num_clusters > 64 and model.flat.cluster_nodes.shape[1] == 2
    lib = synthetic_library(400, num_conformers=8, model_nodes=_model_nodes(model), active_fraction=0.4, seed=110110)
    ref = oracle.oracle_score(model.flat, lib, weights_vector(None), num_threads=os.cpu_count() or 8)
    got, status = _gpu(model, lib)
    assert np.all(status == 0)
    zero = ref == 0
    assert np.all(got[zero] == 0.0) and np.count_nonzero(ref) > 200
    assert rel_err(got[~zero], ref[~zero]).max() < RTOL
    import os as _os
    _os.environ["PMX_TREE_FLAGS"] = "8"
    try:
        exact, _ = _gpu(model, lib)
    finally:
        del _os.environ["PMX_TREE_FLAGS"]
    assert rel_err(exact[~zero], ref[~zero]).max() < RTOL and np.all(exact[zero] == 0.0)

    st = copy.deepcopy(model.__getstate__())
    for clusters in st["node_cluster_dict"].values():
        for cl in clusters:
            cl["node_types"] = sorted(set(cl["node_types"]) | {"Hydrophobic"})
    wide = PharmacophoreModel.__new__(PharmacophoreModel)
    wide.__setstate__(st)
    ref = oracle.oracle_score(wide.flat, lib, weights_vector(None), num_threads=os.cpu_count() or 8)
    got, status = _gpu(wide, lib)
    hyd = np.array([bool(np.any(lib.unpack(i)["typemask"] & 1)) for i in range(len(lib))])  # type id 0 = Hydrophobic
    assert hyd.sum() > 100
    assert np.all(status[hyd] == 1) and np.all(np.isnan(got[hyd]))
    assert np.all(status[~hyd] == 0)
    assert rel_err(got[~hyd], ref[~hyd])[ref[~hyd] != 0].max(initial=0.0) < RTOL and np.all(got[~hyd][ref[~hyd] == 0] == 0.0)


def test_corrupt_library_is_neutralised_not_followed():
    """pmx_library_upload validates every record on the device: a record whose header does not fit its byte range, whose
    cluster ends run backwards or whose type mask has bit 7 set is scored as unsupported (NaN) instead of being read."""
    from pharmaconet_amd import PackedLibrary
    from pharmaconet_amd.engine import DeviceLibrary

    model, lib, _, d = load_golden("set_6oim_c5")
    recs = [bytearray(lib.record(i)) for i in range(6)]
    recs[1][2:4] = (60).to_bytes(2, "little")  # claims 60 conformers: xyz would run past the record
    recs[2][8] |= 0x80                         # type mask with bit 7
    n2, _, k2 = lib.header(3)
    if k2 >= 2:
        recs[3][8 + n2] = 200                  # first cluster end beyond n_nodes
    bad = PackedLibrary.from_records([bytes(r) for r in recs])
    dev = DeviceLibrary(bad)
    assert dev.num_unsupported >= 2
    scores = model.screen(dev).scores.cpu().numpy()
    assert np.isnan(scores[1]) and np.isnan(scores[2])
    assert abs(scores[0] - d["score"][0]) <= RTOL * abs(d["score"][0]) + 1e-30
    assert abs(scores[5] - d["score"][5]) <= RTOL * abs(d["score"][5]) + 1e-30


def test_small_arena_changes_nothing_but_time(monkeypatch):
    """The arena holds the tables of ligands whose tree is split (and of ligands too large for any slice). When it is full
    a tree that runs over its budget is walked by its own wavefront to the end instead of being split: same bits. A ligand
    whose tables fit neither a slice nor the arena at all is reported, not mis-scored."""
    import torch

    from pharmaconet_amd import engine
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import expand_library_on_device, synthetic_library

    model, _, _, _ = load_golden("set_6oim_c8")
    base = synthetic_library(256, num_conformers=8, model_nodes=_model_nodes(model), conformer_noise=0.0, seed=777)
    offsets, data = expand_library_on_device(base, 64, "cuda")  # 16,384 ligands
    lib = DeviceLibrary.from_device_buffers(offsets, data)
    want = model.screen(lib).scores
    engine.release_workspaces()  # the arena is sized when a workspace is created
    try:
        with monkeypatch.context() as mp:
            mp.setenv("PMX_ARENA_MB", "16")
            mp.setenv("PMX_BUDGET", "64")
            got = model.screen(lib).scores
            stats = engine.last_score_stats()
        assert stats["arena_bytes"] > 16 << 20  # more was asked for than there is: otherwise this test shows nothing
        assert torch.equal(got, want)
        engine.release_workspaces()
        with monkeypatch.context() as mp:  # tables that fit nowhere: slices of 4 KB, no large slices to speak of, arena of 16 MB... still fit
            mp.setenv("PMX_SLICE_KB", "4")
            mp.setenv("PMX_BIG_SLICE_MB", "1")
            mp.setenv("PMX_ARENA_MB", "16")
            res = model.screen(lib)
        st = res.status.cpu().numpy()
        sc = res.scores.cpu().numpy()
        ok = st == 0
        assert np.array_equal(sc[ok], want.cpu().numpy()[ok])   # whatever was scored is right
        assert np.all(np.isnan(sc[~ok])) and np.all(st[~ok] == 2)  # the rest says PMX_LIGAND_TOO_LARGE
    finally:
        engine.release_workspaces()


def test_arena_pass_is_retried_with_the_arena_empty(monkeypatch):
    """a ligand whose tables need the arena (larger than any slice) must not be reported as too large because the
    arena happened to be full of other ligands' tables - it is a bump allocator - when its turn came. Four ligands of the
    64-conformer stress set have tables of 1.9 - 2.7 MB; with large slices of 1 MB they go to the arena, and an arena of 6 MB
    holds at most two of them next to the tables of the over-budget trees. Every ligand is scored, to the reference's floats."""
    from pharmaconet_amd import engine

    model, lib, weights, d = load_golden("set_s64_c64")
    engine.release_workspaces()
    try:
        with monkeypatch.context() as mp:
            mp.setenv("PMX_SLICE_KB", "64")
            mp.setenv("PMX_BIG_SLICE_MB", "1")
            mp.setenv("PMX_ARENA_MB", "6")
            got, status = _gpu(model, lib, weights)
        assert np.all(status == 0), status
        assert rel_err(got, d["score"]).max() < RTOL
        engine.release_workspaces()
        with monkeypatch.context() as mp:  # an arena smaller than the largest table: that ligand, and only that kind, is reported
            mp.setenv("PMX_SLICE_KB", "64")
            mp.setenv("PMX_BIG_SLICE_MB", "1")
            mp.setenv("PMX_ARENA_MB", "2")
            got, status = _gpu(model, lib, weights)
        ok = status == 0
        assert 0 < (~ok).sum() <= 4 and np.all(status[~ok] == 2) and np.all(np.isnan(got[~ok]))
        assert rel_err(got[ok], d["score"][ok]).max() < RTOL
    finally:
        engine.release_workspaces()


def test_sixteen_pockets_one_library():
    """BASELINE.json configs[3]: a batch of 16 distinct pockets against one shared library through pmx_score_multi;
    every (pocket, ligand) score against the reference's own (fixture minted by tests/golden/make_golden_pockets.py)."""
    import ctypes

    import torch

    from pharmaconet_amd import PackedLibrary, PharmacophoreModel, _ffi
    from pharmaconet_amd.constants import weights_vector
    from pharmaconet_amd.engine import DeviceLibrary, device_model

    d = np.load(GOLDEN / "pockets16.npz")
    lib = PackedLibrary.load(GOLDEN / "pockets16.pmxlib")
    models = [PharmacophoreModel.load(GOLDEN / "pockets16" / f"model_{k:02d}.pm") for k in range(16)]
    dev = DeviceLibrary(lib)
    handles = (ctypes.c_void_p * 16)(*[device_model(m).handle for m in models])
    out = torch.empty(16 * len(lib), dtype=torch.float32, device="cuda")
    status = torch.empty(len(lib), dtype=torch.int32, device="cuda")
    w = (ctypes.c_float * 7)(*weights_vector(None))
    _ffi.check(_ffi.load().pmx_score_multi(handles, 16, dev.handle, w, 0, len(lib), out.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(16, -1).astype(np.float64)
    assert np.all(status.cpu().numpy() == 0)
    ref = d["score"]
    zero = ref == 0
    assert np.all(got[zero] == 0.0)
    assert rel_err(got[~zero], ref[~zero]).max() < RTOL + 6e-8
    assert np.count_nonzero(ref) > ref.size // 2


def test_dead_entry_test_changes_no_score(monkeypatch):
    """The table phase settles a pair entry as -1 without computing its items when more than half of its node pairs lie outside
    every 2-sigma window of the two model clusters for every conformer (build_tables, DevModel::cwin) - which is what the items
    would have given (match_utils.py:55-61, :71-74). Sixteen pockets x 3 000 ligands (8 conformers) and the 64-node model x 64
    conformers: the same bits with the test off, at its default and on every level pair; and it does settle entries."""
    import torch

    from pharmaconet_amd import PharmacophoreModel, engine
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import synthetic_library

    model8, _, _, _ = load_golden("set_6oim_c8")
    lib8 = DeviceLibrary(synthetic_library(3000, num_conformers=8, model_nodes=_model_nodes(model8), active_fraction=0.3, seed=777))
    model64, _, _, _ = load_golden("set_s64_c64")
    lib64 = DeviceLibrary(synthetic_library(300, num_conformers=64, model_nodes=_model_nodes(model64), active_fraction=0.3, seed=778,
                                            conformer_noise=0.0))
    cases = [(PharmacophoreModel.load(GOLDEN / "pockets16" / f"model_{k:02d}.pm"), lib8) for k in range(16)] + [(model64, lib64)]
    for model, lib in cases:
        got, dead, items = {}, {}, {}
        for tag, env in (("off", {"PMX_TREE_FLAGS": "65536"}), ("default", {}), ("everywhere", {"PMX_DEAD_MIN_ENTRIES": "1"})):
            with monkeypatch.context() as mp:
                for k, v in env.items():
                    mp.setenv(k, v)
                got[tag] = model.screen(lib).scores
                st = engine.last_score_stats()
                dead[tag], items[tag] = st["n_dead_entries"], st["n_items"]
        assert torch.equal(got["off"], got["default"]) and torch.equal(got["off"], got["everywhere"])
        assert dead["off"] == 0 and dead["everywhere"] >= dead["default"] > 0
        assert items["everywhere"] <= items["default"] < items["off"]


def test_sixteen_pockets_at_shard_size(oracle):
    """BASELINE.json configs[3] at the size one GPU of eight holds: 16 distinct pockets x a 1.25 M-ligand shard of the shared
    library (10 M / 8) through pmx_score_multi - 20 M (pocket, ligand) scores. Size-independent properties (finite,
    non-negative, every pocket's row equal to scoring that pocket alone on a window of the library) and a sample of every
    pocket's scores against the CPU oracle."""
    import ctypes

    import torch

    from pharmaconet_amd import PackedLibrary, PharmacophoreModel, _ffi
    from pharmaconet_amd.constants import weights_vector
    from pharmaconet_amd.engine import DeviceLibrary, device_model
    from tools.synthetic import BASE_SEED, expand_library_on_device, synthetic_library

    model6, _, _, _ = load_golden("set_6oim_c8")
    base = synthetic_library(4096, num_conformers=8, model_nodes=_model_nodes(model6), active_fraction=0.1, seed=BASE_SEED, max_nodes=32,
                             conformer_noise=0.0)
    offsets, data = expand_library_on_device(base, 306, "cuda", seed=BASE_SEED + 3)
    lib = DeviceLibrary.from_device_buffers(offsets, data)
    n = len(lib)
    assert n == 1_253_376
    models = [PharmacophoreModel.load(GOLDEN / "pockets16" / f"model_{k:02d}.pm") for k in range(16)]
    handles = (ctypes.c_void_p * 16)(*[device_model(m).handle for m in models])
    out = torch.empty(16 * n, dtype=torch.float32, device="cuda")
    status = torch.empty(n, dtype=torch.int32, device="cuda")
    w = (ctypes.c_float * 7)(*weights_vector(None))
    _ffi.check(_ffi.load().pmx_score_multi(handles, 16, lib.handle, w, 0, n, out.data_ptr(), status.data_ptr(), None))
    torch.cuda.synchronize()
    scores = out.view(16, n)
    assert torch.isfinite(scores).all() and (scores >= 0).all() and (status == 0).all()
    off = offsets.cpu().numpy()
    dat = data.cpu().numpy()
    rng = np.random.default_rng(316)
    worst = 0.0
    for k, m in enumerate(models):
        first = int(rng.integers(0, n - 20_000))
        alone = m.screen(lib, first=first, count=20_000).scores
        assert torch.equal(alone, scores[k, first:first + 20_000]), f"pocket {k}"
        pick = np.sort(rng.choice(n, size=48, replace=False))
        sample = PackedLibrary.from_records([dat[off[i]:off[i + 1]].tobytes() for i in pick])
        ref = oracle.oracle_score(m.flat, sample, weights_vector(None), num_threads=os.cpu_count() or 8)
        got = scores[k, torch.from_numpy(pick).cuda()].cpu().numpy().astype(np.float64)
        zero = ref == 0
        assert np.all(got[zero] == 0.0)
        if (~zero).any():
            worst = max(worst, rel_err(got[~zero], ref[~zero]).max())
    assert worst < RTOL + 6e-8
    print(f"16 pockets x {n} ligands: max rel err vs oracle on 16 x 48 samples {worst:.2e}")


def test_stress_config_at_full_size(oracle, monkeypatch):
    """BASELINE.json configs[4] at its stated size: the 64-node model, 100 352 ligands x 64 conformers. Size-independent
    properties (finite, non-negative, chunk-invariant, reproducible) and a 512-ligand sample against the CPU oracle; then
    the fp32 vs fp16-coordinate sweep on 4 096 ligands with median / p95 / max error and the flip count written to
    profiles/ (PMX_WRITE_PROFILES=1) or a temporary file."""
    import tempfile

    import torch

    from pharmaconet_amd import PackedLibrary
    from pharmaconet_amd.constants import weights_vector
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import expand_library_on_device, synthetic_library

    model, _, _, _ = load_golden("set_s64_c64")
    assert model.flat.num_nodes == 64
    base = synthetic_library(512, num_conformers=64, model_nodes=_model_nodes(model), active_fraction=0.2, seed=6464,
                             max_nodes=32, conformer_noise=0.0)
    offsets, data = expand_library_on_device(base, 196, "cuda", seed=6465)
    lib = DeviceLibrary.from_device_buffers(offsets, data)
    assert len(lib) == 100_352 and lib.max_conformers == 64
    full = model.screen(lib).scores
    assert torch.isfinite(full).all() and (full >= 0).all()
    checksum = full.double().sum().item()
    with monkeypatch.context() as mp:
        mp.setenv("PMX_SUPER", "17000")
        again = model.screen(lib).scores
    assert torch.equal(again, full) and again.double().sum().item() == checksum
    off = offsets.cpu().numpy()
    dat = data.cpu().numpy()
    rng = np.random.default_rng(64)
    pick = np.sort(rng.choice(len(lib), size=512, replace=False))
    sample = PackedLibrary.from_records([dat[off[i] : off[i + 1]].tobytes() for i in pick])
    ref = oracle.oracle_score(model.flat, sample, weights_vector(None), num_threads=os.cpu_count() or 8)
    got = full[torch.from_numpy(pick).cuda()].cpu().numpy().astype(np.float64)
    zero = ref == 0
    assert np.all(got[zero] == 0.0)
    assert rel_err(got[~zero], ref[~zero]).max() < RTOL + 6e-8
    # fp16 coordinates (centred per ligand before rounding), everything else unchanged
    n_sweep = 4096
    recs = []
    for i in range(n_sweep):
        rec = bytearray(dat[off[i] : off[i + 1]].tobytes())
        n, c, k = int.from_bytes(rec[0:2], "little"), int.from_bytes(rec[2:4], "little"), int.from_bytes(rec[4:6], "little")
        o = (8 + n + k + 3) & ~3
        xyz = np.frombuffer(bytes(rec[o : o + 12 * n * c]), dtype=np.float32).reshape(n, 3, c)
        center = xyz.mean(axis=(0, 2), keepdims=True)
        q = ((xyz - center).astype(np.float16).astype(np.float32) + center).astype(np.float32)
        rec[o : o + 12 * n * c] = np.ascontiguousarray(q).tobytes()
        recs.append(bytes(rec))
    half = model.screen(PackedLibrary.from_records(recs)).scores.cpu().numpy().astype(np.float64)
    f32 = full[:n_sweep].cpu().numpy().astype(np.float64)
    nz = f32 > 0
    err = rel_err(half[nz], f32[nz])
    report = {
        "config": "BASELINE.json configs[4]: model_stress64 (64 nodes), 64 conformers per ligand",
        "ligands_scored_fp32": len(lib), "ligands_in_sweep": int(nz.sum()),
        "coordinates": "centred per ligand, rounded to fp16, converted back to fp32; all arithmetic unchanged",
        "rel_err_median": float(np.median(err)), "rel_err_p95": float(np.percentile(err, 95)), "rel_err_max": float(err.max()),
        "flips_beyond_1e-3": int((err > 1e-3).sum()), "flips_beyond_1e-2": int((err > 1e-2).sum()),
        "zero_vs_nonzero_changes": int(((half > 0) != (f32 > 0)).sum()),
    }
    # PMX_WRITE_PROFILES names the directory the report goes to (gpurun_out/ on the GPU box, copied to profiles/ afterwards)
    target = os.path.join(os.environ["PMX_WRITE_PROFILES"], "r4_fp16_sweep.json") if os.environ.get("PMX_WRITE_PROFILES") else tempfile.mktemp(suffix=".json")
    with open(target, "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    assert report["rel_err_median"] < 1e-3 and report["flips_beyond_1e-2"] <= n_sweep // 50


def test_topk_exchange_through_the_c_abi_single_rank():
    """pmx_comm_* / pmx_topk_allgather (RCCL) with one rank: the all-gather is a copy, the merge is the ranking rule of
    screening.py:70. (Two-rank behaviour of the merge is covered on CPU by tests/test_distributed.py with gloo.)"""
    import torch

    from pharmaconet_amd.distributed import TopkExchange, merge_topk

    model, lib, _, _ = load_golden("set_c21_c8")
    k = 40
    res = model.screen(lib, topk=k, index_base=5000)
    ex = TopkExchange("cuda:0")
    assert (ex.rccl_rank, ex.rccl_ranks, ex.rccl_device) == (0, 1, 0)  # (pmx_comm_info: as RCCL reports them)
    top_s, top_i = ex.allgather(res.topk_scores, res.topk_indices, k)
    torch.cuda.synchronize()
    want_s, want_i = merge_topk(res.topk_scores.cpu().numpy(), res.topk_indices.cpu().numpy(), k)
    assert top_i.cpu().numpy().tolist() == want_i.tolist()
    np.testing.assert_array_equal(top_s.cpu().numpy(), want_s)
    ex.close()


def test_release_workspaces_and_score_again():
    """pmx_release_workspaces frees what libpmx caches between calls; scoring afterwards rebuilds it and gives the same bits."""
    import torch

    from pharmaconet_amd import engine

    model, lib, weights, d = load_golden("set_6oim_c5")
    a = model.screen(lib, weights=weights, topk=5)
    engine.release_workspaces()
    b = model.screen(lib, weights=weights, topk=5)
    assert torch.equal(a.scores, b.scores) and torch.equal(a.topk_indices, b.topk_indices)


def _exchange_rank(rank, world, port, k, n_per_rank, out_dir):
    """One rank of test_topk_exchange_over_rccl_on_every_gpu (a spawned process: one per GPU)."""
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    try:
        from pharmaconet_amd.distributed import TopkExchange
        from pharmaconet_amd.engine import topk

        rng = np.random.default_rng(1234 + rank)
        scores = rng.integers(0, 40, size=n_per_rank).astype(np.float32)  # many ties across ranks
        scores[rng.integers(0, n_per_rank, size=5)] = np.nan               # unsupported ligands
        if rank == world - 1:
            scores = scores[: n_per_rank // 2]                             # a short last shard
        t = torch.from_numpy(scores).to(device)
        ls, li = topk(t, k, base_index=rank * n_per_rank)
        ex = TopkExchange(device)
        gs, gi = ex.allgather(ls, li, k)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=scores, gs=gs.cpu().numpy(), gi=gi.cpu().numpy())
        ex.close()
    finally:
        dist.destroy_process_group()


def test_topk_exchange_over_rccl_on_every_gpu(tmp_path):
    """The shipped multi-GPU exchange (`pmx_topk_allgather`: RCCL all-gather + merge on the device) with one process per
    visible GPU (at most 8): every rank ends with the ranking a single-process stable sort of all shards gives - descending
    score, ties by ascending global index, NaN after every real score. On a box with one GPU the communicator has one
    rank: the same calls (communicator from an id, ncclAllGather, merge on the device) with nothing to merge in."""
    import socket

    import torch
    import torch.multiprocessing as mp

    world = max(1, min(torch.cuda.device_count(), 8))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    k, n_per_rank = 500, 20_000
    mp.spawn(_exchange_rank, args=(world, port, k, n_per_rank, str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    all_scores = np.concatenate([np.pad(sh["scores"], (0, 0)) for sh in shards])
    all_index = np.concatenate([r * n_per_rank + np.arange(len(sh["scores"])) for r, sh in enumerate(shards)])
    key = np.where(np.isnan(all_scores), -np.inf, all_scores.astype(np.float64))
    nan_last = np.isnan(all_scores).astype(np.int64)
    order = np.lexsort((all_index, -key, nan_last))[:k]
    for sh in shards:
        assert sh["gi"].tolist() == all_index[order].tolist()
        np.testing.assert_array_equal(sh["gs"], all_scores[order])
