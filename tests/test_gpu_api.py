"""GPU tests of the host API around the kernels: top-k order, ranges, the reference-named entry points,
several models over one library, edge cases, and order/partition invariance on a large synthetic library."""

import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def stable_desc(scores, k):
    return sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)[:k]


def test_topk_matches_stable_descending_sort():
    """screening.py:70 on the device: descending, ties in library order."""
    import torch

    from pharmaconet_amd.engine import topk

    rng = np.random.default_rng(3)
    s = rng.integers(0, 50, size=10_000).astype(np.float32)
    s[5] = np.nan  # unsupported ligand: ranks last
    t = torch.from_numpy(s).cuda()
    ts, ti = topk(t, 300, base_index=1000)
    clean = np.where(np.isnan(s), -np.inf, s)
    want = stable_desc(clean, 300)
    assert (ti.cpu().numpy() - 1000).tolist() == want
    np.testing.assert_array_equal(ts.cpu().numpy(), clean[want])
    # k larger than n pads with -inf / -1
    ts, ti = topk(t[:10], 16)
    assert ti.cpu().numpy()[10:].tolist() == [-1] * 6 and np.all(np.isinf(ts.cpu().numpy()[10:]))


def test_screen_ranges_and_topk():
    model, lib, weights, d = load_golden("set_c21_c8")
    full = model.screen(lib, topk=20)
    got = full.scores.cpu().numpy()
    part = model.screen(lib, first=37, count=50, topk=5, index_base=1_000_000)
    np.testing.assert_array_equal(part.scores.cpu().numpy(), got[37:87])
    want = stable_desc(got, 20)
    assert [i for i, _ in full.ranking()] == want
    assert [i for i, _ in part.ranking()] == [1_000_037 + j for j in stable_desc(got[37:87], 5)]


def test_reference_named_entry_points():
    """`_scoring` returns a Python float per ligand (pharmacophore_model.py:101-106)."""
    model, lib, weights, d = load_golden("set_6oim_c5")
    for i in (0, 3, 17):
        s = model._scoring(lib.record(i), weights)
        assert isinstance(s, float)
        assert abs(s - d["score"][i]) <= 2e-6 * abs(d["score"][i]) + 1e-30
    try:
        import openbabel  # noqa: F401
    except ImportError:
        # file / SMILES entry points need OpenBabel exactly like the reference; they must say so
        with pytest.raises(ImportError, match="OpenBabel"):
            model.scoring_file("ligand.sdf")


def test_weights_argument():
    model, lib, weights, d = load_golden("set_6oim_c8_weights")
    default = model.screen(lib).scores.cpu().numpy()
    override = model.screen(lib, weights=weights).scores.cpu().numpy()
    assert rel_err(override, d["score"]).max() < 2e-6 + 6e-8
    assert np.abs(default - override).max() > 1.0


def test_multi_model_over_one_library():
    import ctypes

    import torch

    from pharmaconet_amd import _ffi
    from pharmaconet_amd.constants import weights_vector
    from pharmaconet_amd.engine import DeviceLibrary, device_model

    m1, lib, _, d1 = load_golden("set_6oim_c8")
    m2, _, _, _ = load_golden("set_c21_c8")
    lib = lib.slice(10, 120)
    dev = DeviceLibrary(lib)
    handles = (ctypes.c_void_p * 2)(device_model(m1).handle, device_model(m2).handle)
    out = torch.empty(2 * len(lib), dtype=torch.float32, device="cuda")
    w = (ctypes.c_float * 7)(*weights_vector(None))
    _ffi.check(_ffi.load().pmx_score_multi(handles, 2, dev.handle, w, 0, len(lib), out.data_ptr(), None, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(2, -1)
    np.testing.assert_array_equal(got[0], m1.screen(dev).scores.cpu().numpy())
    np.testing.assert_array_equal(got[1], m2.screen(dev).scores.cpu().numpy())
    ref = d1["score"][10:130]
    assert np.all(got[0][ref == 0] == 0.0)
    assert rel_err(got[0][ref != 0], ref[ref != 0]).max() < 2e-6 + 6e-8


def test_edge_cases():
    from pharmaconet_amd import PackedLibrary
    from pharmaconet_amd.library import LigandFeatures, pack_ligand

    model, lib, _, d = load_golden("set_c21_c8")
    empty = PackedLibrary.from_records([])
    res = model.screen(empty, topk=3)
    assert res.scores.numel() == 0 and [i for i, _ in res.ranking()] == []
    # zero-feature ligand and a ligand whose only type (Halogen) the model lacks: 0 (graph_match.py:95-99)
    zero = pack_ligand(LigandFeatures([6, 8], [[1], [0]], [], np.zeros((2, 4, 3), np.float32)))
    hal = pack_ligand(LigandFeatures([6, 17], [[1], [0]], [("Halogen", 1, 1)], np.ones((2, 4, 3), np.float32)))
    got = model.screen([zero, hal, lib.record(0)]).scores.cpu().numpy()
    assert got[0] == 0.0 and got[1] == 0.0


def test_order_and_partition_invariance_large():
    """Size-independent properties at a BASELINE-like scale: a ligand's score does not depend on its
    position, its neighbours, or on how the library is chunked into launches."""
    import torch

    from pharmaconet_amd.constants import TYPE_ID
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import expand_library_on_device, synthetic_library

    model, _, _, _ = load_golden("set_6oim_c8")
    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]])
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    base = synthetic_library(512, num_conformers=8, model_nodes=(centers, types), conformer_noise=0.0)
    offsets, data = expand_library_on_device(base, 300, "cuda")  # 153,600 ligands > one chunk
    lib = DeviceLibrary.from_device_buffers(offsets, data)
    full = model.screen(lib).scores
    assert torch.isfinite(full).all() and (full >= 0).all()
    # shifted ranges cut the chunks differently
    for first, count in ((1, 70_000), (65_537, 88_000), (131_071, 2)):
        part = model.screen(lib, first=first, count=count).scores
        assert torch.equal(part, full[first : first + count])
    # a reversed copy of a slice gives reversed scores
    n = 4096
    off = offsets.cpu().numpy()
    dat = data.cpu().numpy()
    from pharmaconet_amd import PackedLibrary

    recs = [dat[off[i] : off[i + 1]].tobytes() for i in range(n)]
    rev = model.screen(PackedLibrary.from_records(recs[::-1])).scores
    assert torch.equal(rev.flip(0), full[:n])


def test_screening_cli_writes_the_reference_csv(tmp_path):
    """`python -m pharmaconet_amd.screening` (screening.py:50-75): CSV sorted like the reference would."""
    from conftest import GOLDEN
    from pharmaconet_amd.screening import main

    model, lib, weights, d = load_golden("set_6oim_c8_weights")
    libfile = tmp_path / "lib.pmxlib"
    lib.save(libfile)
    (tmp_path / "lib.pmxlib.names").write_text("\n".join(f"mol_{i}.sdf" for i in range(len(lib))))
    out = tmp_path / "out.csv"
    main(["-p", str(GOLDEN / "model_6oim_like.pm"), "-d", str(libfile), "-o", str(out), "--hbd", "5", "--hba", "5", "--aromatic", "8"])
    lines = out.read_text().splitlines()
    assert lines[0] == "path,score" and len(lines) == len(lib) + 1
    names = [ln.split(",")[0] for ln in lines[1:]]
    scores = np.array([float(ln.split(",")[1]) for ln in lines[1:]])
    assert np.all(np.diff(scores) <= 0)
    ref = d["score"]
    want = sorted(range(len(ref)), key=lambda i: ref[i], reverse=True)
    # same ranking as the reference wherever reference scores differ by more than the float32 resolution
    got_idx = [int(n.split("_")[1].split(".")[0]) for n in names]
    for pos, (g, w) in enumerate(zip(got_idx, want)):
        assert g == w or abs(ref[g] - ref[w]) <= 2e-6 * abs(ref[w]), pos
    assert rel_err(scores, ref[got_idx]).max() < 2e-6 + 6e-8


def test_float64_scores_are_the_unrounded_float32_scores():
    """`pmx_score_f64` / `pmx_score_multi_f64`: the float64 mean the reference returns (graph_match.py:109). Its float32 rounding IS
    pmx_score's output bit for bit, it is closer to the reference's float64 than a float32 can be where the tables allow, `_scoring`
    hands it out as a Python float, and `topk` is refused with it (the device ranking is a float32 ranking)."""
    import ctypes

    import torch

    from pharmaconet_amd import _ffi
    from pharmaconet_amd.constants import weights_vector
    from pharmaconet_amd.engine import DeviceLibrary, device_model

    for name in ("set_6oim_c8", "set_6oim_c1", "set_s64_c64"):
        model, lib, weights, d = load_golden(name)
        f32 = model.screen(lib, weights=weights).scores.cpu().numpy()
        res = model.screen(lib, weights=weights, float64=True)
        f64 = res.scores.cpu().numpy()
        assert f64.dtype == np.float64 and f32.dtype == np.float32
        np.testing.assert_array_equal(f64.astype(np.float32), f32)
        np.testing.assert_array_equal(res.status.cpu().numpy(), 0)
        ref = d["score"]
        assert np.all(f64[ref == 0] == 0.0) and rel_err(f64[ref != 0], ref[ref != 0]).max() < 2e-6
        assert not np.array_equal(f64, f32.astype(np.float64))  # (it does carry more than float32 digits)
    with pytest.raises(ValueError):
        model.screen(lib, topk=5, float64=True)
    # the multi-model entry point, through the C ABI
    model, lib, _, d = load_golden("set_6oim_c8")
    other, _, _, _ = load_golden("set_c21_c8")
    dev = DeviceLibrary(lib)
    handles = (ctypes.c_void_p * 2)(device_model(model, dev.device).handle, device_model(other, dev.device).handle)
    w = (ctypes.c_float * 7)(*weights_vector(None))
    out = torch.full((2, len(lib)), -1.0, dtype=torch.float64, device="cuda")
    _ffi.check(_ffi.load().pmx_score_multi_f64(handles, 2, dev.handle, w, 0, len(lib), out.data_ptr(), None, None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out[0].cpu().numpy(), model.screen(lib, float64=True).scores.cpu().numpy())
    np.testing.assert_array_equal(out[1].cpu().numpy().astype(np.float32), other.screen(lib).scores.cpu().numpy())
    # _scoring returns the float64
    one = lib.slice(3, 1)
    assert model._scoring(one) == float(out[0, 3])


def test_sharded_screen_merges_to_the_global_ranking():
    """Two contiguous shards scored separately (as two ranks would), per-shard top-k merged: the same
    ranking as one pass over the whole library (screening.py:70 order)."""
    from pharmaconet_amd.distributed import merge_topk, shard_range

    model, lib, _, _ = load_golden("set_6oim_c8")
    k = 25
    full = model.screen(lib, topk=k)
    parts_s, parts_i = [], []
    for rank in range(2):
        first, count = shard_range(len(lib), rank, 2)
        res = model.screen(lib.slice(first, count), topk=k, index_base=first)
        parts_s.append(res.topk_scores.cpu().numpy())
        parts_i.append(res.topk_indices.cpu().numpy())
        np.testing.assert_array_equal(res.scores.cpu().numpy(), full.scores.cpu().numpy()[first : first + count])
    top_s, top_i = merge_topk(np.concatenate(parts_s), np.concatenate(parts_i), k)
    assert top_i.tolist() == full.topk_indices.cpu().numpy().tolist()
    np.testing.assert_array_equal(top_s, full.topk_scores.cpu().numpy())


def test_scores_do_not_depend_on_how_the_work_is_cut(monkeypatch):
    """The same bits whatever the launch structure: small super-chunks, few or many wavefronts per CU, tables that do
    not fit the slices (large-slice and arena passes), trees split into queued subtrees almost at once or never, one
    task round only, subtrees handed over at any depth - and with the bound test of the tree search switched off
    (every subtree walked, as the reference does): dropping or moving subtrees never changes a score."""
    import torch

    from pharmaconet_amd import engine
    from pharmaconet_amd.constants import TYPE_ID
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import expand_library_on_device, synthetic_library

    model, _, _, _ = load_golden("set_6oim_c8")
    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]])
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    base = synthetic_library(256, num_conformers=8, model_nodes=(centers, types), conformer_noise=0.0, seed=4242)
    offsets, data = expand_library_on_device(base, 120, "cuda")  # 30,720 ligands
    lib = DeviceLibrary.from_device_buffers(offsets, data)
    want = model.screen(lib).scores
    stats_default = engine.last_score_stats()
    assert torch.isfinite(want).all()
    assert stats_default["n_tasks"] > 0  # the default run does split trees
    for env in (
        {"PMX_SUPER": "4001"},                       # 8 super-chunks: control block, queue and arena restart eight times
        {"PMX_SUPER": "4001", "PMX_OVERLAP": "2"},   # ... with every chunk's task rounds on the side stream beside the next chunk's ligand kernel
        {"PMX_SUPER": "7000", "PMX_OVERLAP": "2", "PMX_LIG_SHARE": "0.3", "PMX_BUDGET": "32"},
        {"PMX_WAVES_PER_CU": "2"},                   # few persistent wavefronts: each builds and walks many ligands
        {"PMX_SLICE_KB": "8"},                       # most tables overflow the slices: large-slice pass
        {"PMX_SLICE_KB": "8", "PMX_BIG_SLICE_MB": "1", "PMX_BIG_TOTAL_MB": "64"},
        {"PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0"},  # trees are split almost at once, subtrees handed over at any depth
        {"PMX_BUDGET": "64", "PMX_ROUNDS": "1"},     # one task round: queued subtrees are walked to their end
        {"PMX_TREE_FLAGS": "2"},                     # nothing is ever queued
        {"PMX_TREE_FLAGS": "4"},                     # no bound test
        {"PMX_TREE_FLAGS": "32"},                    # the last two levels walked frame by frame instead of fused into the parent's pass
        {"PMX_TREE_FLAGS": "64"},                    # no cache of the children's totals: every return evaluates the frame again
        {"PMX_TREE_FLAGS": "96"},
        {"PMX_TREE_FLAGS": "128"},                   # frames with more candidates than slots are not filtered through the V masks first
        {"PMX_TREE_FLAGS": "256"},
        {"PMX_TREE_FLAGS": "512"},
        {"PMX_TREE_FLAGS": "8192"},                  # the walkers of a split ligand do not trade maxima while they run
        {"PMX_TREE_FLAGS": "4096"},                  # children with fewer than 5 matches are never bound-tested (walked, not probed)
        {"PMX_TREE_FLAGS": "2048"},                  # children visited first to last instead of largest bound first
        {"PMX_TREE_FLAGS": "2048", "PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0"},
        {"PMX_TREE_FLAGS": "1024"},                  # no path-aware bound test (children tested against their W bound only)
        {"PMX_TREE_FLAGS": "1024", "PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0"},
        {"PMX_TREE_FLAGS": "32768"},                 # probes search without the chain lengths (every dropped child is searched, every candidate tried)
        {"PMX_TREE_FLAGS": "32768", "PMX_BUDGET": "16", "PMX_MIN_LEVELS": "0"},
        {"PMX_TREE_FLAGS": "65536"},                 # no dead-entry test in the table phase: every entry that passes the prefilter is computed
        {"PMX_DEAD_MIN_ENTRIES": "1"},               # ... and the test on every level pair, however few entries it has
        {"PMX_DEAD_MIN_ENTRIES": "1", "PMX_SLICE_KB": "8"},
        {"PMX_PATH_KB": "4"},                        # path sums of most ligands do not fit the wave's buffer: those do without the test
        {"PMX_PATH_KB": "4", "PMX_BUDGET": "32"},                   # children tested against the per-level bound instead of their own                   # a ligand's subtrees spread over the queue shards instead of kept in one
    ):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            got = model.screen(lib).scores
            stats = engine.last_score_stats()
        assert torch.equal(got, want), env
        if env.get("PMX_SLICE_KB"):
            assert stats["n_slice_overflow"] > 1000
        if env.get("PMX_TREE_FLAGS") == "2":
            assert stats["n_tasks"] == 0
        if env.get("PMX_TREE_FLAGS") == "32768" and "PMX_BUDGET" not in env:
            assert stats["n_probe_passes"] > 4 * stats_default["n_probe_passes"]  # what the chain lengths keep the probes from
        if env.get("PMX_TREE_FLAGS") == "65536":
            assert stats["n_dead_entries"] == 0 and stats["n_items"] >= stats_default["n_items"]
        if env.get("PMX_DEAD_MIN_ENTRIES") == "1" and "PMX_SLICE_KB" not in env:
            assert stats["n_dead_entries"] > 0 and stats["n_items"] < stats_default["n_items"]
        if env.get("PMX_TREE_FLAGS") == "4":
            assert stats["n_frames"] > 2 * stats_default["n_frames"]  # the bound test is what keeps the trees small


def test_concurrent_callers_are_serialised_not_corrupted():
    """include/pmx.h: pmx_score may be called from any thread; work buffers are kept per (device, stream). Two threads
    screening different libraries on their own streams at once get the scores they get alone."""
    import threading

    import torch

    model, lib, _, expected = load_golden("set_6oim_c8")
    model2, lib2, _, expected2 = load_golden("set_c21_c8")
    alone = [model.screen(lib).scores.cpu().numpy(), model2.screen(lib2).scores.cpu().numpy()]
    got = [None, None]
    errors = []

    def worker(slot, m, l):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(3):
                    got[slot] = m.screen(l).scores.cpu().numpy()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(0, model, lib)), threading.Thread(target=worker, args=(1, model2, lib2))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    np.testing.assert_array_equal(got[0], alone[0])
    np.testing.assert_array_equal(got[1], alone[1])


def test_workspaces_of_abandoned_streams_are_recycled(monkeypatch):
    """Work buffers are cached per (device, stream), some 40 GB each at the defaults, and nothing tells libpmx that a stream is
    gone: at most PMX_MAX_WORKSPACES are kept per device, the least recently used idle one is freed when one more is needed
    (ADVICE r3). Six short-lived streams, a cap of two: the same scores every time, and the device does not fill up."""
    import torch

    from pharmaconet_amd import engine

    model, lib, _, expected = load_golden("set_6oim_c8")
    engine.release_workspaces()
    monkeypatch.setenv("PMX_MAX_WORKSPACES", "2")
    monkeypatch.setenv("PMX_ARENA_MB", "2048")
    want = model.screen(lib).scores.cpu().numpy()
    free = []
    for k in range(6):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            got = model.screen(lib).scores.cpu().numpy()
        np.testing.assert_array_equal(got, want)
        del stream
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert min(free[2:]) > free[1] - (3 << 30), free  # (two workspaces of 2 x 2 GB arenas + slices stay; four more would be 25 GB)
    engine.release_workspaces()


def test_full_task_queue_changes_nothing_but_time(monkeypatch):
    """The queue of exported subtrees is 64 shards of fixed size. When a wave finds its shard full it keeps the subtree
    and walks it itself (and the call reports `queue_overflow`): same bits."""
    import torch

    from pharmaconet_amd import engine
    from pharmaconet_amd.constants import TYPE_ID
    from pharmaconet_amd.engine import DeviceLibrary
    from tools.synthetic import expand_library_on_device, synthetic_library

    model, _, _, _ = load_golden("set_6oim_c8")
    st = model.__getstate__()
    centers = np.array([n["center"] for n in st["nodes"]])
    types = np.array([TYPE_ID[n["type"]] for n in st["nodes"]])
    base = synthetic_library(256, num_conformers=8, model_nodes=(centers, types), conformer_noise=0.0, seed=4242)
    offsets, data = expand_library_on_device(base, 40, "cuda")  # 10,240 ligands
    lib = DeviceLibrary.from_device_buffers(offsets, data)
    want = model.screen(lib).scores
    assert engine.last_score_stats()["queue_overflow"] == 0
    engine.release_workspaces()  # the queue is sized when a workspace is created
    try:
        with monkeypatch.context() as mp:
            mp.setenv("PMX_TASKQ_MB", "1")  # 128 records per shard
            mp.setenv("PMX_BUDGET", "16")   # export early and often
            mp.setenv("PMX_MIN_LEVELS", "0")
            got = model.screen(lib).scores
            stats = engine.last_score_stats()
        assert stats["queue_overflow"] == 1  # otherwise this test shows nothing
        assert torch.equal(got, want)
    finally:
        engine.release_workspaces()
