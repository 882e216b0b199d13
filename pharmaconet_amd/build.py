"""Builds pharmaconet_amd/libpmx.so (HIP, gfx950 only) in-tree with hipcc.

`python -m pharmaconet_amd.build [--force]`. The .so is git-ignored but travels with the tree.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
CSRC = PKG / "csrc"
LIB = PKG / "libpmx.so"
PACK_LIB = PKG / "libpmx_pack.so"  # the packer alone, host-only (no HIP / RCCL runtime)
SOURCES = ("pmx_api.hip", "pmx_screen_debug.hip", "pmx_topk.hip", "pmx_density.hip", "pmx_pack_device.hip", "pmx_pack.cpp", "pmx_sdf.cpp", "pmx_perceive.cpp")
DEPS = ("pmx_screen.hip", "pmx_device.h", "pmx_debug.h")
FLAGS = (
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # distances and thresholds must round like the reference's float32 NumPy code
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
    # [MI355X] configs[1] 117.6 -> 106.7 ms per pass (ligand kernel 90 -> 79 ms): SimplifyCFG's sinking of common instructions into the
    # blocks where the many paths of the table and walker loops meet costs these kernels copies at every join; loop strength reduction
    # trades address arithmetic for live registers they do not have (tools/build_variant.py A/B runs, HISTORY.md round 5)
    "-mllvm", "-simplifycfg-sink-common=false",
    "-mllvm", "-disable-lsr",
    # [MI355X] 100.5-100.7 -> 99.3-99.7 ms: the greedy allocator takes the register class of a live range before its globalness
    # (the scalar ranges of the walker's loop are assigned before the many short vector ones; A/B twice on one box, HISTORY.md round 5)
    "-mllvm", "-greedy-regclass-priority-trumps-globalness",
    # [MI355X] 99.5-99.9 -> 99.0 ms (twice): SimplifyCFG folds a two-entry phi into a select only when one instruction has to be speculated
    "-mllvm", "-phi-node-folding-threshold=1",
)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


STAMP = PKG / "libpmx.stamp"  # what libpmx.so was built from: written by _build(), read by _stale() and by bench.py's csrc_digest()


def hipcc_version() -> str:
    try:
        out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception as e:  # (informational only)
        return f"unknown ({e!r})"
    lines = [ln.strip() for ln in out.splitlines() if "HIP version" in ln or "clang version" in ln]
    return "; ".join(lines) or out.strip()[:200]


def build_digest() -> dict:
    """Content hash of everything that decides what the kernels are: the device and host sources, include/pmx.h, the compile flags and
    PMX_CXXFLAGS. (Not the compiler's version: the GPU box's hipcc may differ from the build container's and must not trigger a rebuild
    there - the .so travels with the tree; the version the build was made with is recorded in the stamp.)"""
    import hashlib

    extra = os.environ.get("PMX_CXXFLAGS", "").split()
    h = hashlib.sha256((" ".join(FLAGS) + "\0" + " ".join(extra)).encode())
    dev = hashlib.sha256((" ".join(FLAGS) + "\0" + " ".join(extra)).encode())
    for f in sorted([CSRC / s for s in SOURCES + DEPS] + [REPO / "include" / "pmx.h"]):
        blob = f.name.encode() + b"\0" + f.read_bytes()
        h.update(blob)
        if f.suffix in (".hip", ".h"):
            dev.update(blob)
    return {"digest": h.hexdigest()[:16], "csrc_sha16": dev.hexdigest()[:16], "flags": list(FLAGS), "cxxflags": extra}


def read_stamp() -> dict | None:
    try:
        import json

        return json.loads(STAMP.read_text())
    except Exception:
        return None


def _stale() -> bool:
    if not LIB.exists() or not PACK_LIB.exists():
        return True
    stamp = read_stamp()
    # (a change of FLAGS or PMX_CXXFLAGS alone rebuilds: four -mllvm switches are worth 15 % of a pass - ADVICE r5)
    return stamp is None or stamp.get("digest") != build_digest()["digest"]


def build_native(force: bool = False, verbose: bool = False) -> Path:
    if not force and not _stale():
        return LIB
    import fcntl

    # several ranks may call build() at once (bench.py under torchrun): build under a lock, re-check after
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(verbose: bool) -> Path:
    cc = hipcc()
    digest = build_digest()  # (of the sources as they are when the compile starts)
    from concurrent.futures import ThreadPoolExecutor

    extra = os.environ.get("PMX_CXXFLAGS", "").split()

    def compile_one(src: str) -> str:
        obj = CSRC / (src.rsplit(".", 1)[0] + ".o")
        cmd = [cc, *FLAGS, *extra, f"-I{REPO / 'include'}", f"-I{CSRC}", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return str(obj)

    # (the two translation units that hold the screening kernels take a minute each: side by side)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    tmp = LIB.with_suffix(".so.tmp")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # librccl is linked without an rpath: a process that has imported torch already holds torch's bundled RCCL / HIP runtime
    # (see _ffi.load), and one that has not finds ROCm's through the loader's usual search path
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(tmp), f"-L{rocm}/lib", "-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    # the host-only packer library
    tmp = PACK_LIB.with_suffix(".so.tmp")
    cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DPMX_PACK_STANDALONE", f"-I{REPO / 'include'}",
           str(CSRC / "pmx_pack.cpp"), str(CSRC / "pmx_sdf.cpp"), str(CSRC / "pmx_perceive.cpp"), "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, PACK_LIB)
    import json

    STAMP.write_text(json.dumps({**digest, "hipcc": hipcc_version()}, indent=1) + "\n")
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
