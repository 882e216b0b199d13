#!/usr/bin/env python3
"""An A/B build of libpmx.so with extra compile flags, next to the product build:

    python tools/build_variant.py w7 -DPMX_SCREEN_WAVES=7 -DPMX_TASK_WAVES=7 -DPMX_TC_LEVELS=2
    PMX_LIBPMX=$PWD/variants/libpmx_w7.so python tools/knob_sweep.py

(variants/*.so are git-ignored and travel to the GPU box like the product's .so)."""
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from pharmaconet_amd.build import CSRC, FLAGS, SOURCES, hipcc  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out = REPO / "variants"
    out.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = out / f"{name}_{src.rsplit('.', 1)[0]}.o"
        # (the two translation units that include pmx_screen.hip - the product kernels and the validation kernels - are always compiled
        # with the variant's flags: a debug kernel built with the base macros under a host side built with the variant's would have
        # another LDS layout and other launch bounds)
        screening = src in ("pmx_api.hip", "pmx_screen_debug.hip")
        base_obj = out / f"base_{src.rsplit('.', 1)[0]}.o"
        fresh = base_obj.exists() and base_obj.stat().st_mtime > max((CSRC / src).stat().st_mtime, (REPO / "include" / "pmx.h").stat().st_mtime)
        if src.endswith(".hip") and not screening and fresh and name != "base":
            obj = out / f"base_{src.rsplit('.', 1)[0]}.o"
        else:  # (the compiles run side by side: the two screening units take a minute each)
            procs.append(subprocess.Popen([hipcc(), *FLAGS, *extra, f"-I{REPO / 'include'}", f"-I{CSRC}", "-c", str(CSRC / src), "-o", str(obj)]))
        objs.append(str(obj))
    if any(pr.wait() != 0 for pr in procs):
        sys.exit("build_variant: a compile failed")
    lib = out / f"libpmx_{name}.so"
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(lib), "-L/opt/rocm/lib", "-lrccl"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
