#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table from the gfx950 code object of libpmx.

    python tools/kernel_resources.py [--out profiles/r4_kernel_resources.json]

Compiles csrc/pmx_api.hip and csrc/pmx_pack_device.hip device-only with the build's flags, unbundles the gfx950 code object and reads the
AMDGPU metadata notes (llvm-readelf --notes)."""
import json, re, subprocess, sys, tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from pharmaconet_amd.build import FLAGS, CSRC, hipcc  # noqa: E402

LLVM = Path("/opt/rocm/lib/llvm/bin")


def main():
    out = None
    if "--out" in sys.argv:
        out = Path(sys.argv[sys.argv.index("--out") + 1])
    notes = ""
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        flags = [f for f in FLAGS if f != "-fPIC"]
        import os
        extra = os.environ.get("PMX_CXXFLAGS", "").split()
        for src in ("pmx_api.hip", "pmx_pack_device.hip"):  # the screening / top-k kernels, and the device packer's
            subprocess.run([hipcc(), *flags, *extra, f"-I{REPO / 'include'}", f"-I{CSRC}", "--cuda-device-only", "-c", str(CSRC / src), "-o", str(td / "dev.o")], check=True)
            subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={td / 'dev.o'}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={td / 'dev.co'}"], check=True)
            notes += subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(td / "dev.co")], check=True, capture_output=True, text=True).stdout
    rows = []
    cur = {}
    keys = ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size")
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k in keys:
            cur[k] = int(v)
        elif k == "symbol":
            cur["symbol"] = v
        elif k == "wavefront_size":  # last key of a kernel's record (keys are sorted)
            if "symbol" in cur:
                rows.append(cur)
            cur = {}
    for r in rows:
        r["demangled"] = subprocess.run(["c++filt", r["symbol"][:-3]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "pmx_pack_device::").split("(")[0]
    rows = [r for r in rows if "rocprim" not in r["demangled"] and "hipcub" not in r["demangled"]]  # (the scan's library kernels are not ours to tune)
    rows.sort(key=lambda r: r["demangled"])
    print(f"{'kernel':60s} {'vgpr':>5s} {'vspill':>6s} {'sgpr':>5s} {'sspill':>6s} {'scratch':>8s} {'lds':>6s}")
    for r in rows:
        print(f"{r['demangled'][:60]:60s} {r.get('vgpr_count', 0):5d} {r.get('vgpr_spill_count', 0):6d} {r.get('sgpr_count', 0):5d} {r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):8d} {r.get('group_segment_fixed_size', 0):6d}")
    if out:
        from pharmaconet_amd.build import hipcc_version, read_stamp

        out.write_text(json.dumps({
            "source": "llvm-readelf --notes of the gfx950 code objects of csrc/pmx_api.hip and csrc/pmx_pack_device.hip (tools/kernel_resources.py)",
            "kernels": rows,
            "hipcc": hipcc_version(),
            "build_stamp": read_stamp(),
            "note": "the four -mllvm switches of pharmaconet_amd/build.py were validated (A/B on the GPU box) with this hipcc; the GPU box's own HIP runtime is 7.0.2 and only loads the code object",
        }, indent=1) + "\n")


if __name__ == "__main__":
    main()
