"""Build libvpk.so (HIP, gfx950 only) in-tree with hipcc.

    python -m vanishing_points_2017_amd.build [--force]

The EM translation unit is compiled with -ffp-contract=off (csrc/em_device.hpp explains why);
everything else with the default contraction.  No CUDA, no hipify, no multi-arch fat binaries.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "_obj")
SO = os.path.join(PKG, "libvpk.so")
ARCH = "gfx950"

# source file -> extra flags
UNITS = {
    "vpk_core.hip": [],
    "vpk_em.hip": ["-ffp-contract=off"],
    "vpk_cnn.hip": [],
    "vpk_raster.hip": ["-ffp-contract=off"],   # the curve samples must round like NumPy's separate ufunc calls
    "vpk_horizon.hip": ["-ffp-contract=off"],
    "vpk_pipeline.hip": [],
    "vpk_lsd.cpp": ["-ffp-contract=off"],      # host code: the front end's line segment detector
}
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libvpk.so can only be built with the ROCm toolchain")
    return exe


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + \
           [os.path.join(PKG, "..", "include", "vpk.h")]


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    deps_mtime = max(os.path.getmtime(d) for d in _deps())
    objs = []
    procs = []
    for src, extra in UNITS.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            raise RuntimeError("missing source " + path)
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        fresh = os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), deps_mtime)
        if fresh and not force:
            continue
        cmd = [cc] + COMMON + extra + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    need_link = force or procs or not os.path.exists(SO) or \
        any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs)
    if need_link:
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", SO]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(SO)
