"""Where a conv tile's time goes (dev tool; run on the GPU box: `gpurun -- python scripts/conv_tile_timing.py 121 1200`).

Builds an instrumented copy of csrc/vpk_cnn.hip under /tmp (device clock read at tile start, after the tile
decode, after the two-stage prologue, after the K loop and after the epilogue; wave 0 / lane 0 accumulate into
a __device__ array for the layer whose K equals the argument), links it with the other objects into a private
libvpk, and runs one forward at B = 102.  Nothing under vanishing_points_2017_amd/ is modified."""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vanishing_points_2017_amd")


def patched_source(k):
    s = open(os.path.join(PKG, "csrc", "vpk_cnn.hip")).read()
    rep = [
        ('#include "vpk_internal.hpp"', '#include "%s/csrc/vpk_internal.hpp"' % PKG),
        ("constexpr int BK = 16; ", "__device__ unsigned long long g_tim[8];\nconstexpr int BK = 16; "),
        ("    int nx = 0;\n    if (tid == 0)    // ONE lane;",
         "    long long T0 = wall_clock64(), T1 = 0, T2 = 0, T3 = 0;\n    int nx = 0;\n    if (tid == 0)    // ONE lane;"),
        ("    const int nk = kt1 - kt0;\n    if (nk > 0) issue(kt0, 0);",
         "    T1 = wall_clock64();\n    const int nk = kt1 - kt0;\n    if (nk > 0) issue(kt0, 0);"),
        ("    __builtin_amdgcn_s_barrier();\n    for (int t = 0; t < nk; ++t) {",
         "    __builtin_amdgcn_s_barrier();\n    T2 = wall_clock64();\n    for (int t = 0; t < nk; ++t) {"),
        ("    const int oplane = d.OHp * d.OWp;\n#pragma unroll\n    for (int j = 0; j < TN; ++j) {",
         "    T3 = wall_clock64();\n    const int oplane = d.OHp * d.OWp;\n#pragma unroll\n    for (int j = 0; j < TN; ++j) {"),
        ("    tile = __builtin_amdgcn_readfirstlane(s_next[parity]);",
         "    if (tid == 0 && d.K == %d) { long long T4 = wall_clock64(); atomicAdd(&g_tim[0], (unsigned long long)(T1 - T0)); "
         "atomicAdd(&g_tim[1], (unsigned long long)(T2 - T1)); atomicAdd(&g_tim[2], (unsigned long long)(T3 - T2)); "
         "atomicAdd(&g_tim[3], (unsigned long long)(T4 - T3)); atomicAdd(&g_tim[4], 1ull); }\n"
         "    tile = __builtin_amdgcn_readfirstlane(s_next[parity]);" % k),
        ('extern "C" {\n\nint vpk_cnn_set_profiling',
         'extern "C" {\n\nvoid vpk_dbg_tim(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tim), 64); '
         'unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tim), z, 64); }\n\nint vpk_cnn_set_profiling'),
    ]
    for a, b in rep:
        if a not in s:
            raise SystemExit("vpk_cnn.hip changed: anchor not found: " + a[:60])
        s = s.replace(a, b, 1)
    return s


def child(k, so):
    sys.path.insert(0, ROOT)
    from vanishing_points_2017_amd import _lib
    _lib.SO_PATH = so
    import torch
    from vanishing_points_2017_amd import cnn
    from vanishing_points_2017_amd.runtime import get_runtime
    rt = get_runtime(0)
    net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
    x = torch.randint(0, 60, (102, 500, 500), dtype=torch.uint8, device=rt.tdev)
    for _ in range(2):
        net.forward_device(x)
    rt.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    rt.lib.vpk_dbg_tim(buf)
    net.forward_device(x)
    rt.synchronize()
    rt.lib.vpk_dbg_tim(buf)
    n = max(buf[4], 1)
    print("K=%d: %d tiles; per tile (us, wave 0): decode %.2f  prologue %.2f  K loop %.2f  epilogue %.2f"
          % ((k, buf[4]) + tuple(buf[i] / n / 100.0 for i in range(4))))


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), sys.argv[3])
    ks = [int(a) for a in sys.argv[1:]] or [121, 1200]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = [os.path.join(PKG, "csrc", "_obj", o) for o in ("vpk_core.o", "vpk_em.o", "vpk_raster.o")]
    for k in ks:
        src, obj, so = "/tmp/vpk_cnn_tim%d.hip" % k, "/tmp/vpk_cnn_tim%d.o" % k, "/tmp/libvpk_tim%d.so" % k
        open(src, "w").write(patched_source(k))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(PKG, "csrc"), "-c", src, "-o", obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [obj, "-o", so])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", str(k), so])


if __name__ == "__main__":
    main()
