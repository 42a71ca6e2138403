#!/usr/bin/env python
"""Micro-benchmark of the trunk's 3x3 conv kernels on the device, through the C ABI (no autograd):
every problem of a stage-4 trunk level alone and the grouped level, old per-tile kernel (conv_tile.hip) against the
persistent one (conv_tile_p.hip) at several strip lengths and transform modes.  python tools/tilebench.py [reps]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from semseg_amd import hip_backend as hb  # noqa: E402

if "--timing" in sys.argv or "--lib" in sys.argv:
    from semseg_amd import _lib
    _name = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else "timing"
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libsemseg_hip.so", "libsemseg_hip_%s.so" % _name)
L = hb.lib()
DEV = "cuda"
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731


class Prob:
    def __init__(self, C, H, W, seed):
        g = torch.Generator().manual_seed(seed)
        self.C, self.H, self.W = C, H, W
        self.x = torch.randn(1, H, W, C, generator=g).to(DEV).to(torch.bfloat16)
        self.x2 = torch.randn(1, H, W, C, generator=g).to(DEV).to(torch.bfloat16)
        self.aux = torch.randn(1, H, W, C, generator=g).to(DEV).to(torch.bfloat16)
        self.w = (torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5)).to(DEV)
        self.wp, _ = hb._packed_filter(self.w, 2, C, 0)
        self.y = torch.empty(1, H, W, C, device=DEV, dtype=torch.bfloat16)
        self.stats = torch.zeros(hb.stat_replicas() * 2 * C, device=DEV, dtype=torch.float64)
        self.xf = torch.rand(5, C, device=DEV) + 0.5
        self.coef = torch.rand(4, C, device=DEV) + 0.5
        self.d = hb._tile_desc(1, H, W, C, C, C, (3, 3), 1, 1, 1, H, W, False)
        self.flops = 2.0 * H * W * C * C * 9
        self.bytes = 2.0 * H * W * C * 2 + 2.0 * C * C * 9

    def old(self):
        hb.check(L.ssa_conv2d_tile(ctypes.byref(self.d), P(self.x), P(self.wp), None, P(self.y), P(self.stats), hb._s()), "tile")

    def new(self, aux=0):
        st = None if aux == 1 else self.stats
        hb.check(L.ssa_conv2d_tile_p(ctypes.byref(self.d), P(self.x), P(self.wp), None, P(self.y), P(st),
                                     P(self.aux) if aux else None, self.C, P(self.coef) if aux == 2 else None, aux, hb._s()),
                 "tile_p")


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    level = [(48, 256, 256), (48, 128, 128), (96, 128, 128), (96, 64, 64), (192, 64, 64), (192, 32, 32), (384, 32, 32), (384, 16, 16)]
    probs = [Prob(C, H, W, i) for i, (C, H, W) in enumerate(level)]
    print("%-16s %8s %8s | persistent by strip units: %s" % ("problem", "old us", "", " ".join("%7d" % u for u in (1, 2, 4, 8, 16))))
    for p in probs:
        t_old = timeit(p.old, reps)
        row = []
        for u in (1, 2, 4, 8, 16):
            L.ssa_conv_tile_strip(u)
            row.append(timeit(p.new, reps))
        L.ssa_conv_tile_strip(0)
        t_auto = timeit(p.new, reps)
        print("%3d @ %3dx%-3d   %8.1f %8s | %s  auto %7.1f   (%.0f TF/s, %.0f GB/s at best)" % (
            p.C, p.H, p.W, t_old, "", " ".join("%7.1f" % t for t in row), t_auto,
            p.flops / min(row + [t_auto]) / 1e6, p.bytes / min(row + [t_auto]) / 1e3))

    def level_old():
        with hb.group():
            for p in probs:
                p.old()

    def level_new(aux=0):
        with hb.group():
            for p in probs:
                p.new(aux)
    fl = sum(p.flops for p in probs)
    by = sum(p.bytes for p in probs)
    t = timeit(level_old, reps)
    print("level (8 problems, %.1f GFLOP, %.1f MB): old grouped %.1f us = %.0f TF/s, %.0f GB/s" % (fl / 1e9, by / 1e6, t, fl / t / 1e6, by / t / 1e3))
    best = (1e9, 4)
    for u in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        L.ssa_conv_tile_strip(u)
        t = timeit(level_new, reps)
        best = min(best, (t, u))
        print("   persistent, strip units %d: %.1f us = %.0f TF/s, %.0f GB/s" % (u, t, fl / t / 1e6, by / t / 1e3))
    L.ssa_conv_tile_strip(best[1])
    for aux in (2, 1):
        t = timeit(lambda: level_new(aux), reps)
        print("   persistent units %d, aux %d: %.1f us" % (best[1], aux, t))
    L.ssa_conv_tile_strip(0)
    # two-branch level (stage 2)
    probs2 = probs[:4]
    for u in (1, 2, 4):
        L.ssa_conv_tile_strip(u)

        def lv():
            with hb.group():
                for p in probs2:
                    p.new()
        t = timeit(lv, reps)
        print("   stage-2 level (4 problems) persistent units %d: %.1f us" % (u, t))

    def lv_old():
        with hb.group():
            for p in probs2:
                p.old()
    print("   stage-2 level old: %.1f us" % timeit(lv_old, reps))
    L.ssa_conv_tile_strip(0)




def timing():
    """Phase stamps (s_memtime, 100 MHz) of wave 0 of workgroup 0 from the -DSSA_TILE_TIMING build."""
    names = ["start", "staged", "barY", "fetched", "mfma0", "bar0", "mfma_done", "epilogue"]
    for (C, H, W), units in (((48, 256, 256), 8), ((96, 128, 128), 8), ((192, 64, 64), 8), ((384, 32, 32), 8), ((48, 256, 256), 1)):
        p = Prob(C, H, W, 1)
        dbg = torch.zeros(24 * 8, dtype=torch.int64, device=DEV)
        L.ssa_conv_tile_strip(units)
        for _ in range(3):
            hb.check(L.ssa_conv2d_tile_p(ctypes.byref(p.d), P(p.x), P(p.wp), None, P(p.y), P(p.stats), None, C,
                                         P(dbg), 0, hb._s()), "tile_p")
        torch.cuda.synchronize()
        t = dbg.cpu().view(24, 8)
        print("== %d @ %dx%d strip units %d: stamps in ticks of s_memtime relative to the iteration start; d(start) = whole iteration" % (C, H, W, units))
        print("   it " + " ".join("%9s" % n for n in names) + "   next-start")
        for i in range(24):
            if int(t[i, 0]) == 0:
                break
            nxt = int(t[i + 1, 0]) - int(t[i, 0]) if i + 1 < 24 and int(t[i + 1, 0]) else -1
            print("   %2d " % i + " ".join("%9d" % (int(t[i, k]) - int(t[i, 0])) for k in range(8)) + "   %d" % nxt)
    L.ssa_conv_tile_strip(0)


if __name__ == "__main__":
    if "--timing" in sys.argv:
        timing()
    else:
        main()
