#!/usr/bin/env python
"""Micro-benchmark of the trunk's halo-staged weight-gradient kernels (csrc/conv_wgrad_tile.hip) through the C ABI: one
grouped launch of `n` layers per channel class at the two scale passes' sizes, strip length from the host's fitter
(hip_backend._fit_tile_strips) or given.   python tools/wgradbench.py [reps] [--strip S] [--lib name]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from semseg_amd import _lib  # noqa: E402
if "--lib" in sys.argv:
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libsemseg_hip.so", "libsemseg_hip_%s.so" % sys.argv[sys.argv.index("--lib") + 1])
from semseg_amd import hip_backend as hb  # noqa: E402
from tilebench import timeit, P  # noqa: E402

L = hb.lib()
DEV = "cuda"


class Job:
    def __init__(self, C, H, W, seed):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(1, H, W, C, generator=g).to(DEV).to(hb.ACT_DTYPE)
        self.dy = torch.randn(1, H, W, C, generator=g).to(DEV).to(hb.ACT_DTYPE)
        self.geom_in, self.k, self.stride, self.dil, self.pad, self.cout_pad = (1, H, W, C), (3, 3), 1, 1, 1, C
        self.C, self.H, self.W = C, H, W
        self.flops = 2.0 * H * W * C * C * 9

    def plan(self, strip):
        C, H, W = self.C, self.H, self.W
        self.d = hb.ConvDesc(1, H, W, C, C, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, strip)
        ns, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        hb.check(L.ssa_conv2d_wgrad_tile_plan(ctypes.byref(self.d), C, ctypes.byref(ns), ctypes.byref(ws)), "plan")
        self.ns = ns.value
        self.partial = torch.empty(ws.value // 4, device=DEV)
        self.dw = torch.empty(C, C, 3, 3, device=DEV)

    def run(self):
        hb.check(L.ssa_conv2d_wgrad_tile(ctypes.byref(self.d), P(self.x), P(self.dy), self.C, self.C, self.ns,
                                         P(self.partial), hb._s()), "wgrad_tile")

    def reduce(self):
        hb.check(L.ssa_conv2d_wgrad_reduce(P(self.partial), self.ns, self.C, self.C, self.C, self.C, 3, 3, P(self.dw), 0, hb._s()), "reduce")


def bench(name, jobs, reps, strip=None):
    fitted = hb._fit_tile_strips(jobs, 8) if strip is None else {id(j): strip for j in jobs}
    for j in jobs:
        j.plan(fitted.get(id(j), 8))

    def go():
        with hb.group():
            for j in jobs:
                j.run()

    def red():
        with hb.group():
            for j in jobs:
                j.reduce()
    t = timeit(go, reps)
    tr = timeit(red, reps)
    fl = sum(j.flops for j in jobs)
    parts = {}
    for j in jobs:
        p_, s_, k_ = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        L.ssa_conv2d_wgrad_tile_geometry(j.C, j.C, ctypes.byref(p_), ctypes.byref(s_), ctypes.byref(k_))
        parts[id(j)] = p_.value
    wgs = sum(j.ns * parts[id(j)] for j in jobs)
    print("%-34s strips %-12s wgs %5d  tile %7.1f us %5.0f TF/s   reduce %6.1f us  (partials %.0f MB)" % (
        name, sorted(set(fitted.values())) if fitted else strip, wgs, t, fl / t / 1e6, tr, sum(j.partial.numel() for j in jobs) * 4 / 1e6), flush=True)


def timing():
    """Phase stamps (s_memtime) of thread 0 of workgroup (0, 0) of ConvWgradTileA from an experiment build that carries
    them (-DSSA_WGRAD_TIMING copy of conv_wgrad_tile.hip, `ssa_wgrad_timing_read`): per tile
    [loop top | barrier passed (tile landed) | next tile's DMAs issued | MFMAs done]."""
    import numpy as np
    jobs = [Job(96, 128, 128, i) for i in range(8)] + [Job(96, 64, 64, 50 + i) for i in range(8)]
    fitted = hb._fit_tile_strips(jobs, 8)
    for j in jobs:
        j.plan(fitted.get(id(j), 8))
    for _ in range(3):
        with hb.group():
            for j in jobs:
                j.run()
    torch.cuda.synchronize()
    fn = L.ssa_wgrad_timing_read
    fn.argtypes = [ctypes.c_void_p]
    buf = np.zeros(256, dtype=np.int64)
    assert fn(buf.ctypes.data) == 0
    t = buf.reshape(64, 4)
    print("strip %s; ticks of s_memtime relative to the loop top of each tile" % sorted(set(fitted.values())))
    print("  tile   landed   issued   mfma_done   next-top")
    for i in range(64):
        if t[i, 0] == 0:
            break
        nxt = t[i + 1, 0] - t[i, 0] if i + 1 < 64 and t[i + 1, 0] else -1
        print("  %3d  %7d  %7d  %9d  %9d" % (i, t[i, 1] - t[i, 0], t[i, 2] - t[i, 0], t[i, 3] - t[i, 0], nxt))


def main():
    if "--timing" in sys.argv:
        return timing()
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    strip = int(sys.argv[sys.argv.index("--strip") + 1]) if "--strip" in sys.argv else None
    n = 8
    mk = lambda C, H, s0: [Job(C, H, H, s0 + i) for i in range(n)] + [Job(C, H // 2, H // 2, s0 + 50 + i) for i in range(n)]  # noqa: E731
    classes = {"48 @ 256^2 + 128^2 x8": mk(48, 256, 0), "96 @ 128^2 + 64^2 x8": mk(96, 128, 100),
               "192 @ 64^2 + 32^2 x8": mk(192, 64, 200), "384 @ 32^2 + 16^2 x8": mk(384, 32, 300)}
    for name, jobs in classes.items():
        bench(name, jobs, reps, strip)
    mixed = classes["96 @ 128^2 + 64^2 x8"] + classes["192 @ 64^2 + 32^2 x8"]
    bench("96 + 192 classes (one launch)", mixed[:32], reps, strip)
    for s in (4, 8, 16, 32):
        bench("96 @ 128^2 + 64^2 x8", classes["96 @ 128^2 + 64^2 x8"], reps, s)


if __name__ == "__main__":
    main()
