#!/usr/bin/env python
"""Micro-benchmark of the head's large-channel conv kernels on the device, through the C ABI (no autograd):
weight gradient (conv_wgrad_head.hip) and forward/data gradient (conv_halo_gemm.hip) on the shapes of one training step
(conv3x3_ocr 720->512 at 256x256 and 128x128, the 1x1 convs of the OCR block).
python tools/headbench.py [reps] [--lib <name>]     (--lib: lib/libsemseg_hip_<name>.so from tools/expbuild.sh)"""
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


class WProb:
    def __init__(self, Cin, Cout, H, W, k, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(1, H, W, Cin, generator=g).to(DEV).to(torch.bfloat16)
        self.dy = torch.randn(1, H, W, Cout, generator=g).to(DEV).to(torch.bfloat16)
        self.d = hb.ConvDesc(1, H, W, Cin, Cin, H, W, Cout, Cout, k, k, 1, k // 2, 1, 0, 0, 0, -1)
        ns, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        hb.check(L.ssa_conv2d_wgrad_head_plan(ctypes.byref(self.d), Cout, ctypes.byref(ns), ctypes.byref(ws)), "plan")
        self.ns = ns.value
        self.partial = torch.empty(ws.value // 4, device=DEV)
        self.Cout = Cout
        self.flops = 2.0 * H * W * Cin * Cout * k * k
        self.name = "%dx%d %4d->%4d @ %3dx%-3d" % (k, k, Cin, Cout, H, W)
        # forward operands
        self.w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)).to(DEV)
        self.y = torch.empty(1, H, W, Cout, device=DEV, dtype=torch.bfloat16)
        self.fd = hb._tile_desc(1, H, W, Cin, Cin, Cout, (k, k), 1, k // 2, 1, H, W, False)
        self.halo = hb.halo_supported(self.fd)
        if self.halo:
            self.wp, _ = hb._packed_filter(self.w, 2, Cin, 0)

    def wgrad(self):
        hb.check(L.ssa_conv2d_wgrad_head(ctypes.byref(self.d), P(self.x), P(self.dy), self.Cout, self.Cout, self.ns,
                                         P(self.partial), hb._s()), "wgrad_head")

    def fwd(self):
        hb.check(L.ssa_conv2d_halo(ctypes.byref(self.fd), P(self.x), P(self.wp), None, P(self.y), None, hb._s()), "halo")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
    shapes = [(720, 512, 256, 256, 3), (720, 512, 128, 128, 3), (512, 256, 128, 128, 3), (256, 256, 128, 128, 3),
              (720, 720, 256, 256, 1), (512, 256, 256, 256, 1), (1024, 512, 256, 256, 1), (512, 512, 256, 256, 1),
              (512, 1024, 256, 256, 1), (256, 512, 256, 256, 1), (720, 720, 128, 128, 1), (1024, 512, 128, 128, 1)]
    if "--only-1x1" in sys.argv:
        shapes = [s for s in shapes if s[4] == 1]
    if "--only-3x3" in sys.argv:      # + the data-gradient shapes of the same convs and the 1.0x attention-head sizes
        shapes = [s for s in shapes if s[4] == 3] + [(512, 720, 256, 256, 3), (512, 720, 128, 128, 3), (512, 256, 256, 256, 3),
                                                     (256, 256, 256, 256, 3), (256, 512, 128, 128, 3)]
    probs = [WProb(*s) for s in shapes]
    for p in probs:
        if "--fwd-only" in sys.argv:
            line = p.name
        else:
            t = timeit(p.wgrad, reps)
            line = "%s  wgrad %8.1f us %6.0f TF/s (splits %3d)" % (p.name, t, p.flops / t / 1e6, p.ns)
        if p.halo:
            try:
                t = timeit(p.fwd, reps)
                line += "   fwd %8.1f us %6.0f TF/s" % (t, p.flops / t / 1e6)
            except Exception as e:  # noqa: BLE001
                line += "   fwd failed: %s" % e
        print(line, flush=True)

    if "--fwd-only" in sys.argv:
        return

    def all3():
        with hb.group():
            for p in probs[:4]:
                p.wgrad()
    t = timeit(all3, reps)
    fl = sum(p.flops for p in probs[:4])
    print("grouped 3x3 wgrads: %.1f us %.0f TF/s" % (t, fl / t / 1e6))


if __name__ == "__main__":
    main()
