#!/usr/bin/env python
"""Micro-benchmark of the backward of the few-channel fp32 upsampling resizes of a 1024 x 1024 training step (class
logits 256^2 -> 1024^2, 128^2 -> 512^2, 512^2 -> 1024^2; the attention map), through the C ABI, each shape replayed from
a hipGraph.  python tools/bilinbench.py [reps]      SSA_BILINEAR_BWD_TILE=0: the per-element gather."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from semseg_amd import hip_backend as hb  # noqa: E402

L = hb.lib()
P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print("SSA_BILINEAR_BWD_TILE =", os.environ.get("SSA_BILINEAR_BWD_TILE", "1"))
for C, hi, wi, ho, wo in [(19, 256, 256, 1024, 1024), (19, 128, 128, 512, 512), (19, 512, 512, 1024, 1024),
                          (1, 128, 128, 1024, 1024), (1, 512, 512, 1024, 1024), (65, 128, 128, 512, 512)]:
    dy = torch.randn(1, ho, wo, C, device="cuda")
    dx = torch.empty(1, hi, wi, C, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run = lambda: hb.check(L.ssa_bilinear_bwd(P(dy), 1, 1, ho, wo, C, C, P(dx), 1, hi, wi, C, ctypes.c_void_p(s.cuda_stream)), "bwd")  # noqa: E731
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                run()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    mb = (dy.numel() + dx.numel()) * 4 / 1e6
    print("C=%2d %4dx%-4d <- %4dx%-4d  %7.1f us  %6.1f MB  %5.2f TB/s" % (C, hi, wi, ho, wo, us, mb, mb / us / 1e6 * 1e6 / 1e6))
