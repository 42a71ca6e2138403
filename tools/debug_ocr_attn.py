#!/usr/bin/env python
"""Debug aid: ocr_attention (network/ocr_utils.py:95-119) on the HIP path against fp32 torch on the device, at the
evaluation shape where the fp16 build's teacher-forced test saw isolated 3 % errors; prints where the worst element
is and what the operands look like there.  SSA_ACT_DTYPE selects the storage format."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from semseg_amd import hip_backend as hb
    dt = hb.ACT_DTYPE
    g = torch.Generator().manual_seed(3)
    H, W, D, K = 256, 512, 256, 19
    for qs, ks in ((1.0, 1.0), (4.0, 1.0), (1.0, 6.0), (8.0, 8.0)):
        q = (torch.relu(torch.randn(1, H, W, D, generator=g)) * qs).cuda().to(dt)
        k = (torch.relu(torch.randn(1, K, D, generator=g)) * ks).cuda().to(dt)
        v = (torch.relu(torch.randn(1, K, D, generator=g)) * ks).cuda().to(dt)
        scale = D ** -0.5
        out = hb.OcrAttnFn.apply(q, k, v, scale).float()
        sim = torch.matmul(q.float().view(1, H * W, D), k.float().transpose(1, 2)) * scale
        p = torch.softmax(sim, dim=-1)
        ref = torch.matmul(p, v.float()).view(1, H, W, D)
        err = (out - ref).abs()
        i = int(err.argmax())
        pix, d = divmod(i, D)
        print("%s q*%g k,v*%g: max err %.4g at pixel %d ch %d (ref %.4g got %.4g) max|ref| %.4g mean err %.3g; sim range [%.3g, %.3g]; finite %s"
              % (dt, qs, ks, float(err.max()), pix, d, float(ref.view(-1)[i]), float(out.view(-1)[i]), float(ref.abs().max()),
                 float(err.mean()), float(sim.min()), float(sim.max()), bool(torch.isfinite(out).all())))
        print("   probs at that pixel:", [round(float(x), 5) for x in p.view(H * W, K)[pix]])


if __name__ == "__main__":
    main()
