"""Shared helpers for the parity tests."""
import os

import torch


import os

# storage format of the product's 16-bit activations in THIS process (semseg_amd/_lib.py: SSA_ACT_DTYPE): the tests
# round their inputs / references to it, so the same test files run against the bf16 and the fp16 build
ACT_DTYPE = torch.float16 if os.environ.get("SSA_ACT_DTYPE", "bf16").lower() in ("fp16", "f16", "float16", "half") else torch.bfloat16


def bf16_round(t):
    """t rounded to the product's 16-bit storage format (bf16 by default; the name is round 1's)."""
    return t.to(ACT_DTYPE).to(torch.float32)


def report(name, got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    mx = err.max().item()
    scale = ref.abs().max().item()
    mean_err = err.mean().item()
    mean_ref = ref.abs().mean().item()
    idx = err.flatten().argmax().item()
    print("[%s] max_err=%.4g (ref_max=%.4g) mean_err=%.4g (ref_mean=%.4g) worst@%d got=%.6g ref=%.6g" % (
        name, mx, scale, mean_err, mean_ref, idx, got.flatten()[idx].item(), ref.flatten()[idx].item()))
    return mx, scale, mean_err, mean_ref


def check_close(name, got, ref, rel_max=1e-2, rel_mean=4e-3):
    """max error <= rel_max * max|ref| and mean error <= rel_mean * mean|ref|.
    Defaults are the bf16 tolerance: one bf16 rounding of the output is 2^-9
    relative (0.2%), plus fp32 accumulation-order noise."""
    assert torch.isfinite(got.detach().float()).all(), name + ": non-finite output"
    mx, scale, mean_err, mean_ref = report(name, got, ref)
    assert mx <= rel_max * scale + 1e-30, "%s: max err %.4g > %.4g" % (name, mx, rel_max * scale)
    assert mean_err <= rel_mean * mean_ref + 1e-30, "%s: mean err %.4g > %.4g" % (name, mean_err, rel_mean * mean_ref)


def check_close_robust(name, got, ref, rel_max=1e-2, rel_mean=4e-3, outliers=1e-4):
    """check_close for gradients that pass through a ReLU mask recomputed from bf16 data: where the
    pre-activation is within rounding of zero the mask may fall on the other side than the oracle's, and
    such an element's gradient differs by its full magnitude.  At most `outliers` of the elements may
    exceed the max-error bound (a wrong tile edge or channel block is orders of magnitude more)."""
    assert torch.isfinite(got.detach().float()).all(), name + ": non-finite output"
    if os.environ.get("SSA_EMU"):
        # the emulation sums an MFMA's 16 products in index order, the hardware does not: a different handful of
        # pre-activations lands on the other side of zero (2 elements of a 9,216-element problem are 2e-4)
        outliers = max(outliers, 5e-4)
    mx, scale, mean_err, mean_ref = report(name, got, ref)
    err = (got.detach().float().cpu() - ref.detach().float().cpu()).abs()
    frac = float((err > rel_max * scale).float().mean())
    assert frac <= outliers, "%s: %.3g of the elements exceed %.4g" % (name, frac, rel_max * scale)
    assert mean_err <= rel_mean * mean_ref + 1e-30, "%s: mean err %.4g > %.4g" % (name, mean_err, rel_mean * mean_ref)


def nhwc(t):
    """NCHW -> NHWC contiguous"""
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()
