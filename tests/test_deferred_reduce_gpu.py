"""ssa_conv2d_wgrad_reduce_batched (the deferred, batched form of the per-layer split-K reduce)
against ssa_conv2d_wgrad_reduce: bit-identical by construction (same summation order); and a
training step with SSA_DEFER_WGRAD_REDUCE against the same step without it."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

# not yet run on hardware (round-1 GPU budget): opt-in until it has
unverified = pytest.mark.skipif(os.environ.get("SSA_TEST_UNVERIFIED", "0") != "1",
                                reason="not yet run on hardware (round-1 GPU budget); set SSA_TEST_UNVERIFIED=1")


@unverified
def test_batched_reduce_is_bit_identical():
    from semseg_amd._lib import lib, check, WgradReduceJob
    L = lib()
    g = torch.Generator().manual_seed(5)
    # (nsplit, cout_pad, Cout, Cin_pad, Cin, KH, KW): odd sizes, padding, 1x1 / 3x3 / 7x7, > 72 jobs
    shapes = [(92, 48, 48, 48, 48, 3, 3), (1, 24, 19, 512, 512, 1, 1), (7, 64, 64, 16, 3, 7, 7),
              (13, 512, 512, 720, 720, 3, 3), (5, 8, 1, 256, 256, 1, 1), (33, 96, 96, 48, 48, 3, 3)]
    shapes += [(3 + i % 9, 48, 48, 48, 48, 3, 3) for i in range(80)]
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    jobs, keep, want = [], [], []
    for ns, cp, co, cip, ci, kh, kw in shapes:
        partial = torch.randn(ns, cp, kh * kw * cip, generator=g).cuda()
        ref = torch.full((co, ci, kh, kw), float("nan"), device="cuda")
        out = torch.full((co, ci, kh, kw), float("nan"), device="cuda")
        check(L.ssa_conv2d_wgrad_reduce(partial.data_ptr(), ns, cp, co, cip, ci, kh, kw, ref.data_ptr(), stream),
              "ssa_conv2d_wgrad_reduce")
        jobs.append(WgradReduceJob(partial.data_ptr(), out.data_ptr(), ns, cp, co, cip, ci, kh, kw, 0))
        keep.append((partial, out))
        want.append(ref)
    arr = (WgradReduceJob * len(jobs))(*jobs)
    check(L.ssa_conv2d_wgrad_reduce_batched(arr, len(jobs), stream), "ssa_conv2d_wgrad_reduce_batched")
    torch.cuda.synchronize()
    for (partial, out), ref, shp in zip(keep, want, shapes):
        assert torch.isfinite(ref).all(), shp
        assert torch.equal(out, ref), shp
        # and both equal the plain sum over the splits, to fp32 summation noise
        ns, cp, co, cip, ci, kh, kw = shp
        plain = partial.sum(0)[:co].view(co, kh * kw, cip)[:, :, :ci].permute(0, 2, 1).reshape(co, ci, kh, kw)
        assert torch.allclose(out, plain, rtol=1e-4, atol=1e-4 * ns ** 0.5), shp


@unverified
def test_training_step_with_deferred_reduces(monkeypatch):
    from semseg_amd import hip_backend, ops
    from semseg_amd.config import cfg
    from semseg_amd.loss import RMILoss
    from semseg_amd.network import ocrnet
    from test_e2e_gpu import _synth
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05
    cfg.MODEL.N_SCALES = None
    images, gts = _synth(1, 256, 256, seed=3)
    inputs = {"images": images.cuda(), "gts": gts.cuda()}
    prev = ops._BACKEND
    ops._set_backend_for_tests(ops.HipBackend())
    try:
        grads = []
        for defer in (False, True):
            monkeypatch.setattr(hip_backend, "_DEFER_WGRAD_REDUCE", defer)
            hip_backend.clear_pack_cache()
            torch.manual_seed(0)
            net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255)).cuda().train()
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout2d):
                    m.p = 0.0
            loss = net(inputs)
            loss.backward()
            torch.cuda.synchronize()
            grads.append((float(loss), {n: p.grad.clone() for n, p in net.named_parameters()}))
    finally:
        ops._set_backend_for_tests(prev)
        cfg.LOSS.SUPERVISED_MSCALE_WT = 0
    (l0, g0), (l1, g1) = grads
    assert abs(l0 - l1) <= 1e-4 * abs(l0)
    rel = sorted(float((g1[n] - g0[n]).norm() / (g0[n].norm() + 1e-30)) for n in g0)
    print("deferred vs per-layer reduce: loss %.6f / %.6f, relative gradient difference median %.3g p99 %.3g max %.3g" % (
        l0, l1, rel[len(rel) // 2], rel[len(rel) * 99 // 100], rel[-1]))
    # same kernels, same summation order (the unit test above shows bit-identity of the reduce itself); what
    # differs between two runs of the step is the order of the fp64 atomics of the BN sums, which a
    # random-weight network can amplify in single layers
    assert rel[len(rel) // 2] < 1e-4 and rel[len(rel) * 99 // 100] < 1e-2
