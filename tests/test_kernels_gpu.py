"""Op-level parity: every HIP kernel (through the C ABI) against the oracle
(oracle/ops.py, CPU fp32/fp64) on identical seeded inputs.

Tolerance (stated once): activations are bf16, accumulation fp32.  Inputs and
weights are rounded to bf16 before BOTH paths, so the only differences are the
fp32 accumulation order and one bf16 rounding of the output (2^-9 relative):
max error <= 1e-2 * max|ref|, mean error <= 4e-3 * mean|ref|.  fp32-out ops
use 2e-3 / 5e-4, fp64 RMI statistics 1e-4.
"""
import math
import os

import pytest
import torch

from util import bf16_round, check_close, report, nhwc, nchw, ACT_DTYPE

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _hb():
    from semseg_amd import hip_backend
    return hip_backend


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_round(torch.randn(*shape, generator=g) * scale)


# ----------------------------------------------------------------- probes
def test_probe_mfma32(out_dir):
    from semseg_amd._lib import lib, check
    import ctypes
    a = _rand(32, 16, seed=1)
    b = _rand(16, 32, seed=2)       # asymmetric on purpose
    a_d = a.to(DEV).to(ACT_DTYPE).contiguous()
    bt_d = b.t().contiguous().to(DEV).to(ACT_DTYPE)
    c_d = torch.zeros(32, 32, device=DEV)
    check(lib().ssa_probe_mfma32(ctypes.c_void_p(a_d.data_ptr()), ctypes.c_void_p(bt_d.data_ptr()),
                                 ctypes.c_void_p(c_d.data_ptr()), None), "probe")
    torch.cuda.synchronize()
    check_close("mfma32", c_d, a @ b, 1e-5, 1e-5)


def test_probe_mfma16():
    """Lane map of v_mfma_f32_16x16x32 (csrc/common.h ssa_mfma16): A rows / B columns l & 15, k = 8 * (l >> 4) + j;
    accumulator column l & 15, rows 4 * (l >> 4) + j."""
    import ctypes
    from semseg_amd._lib import lib, check
    a = _rand(16, 32, seed=1)
    b = _rand(32, 16, seed=2)       # asymmetric on purpose
    a_d = a.to(DEV).to(ACT_DTYPE).contiguous()
    bt_d = b.t().contiguous().to(DEV).to(ACT_DTYPE)
    c_d = torch.zeros(16, 16, device=DEV)
    check(lib().ssa_probe_mfma16(ctypes.c_void_p(a_d.data_ptr()), ctypes.c_void_p(bt_d.data_ptr()),
                                 ctypes.c_void_p(c_d.data_ptr()), None), "probe")
    torch.cuda.synchronize()
    check_close("mfma16", c_d, a @ b, 1e-5, 1e-5)


def test_probe_swap16():
    """v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second."""
    import ctypes
    from semseg_amd._lib import lib, check
    out = torch.zeros(64, 2, dtype=torch.int32, device=DEV)
    check(lib().ssa_probe_swap16(ctypes.c_void_p(out.data_ptr()), None), "probe")
    torch.cuda.synchronize()
    want = []
    for l in range(64):
        if (l >> 4) & 1 == 0:
            want.append((l, l ^ 16))                  # (own a, partner's a)
        else:
            want.append(((l ^ 16) + 100, l + 100))    # (partner's b, own b)
    assert out.cpu().tolist() == [list(t) for t in want]


def test_probe_tr16(out_dir):
    from semseg_amd._lib import lib, check
    import ctypes
    with open(os.path.join(out_dir, "probe_tr16.txt"), "w") as f:
        for mode in (0, 1, 2):
            out = torch.zeros(256, dtype=torch.int16, device=DEV)
            check(lib().ssa_probe_tr16(ctypes.c_void_p(out.data_ptr()), mode, None), "probe_tr16")
            torch.cuda.synchronize()
            v = out.cpu().view(64, 4).tolist()
            f.write("mode %d\n" % mode)
            for l, row in enumerate(v):
                f.write("lane %2d: %s\n" % (l, row))
    assert True


# ------------------------------------------------------------------- conv
CONV_CASES = [
    # B, H, W, Cin(real), Cout, k, stride, pad, dil, bias, out_f32
    (2, 40, 56, 48, 48, 3, 1, 1, 1, False, False),
    (1, 64, 64, 3, 64, 3, 2, 1, 1, False, False),
    (2, 33, 47, 64, 256, 1, 1, 0, 1, False, False),
    (1, 48, 40, 96, 192, 3, 2, 1, 1, False, False),
    (1, 24, 24, 720, 512, 3, 1, 1, 1, True, False),
    (2, 32, 32, 512, 19, 1, 1, 0, 1, True, True),
    (1, 32, 32, 256, 1, 1, 1, 0, 1, False, True),
    (1, 40, 40, 64, 32, 3, 1, 12, 12, False, False),
    (1, 16, 16, 384, 384, 3, 1, 1, 1, False, False),
    (1, 37, 29, 96, 96, 3, 1, 1, 1, False, False),
    (1, 19, 1, 512, 256, 1, 1, 0, 1, False, False),
    # DeepLabV3+/ResNet-50 shapes: 7x7 stride-2 stem, strided 1x1 downsample, ASPP dilations
    (1, 64, 80, 3, 64, 7, 2, 3, 1, False, False),
    (1, 24, 32, 256, 512, 1, 2, 0, 1, False, False),
    (2, 12, 16, 512, 256, 3, 1, 24, 24, False, False),
    (1, 12, 16, 2048, 256, 1, 1, 0, 1, False, False),
    (1, 20, 28, 128, 128, 3, 1, 2, 2, False, False),
    # halo-tile kernel (conv_tile.hip): every (Cin, n-block, tile-width) dispatch class
    (1, 70, 75, 64, 64, 3, 1, 1, 1, False, False),
    (1, 130, 129, 48, 48, 3, 1, 1, 1, True, False),
    (2, 20, 12, 48, 96, 3, 1, 1, 1, False, False),
    (1, 9, 7, 96, 48, 3, 1, 1, 1, False, False),
    (1, 33, 18, 64, 24, 3, 1, 1, 1, False, False),
    (1, 128, 160, 96, 96, 3, 1, 1, 1, False, False),
    (1, 33, 30, 192, 96, 3, 1, 1, 1, False, False),
    (2, 20, 12, 384, 200, 3, 1, 1, 1, True, False),
    (1, 64, 64, 192, 192, 3, 1, 1, 1, False, False),
    # halo-chunk GEMM (conv_halo_gemm.hip): CK 48 / 64, 3x3 / 1x1, ragged tiles, partial n-blocks
    (1, 130, 131, 192, 200, 3, 1, 1, 1, True, False),
    (1, 128, 129, 240, 128, 3, 1, 1, 1, False, False),
    (2, 96, 100, 256, 72, 1, 1, 0, 1, False, False),
    (1, 140, 128, 336, 64, 1, 1, 0, 1, True, False),
    # 3x3 stride-2 down-convs: data gradient by output parity (ssa_conv2d_dgrad_s2) -- odd / even extents,
    # Cout that is no multiple of 32, one-pixel-wide classes
    (1, 37, 45, 48, 96, 3, 2, 1, 1, False, False),
    (2, 33, 64, 96, 200, 3, 2, 1, 1, True, False),
    (1, 128, 128, 64, 64, 3, 2, 1, 1, False, False),
    (1, 2, 3, 48, 24, 3, 2, 1, 1, False, False),
    (1, 1, 9, 48, 48, 3, 2, 1, 1, False, False),
    # the scale-attention head's last conv (network/utils.py:360: 256 -> 1, fp32 out) at the three pass sizes of a
    # 128 x 192 {0.5, 1, 2} evaluation -- round 4's fp16 end-to-end test pointed at the smallest one
    (1, 16, 24, 256, 1, 1, 1, 0, 1, False, True),
    (1, 32, 48, 256, 1, 1, 1, 0, 1, False, True),
    (1, 64, 96, 256, 1, 1, 1, 0, 1, False, True),
]


def _conv_inputs(case, seed=0):
    B, H, W, Cin, Cout, k, s, p, d, bias, out_f32 = case
    x = _rand(B, Cin, H, W, seed=seed)
    w = _rand(Cout, Cin, k, k, seed=seed + 1, scale=1.0 / math.sqrt(Cin * k * k))
    b = _rand(Cout, seed=seed + 2) if bias else None
    return x, w, b


def _to_dev_nhwc(x, cpad=None):
    t = nhwc(x)
    if cpad is not None and cpad > t.shape[3]:
        t = torch.nn.functional.pad(t, (0, cpad - t.shape[3]))
    return t.to(DEV).to(ACT_DTYPE).contiguous()


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case):
    from oracle import ops as O
    hb = _hb()
    B, H, W, Cin, Cout, k, s, p, d, bias, out_f32 = case
    x, w, b = _conv_inputs(case)
    cin_pad = (Cin + 7) // 8 * 8 if Cin >= 8 else 16
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if b is not None else None
    yr = O.conv2d(xr, wr, br, s, p, d)
    gy = _rand(*yr.shape, seed=7)
    yr.backward(gy)

    xd = _to_dev_nhwc(x, cin_pad).requires_grad_(cin_pad == Cin)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if b is not None else None
    hb.clear_pack_cache()
    yd = hb.Conv2dFn.apply(xd, wd, bd, s, p, d, out_f32)
    torch.cuda.synchronize()
    tol = (2e-3, 5e-4) if out_f32 else (1e-2, 4e-3)
    check_close("conv_fwd %s" % (case,), nchw(yd.float()), yr, *tol)
    gyd = nhwc(gy).to(DEV)
    if not out_f32:
        gyd = gyd.to(ACT_DTYPE)
    yd.backward(gyd)
    torch.cuda.synchronize()
    if cin_pad == Cin:
        check_close("conv_dgrad %s" % (case,), nchw(xd.grad.float()), xr.grad)
    # weight grads: fp32 out, but dy is rounded to bf16 for the fp32 heads
    check_close("conv_wgrad %s" % (case,), wd.grad, wr.grad, 1e-2, 4e-3)
    if b is not None:
        check_close("conv_bgrad %s" % (case,), bd.grad, br.grad, 1e-2, 4e-3)


WIDE_CASES = [
    # B, H, W, Cin, Cout, bias, stats, transposed-pack (the data-gradient form: ssa_pack_filter mode 3)
    (1, 16, 32, 96, 264, False, True, False),      # two pixel tiles x two channel tiles, the second 8 channels wide
    (1, 9, 31, 80, 520, True, True, False),        # ragged pixel tile (279 pixels), Cin = 2.5 stages, three channel tiles
    (2, 16, 16, 64, 256, True, False, False),      # one channel tile, exactly two stages (the ring's clamped prefetches)
    (1, 20, 16, 272, 328, False, True, True),      # data-gradient packing, 8.5 stages (ring wraps twice), n-block tail
]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_gemm_wide_1x1(case):
    """csrc/conv_gemm_wide.hip through its own entry point (ssa_conv2d_halo forwards only the large head problems to
    it): every tile / stage / ring edge at sizes the CPU emulation runs in seconds -- ragged pixel tile, Cin that is no
    multiple of the 32-channel stage, channel-tile and n-block tails, bias, the BatchNorm partial sums of the ROUNDED
    outputs, both filter packings."""
    import ctypes
    from oracle import ops as O
    from semseg_amd._lib import check
    hb = _hb()
    B, H, W, Cin, Cout, bias, stats, tr = case
    x = _rand(B, Cin, H, W, seed=3)
    w = _rand(Cout, Cin, 1, 1, seed=4, scale=1.0 / math.sqrt(Cin))
    b = _rand(Cout, seed=5) if bias else None
    ref = O.conv2d(x, w, b, 1, 0, 1)
    xd = _to_dev_nhwc(x)
    hb.clear_pack_cache()
    if tr:      # the weight of the conv whose DATA gradient this is: [Cin_fwd = Cout here][Cout_fwd = Cin here]
        wt = w[:, :, 0, 0].t().contiguous().view(Cin, Cout, 1, 1).to(xd.device)
        wp, _ = hb._packed_filter(wt, 3, 0, Cin)
    else:
        wp, _ = hb._packed_filter(w.to(xd.device), 2, Cin, 0)
    d = hb._tile_desc(B, H, W, Cin, Cin, Cout, (1, 1), 1, 0, 1, H, W, False)
    L = hb.lib()
    assert L.ssa_conv2d_gemm_wide_supported(ctypes.byref(d)) == 0      # (too small for the dispatcher: direct call)
    y = torch.empty(B, H, W, Cout, dtype=ACT_DTYPE, device=xd.device)
    nrep = hb.stat_replicas()
    st = torch.zeros(nrep, 2, Cout, dtype=torch.float64, device=xd.device) if stats else None
    bd = b.to(xd.device) if b is not None else None
    check(L.ssa_conv2d_gemm_wide(ctypes.byref(d), hb._p(xd), hb._p(wp), hb._p(bd), hb._p(y), hb._p(st), hb._s()), "wide")
    if xd.is_cuda:
        torch.cuda.synchronize()
    check_close("gemm_wide %s" % (case,), nchw(y.float()), ref)
    if stats:
        yr = y.float().view(-1, Cout).double()
        got = st.sum(0).cpu()
        check_close("gemm_wide sums %s" % (case,), got[0], yr.sum(0).cpu(), 1e-4, 1e-4)
        check_close("gemm_wide squares %s" % (case,), got[1], (yr * yr).sum(0).cpu(), 1e-4, 1e-4)


HALO_REG_CASES = [
    # B, H, W, Cin, Cout, bias, stats, transposed-pack (the data-gradient form: ssa_pack_filter mode 3)
    (1, 8, 64, 128, 264, True, True, False),       # 2 x 2 pixel tiles, two 64-channel chunks, two channel tiles (2nd: 8 wide)
    (1, 7, 37, 96, 72, False, True, False),        # ragged tile rows / columns, two 48-channel chunks, n-block tail
    (2, 4, 32, 192, 256, True, False, False),      # batch 2, three 64-channel chunks (odd count), one full channel tile
    (1, 9, 33, 144, 320, False, True, True),       # data-gradient packing, three 48-channel chunks, image edge columns
]


@pytest.mark.parametrize("case", HALO_REG_CASES)
def test_halo_reg_3x3(case):
    """csrc/conv_halo_reg.hip through its own entry point (ssa_conv2d_halo forwards the head's 3x3 problems to it): the
    register-fed filter ring, the masked halo DMA (image borders, ragged tiles, the unused pieces of a 48-channel chunk),
    both chunk widths with even / odd chunk counts, channel-tile and n-block tails, bias, the BatchNorm partial sums of the
    ROUNDED outputs, both filter packings -- at sizes the CPU emulation runs in seconds."""
    import ctypes
    from oracle import ops as O
    from semseg_amd._lib import check
    hb = _hb()
    B, H, W, Cin, Cout, bias, stats, tr = case
    x = _rand(B, Cin, H, W, seed=3)
    w = _rand(Cout, Cin, 3, 3, seed=4, scale=1.0 / math.sqrt(9 * Cin))
    b = _rand(Cout, seed=5) if bias else None
    ref = O.conv2d(x, w, b, 1, 1, 1)
    xd = _to_dev_nhwc(x)
    hb.clear_pack_cache()
    if tr:      # the weight of the conv whose DATA gradient this is: flipped taps, [Cin_fwd = Cout here][Cout_fwd = Cin here]
        wt = w.flip(2, 3).permute(1, 0, 2, 3).contiguous().to(xd.device)
        wp, _ = hb._packed_filter(wt, 3, 0, Cin)
    else:
        wp, _ = hb._packed_filter(w.to(xd.device), 2, Cin, 0)
    d = hb._tile_desc(B, H, W, Cin, Cin, Cout, (3, 3), 1, 1, 1, H, W, False)
    L = hb.lib()
    assert L.ssa_conv2d_halo_reg_supported(ctypes.byref(d)) == 1
    y = torch.empty(B, H, W, Cout, dtype=ACT_DTYPE, device=xd.device)
    nrep = hb.stat_replicas()
    st = torch.zeros(nrep, 2, Cout, dtype=torch.float64, device=xd.device) if stats else None
    bd = b.to(xd.device) if b is not None else None
    check(L.ssa_conv2d_halo_reg(ctypes.byref(d), hb._p(xd), hb._p(wp), hb._p(bd), hb._p(y), hb._p(st), hb._s()), "halo_reg")
    if xd.is_cuda:
        torch.cuda.synchronize()
    check_close("halo_reg %s" % (case,), nchw(y.float()), ref)
    if stats:
        yr = y.float().view(-1, Cout).double()
        got = st.sum(0).cpu()
        check_close("halo_reg sums %s" % (case,), got[0], yr.sum(0).cpu(), 1e-4, 1e-4)
        check_close("halo_reg squares %s" % (case,), got[1], (yr * yr).sum(0).cpu(), 1e-4, 1e-4)


def test_stride2_dgrad_by_parity_matches_zero_inserted():
    """ssa_conv2d_dgrad_s2 (four dense parity classes) against the zero-inserted transposed form it
    replaces: same operands, same bf16 rounding of the result; the two differ only in the order of the
    fp32 accumulation (taps that multiply inserted zeros contribute nothing)."""
    hb = _hb()
    for (B, H, W, Cin, Cout) in [(1, 64, 64, 48, 96), (2, 31, 50, 96, 96), (1, 17, 16, 192, 384)]:
        hb.clear_pack_cache()
        w = _rand(Cout, Cin, 3, 3, seed=11, scale=0.05).to(DEV)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dy = nhwc(_rand(B, Cout, Ho, Wo, seed=12)).to(DEV).to(ACT_DTYPE).contiguous()
        assert hb._DGRAD_S2
        got = hb._conv_dgrad((B, H, W, Cin), w, dy, Cout, Cout, 2, 1, 1, (Ho, Wo))
        hb._DGRAD_S2 = False
        try:
            want = hb._conv_dgrad((B, H, W, Cin), w, dy, Cout, Cout, 2, 1, 1, (Ho, Wo))
        finally:
            hb._DGRAD_S2 = True
        torch.cuda.synchronize()
        assert got.shape == want.shape == (B, H, W, Cin)
        check_close("dgrad_s2 %s" % ((B, H, W, Cin, Cout),), got.float(), want.float(), 8e-3, 1e-3)
        # inside a group bracket: the four classes leave as one launch per tile instantiation
        hb.lib().ssa_launch_count(1)
        with hb.group():
            got2 = hb._conv_dgrad((B, H, W, Cin), w, dy, Cout, Cout, 2, 1, 1, (Ho, Wo))
        torch.cuda.synchronize()
        assert hb.lib().ssa_launch_count(1) <= (1 if H % 2 == 0 and W % 2 == 0 else 4)
        assert torch.equal(got2.view(torch.int16), got.view(torch.int16))
    hb.clear_pack_cache()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5])
def test_conv_all_tile_configs(cfg):
    """Every tile configuration on a shape with ragged M and N tails."""
    import ctypes
    from oracle import ops as O
    hb = _hb()
    B, H, W, Cin, Cout = 1, 37, 45, 72, 88
    x = _rand(B, Cin, H, W, seed=3)
    w = _rand(Cout, Cin, 3, 3, seed=4, scale=0.04)
    yr = O.conv2d(x, w, None, 1, 1, 1)
    xd = _to_dev_nhwc(x)
    wd = w.to(DEV)
    hb.clear_pack_cache()
    wp, Kpad = hb._packed_filter(wd, 0, Cin, 0)
    y = hb._igemm(xd, Cin, (B, H, W, Cin), wp, Kpad, None, (H, W), Cout, (3, 3), 1, 1, 1, False, False, cfg=cfg)
    torch.cuda.synchronize()
    check_close("conv cfg%d" % cfg, nchw(y.float()), yr)


@pytest.mark.parametrize("C,H,W", [(48, 50, 70), (96, 17, 33), (64, 128, 128), (192, 96, 96)])
def test_conv_bn_fused_stats(C, H, W):
    """conv -> BN (training) with the batch statistics accumulated in the conv
    epilogue (HipBackend.conv_bn_act) == oracle conv followed by batch_norm."""
    from oracle import ops as O
    from semseg_amd import ops, nn as snn
    hb = _hb()
    B = 2
    x = _rand(B, C, H, W, seed=11)
    conv = snn.Conv2d(C, C, 3, 1, 1, bias=False)
    bn = snn.BatchNorm2d(C)
    with torch.no_grad():
        conv.weight.copy_(_rand(C, C, 3, 3, seed=12, scale=0.05))
        bn.weight.copy_(torch.rand(C) + 0.5)
        bn.bias.copy_(torch.randn(C) * 0.1)
    rm, rv = torch.zeros(C), torch.ones(C)
    yr = O.conv2d(x, conv.weight.detach(), None, 1, 1, 1)
    zr = torch.relu(O.batch_norm(bf16_round(yr), bn.weight.detach(), bn.bias.detach(), rm, rv, True, 0.1, 1e-5))
    conv, bn = conv.to(DEV), bn.to(DEV).train()
    hb.clear_pack_cache()
    be = ops.HipBackend()
    hb.begin_step(torch.device(DEV))
    z = be.conv_bn_act(conv, bn, _to_dev_nhwc(x), relu=True)
    be.end_forward()
    torch.cuda.synchronize()
    assert not hb._PENDING_STATS                 # consumed by the normalisation
    check_close("fused conv-bn", nchw(z.float()), zr, 2e-2, 6e-3)
    check_close("fused running_mean", bn.running_mean, rm, 2e-3, 2e-3)
    check_close("fused running_var", bn.running_var, rv, 2e-3, 2e-3)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("C,B,H,W", [(48, 1, 37, 45), (64, 2, 20, 33), (96, 1, 64, 64), (192, 1, 21, 40), (384, 1, 9, 33)])
def test_wgrad_tile_kernel(C, B, H, W):
    """Halo-staged weight gradient (conv_wgrad_tile.hip, ds_read_b64_tr_b16 fragments):
    every channel configuration against the oracle, through the C ABI (the host glue routes the
    trunk's 3x3 weight gradients here, grouped: tests/test_group_gpu.py)."""
    import ctypes
    from oracle import ops as O
    from semseg_amd._lib import lib, check, ConvDesc
    x = _rand(B, C, H, W, seed=41)
    gy = _rand(B, C, H, W, seed=42)
    w = torch.zeros(C, C, 3, 3, requires_grad=True)
    O.conv2d(x, w, None, 1, 1, 1).backward(gy)
    xd = _to_dev_nhwc(x)
    gd = nhwc(gy).to(DEV).to(ACT_DTYPE).contiguous()
    d = ConvDesc(B, H, W, C, C, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, -1)
    ns, ws = ctypes.c_int(0), ctypes.c_size_t(0)
    L = lib()
    check(L.ssa_conv2d_wgrad_tile_plan(ctypes.byref(d), C, ctypes.byref(ns), ctypes.byref(ws)), "plan")
    part = torch.empty(ws.value // 4, dtype=torch.float32, device=DEV)
    dw = torch.empty(C, C, 3, 3, dtype=torch.float32, device=DEV)
    P = ctypes.c_void_p
    check(L.ssa_conv2d_wgrad_tile(ctypes.byref(d), P(xd.data_ptr()), P(gd.data_ptr()), C, C, ns.value,
                                  P(part.data_ptr()), None), "ssa_conv2d_wgrad_tile")
    check(L.ssa_conv2d_wgrad_reduce(P(part.data_ptr()), ns.value, C, C, C, C, 3, 3, P(dw.data_ptr()), 0, None),
          "ssa_conv2d_wgrad_reduce")
    torch.cuda.synchronize()
    check_close("wgrad tile C=%d" % C, dw, w.grad, 2e-3, 5e-4)


def test_wgrad_tile_grouped_launch_of_twenty_layers():
    """One grouped launch carries up to 32 weight-gradient problems (csrc/group.h MAXJOBS of ConvWgradTile; 16 for every
    other kernel): twenty 48-channel layers of different sizes inside one bracket -> ONE tile launch, one reduce launch
    per 16 parameters, every layer's gradient against the oracle."""
    import ctypes
    from oracle import ops as O
    from semseg_amd._lib import lib, check, ConvDesc
    hb = _hb()
    C, n = 48, 20
    L = lib()
    P = ctypes.c_void_p
    keep, want = [], []
    plans = []
    for i in range(n):
        H, W = 5 + (i % 4) * 3, 33 + 7 * (i % 3)
        x = _rand(1, C, H, W, seed=300 + i)
        gy = _rand(1, C, H, W, seed=400 + i)
        w = torch.zeros(C, C, 3, 3, requires_grad=True)
        O.conv2d(x, w, None, 1, 1, 1).backward(gy)
        want.append(w.grad)
        xd, gd = _to_dev_nhwc(x), nhwc(gy).to(DEV).to(ACT_DTYPE).contiguous()
        d = ConvDesc(1, H, W, C, C, H, W, C, C, 3, 3, 1, 1, 1, 0, 0, 0, 2)        # strips of 2 tiles: several partials
        ns, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        check(L.ssa_conv2d_wgrad_tile_plan(ctypes.byref(d), C, ctypes.byref(ns), ctypes.byref(ws)), "plan")
        part = torch.empty(ws.value // 4, dtype=torch.float32, device=DEV)
        dw = torch.empty(C, C, 3, 3, dtype=torch.float32, device=DEV)
        keep += [xd, gd, part]
        plans.append((d, xd, gd, ns.value, part, dw))
    L.ssa_launch_count(1)
    with hb.group():
        for d, xd, gd, ns, part, dw in plans:
            check(L.ssa_conv2d_wgrad_tile(ctypes.byref(d), P(xd.data_ptr()), P(gd.data_ptr()), C, C, ns,
                                          P(part.data_ptr()), None), "ssa_conv2d_wgrad_tile")
    n_tile = L.ssa_launch_count(1)
    with hb.group():
        for d, xd, gd, ns, part, dw in plans:
            check(L.ssa_conv2d_wgrad_reduce(P(part.data_ptr()), ns, C, C, C, C, 3, 3, P(dw.data_ptr()), 0, None),
                  "ssa_conv2d_wgrad_reduce")
    n_red = L.ssa_launch_count(1)
    torch.cuda.synchronize()
    assert n_tile == 1 and n_red == 2, (n_tile, n_red)
    for i, (pl, ref) in enumerate(zip(plans, want)):
        check_close("grouped wgrad layer %d" % i, pl[5], ref, 2e-3, 5e-4)


@pytest.mark.parametrize("Cin,Cout,k,stride,H,W", [(48, 96, 3, 2, 50, 70), (64, 256, 1, 1, 33, 47), (96, 48, 1, 1, 20, 24)])
def test_conv_bn_fused_stats_igemm(Cin, Cout, k, stride, H, W):
    """Same as test_conv_bn_fused_stats for the shapes that run on the K-pipelined igemm kernel
    (strided 3x3 and small 1x1 convs of the fuse layers / layer1): statistics from its epilogue."""
    from oracle import ops as O
    from semseg_amd import ops, nn as snn
    hb = _hb()
    B = 2
    x = _rand(B, Cin, H, W, seed=51)
    conv = snn.Conv2d(Cin, Cout, k, stride, k // 2, bias=False)
    bn = snn.BatchNorm2d(Cout)
    with torch.no_grad():
        conv.weight.copy_(_rand(Cout, Cin, k, k, seed=52, scale=0.05))
        bn.weight.copy_(torch.rand(Cout) + 0.5)
        bn.bias.copy_(torch.randn(Cout) * 0.1)
    rm, rv = torch.zeros(Cout), torch.ones(Cout)
    yr = O.conv2d(x, conv.weight.detach(), None, stride, k // 2, 1)
    zr = O.batch_norm(bf16_round(yr), bn.weight.detach(), bn.bias.detach(), rm, rv, True, 0.1, 1e-5)
    conv, bn = conv.to(DEV), bn.to(DEV).train()
    hb.clear_pack_cache()
    be = ops.HipBackend()
    hb.begin_step(torch.device(DEV))
    z = be.conv_bn_act(conv, bn, _to_dev_nhwc(x), relu=False)
    be.end_forward()
    torch.cuda.synchronize()
    assert not hb._PENDING_STATS
    check_close("fused igemm conv-bn", nchw(z.float()), zr, 2e-2, 6e-3)
    check_close("fused igemm running_mean", bn.running_mean, rm, 2e-3, 2e-3)
    check_close("fused igemm running_var", bn.running_var, rv, 2e-3, 2e-3)


def test_batched_filter_repack():
    """refresh_packed_filters (one launch for all stale filters) == per-filter packing."""
    hb = _hb()
    hb.clear_pack_cache()
    ws = [torch.randn(48, 48, 3, 3, device=DEV), torch.randn(19, 512, 1, 1, device=DEV),
          torch.randn(96, 48, 3, 3, device=DEV), torch.randn(64, 3, 3, 3, device=DEV),
          torch.randn(512, 720, 3, 3, device=DEV), torch.randn(40, 300, 1, 1, device=DEV),
          torch.randn(24, 3, 7, 7, device=DEV), torch.randn(384, 192, 3, 3, device=DEV)]
    specs = [(0, 48, 0), (1, 0, 48), (0, 512, 0), (1, 0, 24), (0, 48, 0), (1, 0, 96), (0, 16, 0),
             (2, 48, 0), (3, 0, 48), (2, 512, 0), (2, 48, 0), (3, 0, 96), (2, 16, 0),     # fragment-major forms
             (0, 720, 0), (1, 0, 512), (2, 720, 0), (3, 0, 512), (0, 304, 0), (1, 0, 40), (0, 8, 0), (1, 0, 24),
             (0, 192, 0), (1, 0, 384), (2, 192, 0), (3, 0, 384)]
    owners = [0, 0, 1, 1, 2, 2, 3, 0, 0, 1, 2, 2, 3, 4, 4, 4, 4, 5, 5, 6, 6, 7, 7, 7, 7]
    first = [hb._packed_filter(ws[o], *sp)[0] for o, sp in zip(owners, specs)]
    for w in ws:
        w.mul_(-0.5).add_(0.25)                # in-place update, like an optimizer step
    hb.refresh_packed_filters()
    torch.cuda.synchronize()
    tab, = hb._JOB_TABLES.values()
    assert tab["tiles"] is not None and 300 < tab["ntiles"] < 600   # the tile-balanced kernel ran, ONE tile list per
    # source tensor: the 25 operand forms of the 8 tensors are chained (600+ tiles if every form fetched its own)
    batched = [t.clone() for t in first]       # same persistent buffers, refreshed in place
    # ... and the step's form: the first filters on this stream, the rest on a side stream, joined by the first lookup
    # of a late filter (begin_step -> refresh_packed_filters(overlap=True)); same bits as the one-launch form
    saved = [w.clone() for w in ws]
    for w in ws:
        w.mul_(2.0).sub_(0.125)
    early = hb._PACK_EARLY
    hb._PACK_EARLY = 5
    try:
        hb.refresh_packed_filters(overlap=True)
        if ws[0].is_cuda:                                        # (the CPU emulation has no second stream: one launch)
            assert hb._PACK_SIDE["pending"] and len(hb._PACK_SIDE["late"]) == len(specs) - 5
            hb._packed_filter(ws[owners[2]], *specs[2])
            assert hb._PACK_SIDE["pending"]                      # an early filter: nothing to wait for
            hb._packed_filter(ws[owners[-1]], *specs[-1])
            assert not hb._PACK_SIDE["pending"]                  # a late one: the compute stream waits
    finally:
        hb._PACK_EARLY = early
    torch.cuda.synchronize()
    split = [t.clone() for t in first]
    for w in ws:
        w.add_(0.0)                                          # version bump only
    hb.refresh_packed_filters()
    torch.cuda.synchronize()
    for o, sp, got, one in zip(owners, specs, split, first):
        assert torch.equal(got.view(torch.int16), one.view(torch.int16)), ("two launches vs one", o, sp)
    assert any(not torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(split, batched))
    for w, w1 in zip(ws, saved):
        w.copy_(w1)
    hb.refresh_packed_filters()
    torch.cuda.synchronize()
    for got, b0 in zip(first, batched):
        assert torch.equal(got.view(torch.int16), b0.view(torch.int16))
    hb.clear_pack_cache()
    for o, sp, got in zip(owners, specs, batched):
        want = hb._packed_filter(ws[o], *sp)[0]
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (o, sp)
    hb.clear_pack_cache()


def test_conv_channel_slice_input():
    """Input given as a channel slice of a wider NHWC buffer (ld > C)."""
    from oracle import ops as O
    hb = _hb()
    B, H, W = 1, 20, 24
    full = _rand(B, 96, H, W, seed=5)
    w = _rand(64, 48, 3, 3, seed=6, scale=0.05)
    yr = O.conv2d(full[:, 48:96], w, None, 1, 1, 1)
    fd = _to_dev_nhwc(full)
    hb.clear_pack_cache()
    y = hb.Conv2dFn.apply(fd[..., 48:96], w.to(DEV), None, 1, 1, 1, False)
    torch.cuda.synchronize()
    check_close("conv slice", nchw(y.float()), yr)


# --------------------------------------------------------------------- BN
@pytest.mark.parametrize("C,relu,res,post", [(48, True, True, False), (96, True, False, False),
                                             (720, True, False, False), (256, False, False, False),
                                             (512, True, False, True)])
def test_bn_train(C, relu, res, post):
    from oracle import ops as O
    hb = _hb()
    B, H, W = 2, 17, 23
    x = _rand(B, C, H, W, seed=1) * 1.7 + 0.3
    x = bf16_round(x)
    gamma = torch.rand(C) + 0.5
    beta = torch.randn(C) * 0.1
    r = _rand(B, C, H, W, seed=2) if res else None
    pm = None
    if post:
        pm = (torch.rand(B, C) > 0.3).float() / 0.7
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    rr = r.clone().requires_grad_(True) if res else None
    y = O.batch_norm(xr, gr, br, rm, rv, True, 0.1, 1e-5)
    if res:
        y = y + rr
    if relu:
        y = torch.relu(y)
    if post:
        y = y * pm[:, :, None, None]
    gy = _rand(B, C, H, W, seed=3)
    y.backward(gy)

    xd = _to_dev_nhwc(x).requires_grad_(True)
    gd = gamma.to(DEV).requires_grad_(True)
    bd = beta.to(DEV).requires_grad_(True)
    rd = _to_dev_nhwc(r).requires_grad_(True) if res else None
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    pmd = pm.to(DEV) if post else None
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    z = hb.BatchNormActFn.apply(xd, gd, bd, rd, pmd, rmd, rvd, nbt, 0.1, 1e-5, True, relu, False, None)
    z.backward(nhwc(gy).to(DEV).to(ACT_DTYPE))
    torch.cuda.synchronize()
    check_close("bn_fwd", nchw(z.float()), y)
    check_close("bn_running_mean", rmd, rm, 1e-4, 1e-4)
    check_close("bn_running_var", rvd, rv, 1e-4, 1e-4)
    assert int(nbt) == 1
    check_close("bn_dx", nchw(xd.grad.float()), xr.grad, 2e-2, 6e-3)
    check_close("bn_dgamma", gd.grad, gr.grad, 1e-2, 4e-3)
    check_close("bn_dbeta", bd.grad, br.grad, 1e-2, 4e-3)
    if res:
        check_close("bn_dres", nchw(rd.grad.float()), rr.grad)


@pytest.mark.parametrize("C,res,level", [(48, True, False), (96, False, False), (48, True, True)])
def test_bn_bwd_fused_equals_two_launches(C, res, level):
    """The one-launch BatchNorm backward (reduce, grid-wide rendezvous, apply: csrc/bn.hip bn_bwd_fused_body) against
    the two-launch form on the same inputs: the sums are formed by the same per-workgroup partials in another order of
    fp64 atomics, so everything agrees to fp32 rounding of the coefficients (1e-5 relative; the 16-bit outputs then differ
    in at most the last bit); no workgroup may have timed out of the rendezvous.  `level`: several problems in one grouped
    launch, each with its own ticket."""
    import ctypes
    hb = _hb()
    if DEV != "cuda":
        pytest.skip("the rendezvous needs concurrently resident workgroups: not on the CPU emulation")
    shapes = [(1, 40, 56, C)] if not level else [(1, 64, 64, C), (1, 32, 32, 2 * C), (1, 16, 16, 4 * C)]

    def run(fused_on):
        prev = hb._BN_FUSED_BWD
        hb._BN_FUSED_BWD = fused_on
        try:
            outs = []
            xs, gs, bs, rs, zs = [], [], [], [], []
            for k, (B, H, W, c) in enumerate(shapes):
                x = _to_dev_nhwc(_rand(B, c, H, W, seed=10 + k) * 1.3 + 0.2).requires_grad_(True)
                g = (torch.rand(c, generator=torch.Generator().manual_seed(20 + k)) + 0.5).to(DEV).requires_grad_(True)
                b = (torch.randn(c, generator=torch.Generator().manual_seed(30 + k)) * 0.1).to(DEV).requires_grad_(True)
                r = _to_dev_nhwc(_rand(B, c, H, W, seed=40 + k)).requires_grad_(True) if res else None
                xs.append(x); gs.append(g); bs.append(b); rs.append(r)
            for k, (B, H, W, c) in enumerate(shapes):
                rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
                nbt = torch.zeros((), dtype=torch.long, device=DEV)
                zs.append(hb.BatchNormActFn.apply(xs[k], gs[k], bs[k], rs[k], None, rm, rv, nbt, 0.1, 1e-5, True, True, False, None))
            tot = sum((z.float() * nhwc(_rand(*[z.shape[0], z.shape[3], z.shape[1], z.shape[2]], seed=50 + k)).to(DEV)).sum()
                      for k, z in enumerate(zs))
            tot.backward()
            torch.cuda.synchronize()
            for k in range(len(shapes)):
                outs.append((xs[k].grad.float().cpu(), gs[k].grad.cpu(), bs[k].grad.cpu(), rs[k].grad.float().cpu() if res else None))
            return outs
        finally:
            hb._BN_FUSED_BWD = prev

    L = hb.lib()
    t0 = ctypes.c_uint(0)
    assert L.ssa_bn_bwd_fused_timeouts(ctypes.byref(t0)) == 0
    two = run(False)
    one = run(True)
    t1 = ctypes.c_uint(0)
    assert L.ssa_bn_bwd_fused_timeouts(ctypes.byref(t1)) == 0
    assert t1.value == t0.value, "workgroups timed out of the rendezvous: %d" % (t1.value - t0.value)
    assert L.ssa_bn_bwd_fused_capacity() >= 256
    for (dx2, dg2, db2, dr2), (dx1, dg1, db1, dr1) in zip(two, one):
        check_close("fused dx", dx1, dx2, 1.6e-2, 1e-4)          # one 16-bit ulp
        check_close("fused dgamma", dg1, dg2, 1e-5, 1e-5)
        check_close("fused dbeta", db1, db2, 1e-5, 1e-5)
        if res:
            assert torch.equal(dr1, dr2)


def test_bn_deferred_running_stats_two_passes():
    """Two training passes over one BatchNorm layer (the 0.5x and the 1.0x pass, problems of one
    grouped launch): the deferred batched update must equal the reference's sequential in-place
    updates, in issue order."""
    from oracle import ops as O
    from semseg_amd import ops, nn as snn
    hb = _hb()
    C = 48
    bn = snn.BatchNorm2d(C, momentum=0.1)
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    with torch.no_grad():
        bn.running_mean.copy_(rm)
        bn.running_var.copy_(rv)
    xa, xb = _rand(1, C, 12, 20, seed=21), _rand(2, C, 24, 40, seed=22) * 2.0 + 0.5
    one = torch.ones(C)
    zero = torch.zeros(C)
    O.batch_norm(xa, one, zero, rm, rv, True, 0.1, 1e-5)
    O.batch_norm(xb, one, zero, rm, rv, True, 0.1, 1e-5)
    bn = bn.to(DEV).train()
    be = ops.HipBackend()
    hb.begin_step(torch.device(DEV))
    be.batch_norm_act([_to_dev_nhwc(xa), _to_dev_nhwc(xb)], bn)
    be.end_forward()
    torch.cuda.synchronize()
    check_close("deferred running_mean", bn.running_mean, rm, 1e-4, 1e-4)
    check_close("deferred running_var", bn.running_var, rv, 1e-4, 1e-4)
    assert int(bn.num_batches_tracked) == 2


def test_bn_eval():
    from oracle import ops as O
    hb = _hb()
    B, C, H, W = 1, 64, 9, 11
    x = _rand(B, C, H, W, seed=1)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    rm, rv = torch.randn(C) * 0.2, torch.rand(C) + 0.5
    y = torch.relu(O.batch_norm(x, gamma, beta, rm.clone(), rv.clone(), False))
    z = hb.BatchNormActFn.apply(_to_dev_nhwc(x), gamma.to(DEV), beta.to(DEV), None, None, rm.to(DEV),
                                rv.to(DEV), None, 0.1, 1e-5, False, True, False)
    torch.cuda.synchronize()
    check_close("bn_eval", nchw(z.float()), y)


def test_bn_eval_coefficient_registry():
    """Inference forwards take their BatchNorm coefficients from persistent buffers: computed singly the first time a
    layer is seen, refreshed by ONE batched launch at the next begin_step -- and never stale: running statistics and
    affine parameters changed between two forwards (as a training step does, through raw pointers: no version counter
    moves) reach the next forward; under autograd the registry is not used."""
    from oracle import ops as O
    hb = _hb()
    B, H, W = 1, 7, 9
    layers = []
    for k, C in enumerate((48, 64, 720)):
        g = torch.Generator().manual_seed(40 + k)
        layers.append(dict(C=C, x=_rand(B, C, H, W, seed=50 + k), gamma=torch.rand(C, generator=g) + 0.5,
                           beta=torch.randn(C, generator=g) * 0.1, rm=torch.randn(C, generator=g) * 0.2,
                           rv=torch.rand(C, generator=g) + 0.5))
    for l in layers:
        l["dev"] = {k: l[k].to(DEV) for k in ("gamma", "beta", "rm", "rv")}
        l["xd"] = _to_dev_nhwc(l["x"])

    def forward():
        hb.begin_step()
        with torch.no_grad():
            return [hb.BatchNormActFn.apply(l["xd"], l["dev"]["gamma"], l["dev"]["beta"], None, None, l["dev"]["rm"],
                                            l["dev"]["rv"], None, 0.1, 1e-5, False, True, False) for l in layers]

    def check(zs, tag):
        torch.cuda.synchronize()
        for l, z in zip(layers, zs):
            d = l["dev"]
            want = torch.relu(O.batch_norm(l["x"], d["gamma"].cpu(), d["beta"].cpu(), d["rm"].cpu().clone(), d["rv"].cpu().clone(), False))
            check_close("bn_eval registry %s C=%d" % (tag, l["C"]), nchw(z.float()), want)

    reg = hb._BN_EVAL
    z1 = forward()                                   # first sight: single launches, registered
    check(z1, "first")
    mine = [l["dev"]["rm"] for l in layers]
    assert sum(any(e[0]() is t for t in mine) for e in reg.entries.values()) == 3
    z2 = forward()                                   # the batched refresh
    check(z2, "batched")
    assert reg.table is not None and all(e[5] == reg.gen for e in reg.entries.values())
    for a, b in zip(z1, z2):
        assert torch.equal(a, b)
    for l in layers:                                 # what a training step does to a layer, behind autograd's back
        d = l["dev"]
        d["rm"].data.mul_(1.5).add_(0.1)
        d["rv"].data.mul_(0.5).add_(0.2)
        d["gamma"].data.mul_(-0.75)
    z3 = forward()
    check(z3, "after an update")
    assert not torch.equal(z3[0], z2[0])
    hb.begin_step()                                  # a forward that used no inference BatchNorm (a training forward) ...
    hb.begin_step()
    for l in layers:
        l["dev"]["beta"].data.add_(0.25)
    check(forward(), "after a training forward")    # ... then the single launches again, still current
    # under autograd: fresh coefficients per call, the registry untouched
    uses = reg.uses
    xg = layers[0]["xd"].clone().requires_grad_(True)
    d = layers[0]["dev"]
    z = hb.BatchNormActFn.apply(xg, d["gamma"], d["beta"], None, None, d["rm"], d["rv"], None, 0.1, 1e-5, False, True, False)
    assert z.requires_grad and reg.uses == uses


# ---------------------------------------------------------------- pooling
@pytest.mark.parametrize("B,C,H,W", [(2, 64, 48, 64), (1, 64, 33, 47), (1, 8, 5, 7)])
def test_maxpool3x3s2(B, C, H, W):
    """ResNet stem pooling on post-ReLU data (many exact ties at zero): values and
    the gradient routing must match PyTorch's first-maximum rule."""
    from oracle import ops as O
    hb = _hb()
    x = torch.relu(_rand(B, C, H, W, seed=31))
    xr = x.clone().requires_grad_(True)
    yr = O.max_pool_3x3_s2(xr)
    gy = _rand(*yr.shape, seed=32)
    yr.backward(gy)
    xd = _to_dev_nhwc(x).requires_grad_(True)
    yd = hb.MaxPool3x3s2Fn.apply(xd)
    yd.backward(nhwc(gy).to(DEV).to(ACT_DTYPE))
    torch.cuda.synchronize()
    assert torch.equal(nchw(yd.float()).cpu(), yr.detach())
    check_close("maxpool dx", nchw(xd.grad.float()), xr.grad, 1e-2, 4e-3)


def test_global_avg_pool():
    from oracle import ops as O
    hb = _hb()
    x = _rand(2, 2048, 12, 16, seed=33)
    xr = x.clone().requires_grad_(True)
    yr = O.global_avg_pool(xr)
    gy = _rand(*yr.shape, seed=34)
    yr.backward(gy)
    xd = _to_dev_nhwc(x).requires_grad_(True)
    yd = hb.GlobalAvgPoolFn.apply(xd)
    yd.backward(nhwc(gy).to(DEV).to(ACT_DTYPE))
    torch.cuda.synchronize()
    check_close("gap fwd", nchw(yd.float()), yr)
    check_close("gap dx", nchw(xd.grad.float()), xr.grad)


# --------------------------------------------------------------- bilinear
@pytest.mark.parametrize("C,hi,wi,ho,wo,f32", [(48, 16, 20, 64, 80, False), (96, 15, 9, 30, 18, False),
                                               (19, 32, 32, 128, 128, True), (1, 24, 40, 96, 160, True),
                                               (19, 64, 64, 32, 32, True), (192, 8, 8, 64, 64, False),
                                               (19, 33, 45, 67, 91, True),
                                               # dense few-channel fp32 upsampling: the LDS-staged row pass of the separable
                                               # backward (65 = Mapillary; rows of several segments at 2x and 4x; a ratio that
                                               # is not an integer)
                                               (65, 12, 20, 48, 80, True), (19, 6, 1100, 12, 2200, True),
                                               (19, 5, 400, 20, 1600, True), (19, 8, 12, 21, 31, True)])
def test_bilinear(C, hi, wi, ho, wo, f32):
    from oracle import ops as O
    hb = _hb()
    B = 2
    x = _rand(B, C, hi, wi, seed=1)
    xr = x.clone().requires_grad_(True)
    y = O.bilinear(xr, (ho, wo))
    gy = _rand(B, C, ho, wo, seed=2)
    y.backward(gy)
    xd = nhwc(x).to(DEV)
    gyd = nhwc(gy).to(DEV)
    if not f32:
        xd, gyd = xd.to(ACT_DTYPE), gyd.to(ACT_DTYPE)
    xd.requires_grad_(True)
    yd = hb.BilinearFn.apply(xd, ho, wo, f32)
    yd.backward(gyd)
    torch.cuda.synchronize()
    tol = (1e-5, 1e-5) if f32 else (1e-2, 4e-3)
    check_close("bilinear_fwd", nchw(yd.float()), y, *tol)
    check_close("bilinear_bwd", nchw(xd.grad.float()), xr.grad, *tol)


def test_image_resize():
    from oracle import ops as O
    hb = _hb()
    x = torch.randn(2, 3, 64, 96)
    y = hb.image_to_nhwc(x.to(DEV), None)
    torch.cuda.synchronize()
    check_close("image copy", nchw(y.float())[:, :3], bf16_round(x), 1e-6, 1e-6)
    assert float(y[..., 3:].abs().max()) == 0.0
    y2 = hb.image_to_nhwc(x.to(DEV), (32, 48))
    torch.cuda.synchronize()
    check_close("image resize", nchw(y2.float())[:, :3], O.resize_x(x, 0.5), 1e-2, 4e-3)


def test_sum_act():
    hb = _hb()
    ts = [_rand(2, 12, 10, 48, seed=i) for i in range(3)]
    tr = [t.clone().requires_grad_(True) for t in ts]
    y = torch.relu(tr[0] + tr[1] + tr[2])
    gy = _rand(2, 12, 10, 48, seed=9)
    y.backward(gy)
    td = [t.to(DEV).to(ACT_DTYPE).requires_grad_(True) for t in ts]
    z = hb.SumActFn.apply(True, *td)
    z.backward(gy.to(DEV).to(ACT_DTYPE))
    torch.cuda.synchronize()
    check_close("sum_act", z.float(), y)
    for i in range(3):
        check_close("sum_act_grad%d" % i, td[i].grad.float(), tr[i].grad)


# -------------------------------------------------------------------- OCR
def test_ocr_gather():
    from oracle import ops as O
    hb = _hb()
    B, C, K, H, W = 2, 512, 19, 24, 28
    feats = _rand(B, C, H, W, seed=1)
    logits = torch.randn(B, K, H, W) * 2.0
    fr = feats.clone().requires_grad_(True)
    lr = logits.clone().requires_grad_(True)
    ctx = O.spatial_gather(fr, lr)            # [B,C,K,1]
    g = torch.randn(B, C, K, 1)
    ctx.backward(g)
    fd = _to_dev_nhwc(feats).requires_grad_(True)
    ld = nhwc(logits).to(DEV).requires_grad_(True)
    out = hb.OcrGatherFn.apply(fd, ld)        # [B,K,C]
    out.backward(g[..., 0].permute(0, 2, 1).contiguous().to(DEV))
    torch.cuda.synchronize()
    check_close("gather_fwd", out.permute(0, 2, 1), ctx[..., 0], 1e-2, 4e-3)
    check_close("gather_dfeats", nchw(fd.grad.float()), fr.grad, 2e-2, 6e-3)
    check_close("gather_dlogits", nchw(ld.grad), lr.grad, 2e-2, 8e-3)


@pytest.mark.parametrize("K,H,W,fused", [(19, 20, 24, True), (65, 9, 31, True), (19, 3, 5, True), (96, 8, 16, True),
                                          (19, 20, 24, False), (150, 6, 11, True)])
def test_ocr_attention(K, H, W, fused, monkeypatch):
    """ObjectAttentionBlock's softmax(q k^T / sqrt(256)) v (network/ocr_utils.py:100-113): the fused kernel
    (csrc/ocr_attn.hip; 19 = Cityscapes, 65 = Mapillary -> 3 region blocks, 96 = its limit, pixel counts that are not
    multiples of the 128-pixel workgroup tile) and the three-launch form (forced, and what 150 regions fall back to)."""
    from oracle import ops as O
    hb = _hb()
    monkeypatch.setattr(hb, "_OCR_ATTN_FUSED", fused)
    B, D = 2, 256
    q = _rand(B, H * W, D, seed=1)
    k = _rand(B, K, D, seed=2)
    v = _rand(B, K, D, seed=3)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = O.object_attention(qr, kr.permute(0, 2, 1), vr, D)
    g = _rand(B, H * W, D, seed=4)
    out.backward(g)
    qd = q.view(B, H, W, D).to(DEV).to(ACT_DTYPE).requires_grad_(True)
    kd = k.to(DEV).to(ACT_DTYPE).requires_grad_(True)
    vd = v.to(DEV).to(ACT_DTYPE).requires_grad_(True)
    od = hb.OcrAttnFn.apply(qd, kd, vd, D ** -0.5)
    od.backward(g.view(B, H, W, D).to(DEV).to(ACT_DTYPE))
    torch.cuda.synchronize()
    # one rounding of the output (the probabilities are rounded to 16 bit on both paths before the second product)
    check_close("attn_fwd", od.float().view(B, H * W, D), out, 1e-2, 4e-3)
    check_close("attn_dq", qd.grad.float().view(B, H * W, D), qr.grad, 3e-2, 1.5e-2)
    check_close("attn_dk", kd.grad.float(), kr.grad, 3e-2, 1.5e-2)
    check_close("attn_dv", vd.grad.float(), vr.grad, 2e-2, 8e-3)


# ----------------------------------------------------------------- fusion
@pytest.mark.parametrize("shape", [(2, 16, 20, 19), (1, 5, 7, 19), (1, 9, 30, 3), (1, 4, 70, 65)])
def test_scale_fusion_ops(shape):
    hb = _hb()
    B, H, W, C = shape      # ragged last block of the 256-pixel tile kernel; 65 classes: the wave-per-pixel kernel
    a = torch.rand(B, H, W, 1)
    lo = torch.randn(B, H, W, C)
    hi = torch.randn(B, H, W, C)
    x = torch.randn(B, H, W, 1)
    ar, lor, hir, xr = (t.clone().requires_grad_(True) for t in (a, lo, hi, x))
    s = torch.sigmoid(xr)
    m = ar * lor
    j = m + (1 - s) * hir
    gj = torch.randn(B, H, W, C)
    j.backward(gj)
    ad, lod, hid, xd = (t.to(DEV).requires_grad_(True) for t in (a, lo, hi, x))
    sd = hb.SigmoidFn.apply(xd)
    md = hb.BcastMulFn.apply(ad, lod)
    jd = hb.AttnBlendFn.apply(md, sd, hid)
    jd.backward(gj.to(DEV))
    torch.cuda.synchronize()
    check_close("fusion_fwd", jd, j, 1e-5, 1e-5)
    for n, d, r in (("a", ad, ar), ("lo", lod, lor), ("hi", hid, hir), ("x", xd, xr)):
        check_close("fusion_d" + n, d.grad, r.grad, 1e-4, 1e-4)


# ------------------------------------------------------------------ losses
def _labels(B, H, W, C, seed=0):
    g = torch.Generator().manual_seed(seed)
    blocks = torch.randint(0, C, (B, (H + 7) // 8, (W + 7) // 8), generator=g)
    lab = blocks.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :H, :W].clone()
    ign = torch.rand(B, H, W, generator=g) < 0.1
    lab[ign] = 255
    return lab.long()


def test_cross_entropy():
    from oracle import ops as O
    hb = _hb()
    B, C, H, W = 2, 19, 40, 56
    logits = torch.randn(B, C, H, W) * 3
    lab = _labels(B, H, W, C)
    lr = logits.clone().requires_grad_(True)
    loss = O.cross_entropy(lr, lab, 255)
    (loss * 1.7).backward()
    ld = nhwc(logits).to(DEV).requires_grad_(True)
    out = hb.CrossEntropyFn.apply(ld, lab.to(DEV), 255)
    (out * 1.7).backward()
    torch.cuda.synchronize()
    check_close("ce", out.view(1), loss.view(1), 1e-5, 1e-5)
    check_close("ce_grad", nchw(ld.grad), lr.grad, 1e-4, 1e-4)


# (2, 64, 96): dense, element count a multiple of four -- the forward saves no gradient, the backward recomputes it
# (ssa_bce_bwd / inside ssa_rmi_bwd_logits_bce); (1, 65, 97): an odd pixel count -- the saved-gradient path
@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 65, 97)])
@pytest.mark.parametrize("do_rmi", [False, True])
def test_bce_rmi(do_rmi, shape):
    from oracle import ops as O
    hb = _hb()
    C = 19
    B, H, W = shape
    logits = torch.randn(B, C, H, W) * 2
    lab = _labels(B, H, W, C, seed=3)
    lr = logits.clone().requires_grad_(True)
    loss = O.rmi_loss(lr, lab, C, do_rmi=do_rmi)
    (loss * 0.4).backward()
    ld = nhwc(logits).to(DEV).requires_grad_(True)
    out = hb.BceRmiFn.apply(ld, lab.to(DEV), do_rmi, 0.5)
    (out * 0.4).backward()
    torch.cuda.synchronize()
    check_close("bce_rmi(%s)" % do_rmi, out.view(1), loss.view(1), 1e-4, 1e-4)
    check_close("bce_rmi_grad(%s)" % do_rmi, nchw(ld.grad), lr.grad, 2e-3, 1e-3)
