"""The 48-channel-block geometry of the trunk conv (csrc/conv_tile_q.hip, opt-in: SSA_TILE_Q=1) against the oracle
and against the default kernel (csrc/conv_tile_p.hip) on the same inputs: the two lane maps it relies on
(v_mfma_f32_16x16x32, v_permlane16_swap), its filter layout through all three pack kernels, forward + BatchNorm
statistics, the data gradient with both fused epilogues, ragged images, every wave shape (pb = 4 / 2 / 1), strips of
several tiles, and a grouped level that mixes the wave shapes in one launch.  network/hrnetv2.py:31-66 (BasicBlock).

Same tolerance as tests/test_kernels_gpu.py (16-bit operands rounded before both paths, fp32 accumulation, one
rounding of the output): max error <= 1e-2 max|ref|, mean error <= 4e-3 mean|ref|.  Runs on the CPU emulation of
the kernels under SSA_EMU=1 (tests/test_emu_selected_cpu.py runs a selection in the default CPU suite)."""
import ctypes
import os

import pytest
import torch

from util import bf16_round, check_close, nchw, ACT_DTYPE

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _hb():
    from semseg_amd import hip_backend
    return hip_backend


# SSA_TILE_Q_THREE=1: the same tests on the three-workgroups-per-CU form of the kernel (ssa_conv_tile_q_config(1): filter
# in three stages through two 15 KB buffers, pb <= 2) -- run on the CPU emulation by tests/test_emu_selected_cpu.py; the
# form has not been on the device yet, so the GPU suite does not select it
THREE = os.environ.get("SSA_TILE_Q_THREE", "0") == "1"


@pytest.fixture(autouse=True)
def _q_on(monkeypatch):
    hb = _hb()
    monkeypatch.setattr(hb, "_TILE_Q", True)
    hb.clear_pack_cache()
    hb.lib().ssa_conv_tile_q_config(1 if THREE else 0)
    yield
    hb.lib().ssa_conv_tile_q_config(0)
    hb.clear_pack_cache()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_round(torch.randn(*shape, generator=g) * scale)


def _dev_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(ACT_DTYPE).to(DEV)


def _sync():
    torch.cuda.synchronize()


# ----------------------------------------------------------------- lane maps
def test_probe_mfma16():
    from semseg_amd._lib import lib, check
    a = _rand(16, 32, seed=1)
    b = _rand(32, 16, seed=2)       # asymmetric on purpose
    a_d = a.to(DEV).to(ACT_DTYPE).contiguous()
    bt_d = b.t().contiguous().to(DEV).to(ACT_DTYPE)
    c_d = torch.zeros(16, 16, device=DEV)
    check(lib().ssa_probe_mfma16(ctypes.c_void_p(a_d.data_ptr()), ctypes.c_void_p(bt_d.data_ptr()),
                                 ctypes.c_void_p(c_d.data_ptr()), None), "probe")
    _sync()
    check_close("mfma16", c_d, a @ b, 1e-5, 1e-5)


def test_probe_swap16():
    """v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second."""
    from semseg_amd._lib import lib, check
    out = torch.zeros(64, 2, dtype=torch.int32, device=DEV)
    check(lib().ssa_probe_swap16(ctypes.c_void_p(out.data_ptr()), None), "probe")
    _sync()
    want = []
    for l in range(64):
        if (l >> 4) & 1 == 0:
            want.append((l, l ^ 16))                  # (own a, partner's a)
        else:
            want.append(((l ^ 16) + 100, l + 100))    # (partner's b, own b)
    assert out.cpu().tolist() == [list(t) for t in want]


# ----------------------------------------------------------------- filter layout
def _q_layout_reference(w, transposed):
    """[Cout/48][Cin/48][14][3][64][8] of csrc/conv_igemm.hip:frag_offset layout 1, in plain loops."""
    Cout, Cin = w.shape[:2]
    rows, cols = (Cin, Cout) if transposed else (Cout, Cin)
    nt, nc = -(-rows // 48), cols // 48
    out = torch.zeros(nt, nc, 14, 3, 64, 8)
    for kk in range(432):
        tap, c = divmod(kk, 48)
        ks, q, j = kk >> 5, (kk & 31) >> 3, kk & 7
        kh, kw = divmod(tap, 3)
        for cc in range(nc):
            if transposed:      # operand row = input channel, k runs over (flipped tap, output channel)
                col = w[cc * 48 + c, :, 2 - kh, 2 - kw]
            else:
                col = w[:, cc * 48 + c, kh, kw]
            for r in range(rows):
                out[r // 48, cc, ks, (r % 48) >> 4, q * 16 + (r & 15), j] = col[r]
    return out.reshape(nt * 48, nc * 448)


@pytest.mark.parametrize("Cout,Cin,mode", [(48, 96, 10), (96, 48, 11), (96, 96, 10)])
def test_q_filter_layout_single_and_batched(Cout, Cin, mode):
    hb = _hb()
    w = _rand(Cout, Cin, 3, 3, seed=5, scale=0.1).to(DEV).requires_grad_(True)
    cin_pad, cout_pad = (Cin, 0) if mode == 10 else (0, Cout)
    wp, Kpad = hb._packed_filter(w, mode, cin_pad, cout_pad)
    _sync()
    want = _q_layout_reference(w.detach().cpu(), mode == 11)
    assert tuple(wp.shape) == tuple(want.shape) and Kpad == want.shape[1]
    assert torch.equal(wp.float().cpu(), want), "ssa_pack_filter"
    # the per-step refresh (tiled, then batched) after a parameter update must produce the same layout
    with torch.no_grad():
        w.mul_(0.5)
    want = _q_layout_reference(bf16_round(w.detach().cpu()), mode == 11)
    for tiled in (True, False):
        hb.invalidate_packed_filters()
        hb._PACK_TILED = tiled
        hb._JOB_TABLE.update(key=None)
        try:
            hb.refresh_packed_filters()
        finally:
            hb._PACK_TILED = True
        _sync()
        wp2, _ = hb._packed_filter(w, mode, cin_pad, cout_pad)
        assert wp2.data_ptr() == wp.data_ptr()
        assert torch.equal(wp2.float().cpu(), want), "tiled refresh" if tiled else "batched refresh"


# ----------------------------------------------------------------- forward / data gradient
def _oracle_conv(x, w):
    from oracle import ops as O
    return O.conv2d(x, w, None, 1, 1, 1)


# (C, B, H, W): pb = 2 up to 192 channels, 1 for 384 (test_q_forced_wave_shapes: the others); ragged right / bottom
# edges; images lower than a tile (pb halves); several tiles per strip
FWD_CASES = [(48, 1, 16, 16), (48, 2, 21, 37), (96, 1, 16, 32), (96, 1, 7, 19), (192, 1, 8, 16), (192, 1, 11, 18),
             (384, 1, 4, 16), (384, 1, 6, 17)]


@pytest.mark.parametrize("C,B,H,W", FWD_CASES)
def test_q_forward_and_statistics(C, B, H, W):
    hb = _hb()
    x = _rand(B, C, H, W, seed=21)
    w = _rand(C, C, 3, 3, seed=22, scale=(2.0 / (9 * C)) ** 0.5)
    yr = _oracle_conv(x, w)
    xd = _dev_nhwc(x)
    hb.begin_step(torch.device(DEV))
    d = hb._tile_desc(B, H, W, C, C, C, (3, 3), 1, 1, 1, H, W, False)
    assert hb.tile_q_supported(d)
    hb.lib().ssa_launch_count(1)
    y, stats = hb._conv_fwd(xd, C, w.to(DEV), None, 1, 1, 1, False, True)
    _sync()
    check_close("q forward %d" % C, nchw(y.float()), yr)
    # statistics of the ROUNDED outputs, as the default kernel's epilogue produces them
    yf = y.double().cpu().reshape(-1, C)
    got = stats.double().cpu().view(hb.stat_replicas(), 2, C).sum(0)
    want = torch.stack([yf.sum(0), (yf * yf).sum(0)])
    scale = torch.stack([yf.abs().sum(0), (yf * yf).sum(0)]).clamp_min(1e-30)
    err = ((got - want).abs() / scale).max().item()
    print("statistics: max |sum - reference| / sum|terms| = %.3g" % err)
    assert err < 2e-5
    hb._PENDING_STATS.clear()


@pytest.mark.parametrize("C,B,H,W", [(48, 1, 16, 16), (96, 2, 9, 21), (192, 1, 8, 16), (384, 1, 5, 16)])
def test_q_dgrad_epilogues_equal_the_default_kernel(C, B, H, W):
    """Data gradient, plain / + residual gradient (aux_mode 1) / + bn1 backward sums (aux_mode 2): against the
    oracle, and against conv_tile_p.hip on the same inputs (same rounding points: outputs may differ by fp32
    summation order only, i.e. by at most one rounding of a few elements)."""
    from oracle import ops as O
    hb = _hb()
    g = torch.Generator().manual_seed(31)
    dy = _rand(B, C, H, W, seed=32)
    w = _rand(C, C, 3, 3, seed=33, scale=(2.0 / (9 * C)) ** 0.5)
    res = _rand(B, C, H, W, seed=34)
    coef = torch.stack([torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.5,
                        torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).to(DEV)
    dxr = torch.nn.grad.conv2d_input((B, C, H, W), w, dy, 1, 1, 1)
    dyd, resd, wd = _dev_nhwc(dy), _dev_nhwc(res), w.to(DEV)
    nrep = hb.stat_replicas()
    outs = {}
    for q in (True, False):
        hb._TILE_Q = q
        hb.clear_pack_cache()
        plain = hb._conv_dgrad((B, H, W, C), wd, dyd, C, C, 1, 1, 1, (H, W))
        added = hb._conv_dgrad((B, H, W, C), wd, dyd, C, C, 1, 1, 1, (H, W), aux=resd, ldaux=C, mode=1)
        sums = torch.zeros(nrep * 2 * C, dtype=torch.float64, device=DEV)
        dz = hb._conv_dgrad((B, H, W, C), wd, dyd, C, C, 1, 1, 1, (H, W), aux=resd, ldaux=C, coef=coef, mode=2, stats=sums)
        _sync()
        outs[q] = (plain, added, dz, sums.view(nrep, 2, C).sum(0))
    hb._TILE_Q = True
    plain, added, dz, sums = outs[True]
    check_close("q dgrad %d" % C, nchw(plain.float()), dxr)
    assert torch.equal(dz, plain)                                              # dz itself is untouched
    assert torch.equal(added, (plain.float() + resd.float()).to(ACT_DTYPE))   # what autograd's 16-bit add produces
    # against the default kernel
    p0, a0, z0, s0 = outs[False]
    check_close("q vs p", plain.float(), p0.float(), 8e-3, 1e-4)
    # bn1 backward sums from (x tile, dz) of THIS kernel's dz, in fp64
    xf, dzf, c = resd.double().cpu(), dz.double().cpu(), coef.double().cpu()
    m = (resd.float().cpu() * coef[0].cpu() + coef[1].cpu() > 0).double()
    want = torch.stack([(m * dzf).sum((0, 1, 2)), (m * dzf * (xf - c[2]) * c[3]).sum((0, 1, 2))])
    scale = (m * dzf).abs().sum((0, 1, 2)).clamp_min(1e-30)
    err = ((sums.cpu() - want).abs() / scale).max().item()
    print("bn1 backward sums vs fp64: max |sum - reference| / sum|terms| = %.3g" % err)
    assert err < 1e-4


@pytest.mark.parametrize("pb", [4, 1])
@pytest.mark.parametrize("C,B,H,W", [(48, 1, 21, 37), (96, 1, 18, 16)])
def test_q_forced_wave_shapes(monkeypatch, pb, C, B, H, W):
    """SSA_TILE_Q_PB (read per launch) forces the wave shape: 16 x 16-pixel tiles (pb = 4) and 16 x 4 (pb = 1) on the
    problems that run at pb = 2 by default; forward + statistics, and the fused-sums data gradient (pb = 4 is not
    compiled for aux_mode 2: it runs at 2)."""
    if THREE and pb == 4:
        pytest.skip("the three-per-CU form runs at pb <= 2")
    monkeypatch.setenv("SSA_TILE_Q_PB", str(pb))
    hb = _hb()
    x = _rand(B, C, H, W, seed=71)
    w = _rand(C, C, 3, 3, seed=72, scale=(2.0 / (9 * C)) ** 0.5)
    xd, wd = _dev_nhwc(x), w.to(DEV)
    hb.begin_step(torch.device(DEV))
    y, stats = hb._conv_fwd(xd, C, wd, None, 1, 1, 1, False, True)
    _sync()
    check_close("q forced pb %d" % pb, nchw(y.float()), _oracle_conv(x, w))
    yf = y.double().cpu().reshape(-1, C)
    got = stats.double().cpu().view(hb.stat_replicas(), 2, C).sum(0)
    want = torch.stack([yf.sum(0), (yf * yf).sum(0)])
    scale = torch.stack([yf.abs().sum(0), (yf * yf).sum(0)]).clamp_min(1e-30)
    assert ((got - want).abs() / scale).max().item() < 2e-5
    hb._PENDING_STATS.clear()
    monkeypatch.delenv("SSA_TILE_Q_PB")
    y2, _ = hb._conv_fwd(xd, C, wd, None, 1, 1, 1, False, False)
    _sync()
    check_close("forced vs default wave shape", y.float(), y2.float(), 8e-3, 1e-4)


def test_q_strips_of_several_tiles():
    """A strip budget that puts several tiles (and all chunks of each) on one workgroup; the result does not depend
    on the budget."""
    hb = _hb()
    C, B, H, W = 96, 1, 40, 24
    x = _rand(B, C, H, W, seed=41)
    w = _rand(C, C, 3, 3, seed=42, scale=0.03)
    xd, wd = _dev_nhwc(x), w.to(DEV)
    L = hb.lib()
    d = hb._tile_desc(B, H, W, C, C, C, (3, 3), 1, 1, 1, H, W, False)
    ys = []
    for budget in (0, 8, 24, 64):
        L.ssa_conv_tile_q_strip(budget)
        try:
            n = L.ssa_conv_tile_q_wgs(ctypes.byref(d), budget, 0)
            y, _ = hb._conv_fwd(xd, C, wd, None, 1, 1, 1, False, False)
        finally:
            L.ssa_conv_tile_q_strip(0)
        _sync()
        ys.append(y)
        print("budget %d: %d workgroups" % (budget, n))
    assert L.ssa_conv_tile_q_wgs(ctypes.byref(d), 64, 0) < L.ssa_conv_tile_q_wgs(ctypes.byref(d), 8, 0)
    for y in ys[1:]:
        assert torch.equal(y, ys[0])
    check_close("q strips", nchw(ys[0].float()), _oracle_conv(x, w))


def test_q_grouped_level_mixes_wave_shapes():
    """Four branches of a trunk level (48 / 96 / 192 / 384 channels: pb 2, 2, 2, 1) in ONE grouped launch, bit-identical
    to the four separate launches."""
    hb = _hb()
    shapes = [(48, 32, 32), (96, 16, 16), (192, 8, 16), (384, 4, 16)]
    xs = [_dev_nhwc(_rand(1, C, H, W, seed=50 + i)) for i, (C, H, W) in enumerate(shapes)]
    ws = [_rand(C, C, 3, 3, seed=60 + i, scale=(2.0 / (9 * C)) ** 0.5).to(DEV) for i, (C, H, W) in enumerate(shapes)]
    hb.begin_step(torch.device(DEV))
    single = [hb._conv_fwd(x, x.shape[3], w, None, 1, 1, 1, False, True) for x, w in zip(xs, ws)]
    _sync()
    descs = [hb._tile_desc(1, H, W, C, C, C, (3, 3), 1, 1, 1, H, W, False) for C, H, W in shapes]
    hb.lib().ssa_launch_count(1)
    with hb.tile_strip(descs), hb.group():
        grouped = [hb._conv_fwd(x, x.shape[3], w, None, 1, 1, 1, False, True) for x, w in zip(xs, ws)]
    _sync()
    assert hb.lib().ssa_launch_count(1) == 1
    for (y1, s1), (y2, s2), (C, H, W) in zip(single, grouped, shapes):
        assert torch.equal(y1, y2), C
        a = s1.view(hb.stat_replicas(), 2, C).sum(0)
        b = s2.view(hb.stat_replicas(), 2, C).sum(0)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), C
    hb._PENDING_STATS.clear()
