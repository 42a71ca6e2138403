"""ssa_conv2d_tile_aux (conv_tile.hip, AUX variants): the tile data-gradient kernel with a fused
epilogue tile, against ssa_conv2d_tile + the separate passes it replaces.  (The residual-block Function
that uses both epilogues is checked end to end in tests/test_group_gpu.py.)"""
import ctypes
import os

import pytest
import torch

from util import ACT_DTYPE

pytestmark = pytest.mark.gpu


SHAPES = [(1, 256, 256, 48), (2, 64, 96, 48), (1, 128, 128, 96), (1, 37, 53, 96), (2, 64, 64, 192),
          (1, 32, 32, 384), (1, 9, 20, 64), (1, 128, 128, 64)]


def _setup(B, H, W, C, seed):
    from semseg_amd import hip_backend as hb
    from semseg_amd._lib import ConvDesc
    g = torch.Generator().manual_seed(seed)
    dy = (torch.randn(B, H, W, C, generator=g)).to(ACT_DTYPE).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    aux = torch.randn(B, H, W, C, generator=g).to(ACT_DTYPE).cuda()
    hb.clear_pack_cache()
    wpt, _ = hb._packed_filter(w, 3, 0, C)
    d = hb._tile_desc(B, H, W, C, C, C, (3, 3), 1, 1, 1, H, W, False)
    assert hb.tile_supported(d)
    return hb, d, dy, wpt, aux, g


@pytest.mark.parametrize("B,H,W,C", SHAPES)
def test_aux_add_is_the_unfused_add(B, H, W, C):
    hb, d, dy, wpt, aux, _ = _setup(B, H, W, C, 1)
    ref = hb._tile_conv(d, dy, wpt, None, None)
    out = hb._tile_conv_aux(d, dy, wpt, None, aux, C, None, 1)
    torch.cuda.synchronize()
    want = (ref.float() + aux.float()).to(ACT_DTYPE)       # what autograd's bf16 add produces
    assert torch.equal(out, want)


@pytest.mark.parametrize("B,H,W,C", SHAPES)
def test_aux_bn_backward_sums(B, H, W, C):
    hb, d, dy, wpt, x, g = _setup(B, H, W, C, 2)
    coef = torch.stack([torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.5,
                        torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).cuda()
    nrep = hb.stat_replicas()
    sums = torch.zeros(nrep * 2 * C, dtype=torch.float64, device="cuda")
    ref = hb._tile_conv(d, dy, wpt, None, None)
    out = hb._tile_conv_aux(d, dy, wpt, sums, x, C, coef, 2)
    # the separate pass this replaces, on the same dz
    sums2 = torch.zeros(nrep * 2 * C, dtype=torch.float64, device="cuda")
    from semseg_amd._lib import lib, check
    P = B * H * W
    check(lib().ssa_bn_bwd_reduce(x.data_ptr(), C, ref.data_ptr(), C, None, C, P, C, coef[2].data_ptr(),
                                  coef[3].data_ptr(), 1, None, H * W, sums2.data_ptr(), nrep, 0,
                                  coef[0].data_ptr(), coef[1].data_ptr(), None,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ssa_bn_bwd_reduce")
    torch.cuda.synchronize()
    assert torch.equal(out, ref)                                  # dz itself is untouched
    got = sums.view(nrep, 2, C).sum(0)
    sep = sums2.view(nrep, 2, C).sum(0)
    xf, dz = x.double(), ref.double()
    c = coef.double()
    m = (x.float() * coef[0] + coef[1] > 0).double()
    want = torch.stack([(m * dz).sum((0, 1, 2)), (m * dz * (xf - c[2]) * c[3]).sum((0, 1, 2))])
    scale = (m * dz).abs().sum((0, 1, 2)).clamp_min(1e-30)
    # fused epilogue vs the separate pass: same expressions, same mask -> only the summation order differs
    err = ((got - sep).abs() / scale).max().item()
    print("fused epilogue vs separate pass: max |difference| / sum|terms| = %.3g" % err)
    assert err < 2e-6
    # both vs an fp64 evaluation: an element whose scale*x+shift is within an ulp of zero may fall on the
    # other side of the mask there (fma vs mul+add), which moves a sum by one term of ~P
    for name, v in (("fused epilogue", got), ("separate pass", sep)):
        err = ((v - want).abs() / scale).max().item()
        print("%s vs fp64: max |sum - reference| / sum|terms| = %.3g" % (name, err))
        assert err < 1e-4, name
