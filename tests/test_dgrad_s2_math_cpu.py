"""The arithmetic behind ssa_conv2d_dgrad_s2 (csrc/conv_igemm.hip), restated in torch on the CPU and pinned to
autograd: the data gradient of a 3x3, stride-2, pad-1 convolution (the fuse / transition down-convs,
network/hrnetv2.py:218-250 of the reference) decomposed by OUTPUT PARITY.  Along one axis

    dx[2m]   = w[1] * dy[m]
    dx[2m+1] = w[2] * dy[m] + w[0] * dy[m+1]

so the pixels (2m+py, 2n+px) of dx are a dense stride-1 correlation of dy with (1+py) x (1+px) taps; tap j of
class p is forward tap s2_tap(p, j) -- the mapping `pack_one` (modes 4..7) and the kernel's gather (pad 0, taps
at +j) implement.  Also: the work list of the tile-balanced filter repack covers every filter element once."""
import pytest
import torch
import torch.nn.functional as F


def s2_tap(p, j):
    return 1 if p == 0 else 2 - 2 * j


def dgrad_s2_by_parity(dy, w, H, W):
    """dy [B,Cout,Ho,Wo], w [Cout,Cin,3,3] -> dx [B,Cin,H,W] by four dense class correlations."""
    B, Cout, Ho, Wo = dy.shape
    Cin = w.shape[1]
    dx = torch.zeros(B, Cin, H, W, dtype=dy.dtype)
    for py in (0, 1):
        for px in (0, 1):
            Hc, Wc = (H - py + 1) // 2, (W - px + 1) // 2        # pixels of dx with this parity
            if Hc <= 0 or Wc <= 0:
                continue
            # class operand [Cin][(jy, jx, co)] as ssa_pack_filter(mode 4 + 2*py + px) lays it out
            taps = [(jy, jx) for jy in range(1 + py) for jx in range(1 + px)]
            acc = torch.zeros(B, Cin, Hc, Wc, dtype=dy.dtype)
            for jy, jx in taps:
                wk = w[:, :, s2_tap(py, jy), s2_tap(px, jx)]       # [Cout, Cin]
                # dy[m + jy, n + jx], zero beyond the last row / column (the kernel's bounds check)
                sl = torch.zeros(B, Cout, Hc, Wc, dtype=dy.dtype)
                hh, ww = min(Hc, Ho - jy), min(Wc, Wo - jx)
                if hh > 0 and ww > 0:
                    sl[:, :, :hh, :ww] = dy[:, :, jy:jy + hh, jx:jx + ww]
                acc += torch.einsum("bohw,oi->bihw", sl, wk)
            dx[:, :, py::2, px::2] = acc                           # the strided output map (o_mul = 2)
    return dx


@pytest.mark.parametrize("shape", [(1, 8, 8), (2, 7, 10), (1, 9, 9), (1, 1, 6), (1, 2, 1), (1, 31, 50)])
def test_parity_classes_equal_the_transposed_convolution(shape):
    B, H, W = shape
    Cin, Cout = 5, 7
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, stride=2, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (want,) = torch.autograd.grad(y, x, dy)
    got = dgrad_s2_by_parity(dy, w, H, W)
    assert y.shape[2:] == ((H - 1) // 2 + 1, (W - 1) // 2 + 1)       # the size check of the C entry point
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)
    # every pixel of dx belongs to exactly one class; the classes use 1 + 2 + 2 + 4 = 9 taps in total
    assert sum((1 + py) * (1 + px) for py in (0, 1) for px in (0, 1)) == 9


def test_repack_tile_list_covers_every_filter_element_once():
    from semseg_amd import _lib
    L = _lib.lib()
    for (Cout, Cin, KH, KW) in [(48, 48, 3, 3), (19, 512, 1, 1), (512, 720, 3, 3), (24, 3, 7, 7), (40, 300, 1, 1)]:
        ct = L.ssa_pack_tile_channels(KH, KW)
        assert ct >= 8 and ct % 8 == 0 and 32 * ((ct * KH * KW) | 1) * 4 <= 64 * 1024
        seen = torch.zeros(Cout, Cin, dtype=torch.int32)
        for co0 in range(0, Cout, 32):
            for ci0 in range(0, Cin, ct):
                seen[co0:co0 + 32, ci0:ci0 + ct] += 1
        assert int(seen.min()) == 1 and int(seen.max()) == 1
    assert L.ssa_pack_tile_channels(9, 9) == 0                         # 81 taps: falls back to the batched kernel
