"""Grouped (multi-problem) launches and the Functions built on them, against the oracle.

The HRNet trunk runs the (branch x scale pass) problems of a depth level as ONE autograd node whose
launches leave as one launch per kernel instantiation (csrc/group.h); parameter gradients are
accumulated into the gradient arena by the kernels and published at the end of backward.  Checked
here, through the C ABI, on the real heterogeneous mixes of a HighResolutionModule level:
  * a grouped launch == the same problems launched one by one, BIT FOR BIT (same kernel body, only the
    block index is virtual);
  * ConvGroupFn / BnActGroupFn / BasicBlockGroupFn / SumActGroupFn / BilinearGroupFn forward,
    data gradients and PUBLISHED parameter gradients (two passes through one layer accumulate)
    against oracle/ops.py at the one-bf16-rounding tolerance of tests/test_kernels_gpu.py.
"""
import math

import pytest
import torch

from util import bf16_round, check_close, check_close_robust, nhwc, nchw, ACT_DTYPE

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (C, H, W) of the four branches of an HRNet-W48 stage-4 module at a 128x96 crop, plus the 0.5x pass
LEVEL = [(48, 32, 24), (96, 16, 12), (192, 8, 6), (384, 4, 3), (48, 16, 12), (96, 8, 6), (192, 4, 3), (384, 2, 2)]
# larger problems: several workgroups per problem, ragged tiles, both 48-channel dispatch classes
LEVEL_BIG = [(48, 256, 128), (48, 70, 75), (96, 64, 64), (192, 40, 33), (384, 17, 20), (64, 37, 41)]


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return bf16_round(torch.randn(*shape, generator=g) * scale)


def _dev(x):
    return nhwc(x).to(DEV).to(ACT_DTYPE).contiguous()


def _hb():
    from semseg_amd import hip_backend as hb
    hb.clear_pack_cache()
    hb.begin_step(torch.device(DEV))
    return hb


@pytest.mark.parametrize("level", [LEVEL, LEVEL_BIG])
def test_grouped_launch_is_bit_identical_to_single_launches(level):
    hb = _hb()
    from semseg_amd._lib import lib
    xs = [_dev(_rand(1, C, H, W, seed=i)) for i, (C, H, W) in enumerate(level)]
    ws = [(_rand(C, C, 3, 3, seed=50 + i, scale=1.0 / math.sqrt(9 * C))).to(DEV) for i, (C, H, W) in enumerate(level)]
    spec = tuple((1, 1, 1, False, True) for _ in level)
    flat = []
    for x, w in zip(xs, ws):
        flat += [x, w, None]
    for (C, H, W), w in zip(level, ws):
        hb._packed_filter(w, 2, C, 0)             # warm the filter cache: only conv launches are counted below
    lib().ssa_launch_count(1)
    grouped = hb.ConvGroupFn.apply(spec, *flat)
    n_grouped = lib().ssa_launch_count(1)
    stats_g = [hb._PENDING_STATS.pop(y.data_ptr())[0].clone() for y in grouped]
    single = [hb.ConvGroupFn.apply((spec[i],), xs[i], ws[i], None)[0] for i in range(len(level))]
    n_single = lib().ssa_launch_count(1)
    stats_s = [hb._PENDING_STATS.pop(y.data_ptr())[0].clone() for y in single]
    torch.cuda.synchronize()
    print("launches: grouped %d, one by one %d" % (n_grouped, n_single))
    assert n_grouped < n_single == len(level)
    for yg, ys in zip(grouped, single):
        assert torch.equal(yg, ys)
    for sg, ss, (C, H, W) in zip(stats_g, stats_s, level):
        # fp64 atomics of fp32 partials: the order differs, the sums agree to fp64 rounding
        a, b = sg.view(-1, 2, C).sum(0), ss.view(-1, 2, C).sum(0)
        assert (a - b).abs().max() <= 1e-9 * b.abs().max() + 1e-12
    # BatchNorm apply, sums, bilinear: grouped == single, bit for bit
    zs_g = hb.SumActGroupFn.apply(True, tuple(2 for _ in level), *[t for y, x in zip(grouped, xs) for t in (y, x)])
    zs_s = [hb.SumActGroupFn.apply(True, (2,), y, x)[0] for y, x in zip(single, xs)]
    up_g = hb.BilinearGroupFn.apply(tuple((2 * H, 2 * W + 1, False) for C, H, W in level), *grouped)
    up_s = [hb.BilinearGroupFn.apply(((2 * H, 2 * W + 1, False),), y)[0] for y, (C, H, W) in zip(single, level)]
    torch.cuda.synchronize()
    for a, b in zip(list(zs_g) + list(up_g), zs_s + up_s):
        assert torch.equal(a, b)


@pytest.mark.parametrize("level", [LEVEL, LEVEL_BIG])
def test_conv_group_against_oracle_with_published_weight_gradients(level):
    """Two 'passes' through every layer (the same Parameter in two problems of different size): the
    weight gradient that appears in .grad is the SUM over both, accumulated by ONE reduce."""
    from oracle import ops as O
    hb = _hb()
    n = len(level)
    params, xs_ref, xs_dev, ws_ref = [], [], [], []
    for i, (C, H, W) in enumerate(level):
        w = _rand(C, C, 3, 3, seed=70 + i, scale=1.0 / math.sqrt(9 * C))
        params.append(torch.nn.Parameter(w.to(DEV)))
        ws_ref.append(w.clone().requires_grad_(True))
    jobs = [(i, 1) for i in range(n)] + [(i, 2) for i in range(n)]          # (layer, pass): pass 2 = half size
    flat, spec, outs_ref = [], [], []
    for k, (i, p) in enumerate(jobs):
        C, H, W = level[i]
        h, w_ = (H, W) if p == 1 else (max(H // 2, 1), max(W // 2, 1))
        x = _rand(1, C, h, w_, seed=200 + k)
        xr = x.clone().requires_grad_(True)
        xs_ref.append(xr)
        xd = _dev(x).requires_grad_(True)
        xs_dev.append(xd)
        flat += [xd, params[i], None]
        spec.append((1, 1, 1, False, False))
        outs_ref.append(O.conv2d(xr, ws_ref[i], None, 1, 1, 1))
    outs = hb.ConvGroupFn.apply(tuple(spec), *flat)
    gys = [_rand(*o.shape, seed=300 + k) for k, o in enumerate(outs_ref)]
    torch.autograd.backward(outs_ref, gys)
    torch.autograd.backward(list(outs), [_dev(g) for g in gys])
    torch.cuda.synchronize()
    assert not hb._WGRAD_Q and not hb._GRADS.slots
    for k, (i, p) in enumerate(jobs):
        check_close("conv fwd job %d" % k, nchw(outs[k].float()), outs_ref[k])
        check_close("conv dgrad job %d" % k, nchw(xs_dev[k].grad.float()), xs_ref[k].grad)
    for i in range(n):
        assert params[i].grad is not None
        check_close("published wgrad layer %d (C=%d)" % (i, level[i][0]), params[i].grad, ws_ref[i].grad, 1e-2, 4e-3)
    # a second backward without zero_grad accumulates into the published .grad
    outs2 = hb.ConvGroupFn.apply(tuple(spec), *flat)
    torch.autograd.backward(list(outs2), [_dev(g) for g in gys])
    torch.cuda.synchronize()
    check_close("accumulated wgrad", params[0].grad, 2 * ws_ref[0].grad, 1e-2, 4e-3)


def _block(C, seed):
    from semseg_amd.network.hrnetv2 import BasicBlock
    from semseg_amd.config import cfg
    cfg.MODEL.BNFUNC = None
    blk = BasicBlock(C, C)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for conv in (blk.conv1, blk.conv2):
            conv.weight.copy_(bf16_round(torch.randn(conv.weight.shape, generator=g) * (1.4 / math.sqrt(9 * C))))
        for bn in (blk.bn1, blk.bn2):
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
    return blk


def _oracle_block(blk, x, p):
    """The residual block in fp32 with the product's storage rounding (conv outputs and the BN
    output feeding conv2 are bf16 tensors on the HIP path)."""
    from oracle import ops as O
    from bf16_emu_backend import R
    rm, rv = torch.zeros(x.shape[1]), torch.ones(x.shape[1])
    y1 = R(O.conv2d(x, p["w1"], None, 1, 1, 1))
    a1 = R(torch.relu(O.batch_norm(y1, p["g1"], p["b1"], rm.clone(), rv.clone(), True, 0.1, 1e-5)))
    y2 = R(O.conv2d(a1, p["w2"], None, 1, 1, 1))
    return torch.relu(O.batch_norm(y2, p["g2"], p["b2"], rm.clone(), rv.clone(), True, 0.1, 1e-5) + x)


@pytest.mark.parametrize("level", [[(48, 32, 24), (96, 16, 12), (192, 16, 12), (384, 8, 6)],
                                   [(48, 128, 160), (96, 37, 41), (192, 33, 30), (384, 16, 16)],
                                   # every problem (both passes) wide enough for the persistent kernel
                                   [(48, 12, 72), (96, 9, 40), (192, 8, 34), (384, 6, 32)]])
def test_basic_block_group_against_oracle(level):
    """BasicBlockGroupFn: forward, data gradient, and the published gradients of all six parameters per
    block, with every block used by TWO problems (the scale passes) of different size."""
    from semseg_amd import ops
    hb = _hb()
    be = ops.HipBackend()
    blocks = [_block(C, 400 + i) for i, (C, H, W) in enumerate(level)]
    refs = [{"w1": b.conv1.weight.detach().clone().requires_grad_(True), "g1": b.bn1.weight.detach().clone().requires_grad_(True),
             "b1": b.bn1.bias.detach().clone().requires_grad_(True), "w2": b.conv2.weight.detach().clone().requires_grad_(True),
             "g2": b.bn2.weight.detach().clone().requires_grad_(True), "b2": b.bn2.bias.detach().clone().requires_grad_(True)}
            for b in blocks]
    blocks = [b.to(DEV).train() for b in blocks]
    jobs = [(i, 1) for i in range(len(level))] + [(i, 2) for i in range(len(level))]
    xs_ref, xs_dev, outs_ref = [], [], []
    for k, (i, p) in enumerate(jobs):
        C, H, W = level[i]
        h, w_ = (H, W) if p == 1 else (max(H // 2, 2), max(W // 2, 2))
        x = _rand(1, C, h, w_, seed=500 + k)
        xr = x.clone().requires_grad_(True)
        xs_ref.append(xr)
        xs_dev.append(_dev(x).requires_grad_(True))
        outs_ref.append(_oracle_block(blocks[i], xr, refs[i]))
    outs = be.basic_block([blocks[i] for i, p in jobs], xs_dev)
    be.end_forward()
    gys = [_rand(*o.shape, seed=600 + k) for k, o in enumerate(outs_ref)]
    torch.autograd.backward(outs_ref, gys)
    torch.autograd.backward(outs, [_dev(g) for g in gys])
    torch.cuda.synchronize()
    assert not hb._WGRAD_Q and not hb._GRADS.slots
    for k in range(len(jobs)):
        check_close("block out job %d" % k, nchw(outs[k].float()), outs_ref[k], 2e-2, 6e-3)
        check_close_robust("block dx job %d" % k, nchw(xs_dev[k].grad.float()), xs_ref[k].grad, 3e-2, 1e-2)
    for i, b in enumerate(blocks):
        C = level[i][0]
        for name, p in (("w1", b.conv1.weight), ("g1", b.bn1.weight), ("b1", b.bn1.bias), ("w2", b.conv2.weight),
                        ("g2", b.bn2.weight), ("b2", b.bn2.bias)):
            assert p.grad is not None, (i, name)
            # a ReLU-mask flip of one a1 element moves the 9*C weight-gradient entries its pixel touches
            check_close_robust("block %d (C=%d) d%s" % (i, C, name), p.grad, refs[i][name].grad, 3e-2, 1e-2, 1e-2)
        assert int(b.bn1.num_batches_tracked) == 2 and int(b.bn2.num_batches_tracked) == 2


def test_basic_block_group_equals_the_unfused_composition():
    """Same block, same inputs: the one-node block (fused backward epilogues, queued weight
    gradients) against conv_bn_act twice (generic grouped Functions, autograd's own residual add).
    The data gradient's residual add is bit-identical to the unfused bf16 add; the BatchNorm sums
    differ in summation order only."""
    from semseg_amd import ops
    hb = _hb()
    be = ops.HipBackend()
    res = []
    for fused in (True, False):
        blk = _block(96, 7).to(DEV).train()
        x = _dev(_rand(2, 96, 40, 56, seed=9)).requires_grad_(True)
        hb.begin_step(torch.device(DEV))
        if fused:
            out = be.basic_block([blk], [x])[0]
        else:
            mid = be.conv_bn_act(blk.conv1, blk.bn1, x, relu=True)
            out = be.conv_bn_act(blk.conv2, blk.bn2, mid, residual=x, relu=True)
        out.backward(_dev(_rand(2, 96, 40, 56, seed=10)))
        torch.cuda.synchronize()
        res.append((out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in blk.parameters()]))
    (o0, dx0, g0), (o1, dx1, g1) = res
    assert torch.equal(o0, o1)
    check_close_robust("dx fused vs unfused", dx0.float(), dx1.float(), 1e-2, 1e-3)
    for a, b in zip(g0, g1):
        check_close("param grad fused vs unfused", a, b, 2e-3, 5e-4)


def test_bn_group_against_oracle():
    from oracle import ops as O
    from semseg_amd import ops, nn as snn
    hb = _hb()
    be = ops.HipBackend()
    shapes = [(48, 2, 17, 23), (96, 1, 9, 11), (720, 1, 12, 16), (48, 1, 30, 20)]
    bns = [snn.BatchNorm2d(C) for C, _, _, _ in shapes[:3]]
    bns.append(bns[0])                                        # two passes through one layer
    for bn in bns[:3]:
        with torch.no_grad():
            bn.weight.copy_(torch.rand(bn.num_features) + 0.5)
            bn.bias.copy_(torch.randn(bn.num_features) * 0.1)
    refs = [(bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)) for bn in bns[:3]]
    refs.append(refs[0])
    xs, xrs, ress, rrs, ys = [], [], [], [], []
    for k, (C, B, H, W) in enumerate(shapes):
        x = bf16_round(_rand(B, C, H, W, seed=k) * 1.5 + 0.2)
        xr = x.clone().requires_grad_(True)
        r = _rand(B, C, H, W, seed=20 + k) if k % 2 == 0 else None
        rr = r.clone().requires_grad_(True) if r is not None else None
        y = O.batch_norm(xr, refs[k][0], refs[k][1], torch.zeros(C), torch.ones(C), True, 0.1, 1e-5)
        if rr is not None:
            y = y + rr
        ys.append(torch.relu(y))
        xs.append(_dev(x).requires_grad_(True))
        xrs.append(xr)
        ress.append(_dev(r).requires_grad_(True) if r is not None else None)
        rrs.append(rr)
    bns_d = [bn.to(DEV).train() for bn in bns]
    zs = be.batch_norm_act(xs, bns_d, ress, True, None)
    be.end_forward()
    gys = [_rand(*y.shape, seed=40 + k) for k, y in enumerate(ys)]
    torch.autograd.backward(ys, gys)
    torch.autograd.backward(zs, [_dev(g) for g in gys])
    torch.cuda.synchronize()
    for k in range(len(shapes)):
        check_close("bn fwd %d" % k, nchw(zs[k].float()), ys[k])
        check_close("bn dx %d" % k, nchw(xs[k].grad.float()), xrs[k].grad, 2e-2, 6e-3)
        if ress[k] is not None:
            check_close("bn dres %d" % k, nchw(ress[k].grad.float()), rrs[k].grad)
    for k in range(3):
        check_close("bn dgamma %d (published)" % k, bns_d[k].weight.grad, refs[k][0].grad, 1e-2, 4e-3)
        check_close("bn dbeta %d (published)" % k, bns_d[k].bias.grad, refs[k][1].grad, 1e-2, 4e-3)
    assert int(bns_d[0].num_batches_tracked) == 2


def test_hr_module_lockstep_against_oracle():
    """A whole HighResolutionModule (4 branches x 2 blocks, all fuse layers) on two scale passes in
    lockstep against the same module on the oracle's operators: outputs and input gradients."""
    from semseg_amd import ops
    from semseg_amd.config import cfg
    from semseg_amd.network.hrnetv2 import HighResolutionModule, BasicBlock
    from bf16_emu_backend import Bf16EmuBackend
    cfg.MODEL.BNFUNC = None
    torch.manual_seed(3)
    ch = [48, 96, 192, 384]
    mod = HighResolutionModule(4, BasicBlock, [2, 2, 2, 2], ch, ch).train()
    for m in mod.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
            with torch.no_grad():
                m.weight.copy_(bf16_round(m.weight * 0.7))
    sizes = [[(128, 96), (64, 48), (32, 24), (16, 12)], [(64, 48), (32, 24), (16, 12), (8, 6)]]
    xs = [[_rand(1, ch[i], *sizes[p][i], seed=10 * p + i) for p in range(2)] for i in range(4)]

    def run(backend, device):
        prev = ops._BACKEND
        ops._set_backend_for_tests(backend)
        try:
            m = mod if device == "cpu" else __import__("copy").deepcopy(mod).to(device)
            if device != "cpu":
                _hb()
            ins = [[(nhwc(x).to(device).to(backend.act_dtype).contiguous()).requires_grad_(True) for x in br] for br in xs]
            outs = m(ins)
            flat = [o for br in outs for o in br]
            gys = [nhwc(_rand(1, o.shape[3], o.shape[1], o.shape[2], seed=90 + k)) for k, o in enumerate(flat)]
            torch.autograd.backward(flat, [g.to(device).to(o.dtype) for g, o in zip(gys, flat)])
            if device != "cpu":
                torch.cuda.synchronize()
            return ([o.detach().float().cpu() for o in flat], [x.grad.float().cpu() for br in ins for x in br],
                    {n: p.grad.detach().float().cpu() for n, p in m.named_parameters()})
        finally:
            ops._set_backend_for_tests(prev)

    ro, rg, rp = run(Bf16EmuBackend(), "cpu")
    ho, hg, hp = run(ops.HipBackend(), DEV)
    for k, (a, b) in enumerate(zip(ho, ro)):
        check_close_robust("module out %d" % k, a, b, 6e-2, 2e-2, 1e-2)
    for k, (a, b) in enumerate(zip(hg, rg)):
        # two residual blocks + the fuse layers deep, training-mode BatchNorm over as few as 48 pixels: the
        # bf16 noise of the emulation itself is a few percent here; a wiring error is O(1)
        check_close_robust("module dx %d" % k, a, b, 1.5e-1, 6e-2, 2e-2)
    bad = []
    for n in rp:
        cos = float((hp[n] * rp[n]).sum() / (hp[n].norm() * rp[n].norm() + 1e-30))
        ratio = float(hp[n].norm() / (rp[n].norm() + 1e-30))
        if cos < 0.98 or not 0.9 < ratio < 1.1:       # a wrong kernel / wiring gives cos ~ 0 for what it touches
            bad.append((n, cos, ratio))
    assert not bad, bad[:8]


def test_upsample_cat_equals_resize_then_cat():
    """UpsampleCatGroupFn (the resize kernel writes into its channel slice of the concatenated buffer) == BilinearGroupFn
    followed by torch.cat, bit for bit, forward and backward -- two groups of different size (the scale passes)."""
    hb = _hb()
    shapes = [[(48, 32, 24), (96, 16, 12), (192, 8, 6), (384, 4, 3)], [(48, 16, 12), (96, 8, 6), (192, 4, 3), (384, 2, 2)]]
    xs = [[_dev(_rand(1, C, H, W, seed=900 + 10 * g + i)) for i, (C, H, W) in enumerate(grp)] for g, grp in enumerate(shapes)]
    gys = [_dev(_rand(1, 720, grp[0][1], grp[0][2], seed=950 + g)) for g, grp in enumerate(shapes)]

    def run(fused):
        leaves = [[x.clone().requires_grad_(True) for x in grp] for grp in xs]
        if fused:
            outs = hb.UpsampleCatGroupFn.apply(tuple(len(g) for g in leaves), *[t for g in leaves for t in g])
        else:
            outs = []
            for grp in leaves:
                size = tuple(grp[0].shape[1:3])
                ups = hb.BilinearGroupFn.apply(tuple((size[0], size[1], False) for _ in grp[1:]), *grp[1:])
                outs.append(torch.cat([grp[0]] + list(ups), dim=3))
        torch.autograd.backward(list(outs), gys)
        return [o.detach() for o in outs], [[t.grad for t in grp] for grp in leaves]

    o_f, g_f = run(True)
    o_r, g_r = run(False)
    for a, b in zip(o_f, o_r):
        assert a.shape == b.shape and torch.equal(a, b)
    for ga, gb in zip(g_f, g_r):
        for a, b in zip(ga, gb):
            assert torch.equal(a.contiguous(), b.contiguous())


# ----------------------------------------------------------------- concatenation without a copy (ops.cat_slots)
@pytest.mark.parametrize("training", [True, False])
def test_conv_bn_into_cat_slots_equals_torch_cat(training):
    """SpatialOCR_Module's torch.cat([context, feats], 1) (network/ocr_utils.py:151): with ops.cat_slots the two
    producers (conv + BatchNorm + ReLU, two scale passes as a list) write channel slices of ONE buffer per pass and the
    concatenation is that buffer (hip_backend.CatViewFn) -- forward values, the consumer's output and every gradient
    must equal the copying form bit for bit (same kernels, only the output pixel stride differs)."""
    from semseg_amd import ops, nn as snn
    hb = _hb()
    be = ops.HipBackend()
    torch.manual_seed(3)
    ca, cb, cin = 24, 40, 32
    shapes = [(1, 12, 20), (1, 6, 10)]                      # the 1.0x and the 0.5x pass
    xs = [_rand(b, cin, h, w, seed=70 + i) for i, (b, h, w) in enumerate(shapes)]

    def build():
        torch.manual_seed(5)
        convs = [snn.Conv2d(cin, ca, 1, bias=False), snn.Conv2d(cin, cb, 3, padding=1, bias=True),
                 snn.Conv2d(ca + cb, 16, 1, bias=False)]
        bns = [snn.BatchNorm2d(ca), snn.BatchNorm2d(cb), snn.BatchNorm2d(16)]
        mods = torch.nn.ModuleList(convs + bns).to(DEV).train(training)
        return mods[:3], mods[3:]

    def run(use_slots):
        convs, bns = build()
        hb.clear_pack_cache()
        hb.begin_step(torch.device(DEV))
        xin = [_dev(x).requires_grad_(True) for x in xs]
        slots = be.cat_slots(xin, (ca, cb)) if use_slots else None
        a = be.conv_bn_act(convs[0], bns[0], xin, relu=True, **({"out": [s[0] for s in slots]} if slots else {}))
        b = be.conv_bn_act(convs[1], bns[1], xin, relu=True, **({"out": [s[1] for s in slots]} if slots else {}))
        cat = [be.cat([ai, bi]) for ai, bi in zip(a, b)]
        if use_slots:
            assert all(c.data_ptr() == s[0].data_ptr() and c.is_contiguous() for c, s in zip(cat, slots)), "a copy was made"
        y = be.conv_bn_act(convs[2], bns[2], cat, relu=False)
        be.end_forward()
        outs = [t.detach().float().cpu() for t in list(a) + list(b) + list(cat) + list(y)]
        grads = []
        if training:
            g = [_dev(_rand(*t.permute(0, 3, 1, 2).shape, seed=90 + i)) for i, t in enumerate(y)]
            torch.autograd.backward(list(y), g)          # (the arena publishes the parameter gradients at its end)
            if DEV == "cuda":
                torch.cuda.synchronize()
            grads = [t.grad.detach().float().cpu() for t in xin]
            grads += [p.grad.detach().float().cpu() for m in list(convs) + list(bns) for p in m.parameters()
                      if p.grad is not None]
        elif DEV == "cuda":
            torch.cuda.synchronize()
        return outs, grads

    o1, g1 = run(True)
    o0, g0 = run(False)
    assert len(o1) == len(o0) and len(g1) == len(g0) and (not training or len(g1) >= 2 + 5)
    for i, (p, q) in enumerate(zip(o1, o0)):
        assert torch.equal(p, q), ("forward tensor %d differs" % i, float((p - q).abs().max()))
    for i, (p, q) in enumerate(zip(g1, g0)):
        check_close("cat-slot gradient %d" % i, p, q, 1e-6, 1e-6)


def test_inference_conv_bn_as_one_launch_equals_the_two_launches():
    """Inference (autograd off, BatchNorm in evaluation mode): the trunk's 3x3 convs carry normalisation, residual add
    and ReLU as their epilogue (ssa_conv2d_tile_p aux_mode 3 / 4, hip_backend.conv_bn_infer_group).  The epilogue works on
    the 16-bit-rounded conv output with the apply pass's own arithmetic, so the result must equal conv -> ssa_bn_apply
    BIT FOR BIT -- for a BasicBlock level as the evaluation forward issues it (conv1: ReLU, conv2: residual + ReLU),
    without ReLU, with ragged tiles, and next to problems the fused kernel does not take (64 channels, a 1x1 conv, a
    biased conv), which must come out of the same call through the separate launches."""
    from semseg_amd import ops, nn as snn
    hb = _hb()
    be = ops.HipBackend()
    level = [(48, 40, 36), (96, 20, 18), (192, 17, 33), (384, 16, 16), (64, 24, 24)]

    def build():
        torch.manual_seed(11)
        convs = [snn.Conv2d(C, C, 3, padding=1, bias=False) for C, _, _ in level]
        convs += [snn.Conv2d(48, 96, 1, bias=False), snn.Conv2d(96, 96, 3, padding=1, bias=True),
                  snn.Conv2d(48, 96, 3, stride=2, padding=1, bias=False), snn.Conv2d(40, 72, 1, bias=True)]
        bns = [snn.BatchNorm2d(c.out_channels) for c in convs]
        for i, b in enumerate(bns):
            g = torch.Generator().manual_seed(200 + i)
            b.running_mean.copy_(torch.randn(b.num_features, generator=g) * 0.2)
            b.running_var.copy_(torch.rand(b.num_features, generator=g) + 0.5)
            b.weight.data.copy_(torch.rand(b.num_features, generator=g) + 0.5)
            b.bias.data.copy_(torch.randn(b.num_features, generator=g) * 0.1)
        mods = torch.nn.ModuleList(convs + bns).to(DEV).eval()
        return list(mods[:len(convs)]), list(mods[len(convs):])

    xs = [_dev(_rand(1, C, H, W, seed=300 + i)) for i, (C, H, W) in enumerate(level)]
    xs += [_dev(_rand(1, 48, 12, 20, seed=310)), _dev(_rand(1, 96, 20, 18, seed=311)), _dev(_rand(1, 48, 21, 19, seed=312)),
           _dev(_rand(2, 40, 7, 9, seed=313))]
    # (the 1x1, the stride-2 and the biased 40 -> 72 conv run on the implicit-GEMM kernel: epilogue there too, the last
    # two with a residual of the OUTPUT's shape; 64 channels and the biased 3x3 run on conv_tile.hip: two launches)
    ress = [_dev(_rand(1, C, H, W, seed=320 + i)) for i, (C, H, W) in enumerate(level)] + \
        [None, None, _dev(_rand(1, 96, 11, 10, seed=330)), _dev(_rand(2, 72, 7, 9, seed=331))]

    def run(fused):
        convs, bns = build()
        hb.clear_pack_cache()
        old = hb._FUSE_EVAL_BN
        hb._FUSE_EVAL_BN = fused
        try:
            outs = []
            with torch.no_grad():
                for _ in range(2):                  # second forward: coefficients from the batched refresh
                    hb.begin_step(torch.device(DEV))
                    a = be.conv_bn_act(convs, bns, xs, relu=True)
                    b = be.conv_bn_act(convs, bns, xs, residual=ress, relu=True)
                    c = be.conv_bn_act(convs, bns, xs, residual=ress, relu=False)
                    be.end_forward()
                    outs = list(a) + list(b) + list(c)
            if DEV == "cuda":
                torch.cuda.synchronize()
            return [t.float().cpu() for t in outs]
        finally:
            hb._FUSE_EVAL_BN = old

    modes, affine = [], []
    tile_conv, igemm = hb._tile_conv, hb._igemm

    def spy_igemm(*a, **k):
        affine.append(k.get("affine") is not None)
        return igemm(*a, **k)

    def spy(*a, **k):
        modes.append(k.get("mode", 0))
        return tile_conv(*a, **k)
    hb._tile_conv, hb._igemm = spy, spy_igemm
    try:
        two = run(False)
        assert not any(m >= 3 for m in modes) and not any(affine)
        del modes[:], affine[:]
        one = run(True)
    finally:
        hb._tile_conv, hb._igemm = tile_conv, igemm
    assert sum(affine) == 2 * 3 * 3 and all(affine), affine       # the three implicit-GEMM problems, every call
    # two forwards x (ReLU, residual + ReLU, residual) x the four trunk problems went through the epilogue
    assert sum(m == 4 for m in modes) == 2 * 2 * 4 and sum(m == 3 for m in modes) == 2 * 4, modes
    for i, (a, b) in enumerate(zip(one, two)):
        assert torch.equal(a, b), "output %d: max |diff| %g" % (i, float((a - b).abs().max()))
    assert all(float(t.abs().max()) > 0.1 for t in one)
    # under autograd the separate launches stay (the backward wants the conv output)
    convs, bns = build()
    hb.begin_step(torch.device(DEV))
    x = xs[0].clone().requires_grad_(True)
    z = be.conv_bn_act(convs[0], bns[0], x, relu=True)
    assert z.requires_grad
