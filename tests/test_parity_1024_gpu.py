"""Parity AT THE BENCHMARKED CONFIGURATION: HRNet-OCR-MScale, 1x3x1024x1024, two scales, RMI loss
(network/ocrnet.py:264-334, loss/rmi.py:70-215) -- the graph bench.py times.

1. test_teacher_forced_ops_at_1024: every operator call of the training step (forward AND backward),
   teacher-forced (tests/teacher_backend.py): each HIP op gets the oracle's bf16-rounded input at its
   real 1024^2-crop shape and must match the oracle's output to ONE-ROUNDING tolerance (bf16 outputs
   1e-2 max / 4e-3 mean relative, fp32 outputs 2e-3 / 5e-4, fused conv+BN / residual blocks 3e-2 / 6e-3,
   parameter gradients 5e-2 max / 8e-3 mean -- the mean is two bf16 roundings, the max allows single
   ReLU-mask flips, tests/teacher_backend.py).  No error accumulates, so a 1 % error of any kernel at any real
   shape fails.  The launched kernel instantiations are recorded and the shape-dependent dispatch
   classes (head halo GEMM, head weight gradient, 256-pixel/two-n-block 48-channel tile, grouped
   weight-gradient tile kernel) must be among them.
2. test_captured_train_step_at_1024: the bench.py step itself (zero_grad + forward + backward + fused SGD,
   captured in a hipGraph and replayed) against oracle.model.Net on the same weights and batch:
   loss, BatchNorm running statistics, and the per-parameter gradient cosines against the bf16-storage
   emulation (the tolerance rule of tests/test_e2e_gpu.py -- end to end the logits of a random-weight
   network carry the bf16 noise floor, north_star's 1e-3 is met per op, not across 450 layers).
"""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CROP = int(os.environ.get("SSA_PARITY_CROP", "1024"))


def _bench_batch(crop):
    sys.path.insert(0, ROOT)
    import bench
    return bench.synth_batch(1, crop, crop, 0, "cpu")


def _build(sd=None):
    from semseg_amd.config import cfg
    from semseg_amd.loss import RMILoss
    from semseg_amd.network import ocrnet
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05
    cfg.LOSS.OCR_AUX_RMI = False
    cfg.MODEL.N_SCALES = None
    cfg.MODEL.BNFUNC = None
    net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255))
    if sd is None:
        from test_e2e_gpu import parity_state_dict
        sd = parity_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0)
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    return net.train(), sd


def test_teacher_forced_ops_at_1024():
    from semseg_amd import ops, hip_backend as hb
    from teacher_backend import TeacherBackend
    images, gts = _bench_batch(CROP)
    cpu_net, sd = _build()
    hip_net = copy.deepcopy(cpu_net).cuda().train()
    tb = TeacherBackend(cpu_net, hip_net)
    prev = ops._BACKEND
    ops._set_backend_for_tests(tb)
    hb.clear_pack_cache()
    hb.profile_begin()
    # fp16 storage (tests/test_amp_fp16_gpu.py, SSA_ACT_DTYPE=fp16): the backward pass starts from the loss scale, as
    # amp.scale_loss makes it -- the teacher's gradients (rounded to fp16 like the HIP path's) would flush otherwise
    from util import ACT_DTYPE
    S = 65536.0 if ACT_DTYPE == torch.float16 else 1.0
    if S != 1.0:
        hb.enable_fp16_training()
    try:
        loss = cpu_net({"images": images, "gts": gts})
        n_fwd = tb.rec.n_ops
        (loss * S).backward()
        torch.cuda.synchronize()
    finally:
        kernels = hb.profile_end()
        ops._set_backend_for_tests(prev)
    print(tb.rec.summary(20))
    names = sorted(k["kernel"] for k in kernels)
    print("kernel instantiations launched: %d" % len(names))
    for n in names:
        print("   ", n)
    bwd_cmp = sum(1 for r in tb.rec.rows if r[2].startswith("d"))
    print("forward ops %d, backward comparisons %d" % (n_fwd, bwd_cmp))
    assert n_fwd > 100 and bwd_cmp > 1500
    if CROP >= 1024:
        fams = {n.split("<")[0] for n in names}
        halo3 = "ConvHaloGemm3" if os.environ.get("SSA_HALO3_REG") == "0" else "ConvHaloReg3"      # (csrc/conv_halo_gemm.hip's switch)
        for f in (halo3, "ConvHaloGemm1", "ConvGemmWide1", "ConvWgradHead3", "ConvWgradHead", "ConvTile", "ConvWgradTile",
                  "ConvWgradTileA", "ConvIgemm", "ConvWgradTr", "BnApplyTrainK", "BnBwdReduceK", "BnBwdReduceXK", "BnBwdApplyK",
                  "BnBwdApplyXK"):
            assert f in fams, "dispatch class %s is not on the traced path" % f
        # the trunk levels' 3x3 convs: the resident (48 channels) and the streamed instantiation behind one kernel,
        # conv_tile_p.hip: plain forward <0>, data gradient with the residual gradient <1> / the BatchNorm-backward
        # sums <2> in the epilogue
        for inst in ("ConvTilePK<0>", "ConvTilePK<1>", "ConvTilePK<2>"):
            assert inst in names, "%s is not on the path" % inst
        # the trunk's 3x3 weight gradients: 48 channels on the 4-wave tile kernel, the 96-channel blocks (96 / 192 / 384
        # channels) on the all-taps form (round 6), the large 1x1 head convs on the 256 x 256 GEMM
        assert any(n.startswith("ConvWgradTile<48,") for n in names) and "ConvWgradTileA<0>" in names
    assert not tb.rec.failures(), tb.rec.summary(30)


def test_captured_train_step_at_1024():
    from semseg_amd import ops, hip_backend as hb
    from semseg_amd.loss.optimizer import FusedSGD
    from oracle_backend import OracleBackend
    from bf16_emu_backend import Bf16EmuBackend
    from test_e2e_gpu import _run
    images, gts = _bench_batch(CROP)
    net, sd = _build()
    lr_, gr, sr = _run(OracleBackend(), sd, images, gts, True)
    le, ge, _ = _run(Bf16EmuBackend(), sd, images, gts, True)

    prev = ops._BACKEND
    ops._set_backend_for_tests(ops.HipBackend())
    hb.clear_pack_cache()
    try:
        net = net.cuda()
        init = {k: v.clone() for k, v in net.state_dict().items()}
        # lr = 0: the captured program is bench.py's (optimizer kernels included), the weights stay put
        optim = FusedSGD(net.parameters(), lr=0.0, momentum=0.9, weight_decay=0.0)
        inputs = {"images": images.cuda(), "gts": gts.cuda()}
        static_loss = torch.zeros((), device="cuda")

        def step():
            optim.zero_grad(set_to_none=True)
            loss = net(inputs)
            loss.backward()
            optim.step()
            static_loss.copy_(loss.detach())

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        optim.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            step()
        torch.cuda.synchronize()
        with torch.no_grad():                       # running statistics back to the initial state, one replay
            for k, v in net.state_dict().items():
                v.copy_(init[k])
        graph.replay()
        torch.cuda.synchronize()
        lh = float(static_loss)
        gh = {n: p.grad.detach().float().cpu() for n, p in net.named_parameters() if p.grad is not None}
        sh = {k: v.detach().float().cpu() for k, v in net.state_dict().items() if "running_" in k}
        for n, p in net.named_parameters():
            assert torch.equal(p.detach().cpu(), init[n].cpu()), n
    finally:
        ops._set_backend_for_tests(prev)
    print("crop %d train loss: hip (captured replay) %.6f emu %.6f oracle %.6f" % (CROP, lh, le, lr_))
    assert abs(lh - lr_) <= 2e-3 * abs(lr_) + 2 * abs(le - lr_)
    assert set(gh) == set(gr)

    def cosines(g):
        out = {}
        for name, r in gr.items():
            if float(r.norm()) < 1e-10:
                continue
            assert torch.isfinite(g[name]).all(), name
            out[name] = float((g[name] * r).sum() / (g[name].norm() * r.norm() + 1e-30))
        return out
    ch, ce = cosines(gh), cosines(ge)
    vh, ve = sorted(ch.values()), sorted(ce.values())
    print("grad cosine vs oracle: hip min %.4f p10 %.4f median %.4f | emu min %.4f p10 %.4f median %.4f (n=%d)" % (
        vh[0], vh[len(vh) // 10], vh[len(vh) // 2], ve[0], ve[len(ve) // 10], ve[len(ve) // 2], len(vh)))
    nr = {k: float(gh[k].norm() / (gr[k].norm() + 1e-30)) for k in ch}
    bad = [(k, ch[k], ce[k], nr[k]) for k in ch if ce[k] >= 0.5 and (ch[k] < 0.5 * ce[k] or not 0.6 <= nr[k] <= 1.6)]
    assert not bad, bad[:5]
    assert vh[len(vh) // 2] >= ve[len(ve) // 2] - 0.05
    assert vh[len(vh) // 10] >= ve[len(ve) // 10] - 0.05
    worst = max(float((sh[k] - sr[k]).abs().max() / (sr[k].abs().max() + 1e-12)) for k in sr)
    print("running stats worst rel %.4g" % worst)
    assert worst < 3e-2
