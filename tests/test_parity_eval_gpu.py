"""Teacher-forced parity of the EVALUATION path at the shapes BASELINE.json's eval configurations name
(network/ocrnet.py:185-262 `nscale_forward`, utils/trnval_utils.py:82-198), at FULL size:

  configs[1]  HRNet-OCR, single scale, 1 x 3 x 1024 x 2048 (Cityscapes val)
  configs[2]  HRNet-OCR-MScale, scales {0.5, 1.0, 2.0} of a 1024 x 2048 image: the 2.0x pass is 2048 x 4096
  configs[4]  Mapillary: 65 classes, scales {0.5, 1.0, 2.0} of a 1152 x 1536 image (the 2.0x pass is 2304 x 3072 =
              7 Mpixel: the 65-wide fp32 heads, the OCR gather / attention over 65 object regions, the n-block
              tails of 65 output channels -- 0.46 G elements of logits; the reference's full-size recipe is a
              once-per-round run of tools/eval_bench.py)

Every operator call of the forward pass runs the HIP op on the teacher's (storage-rounded) inputs at its REAL shape
and must match the teacher's output to one-rounding tolerance (tests/teacher_backend.py) -- eval-mode BatchNorm, the
32-bit-offset guards of the halo kernels, the tile dispatch at 512 x 1024 ... 1024 x 2048 grids, none of which the
128 x 192 end-to-end eval test reaches.  The teacher is the fp32 oracle run ON THE DEVICE (oracle/ is plain torch:
SURVEY.md section 8c's second oracle), which is what makes these shapes affordable inside the driver's GPU run;
`test_device_oracle_equals_cpu_oracle` pins it to the CPU oracle (the one the golden fixtures pin to the reference).
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _image(h, w, seed):
    """A mean/std-normalised image with the 1/f spectrum of a photograph (octaves of noise, each twice as coarse and
    1.6x as strong as the one before): white noise loses most of its variance when it is resampled, so the 0.5x /
    2.0x passes of the hierarchical evaluation would see inputs 3-4x weaker than the 1.0x pass -- statistics no
    BatchNorm buffer can describe at once (measured: the 1.0x pass then reaches the OCR head 4.6x hotter than the
    others, its attention softmax one-hot).  A natural image looks alike at every scale; so does this one."""
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(1, 3, h, w)
    amp, step = 1.0, 1
    while min(h, w) // step >= 4:
        n = torch.randn(1, 3, -(-h // step), -(-w // step), generator=g)
        if step > 1:
            n = torch.nn.functional.interpolate(n, size=(h, w), mode="bilinear", align_corners=False)
        img += amp * n
        amp *= 1.6
        step *= 2
    return (img - img.mean((2, 3), keepdim=True)) / img.std((2, 3), keepdim=True)


def calibrate_eval_bn(net, image, device):
    """BatchNorm running statistics := the statistics of every BatchNorm input POOLED over all scale passes of one
    evaluation of `net` on `image` -- what a trained checkpoint's buffers are for its data distribution (the recipe of
    tests/golden/make_golden.py's `calib_buffers`, tests/test_e2e_gpu.py's fixture, tests/test_siblings_cpu.py::calibrate,
    extended to several passes).  `seeded_state_dict` leaves the buffers near (0, 1): with random kaiming weights an
    eval-mode network then multiplies its activations by ~1.02 per conv + BN and arrives at the OCR head with
    |x| ~ 1e3, q.k^T ~ 1e8 and one-hot softmaxes (round-4 review) -- a regime no trained checkpoint is in, where
    `ocr_attention` is an argmax select decided by fp32 summation order.

    The calibration pass runs the network's EVALUATION path (`nscale_forward` / the two-scale eval branch,
    network/ocrnet.py:185-262,321-326) on the oracle's operators with every BatchNorm normalising by its batch
    statistics, and accumulates (n, sum x, sum x^2) per layer over ALL its calls.  Pooling matters for the
    BatchNorms of the OCR proxy branch (f_object / f_down, network/ocr_utils.py:77-93): they see 19 samples per pass,
    and statistics of ONE pass (or of another image) put the other passes' proxies tens of standard deviations out.
    Same image, same size as the comparison pass: the statistics then describe exactly the tensors under test."""
    from semseg_amd import ops
    from oracle_backend import OracleBackend

    class _Calib(OracleBackend):
        """Every BatchNorm normalises by the statistics of its own input (what training mode does), computed here with
        torch.var_mean -- MIOpen's training-mode BatchNorm is not involved (it crashed on the [1, 65, 1, 256] proxy
        tensors of the Mapillary head) -- and the per-call (n, mean, M2) are merged over the calls (Chan et al.)."""

        def __init__(self):
            self.acc = {}

        def _batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
            xf = x.detach().float()
            var, mean = torch.var_mean(xf, dim=(0, 1, 2), unbiased=False)
            n = xf.numel() // xf.shape[-1]
            m2 = var.double() * n
            if id(bn) in self.acc:
                n0, mean0, m20 = self.acc[id(bn)]
                d = mean.double() - mean0
                tot = n0 + n
                self.acc[id(bn)] = (tot, mean0 + d * (n / tot), m20 + m2 + d * d * (n0 * n / tot))
            else:
                self.acc[id(bn)] = (n, mean.double(), m2)
            y = (xf - mean) * torch.rsqrt(var + bn.eps) * bn.weight.detach().float() + bn.bias.detach().float()
            if residual is not None:
                y = y + residual
            if relu:
                y = torch.relu(y)
            if post is not None:
                y = y * post[:, None, None, :]
            return y.to(x.dtype)

    prev = ops._BACKEND
    cal = _Calib()
    ops._set_backend_for_tests(cal)
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    try:
        net.eval().to(device)
        with torch.no_grad():
            net({"images": image.to(device)})
            for m in bns:
                n, mean, m2 = cal.acc[id(m)]
                m.running_mean.copy_(mean)
                m.running_var.copy_(m2 / max(n - 1, 1))
    finally:
        net.eval()
        ops._set_backend_for_tests(prev)
    return net


def _teacher_eval(factory, num_classes, n_scales, h, w, teacher_device="cuda"):
    import teacher_backend
    from teacher_backend import TeacherBackend
    from semseg_amd import ops, hip_backend as hb
    from semseg_amd.config import cfg
    from test_e2e_gpu import parity_state_dict
    saved = (cfg.DATASET.NUM_CLASSES, cfg.MODEL.N_SCALES, cfg.MODEL.BNFUNC)
    cfg.DATASET.NUM_CLASSES = num_classes
    cfg.MODEL.N_SCALES = n_scales
    cfg.MODEL.BNFUNC = None
    teacher_backend.F32_CHANNELS.add(num_classes)
    prev = ops._BACKEND
    import time
    t0 = time.time()
    try:
        from semseg_amd.network import ocrnet
        cpu_net = getattr(ocrnet, factory)(num_classes, None)
        sd = parity_state_dict([(k, tuple(v.shape)) for k, v in cpu_net.state_dict().items()], seed=0)
        # the attention's temperature: the key branch ends in a BatchNorm over 19 (65) proxies per pass, so the spread
        # of q.k^T between object regions of a random-weight network wanders 0.4 ... 2.4 from pass to pass; half the
        # gamma keeps every pass's softmax between flat and one-hot (the harness asserts it: MAX_SOFTMAX_PEAK)
        sd["ocr.ocr_distri_head.object_context_block.f_object.3.0.weight"] *= 0.5
        cpu_net.load_state_dict(sd)
        tf32 = torch.backends.cudnn.allow_tf32
        torch.backends.cudnn.allow_tf32 = False        # the device teacher is fp32 arithmetic, nothing less
        calibrate_eval_bn(cpu_net, _image(h, w, 11), teacher_device)      # (the image of the comparison pass below)
        hip_net = copy.deepcopy(cpu_net).cuda().eval()
        cpu_net = cpu_net.to(teacher_device)           # (the name is round 2's: the teacher, wherever it runs)
        tb = TeacherBackend(cpu_net, hip_net, teacher_device=teacher_device)
        ops._set_backend_for_tests(tb)
        hb.clear_pack_cache()
        hb.profile_begin()
        t1 = time.time()
        try:
            with torch.no_grad():
                out = cpu_net({"images": _image(h, w, 11).to(teacher_device)})
            torch.cuda.synchronize()
            print("%s %d classes %dx%d scales %s: build %.1f s, teacher-forced forward %.1f s" % (
                factory, num_classes, h, w, n_scales, t1 - t0, time.time() - t1))
        finally:
            kernels = hb.profile_end()
            torch.backends.cudnn.allow_tf32 = tf32
    finally:
        ops._set_backend_for_tests(prev)
        if num_classes != 19:
            teacher_backend.F32_CHANNELS.discard(num_classes)
        cfg.DATASET.NUM_CLASSES, cfg.MODEL.N_SCALES, cfg.MODEL.BNFUNC = saved
        hb.clear_pack_cache()
    print(tb.rec.summary(12))
    print("kernel instantiations:", sorted({k["kernel"].split("<")[0] for k in kernels}))
    assert tuple(out["pred"].shape) == (1, num_classes, h, w)
    assert tb.rec.n_ops > 60
    assert not tb.rec.failures(), tb.rec.summary(30)
    return tb, out


def test_device_oracle_equals_cpu_oracle():
    """The oracle's operators (plain torch fp32, NO storage rounding: the rounded teacher is chaotic in the last ulp
    of every stored tensor, its two runs differ by 1e-2) on the CPU and on the device: same evaluation outputs to
    fp32 summation-order noise -- the device teacher of the tests below IS the oracle the golden fixtures pin."""
    from semseg_amd import ops
    from semseg_amd.config import cfg
    from semseg_amd.network import ocrnet
    from oracle_backend import OracleBackend
    from test_e2e_gpu import parity_state_dict
    saved = (cfg.MODEL.N_SCALES, cfg.MODEL.BNFUNC)
    cfg.MODEL.N_SCALES, cfg.MODEL.BNFUNC = None, None
    prev = ops._BACKEND
    tf32 = torch.backends.cudnn.allow_tf32
    torch.backends.cudnn.allow_tf32 = False
    ops._set_backend_for_tests(OracleBackend())
    try:
        net = ocrnet.HRNet(19, None)
        net.load_state_dict(parity_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()], seed=0))
        img = _image(256, 384, 5)
        calibrate_eval_bn(net, img, "cpu")          # (the regime of the tests below; it restores the backend it found)
        net.eval()
        with torch.no_grad():
            a = net({"images": img})["pred"].float()
            b = net.cuda()({"images": img.cuda()})["pred"].float().cpu()
            t = net.cpu().double()({"images": img.double()})["pred"].float()     # the same operators in fp64: the truth
    finally:
        ops._set_backend_for_tests(prev)
        torch.backends.cudnn.allow_tf32 = tf32
        cfg.MODEL.N_SCALES, cfg.MODEL.BNFUNC = saved
    scale = float(t.abs().max()) + 1e-30
    e_cpu, e_dev, e_ab = (float((x - y).abs().max()) / scale for x, y in ((a, t), (b, t), (a, b)))
    print("max |d pred| / max |pred|: CPU fp32 vs fp64 %.2e, device fp32 vs fp64 %.2e, device vs CPU %.2e" % (e_cpu, e_dev, e_ab))
    # the calibrated random-weight network amplifies a relative perturbation ~100x (measured: 1e-6 on the image ->
    # 1.3e-4 on pred), so two fp32 runs that differ in summation order sit ~4e-5 from the fp64 result and from each other
    assert e_dev <= 3.0 * e_cpu + 2e-5, (e_dev, e_cpu)
    assert e_ab <= 1e-3, e_ab


def test_eval_mscale_three_scales_small():
    """The hierarchical evaluation at the end-to-end test's size (128 x 192: passes of 64 x 96, 128 x 192, 256 x 384 --
    the trunk's smallest branch is 2 x 3 pixels): the narrow-image dispatch classes, teacher-forced."""
    _teacher_eval("HRNet_Mscale", 19, [0.5, 1.0, 2.0], 128, 192)


def test_eval_hrnet_ocr_single_scale_1024x2048():
    """BASELINE configs[1]."""
    _teacher_eval("HRNet", 19, None, 1024, 2048)


def test_eval_mscale_three_scales_1024x2048():
    """BASELINE configs[2]: {0.5, 1.0, 2.0} hierarchical attention at the full Cityscapes size (2.0x pass 2048 x 4096);
    keys of the reference's output dict."""
    tb, out = _teacher_eval("HRNet_Mscale", 19, [0.5, 1.0, 2.0], 1024, 2048)
    for k in ("pred_0.5x", "pred_1.0x", "pred_2.0x", "attn_0.5x", "attn_1.0x"):
        assert k in out, sorted(out)


def test_eval_mapillary_65_classes_three_scales():
    """BASELINE configs[4]: 65 classes, {0.5, 1.0, 2.0} on a Mapillary-shaped (4:3) image.  1152 x 1536, the 2.0x pass
    2304 x 3072 = 7 Mpixel: the device teacher's fp32 convs (MIOpen, compiled at first use) take 100 s here and 140 s at
    1536 x 2048 (round 4, call H); the shapes that matter -- 65-wide heads and OCR regions (three region blocks of the
    fused attention kernel), n-block tails, 0.46 G-element logit tensors -- are the same.  The reference's full recipe
    (eval_mapillary.yml: pre_size 2177, four scales, the 2.0x pass 14 Mpixel) runs once per round through
    `tools/eval_bench.py 3 mapillary-ref` (profiles/r05_eval_bench.json: time, peak memory, finite outputs -- no teacher
    at that size)."""
    tb, out = _teacher_eval("HRNet_Mscale", 65, [0.5, 1.0, 2.0], 1152, 1536)
    for k in ("pred_0.5x", "pred_2.0x", "attn_1.0x"):
        assert k in out, sorted(out)


def test_eval_mapillary_65_classes_four_scales():
    """BASELINE configs[4] as the reference's recipe chains it (scripts/eval_mapillary.yml:13-18:
    n_scales 0.25,0.5,1.0,2.0): `nscale_forward` (network/ocrnet.py:185-262) runs 2.0 -> 1.0 -> 0.5 -> 0.25 and fuses
    TWICE through the `s < 1.0` branch (attention-weighted logits resampled UP to the running prediction).  896 x 1152:
    the 0.25x pass is 224 x 288 (its trunk ends at 7 x 9 pixels), the 2.0x pass 1792 x 2304; 65 classes.  The oracle's
    four-scale chain is pinned to the real reference by tests/golden/nscale4_golden.pt
    (tests/test_oracle_golden.py::test_mscale_eval_four_scales_65_classes_matches_reference); this test runs on both
    storage builds (tests/test_fp16_storage_gpu.py runs it on the fp16 build, the format BASELINE names)."""
    tb, out = _teacher_eval("HRNet_Mscale", 65, [0.25, 0.5, 1.0, 2.0], 896, 1152)
    for k in ("pred_0.25x", "pred_0.5x", "pred_1.0x", "pred_2.0x", "attn_0.25x", "attn_0.5x", "attn_1.0x"):
        assert k in out, sorted(out)
    assert "attn_2.0x" not in out
