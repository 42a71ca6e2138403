"""Teacher-forced parity of the EVALUATION path at the shapes BASELINE.json's eval configurations name
(network/ocrnet.py:185-262 `nscale_forward`, utils/trnval_utils.py:82-198):

  configs[1]  HRNet-OCR, single scale, 1 x 3 x 1024 x 2048 (Cityscapes val)
  configs[2]  HRNet-OCR-MScale, scales {0.5, 1.0, 2.0}: the 2.0x pass of a 1024 x 2048 image is 2048 x 4096
  configs[4]  Mapillary: 65 classes (the shape-dependent pieces are the 65-wide fp32 heads, the OCR gather /
              attention over 65 object regions and the n-block tails of 65 output channels)

Every operator call of the forward pass runs the HIP op on the oracle's (bf16-rounded) inputs at its REAL shape and
must match the oracle's output to one-rounding tolerance (tests/teacher_backend.py) -- eval-mode BatchNorm, the
32-bit-offset guards of the halo kernels, the tile dispatch at 512 x 1024 / 1024 x 2048 grids, none of which the
128 x 192 end-to-end eval test reaches.  The CPU teacher needs a few seconds per TFLOP: configs[2] and [4] run at
HALF the linear size by default (SSA_PARITY_EVAL_FULL=1: the full 1024 x 2048), configs[1] at full size always.
"""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

FULL = os.environ.get("SSA_PARITY_EVAL_FULL", "0") == "1"


def _image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 3, h, w, generator=g)


def _teacher_eval(factory, num_classes, n_scales, h, w):
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
    try:
        from semseg_amd.network import ocrnet
        cpu_net = getattr(ocrnet, factory)(num_classes, None)
        sd = parity_state_dict([(k, tuple(v.shape)) for k, v in cpu_net.state_dict().items()], seed=0)
        cpu_net.load_state_dict(sd)
        cpu_net.eval()
        hip_net = copy.deepcopy(cpu_net).cuda().eval()
        tb = TeacherBackend(cpu_net, hip_net)
        ops._set_backend_for_tests(tb)
        hb.clear_pack_cache()
        hb.profile_begin()
        try:
            with torch.no_grad():
                out = cpu_net({"images": _image(h, w, 11)})
            torch.cuda.synchronize()
        finally:
            kernels = hb.profile_end()
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


def test_eval_hrnet_ocr_single_scale_1024x2048():
    """BASELINE configs[1]."""
    _teacher_eval("HRNet", 19, None, 1024, 2048)


def test_eval_mscale_three_scales():
    """BASELINE configs[2]: {0.5, 1.0, 2.0} hierarchical attention; keys of the reference's output dict."""
    h, w = (1024, 2048) if FULL else (512, 1024)
    tb, out = _teacher_eval("HRNet_Mscale", 19, [0.5, 1.0, 2.0], h, w)
    for k in ("pred_0.5x", "pred_1.0x", "pred_2.0x", "attn_0.5x", "attn_1.0x"):
        assert k in out, sorted(out)


def test_eval_mapillary_65_classes():
    """BASELINE configs[4]'s class count on the single-scale model."""
    h, w = (1024, 2048) if FULL else (512, 1024)
    _teacher_eval("HRNet", 65, None, h, w)
