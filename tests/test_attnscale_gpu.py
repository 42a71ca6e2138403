"""network/attnscale.py on the HIP kernels: the training step of the three factories, teacher-forced op
by op (tests/teacher_backend.py) -- includes the ops only these heads use: the padding=1 1x1 conv, the
conv -> ReLU without BatchNorm, the fp32 elementwise add / mul / div of the attention normalisation."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,scales,training", [("attnscale.DeepV3R50", [0.5, 1.0, 2.0], True),
                                                  ("attnscale.DeepV3R50B", [0.5, 1.0], True),
                                                  ("attnscale.DeepV3R50BP", [0.5, 1.0], True),
                                                  ("attnscale.DeepV3R50BP", [0.5, 1.0, 2.0], False)])
def test_attnscale_teacher_forced(name, scales, training):
    from semseg_amd import ops, hip_backend as hb
    from semseg_amd.config import cfg
    from teacher_backend import TeacherBackend
    from test_attnscale_cpu import build
    from test_e2e_gpu import _synth
    gold = {"scales": scales, "wt": 0.05 if name == "attnscale.DeepV3R50" else 0, "seed": 3}
    cpu_net = build(name, gold, training).float()
    sd = cpu_net.state_dict()
    for k in sd:                               # as tests/test_deepv3_gpu.py: tame the B=2 image-pooling BN
        if k.endswith("aspp.img_conv.1.weight"):
            sd[k].mul_(0.05)
        if k.startswith("scale_attn") and k.endswith("6.weight"):
            sd[k].mul_(0.05)                   # keep the sigmoid attention away from exact 0 (0/0 in the normalisation)
    images, gts = _synth(2, 128, 192, seed=17)
    if not training:
        # eval: BN running statistics calibrated on this batch first, POOLED over the three scale passes (the statistics
        # of one pass put the others' activations at several hundred by layer4: tests/test_parity_eval_gpu.py)
        from test_parity_eval_gpu import calibrate_eval_bn
        calibrate_eval_bn(cpu_net, images, "cpu")
    hip_net = copy.deepcopy(cpu_net).cuda().train(training)
    tb = TeacherBackend(cpu_net, hip_net)
    prev = ops._BACKEND
    ops._set_backend_for_tests(tb)
    hb.clear_pack_cache()
    try:
        if training:
            out = cpu_net({"images": images, "gts": gts})
            loss = out["pred"] if isinstance(out, dict) else out
            loss.backward()
        else:
            with torch.no_grad():
                pred, attn = cpu_net({"images": images})["pred"]
            assert tuple(pred.shape) == (2, 19, 128, 192)
        torch.cuda.synchronize()
    finally:
        ops._set_backend_for_tests(prev)
        cfg.MODEL.N_SCALES = None
        cfg.LOSS.SUPERVISED_MSCALE_WT = 0
    print(tb.rec.summary(6))
    assert tb.rec.n_ops > 60
    assert any(r[1] == "ewise" for r in tb.rec.rows)
    assert not tb.rec.failures(), tb.rec.summary(30)
