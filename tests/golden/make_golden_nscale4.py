"""Golden vectors for the FOUR-scale hierarchical evaluation of BASELINE configs[4] (Mapillary), from the REAL reference.

scripts/eval_mapillary.yml:13-18 evaluates with `n_scales: "0.25,0.5,1.0,2.0"` on 65 classes: network/ocrnet.py:185-262
`nscale_forward` then runs high -> low (2.0, 1.0, 0.5, 0.25) and performs TWO consecutive `s < 1.0` fusions, the branch the
three-scale fixtures of make_golden.py exercise once.  Run in the build container (where /root/reference exists):
    python tests/golden/make_golden_nscale4.py
Output (small, committed): nscale4_golden.pt
    shapes         the 65-class state_dict inventory (key, shape) -- seeded_state_dict(shapes, seed=3) rebuilds the weights
    images         1 x 3 x 256 x 320 (passes of 512 x 640, 256 x 320, 128 x 160, 64 x 80: the 0.25x trunk ends at 2 x 3 pixels)
    calib_buffers  BatchNorm running statistics (batch statistics of one training-mode pass, momentum 1.0)
    eval_nscale4   every key of the reference's output dict, sampled (see `sample`)
    eval_nscale3   the same weights at {0.5, 1.0, 2.0} (the three-scale chain at 65 classes)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_bootstrap import bootstrap  # noqa: E402
from oracle.model import seeded_state_dict  # noqa: E402
from make_golden import synth_batch  # noqa: E402

NUM_CLASSES = 65
SCALES4 = [0.25, 0.5, 1.0, 2.0]
SCALES3 = [0.5, 1.0, 2.0]


def sample(v):
    """65-channel outputs every 16th pixel, the 1-channel attention maps every 8th (the fixture stays ~2 MB)"""
    st = 16 if v.shape[1] > 1 else 8
    return v[:, :, ::st, ::st].clone()


def main():
    cfg = bootstrap(NUM_CLASSES)
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from loss.utils import CrossEntropyLoss2d
    import network.ocrnet as ocrnet

    net = ocrnet.HRNet_Mscale(NUM_CLASSES, CrossEntropyLoss2d(ignore_index=255))
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = seeded_state_dict(shapes, seed=3)
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    images, gts = synth_batch(1, 256, 320, C=NUM_CLASSES, seed=4321)
    gold = {"shapes": shapes, "images": images, "seed": 3, "num_classes": NUM_CLASSES,
            "scales4": SCALES4, "scales3": SCALES3}
    # BatchNorm buffers := batch statistics of one training-mode pass (make_golden.py's recipe)
    net.train()
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    with torch.no_grad():
        net({"images": images, "gts": gts})
    for m in bns:
        m.momentum = 0.1
    gold["calib_buffers"] = {k: v.clone() for k, v in net.state_dict().items()
                             if k.endswith("running_mean") or k.endswith("running_var")}
    net.eval()
    with torch.no_grad():
        cfg.MODEL.N_SCALES = SCALES4
        o = net({"images": images})
        gold["eval_nscale4"] = {k: sample(v) for k, v in o.items()}
        cfg.MODEL.N_SCALES = SCALES3
        o = net({"images": images})
        gold["eval_nscale3"] = {k: sample(v) for k, v in o.items()}
        cfg.MODEL.N_SCALES = None
    torch.save(gold, os.path.join(HERE, "nscale4_golden.pt"))
    print("keys", sorted(gold["eval_nscale4"]), "pred max", float(gold["eval_nscale4"]["pred"].abs().max()))


if __name__ == "__main__":
    main()
