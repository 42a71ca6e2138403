"""Golden fixture for the DeepLabV3+/ResNet-50 plumbing config (BASELINE.json
configs[0]; network/deepv3.py:73-93, network/utils.py:48-99,162-218) generated
from the REAL reference.  Run in the build container:
    python tests/golden/make_golden_deepv3.py
Writes deepv3_golden.pt (train loss, sampled parameter gradients, BN running-stat
samples, sub-sampled eval logits) and keys_deepv3.txt (state_dict keys+shapes)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_bootstrap import bootstrap  # noqa: E402
from make_golden import synth_batch, sample_idx  # noqa: E402
from oracle.model import seeded_state_dict  # noqa: E402


def main():
    bootstrap(19)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import network.Resnet as Resnet
    Resnet.resnet50.__defaults__ = (False,)          # no checkpoint download
    import network.deepv3 as deepv3
    from loss.utils import CrossEntropyLoss2d
    net = deepv3.DeepV3PlusR50(19, CrossEntropyLoss2d(ignore_index=255))
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    with open(os.path.join(HERE, "keys_deepv3.txt"), "w") as f:
        for k, s in shapes:
            f.write("%s %s\n" % (k, ",".join(map(str, s))))
    sd = seeded_state_dict(shapes, seed=3)
    net.load_state_dict(sd)
    images, gts = synth_batch(2, 96, 128, seed=4321)
    gold = {"images": images, "gts": gts, "seed": 3}
    net.train()
    loss = net({"images": images, "gts": gts})
    loss.backward()
    gold["train_loss"] = loss.detach()
    grads = {}
    for name, p in net.named_parameters():
        flat = p.grad.flatten()
        idx = sample_idx(flat.numel())
        grads[name] = (idx, flat[idx].clone(), flat.norm().clone())
    gold["grads"] = grads
    gold["running_sample"] = {k: v.flatten()[:4].clone() for k, v in net.state_dict().items()
                              if k.endswith("running_mean") or k.endswith("running_var")}
    # eval on BN statistics calibrated on this batch (momentum 1.0), as make_golden.py does
    net.load_state_dict(sd)
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
        gold["eval_pred"] = net({"images": images})["pred"][:, :, ::8, ::8].clone()
    torch.save(gold, os.path.join(HERE, "deepv3_golden.pt"))
    print("train_loss", float(gold["train_loss"]), "keys", len(shapes), "params",
          sum(p.numel() for p in net.parameters()))


if __name__ == "__main__":
    main()
