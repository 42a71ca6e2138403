"""Golden fixtures for the attention-to-scale heads (network/attnscale.py of the REAL reference):
`DeepV3R50` (joint attention, non-BN head with its padding=1 1x1 conv), `DeepV3R50B` (BN head),
`DeepV3R50BP` (paired attention).  Run in the build container:
    python tests/golden/make_golden_attnscale.py
Writes attnscale_golden.pt (seeded inputs, train loss of `_forward_fused` / `_forward_paired`, sampled
gradient entries + norms, BN running-stat samples, sub-sampled eval outputs incl. the 3-scale paired
inference with its attention normalisation) and keys_attnscale.txt.  fp64, like the sibling fixtures."""
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

CONFIGS = (("attnscale.DeepV3R50", "DeepV3R50", [0.5, 1.0, 2.0], 0.05),
           ("attnscale.DeepV3R50B", "DeepV3R50B", [0.5, 1.0], 0),
           ("attnscale.DeepV3R50BP", "DeepV3R50BP", [0.5, 1.0, 2.0], 0))


def main():
    cfg = bootstrap(19)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import network.Resnet as Resnet
    Resnet.resnet50.__defaults__ = (False,)
    import network.attnscale as A
    from loss.utils import CrossEntropyLoss2d
    gold, keys = {}, []
    for ci, (name, factory, scales, wt) in enumerate(CONFIGS):
        cfg.MODEL.N_SCALES = list(scales)
        cfg.LOSS.SUPERVISED_MSCALE_WT = wt
        net = getattr(A, factory)(19, CrossEntropyLoss2d(ignore_index=255))
        shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        keys.append("# " + name)
        keys += ["%s %s" % (k, ",".join(map(str, s))) for k, s in shapes]
        sd = seeded_state_dict(shapes, seed=40 + ci)
        net.load_state_dict(sd)
        net = net.double()
        images, gts = synth_batch(2, 64, 96, seed=900 + ci)
        inputs = {"images": images.double(), "gts": gts}
        g = {"images": images.clone(), "gts": gts.to(torch.uint8), "seed": 40 + ci, "scales": list(scales), "wt": wt}
        fused = hasattr(net, "_forward_fused")

        def run_train(n):
            return n._forward_fused(inputs) if fused else n(inputs)

        def run_eval(n):
            return n._forward_fused(inputs) if fused else n(inputs)["pred"]
        net.train()
        loss = run_train(net)
        loss.backward()
        g["train_loss"] = loss.detach().clone()
        samples, norms = [], []
        for pname, p in net.named_parameters():
            flat = p.grad.flatten()
            samples.append(flat[sample_idx(flat.numel())].clone())
            norms.append(flat.norm().clone())
        g["grad_samples"] = torch.cat(samples)
        g["grad_norms"] = torch.stack(norms)
        g["running_sample"] = torch.cat([v.flatten()[:4] for k, v in net.state_dict().items()
                                         if k.endswith("running_mean") or k.endswith("running_var")])
        net.load_state_dict(sd)
        net.train()
        bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        for m in bns:
            m.momentum = 1.0
        with torch.no_grad():
            run_train(net)
        for m in bns:
            m.momentum = 0.1
        net.eval()
        with torch.no_grad():
            out, attn = run_eval(net)
        g["eval_pred"] = out[:, :, ::8, ::8].clone()
        g["eval_attn_shape"] = tuple(attn.shape)
        g["eval_attn"] = attn[:, :, ::4, ::4].clone()
        gold[name] = g
        print(name, "train_loss", float(g["train_loss"]), "keys", len(shapes), "attn", tuple(attn.shape))
    torch.save(gold, os.path.join(HERE, "attnscale_golden.pt"))
    with open(os.path.join(HERE, "keys_attnscale.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")


if __name__ == "__main__":
    main()
