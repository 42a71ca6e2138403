"""Golden fixtures for the sibling architectures of SURVEY.md 8f rank 4, generated
from the REAL reference (network/mscale.py, network/mscale2.py,
network/ocrnet.py:125-155).  Run in the build container:
    python tests/golden/make_golden_siblings.py
Writes siblings_golden.pt: per configuration the seeded inputs, train loss, 16
sampled gradient entries + the norm per parameter, BN running-stat samples and
sub-sampled eval outputs (two-scale and, where the reference supports it,
N-scale); keys_siblings.txt: state_dict keys + shapes per architecture.
All configurations run the reference in fp64, so that the wiring is pinned far
below the rounding-noise amplification of these small cases (in fp32 the
reference's own gradients move by ~20% on the stride-32 branch of the 0.5x pass:
1x2 pixels, BatchNorm over 4 samples).  loss/rmi.py computes parts of the RMI
criterion in fp32 whatever the input type; that configuration is pinned to 1e-5."""
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

# name, module, constructor, criterion, dtype, N-scale eval, SUPERVISED_MSCALE_WT
CONFIGS = (
    ("mscale.HRNet", "mscale", lambda m, c: m.HRNet(19, c), "rmi", torch.float64, True, 0.05),
    ("mscale.HRNet_ASP", "mscale", lambda m, c: m.HRNet_ASP(19, c), "ce", torch.float64, False, 0),
    ("mscale.DeepV3R50", "mscale", lambda m, c: m.DeepV3R50(19, c), "ce", torch.float64, True, 0.05),
    ("mscale.MscaleV3Plus.fuse2b", "mscale",
     lambda m, c: m.MscaleV3Plus(19, trunk="resnet-50", criterion=c, fuse_aspp=True, attn_2b=True), "ce",
     torch.float64, False, 0),
    ("mscale2.DeepV3R50", "mscale2", lambda m, c: m.DeepV3R50(19, c), "ce", torch.float64, True, 0),
    ("ocrnet.OCRNetASPP", "ocrnet", lambda m, c: m.OCRNetASPP(19, criterion=c), "ce", torch.float64, False, 0),
    # BASELINE.json configs[1]: HRNet-OCR single scale
    ("ocrnet.HRNet", "ocrnet", lambda m, c: m.HRNet(19, c), "rmi", torch.float64, False, 0),
)


def main():
    cfg = bootstrap(19)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import importlib
    import network.Resnet as Resnet
    Resnet.resnet50.__defaults__ = (False,)          # no checkpoint download
    from loss.rmi import RMILoss
    from loss.utils import CrossEntropyLoss2d
    gold, keys = {}, []
    for ci, (name, modname, make, crit, dtype, nscale, wt) in enumerate(CONFIGS):
        cfg.LOSS.SUPERVISED_MSCALE_WT = wt
        cfg.MODEL.N_SCALES = None
        mod = importlib.import_module("network." + modname)
        criterion = RMILoss(num_classes=19, ignore_index=255) if crit == "rmi" else CrossEntropyLoss2d(ignore_index=255)
        net = make(mod, criterion)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
        keys.append("# " + name)
        keys += ["%s %s" % (k, ",".join(map(str, s))) for k, s in shapes]
        sd = seeded_state_dict(shapes, seed=10 + ci)
        net.load_state_dict(sd)
        net = net.to(dtype)
        images, gts = synth_batch(2, 64, 96, seed=777 + ci)
        g = {"images": images.clone(), "gts": gts.to(torch.uint8),      # stored fp32 / uint8; the test casts back
             "seed_batch": 777 + ci}
        images = images.to(dtype)
        g.update({ "seed": 10 + ci, "crit": crit, "wt": wt, "dtype": dtype})
        net.train()
        loss = net({"images": images, "gts": gts})
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
        # (the calibrated buffers are not stored: the test repeats this calibration on its side)
        net.eval()
        with torch.no_grad():
            o = net({"images": images, "gts": gts})
            g["eval"] = {k: v[:, :, ::8, ::8].clone() for k, v in o.items()}
            if nscale:
                cfg.MODEL.N_SCALES = [0.5, 1.0, 2.0]
                o = net({"images": images, "gts": gts})
                g["eval_nscale"] = {k: v[:, :, ::8, ::8].clone() for k, v in o.items()}
                cfg.MODEL.N_SCALES = None
        gold[name] = g
        print(name, "train_loss", float(g["train_loss"]), "keys", len(shapes), "eval", sorted(g["eval"]),
              sorted(g.get("eval_nscale", {})))
    torch.save(gold, os.path.join(HERE, "siblings_golden.pt"))
    with open(os.path.join(HERE, "keys_siblings.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")


if __name__ == "__main__":
    main()
