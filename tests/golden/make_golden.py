"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container (where /root/reference exists):
    python tests/golden/make_golden.py
Outputs (small, committed):
    rmi_golden.pt        RMILoss values + logits-gradient samples (loss/rmi.py)
    ce_golden.pt         CrossEntropyLoss2d values (loss/utils.py)
    mscale_golden.pt     HRNet_Mscale (network/ocrnet.py) train loss, sampled
                         gradients per parameter, eval outputs, on seeded
                         weights (oracle.model.seeded_state_dict) and inputs.
    keys.txt             the reference's 1,903 state_dict keys + shapes
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


def synth_batch(B, H, W, C=19, seed=1234):
    """Synthetic Cityscapes-shaped batch: N(0,1) image,
    block-constant labels with ~10% ignore."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g)
    bs = 16
    blocks = torch.randint(0, C, (B, (H + bs - 1) // bs, (W + bs - 1) // bs), generator=g)
    gts = blocks.repeat_interleave(bs, 1).repeat_interleave(bs, 2)[:, :H, :W].clone()
    gts[torch.rand(B, H, W, generator=g) < 0.1] = 255
    return images, gts.long()


def sample_idx(n, k=16, seed=0):
    g = torch.Generator().manual_seed(seed + n)
    return torch.randint(0, n, (min(k, n),), generator=g)


def main():
    cfg = bootstrap(19)
    cfg.LOSS.SUPERVISED_MSCALE_WT = 0.05
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from loss.rmi import RMILoss
    from loss.utils import CrossEntropyLoss2d
    import network.ocrnet as ocrnet

    # ---------------- losses
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(2, 19, 32, 48, generator=g) * 2
    _, gts = synth_batch(2, 32, 48, seed=11)
    out = {"logits": logits, "gts": gts}
    crit = RMILoss(num_classes=19, ignore_index=255)
    for do_rmi in (False, True):
        lg = logits.clone().requires_grad_(True)
        loss = crit(lg, gts, do_rmi=do_rmi)
        loss.backward()
        out["loss_rmi%d" % do_rmi] = loss.detach()
        out["grad_rmi%d" % do_rmi] = lg.grad.clone()
    torch.save(out, os.path.join(HERE, "rmi_golden.pt"))
    ce = CrossEntropyLoss2d(ignore_index=255)
    lg = logits.clone().requires_grad_(True)
    l = ce(lg, gts)
    l.backward()
    torch.save({"logits": logits, "gts": gts, "loss": l.detach(), "grad": lg.grad.clone()},
               os.path.join(HERE, "ce_golden.pt"))

    # ---------------- full network
    net = ocrnet.HRNet_Mscale(19, RMILoss(num_classes=19, ignore_index=255))
    shapes = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    with open(os.path.join(HERE, "keys.txt"), "w") as f:
        for k, s in shapes:
            f.write("%s %s\n" % (k, ",".join(map(str, s))))
    sd = seeded_state_dict(shapes, seed=0)
    net.load_state_dict(sd)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0   # parity hygiene
    images, gts = synth_batch(1, 128, 128, seed=1234)
    gold = {"images": images, "gts": gts, "seed": 0}
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
    # running stats after one training step (two BN passes: 0.5x then 1.0x)
    rs = {}
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            rs[k] = v.flatten()[:4].clone()
    gold["running_sample"] = rs
    # eval: calibrate the BN running stats on this input first (momentum 1.0:
    # running = batch statistics of the last (1.0x) pass), otherwise eval-mode
    # activations of a random-weight residual net overflow (~1e10 logits).
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
        o = net({"images": images, "gts": gts})
        gold["eval"] = {k: v[:, :, ::8, ::8].clone() for k, v in o.items()}
        cfg.immutable(False) if hasattr(cfg, "immutable") else None
        cfg.MODEL.N_SCALES = [0.5, 1.0, 2.0]
        o = net({"images": images, "gts": gts})
        gold["eval_nscale"] = {k: v[:, :, ::8, ::8].clone() for k, v in o.items()}
        cfg.MODEL.N_SCALES = None
    torch.save(gold, os.path.join(HERE, "mscale_golden.pt"))
    print("train_loss", float(gold["train_loss"]), "keys", len(shapes))


if __name__ == "__main__":
    main()
