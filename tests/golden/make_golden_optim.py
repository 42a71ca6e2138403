"""Golden LR sequences and SGD trajectories from the REAL reference's
loss/optimizer.py:get_optimizer (SGD + LambdaLR).  Run in the build container:
    python tests/golden/make_golden_optim.py
Writes optim_golden.json."""
import argparse
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_bootstrap import bootstrap  # noqa: E402

CASES = (
    dict(lr_schedule="poly", lr=0.01, max_epoch=12, poly_exp=2.0, poly_step=110, rescale=1.0, repoly=1.5, rbe=-1),
    dict(lr_schedule="poly", lr=0.005, max_epoch=175, poly_exp=1.0, poly_step=110, rescale=1.0, repoly=1.5, rbe=-1),
    dict(lr_schedule="poly2", lr=0.02, max_epoch=20, poly_exp=1.0, poly_step=8, rescale=1.0, repoly=1.5, rbe=-1),
    dict(lr_schedule="scl-poly", lr=0.01, max_epoch=30, poly_exp=1.0, poly_step=110, rescale=0.5, repoly=1.5, rbe=10),
)


def main():
    cfg = bootstrap(19)
    from loss.optimizer import get_optimizer
    out = []
    for c in CASES:
        cfg.REDUCE_BORDER_EPOCH = c["rbe"]
        args = argparse.Namespace(optimizer="sgd", weight_decay=1e-4, momentum=0.9, amsgrad=False, **c)
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        opt, sch = get_optimizer(args, net)
        lrs, traj = [], []
        g = torch.Generator().manual_seed(9)
        for epoch in range(c["max_epoch"]):
            lrs.append(opt.param_groups[-1]["lr"])
            if epoch < 4:            # one SGD step per epoch on seeded gradients
                for p in net.parameters():
                    p.grad = torch.randn(p.shape, generator=g)
                opt.step()
                traj.append([p.detach().flatten().tolist() for p in net.parameters()])
            sch.step()
        out.append({"case": c, "lrs": lrs, "init": None, "traj": traj})
        torch.manual_seed(3)
        net0 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        out[-1]["init"] = [p.detach().flatten().tolist() for p in net0.parameters()]
    with open(os.path.join(HERE, "optim_golden.json"), "w") as f:
        json.dump(out, f)
    print(len(out), "cases;", [len(o["lrs"]) for o in out])


if __name__ == "__main__":
    main()
