"""Optimizer + LR schedule: the oracle (oracle/optim.py)
against the REAL reference's loss/optimizer.py:get_optimizer trajectories
(tests/golden/optim_golden.json) and against torch.optim.SGD; the product's
get_optimizer LR sequences and state_dict layout (no GPU needed for those)."""
import argparse
import json
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gold():
    with open(os.path.join(G, "optim_golden.json")) as f:
        return json.load(f)


def _args(c):
    c = dict(c)
    c.pop("rbe")
    return argparse.Namespace(optimizer="sgd", weight_decay=1e-4, momentum=0.9, amsgrad=False, **c)


def test_oracle_lr_and_sgd_match_the_reference():
    from oracle.optim import lr_multiplier, sgd_step
    for case in _gold():
        c = case["case"]
        for epoch, lr in enumerate(case["lrs"]):
            mult = lr_multiplier(c["lr_schedule"], epoch, c["max_epoch"], c["poly_exp"], c["poly_step"],
                                 c["rescale"], c["repoly"], c["rbe"])
            assert abs(c["lr"] * mult - lr) <= 1e-15 + 1e-12 * abs(lr), (c, epoch)
        params = [np.array(p, dtype=np.float32) for p in case["init"]]
        bufs = [None] * len(params)
        g = torch.Generator().manual_seed(9)
        shapes = [(5, 7), (5,), (3, 5), (3,)]
        for step, want in enumerate(case["traj"]):
            grads = [torch.randn(s, generator=g).numpy().reshape(-1) for s in shapes]
            bufs = sgd_step(params, grads, bufs, case["lrs"][step], 0.9, 1e-4)
            for p, w in zip(params, want):
                np.testing.assert_allclose(p, np.array(w, dtype=np.float32), rtol=2e-6, atol=1e-7)


def test_oracle_sgd_matches_torch_sgd():
    from oracle.optim import sgd_step
    g = torch.Generator().manual_seed(1)
    for momentum, wd, nesterov in ((0.9, 1e-4, False), (0.0, 1e-4, False), (0.9, 0.0, True), (0.5, 1e-2, True)):
        ps = [torch.randn(n, generator=g).requires_grad_(True) for n in (1, 7, 4096, 5000)]
        opt = torch.optim.SGD(ps, lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
        mine = [p.detach().numpy().copy() for p in ps]
        bufs = [None] * len(ps)
        for _ in range(4):
            grads = [torch.randn(p.shape, generator=g) for p in ps]
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
            opt.step()
            bufs = sgd_step(mine, [gr.numpy() for gr in grads], bufs, 0.05, momentum, wd, nesterov)
            for p, m in zip(ps, mine):
                np.testing.assert_allclose(m, p.detach().numpy(), rtol=2e-6, atol=1e-7)


def test_product_get_optimizer_schedules_and_state_layout():
    from semseg_amd.config import cfg
    from semseg_amd.loss.optimizer import FusedSGD, get_optimizer
    for case in _gold():
        c = case["case"]
        cfg.REDUCE_BORDER_EPOCH = c["rbe"]
        try:
            net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
            opt, sch = get_optimizer(_args(c), net)
            assert isinstance(opt, FusedSGD)
            for epoch, lr in enumerate(case["lrs"]):
                assert abs(opt.param_groups[-1]["lr"] - lr) <= 1e-15 + 1e-12 * abs(lr), (c, epoch)
                opt._step_count = 1          # silence LambdaLR's "step order" warning: no GPU step here
                sch.step()
        finally:
            cfg.REDUCE_BORDER_EPOCH = -1
    # torch.optim.SGD checkpoints restore into FusedSGD and back (same state_dict layout)
    net = torch.nn.Linear(4, 3)
    ref = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    ref.step()
    mine = FusedSGD(net.parameters(), lr=0.3, momentum=0.9, weight_decay=1e-4)
    mine.load_state_dict(ref.state_dict())
    assert mine.param_groups[0]["lr"] == 0.1
    bufs = [mine.state[p]["momentum_buffer"] for p in net.parameters()]
    assert all(torch.equal(b, ref.state[p]["momentum_buffer"]) for b, p in zip(bufs, net.parameters()))
    back = torch.optim.SGD(net.parameters(), lr=0.5, momentum=0.9)
    back.load_state_dict(mine.state_dict())
    assert back.param_groups[0]["lr"] == 0.1 and back.param_groups[0]["weight_decay"] == 1e-4
    # arguments the accelerated path refuses, loudly
    for bad in (dict(dampening=0.1), dict(nesterov=True, momentum=0.0)):
        try:
            FusedSGD(net.parameters(), lr=0.1, **bad)
        except ValueError:
            continue
        raise AssertionError(bad)
    # CPU parameters: no silent fallback
    try:
        mine.step()
    except RuntimeError:
        pass
    else:
        raise AssertionError("FusedSGD.step() on CPU tensors must raise")
