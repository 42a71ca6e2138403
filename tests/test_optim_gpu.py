"""ssa_sgd_momentum_step through semseg_amd.loss.optimizer.FusedSGD against the
oracle (oracle/optim.py, pinned to the reference's optimizer trajectories in
tests/test_optim_cpu.py) and against torch.optim.SGD on the device."""
import os

import numpy as np
import pytest
import torch

from util import ACT_DTYPE

pytestmark = pytest.mark.gpu

# Run on an MI355X in round 1: the first parametrisation (profiles/r01_optim_gpu_test.log: parameters

SIZES = (1, 3, 7, 19, 4095, 4096, 4097, 720 * 512 * 9, 100003) + tuple(range(5, 5 + 120))   # > 96 tensors: 3 launches


@pytest.mark.parametrize("momentum,wd,nesterov", [(0.9, 1e-4, False),
                                                  (0.0, 1e-4, False),
                                                  (0.9, 0.0, True)])
def test_fused_sgd_matches_oracle_and_torch(momentum, wd, nesterov):
    from semseg_amd.loss.optimizer import FusedSGD
    from oracle.optim import sgd_step
    g = torch.Generator().manual_seed(11)
    init = [torch.randn(n, generator=g) for n in SIZES]
    mine = [t.clone().cuda().requires_grad_(True) for t in init]
    ref = [t.clone().cuda().requires_grad_(True) for t in init]
    # one unaligned view (scalar path) and one parameter that never gets a gradient
    base = torch.randn(1001, generator=g).cuda()
    mine.append(base.clone()[1:].requires_grad_(True))
    ref.append(base.clone()[1:].requires_grad_(True))
    idle_m, idle_r = torch.ones(8, device="cuda", requires_grad=True), torch.ones(8, device="cuda", requires_grad=True)
    opt_m = FusedSGD(mine + [idle_m], lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    opt_r = torch.optim.SGD(ref + [idle_r], lr=0.05, momentum=momentum, weight_decay=wd, nesterov=nesterov)
    ora = [p.detach().cpu().numpy().copy() for p in mine]
    bufs = [None] * len(ora)
    lrs = (0.05, 0.05, 0.02, 0.01)
    for step, lr in enumerate(lrs):
        grads = [torch.randn(p.shape, generator=g) for p in mine]
        for p, q, gr in zip(mine, ref, grads):
            p.grad = gr.cuda()
            q.grad = gr.cuda()
        for o in (opt_m, opt_r):
            o.param_groups[0]["lr"] = lr
        opt_m.step()
        opt_r.step()
        bufs = sgd_step(ora, [gr.numpy() for gr in grads], bufs, lr, momentum, wd, nesterov)
        torch.cuda.synchronize()
        worst_t = worst_o = 0.0
        for p, q, o in zip(mine, ref, ora):
            a = p.detach().cpu().numpy()
            np.testing.assert_allclose(a, o, rtol=2e-6, atol=2e-7)
            np.testing.assert_allclose(a, q.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
            worst_o = max(worst_o, float(np.abs(a - o).max()))
            worst_t = max(worst_t, float(np.abs(a - q.detach().cpu().numpy()).max()))
        print("step %d lr %.3f: max |fused - oracle| %.3g, max |fused - torch| %.3g" % (step, lr, worst_o, worst_t))
    assert torch.equal(idle_m, torch.ones(8, device="cuda"))
    if momentum:
        for p, q in zip(mine, ref):
            np.testing.assert_allclose(opt_m.state[p]["momentum_buffer"].cpu().numpy(),
                                       opt_r.state[q]["momentum_buffer"].cpu().numpy(), rtol=2e-6, atol=2e-6)


def test_fused_sgd_step_refreshes_packed_filters():
    """A conv after FusedSGD.step() must see the updated weights: the bf16 operand cache of the HIP
    backend is keyed on the parameter's version counter, which the raw-pointer update has to bump."""
    from semseg_amd import ops
    from semseg_amd.loss.optimizer import FusedSGD
    from semseg_amd.nn import Conv2d
    torch.manual_seed(0)
    conv = Conv2d(16, 16, kernel_size=3, padding=1, bias=False).cuda()
    x = torch.randn(1, 8, 8, 16, device="cuda").to(ACT_DTYPE)
    B = ops.HipBackend()
    B.begin_step(x.device)
    y0 = B.conv2d(x, conv.weight, None, 1, 1, 1).float()
    opt = FusedSGD(conv.parameters(), lr=1.0, momentum=0.0)
    conv.weight.grad = conv.weight.detach().clone()          # p <- p - 1.0 * p = 0
    opt.step()
    B.begin_step(x.device)
    y1 = B.conv2d(x, conv.weight, None, 1, 1, 1).float()
    assert float(y0.abs().max()) > 0.1 and float(y1.abs().max()) == 0.0


def test_fused_sgd_lr_from_device_scalar():
    """sync_lr() is what a captured step relies on: the kernel must read the device scalar."""
    from semseg_amd.loss.optimizer import FusedSGD
    p = torch.zeros(1000, device="cuda", requires_grad=True)
    opt = FusedSGD([p], lr=1.0, momentum=0.0)
    p.grad = torch.ones_like(p)
    v0 = p._version
    opt.step()
    assert p._version > v0        # refresh_packed_filters and autograd rely on the version counter
    assert torch.equal(p.detach(), torch.full_like(p, -1.0))
    opt.param_groups[0]["lr"] = 0.25
    opt.sync_lr()
    assert float(opt._lr_dev[0][0]) == 0.25
    opt.step()
    assert torch.equal(p.detach(), torch.full_like(p, -1.25))
