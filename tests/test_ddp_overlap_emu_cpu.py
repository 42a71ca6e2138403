"""Overlapped gradient exchange of the data-parallel wrapper (semseg_amd.parallel + hip_backend's gradient arena),
two gloo ranks on CPU, REAL kernel arithmetic through the CPU emulation build of the kernel sources
(tests/emu_util.py): the arena ranges exchanged while backward goes on (weight gradients flushed every few layers,
SSA_DDP_FLUSH_AT) give bit for bit the gradients of ONE exchange at the end of backward, and both equal the mean
of the ranks' local gradients.  (On the device the partial exchanges run on a communication stream over a second
RCCL communicator, concurrently with the remaining backward kernels; here they are synchronous gloo calls.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    from semseg_amd.network.hrnetv2 import BasicBlock
    from semseg_amd.config import cfg
    cfg.MODEL.BNFUNC = None
    torch.manual_seed(3)
    net = torch.nn.Sequential(*[BasicBlock(48, 48) for _ in range(3)])
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 4:
                p.copy_(torch.randn_like(p) * 0.05)
    return net.train()


def _step(net, fwd, rank):
    from semseg_amd import hip_backend as hb
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(1, 8, 32, 48, generator=g).to(torch.bfloat16)
    gy = torch.randn(1, 8, 32, 48, generator=g).to(torch.bfloat16)
    for p in net.parameters():
        p.grad = None
    hb.begin_step(torch.device("cpu"))
    y = fwd(x)
    y.backward(gy)
    return [p.grad.clone() for p in net.parameters()]


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_util import emu_backend
    from semseg_amd import ops, hip_backend as hb
    from semseg_amd.parallel import DistributedDataParallel
    with emu_backend():
        ops._set_backend_for_tests(ops.HipBackend())
        net = _net()
        local = _step(net, net, rank)                       # no wrapper: this rank's own gradients
        ddp = DistributedDataParallel(net)
        assert ddp.active
        hb._DDP_FLUSH_AT = 100000
        single = _step(net, ddp, rank)
        n_single = ddp.exchanges
        hb._DDP_FLUSH_AT = 2                                # a flush + a range exchange every two queued layers
        chunked = _step(net, ddp, rank)
        n_chunked = ddp.exchanges
        hb.set_grad_sink(None)
    q.put((rank, [g.numpy() for g in local], [g.numpy() for g in single], [g.numpy() for g in chunked], n_single, n_chunked))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_chunked_exchange_equals_single_exchange_equals_mean_of_ranks():
    from emu_util import build_emu
    build_emu()                                             # once, before the ranks race for it
    world = 2
    port = 23500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, local, single, chunked, n1, nc = q.get(timeout=500)
        res[r] = (local, single, chunked, n1, nc)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][3] == 1 and res[0][4] >= 3, (res[0][3], res[0][4])      # one exchange vs several ranges
    nparam = len(res[0][0])
    for i in range(nparam):
        mean = (torch.from_numpy(res[0][0][i]) + torch.from_numpy(res[1][0][i])) / 2
        for r in range(world):
            single, chunked = torch.from_numpy(res[r][1][i]), torch.from_numpy(res[r][2][i])
            assert torch.equal(single, chunked), "parameter %d: chunked exchange differs from the single exchange" % i
            assert torch.allclose(single, mean, rtol=1e-6, atol=1e-7), "parameter %d: not the mean over ranks" % i


# ---------------------------------------------------------------- fp16 training under data parallelism
def _worker_amp(rank, world, port, q):
    """Two ranks, the gradient exchange + FusedSGD with a loss scaler on the emulated kernels: an inf in ONE rank's local
    gradients reaches every rank through the exchange, so every rank skips the step and halves its scale."""
    import contextlib
    sys.path[:0] = [ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_util import emu_backend
    from semseg_amd import ops, hip_backend as hb
    from semseg_amd.amp import LossScaler
    from semseg_amd.loss import optimizer as sopt
    from semseg_amd.parallel import DistributedDataParallel
    sopt._on_gpu = lambda p: True
    sopt._launch_scope = lambda device: contextlib.nullcontext((None, False))
    with emu_backend():
        ops._set_backend_for_tests(ops.HipBackend())
        net = _net()
        ddp = DistributedDataParallel(net)
        opt = sopt.FusedSGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        opt.loss_scaler = LossScaler(torch.device("cpu"), init_scale=64.0, growth_interval=1000)
        out = []
        for step in range(3):
            S = opt.loss_scaler.loss_scale()
            g = torch.Generator().manual_seed(70 + 10 * step + rank)
            x = torch.randn(1, 8, 32, 48, generator=g).to(torch.bfloat16)
            gy = (torch.randn(1, 8, 32, 48, generator=g) * S).to(torch.bfloat16)
            if step == 1 and rank == 1:
                gy[0, 3, 5, 7] = float("inf")              # one rank, one element
            for p in net.parameters():
                p.grad = None
            hb.begin_step(torch.device("cpu"))
            ddp(x).backward(gy)
            opt.step()
            out.append(([p.detach().clone().numpy() for p in net.parameters()], opt.loss_scaler.state.tolist()))
        hb.set_grad_sink(None)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_loss_scaler_takes_the_same_decision_on_every_rank():
    from emu_util import build_emu
    build_emu()
    world = 2
    port = 25500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_amp, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, out = q.get(timeout=500)
        res[r] = out
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import numpy as np
    for step in range(3):
        (p0, s0), (p1, s1) = res[0][step], res[1][step]
        assert s0 == s1, (step, s0, s1)                                       # same scale, same counters
        for a, b in zip(p0, p1):
            assert np.array_equal(a, b), "step %d: the ranks' parameters diverged" % step
    # step 0 clean (scale stays), step 1 skipped on BOTH ranks (parameters as after step 0, scale halved), step 2 moves again
    assert res[0][0][1][0] == 64.0 and res[0][1][1][:3] == [32.0, 0.0, 0.0] and res[0][2][1][0] == 32.0
    assert all(np.array_equal(a, b) for a, b in zip(res[0][0][0], res[0][1][0]))
    assert not all(np.array_equal(a, b) for a, b in zip(res[0][1][0], res[0][2][0]))
