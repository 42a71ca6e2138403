"""N>1 path on CPU: two gloo ranks.
 * DistributedDataParallel (semseg_amd.parallel): bucketed, hook-driven gradient
   all-reduce == gradients of the mean loss over the concatenated global batch.
 * allreduce_bn_sums: SyncBN statistics == BatchNorm over the concatenated batch.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.ReLU(),
                               torch.nn.Conv2d(8, 4, 1))


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(2, 3, 12, 12, generator=g), torch.randn(2, 4, 12, 12, generator=g)


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semseg_amd.parallel import DistributedDataParallel, allreduce_bn_sums
    net = _model()
    if rank == 1:   # ranks start from different weights; the wrapper must broadcast rank 0's
        for p in net.parameters():
            p.data.add_(1.0)
    ddp = DistributedDataParallel(net)
    for it in range(2):      # two iterations: hooks re-arm
        net.zero_grad(set_to_none=True)
        x, y = _data(rank)
        loss = ((ddp(x) - y) ** 2).mean()
        loss.backward()
    grads = [p.grad.clone() for p in net.parameters()]
    # SyncBN statistics
    x, _ = _data(rank)
    sums = torch.cat([x.double().sum((0, 2, 3)), (x.double() ** 2).sum((0, 2, 3))])
    allreduce_bn_sums(sums)
    count = x.shape[0] * x.shape[2] * x.shape[3] * world
    mean = sums[:3] / count
    var = sums[3:] / count - mean ** 2
    q.put((rank, [g.numpy() for g in grads], mean.numpy(), var.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ddp_and_syncbn_two_ranks():
    world = 2
    port = 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, grads, mean, var = q.get(timeout=100)
        res[r] = (grads, mean, var)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single-process reference on the concatenated batch
    net = _model()
    xs, ys = zip(*[_data(r) for r in range(world)])
    x, y = torch.cat(xs), torch.cat(ys)
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()
    for r in range(world):
        for g, p in zip(res[r][0], net.parameters()):
            assert torch.allclose(torch.from_numpy(g), p.grad, rtol=1e-5, atol=1e-7)
        assert torch.allclose(torch.from_numpy(res[r][1]).float(), x.mean((0, 2, 3)), atol=1e-6)
        assert torch.allclose(torch.from_numpy(res[r][2]).float(), x.var((0, 2, 3), unbiased=False), atol=1e-6)
