"""The peer-to-peer SyncBN exchange on the device: two processes on cuda:0 (the GPU box has one device), each with its
own fine-grained exchange buffer, the peer's opened through hipIpc (semseg_amd/p2p.py), csrc/p2p.hip's kernel writing
both buffers and polling its own.  Checks: the sum on every rank = the rank-ordered sum of the contributions, identical
on both ranks, over sizes from 1 element to a full slot and 24 consecutive collectives (both parities reused a dozen
times); the same five collectives captured in a hipGraph and replayed three times with new inputs (the sequence number
lives in device memory and counts on); the route through parallel.allreduce_bn_sums with the switch on; no rank timed
out of a wait.  What one GPU cannot show is visibility across xGMI -- that rests on the kernel's system-scope
atomics and the fine-grained allocation."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["SSA_SYNCBN_P2P"] = "1"
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from semseg_amd import p2p, parallel
        x = p2p.exchange()
        assert x is not None, "the exchange could not be set up"
        sizes = [1, 7, 1000, 23040, p2p.SLOT_DOUBLES] * 4 + [333, 4096, 17, 23040]
        sums = []
        for k, n in enumerate(sizes):
            t = (torch.arange(n, dtype=torch.float64) * (rank + 1) + k).cuda()
            x.all_reduce_sum_(t)
            want = sum(torch.arange(n, dtype=torch.float64) * (r + 1) + k for r in range(world))
            torch.cuda.synchronize()
            assert torch.equal(t.cpu(), want), (rank, k, n)
            sums.append(float(t.sum()))
        # captured: five collectives per replay
        bufs = [torch.zeros(n, dtype=torch.float64, device="cuda") for n in (48 * 16, 96 * 16, 23040, 5, 720 * 16)]
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g):
                for b in bufs:
                    x.all_reduce_sum_(b)
        for rep in range(3):
            for i, b in enumerate(bufs):
                b.copy_(torch.arange(b.numel(), dtype=torch.float64) * (rank + 2) + rep + i)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for i, b in enumerate(bufs):
                want = sum(torch.arange(b.numel(), dtype=torch.float64) * (r + 2) + rep + i for r in range(world))
                assert torch.equal(b.cpu(), want), (rank, "replay", rep, i)
        # the product's route: SyncBN sums through parallel.allreduce_bn_sums
        t = (torch.ones(2 * 8 * 48, dtype=torch.float64) * (rank + 1)).cuda()
        calls = x.calls
        parallel.allreduce_bn_sums(t)
        torch.cuda.synchronize()
        assert x.calls == calls + 1, "allreduce_bn_sums did not take the peer-to-peer route"
        assert torch.equal(t.cpu(), torch.ones(2 * 8 * 48, dtype=torch.float64) * sum(r + 1 for r in range(world)))
        assert x.timeouts() == 0
        dist.barrier()
        q.put((rank, "ok", sums))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, "error", "%s\n%s" % (e, traceback.format_exc())))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_processes_one_gpu_exchange():
    world, port = 2, 29500 + os.getpid() % 400
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, status, payload = q.get(timeout=300)
        res[r] = (status, payload)
    for p in procs:
        p.join(60)
    for r in range(world):
        assert res[r][0] == "ok", "rank %d: %s" % (r, res[r][1])
    assert res[0][1] == res[1][1]
