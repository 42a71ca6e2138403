"""The peer-to-peer SyncBN exchange on the device.

test_ranks_as_streams_of_one_process: two "ranks" as concurrent streams of ONE process, every rank with its
own uncached exchange buffer (ssa_p2p_vmm_alloc) and its own device-side sequence number -- csrc/p2p.hip's kernel, flags
and two-parity protocol with real concurrency between single-workgroup kernels, independent of what the container
allows between processes.

test_two_processes_one_gpu_exchange: two processes on cuda:0 (the GPU box has one device), each with its
own exchange buffer, the peer's mapped from a file descriptor sent over a unix socket (or opened through hipIpc where
the container permits pidfd_getfd; semseg_amd/p2p.py), csrc/p2p.hip's kernel writing both buffers and polling its own.  Checks: the sum on every rank = the rank-ordered sum of the contributions, identical
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
        if x is None:
            q.put((rank, "unavailable", ""))
            return
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
        q.put((rank, "ok", (x.route, sums)))
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
    if all(res[r][0] == "unavailable" for r in range(world)):
        pytest.skip("neither a file-descriptor mapping nor hipIpc of device memory is permitted between processes here")
    for r in range(world):
        assert res[r][0] == "ok", "rank %d: %s" % (r, res[r][1])
    assert res[0][1] == res[1][1]
    print("route:", res[0][1][0])


def test_ranks_as_streams_of_one_process(world=2):
    # two ranks only: with more, two of the streams may share a hardware queue (GPU_MAX_HW_QUEUES = 4 with the default
    # stream in it) and a kernel then waits for a peer queued BEHIND it -- measured with 4: 67 bounded waits ran out
    import ctypes
    for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from semseg_amd._lib import lib, check
    L = lib()
    slot = 4096
    nb = ctypes.c_size_t(0)
    check(L.ssa_p2p_buffer_bytes(world, slot, ctypes.byref(nb)), "ssa_p2p_buffer_bytes")
    bufs = []
    for _ in range(world):
        p, fd, mapped = ctypes.c_void_p(), ctypes.c_int(-1), ctypes.c_size_t(0)
        check(L.ssa_p2p_vmm_alloc(nb.value, ctypes.byref(p), ctypes.byref(fd), ctypes.byref(mapped)), "ssa_p2p_vmm_alloc")
        os.close(fd.value)
        assert mapped.value >= nb.value and p.value
        bufs.append((p, mapped.value))
    try:
        peers = torch.tensor([p.value for p, _ in bufs], dtype=torch.int64, device="cuda")
        seqs = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in range(world)]
        t0 = ctypes.c_uint(0)

        def timeouts():
            L.ssa_p2p_timeouts(ctypes.byref(t0))
            return t0.value

        # The two "ranks" must really run side by side: streams that the runtime put on ONE hardware queue serialise,
        # the first kernel then waits (bounded: ~2 s) for a peer queued behind it.  Probe with one collective and take
        # other streams of torch's pool if that happened; a box that never runs two of them concurrently cannot run
        # this form of the test (the two-process test above does not depend on it).
        streams, probes = None, 0
        for _attempt in range(4):
            cand = [torch.cuda.Stream() for _ in range(world)]
            before = timeouts()
            probe = [torch.full((8,), float(r + 1), dtype=torch.float64, device="cuda") for r in range(world)]
            torch.cuda.synchronize()
            for r in range(world):
                with torch.cuda.stream(cand[r]):
                    check(L.ssa_p2p_allreduce_f64(ctypes.c_void_p(probe[r].data_ptr()), 8, ctypes.c_void_p(peers.data_ptr()),
                                                  r, world, ctypes.c_void_p(seqs[r].data_ptr()), slot,
                                                  ctypes.c_void_p(cand[r].cuda_stream)), "ssa_p2p_allreduce_f64")
            torch.cuda.synchronize()
            probes += 1
            if timeouts() == before:
                assert all(torch.equal(t.cpu(), torch.full((8,), float(sum(range(1, world + 1))), dtype=torch.float64)) for t in probe)
                streams = cand
                break
        if streams is None:
            pytest.skip("no two streams of this process ran concurrently in %d attempts" % probes)
        timeouts()
        torch.cuda.synchronize()
        sizes = [1, 7, 1000, slot, 333] * 5                     # 25 collectives: both parities a dozen times
        data = [[(torch.arange(n, dtype=torch.float64) * (r + 1) + k).cuda() for k, n in enumerate(sizes)] for r in range(world)]
        torch.cuda.synchronize()
        for k, n in enumerate(sizes):
            for r in range(world):
                with torch.cuda.stream(streams[r]):
                    check(L.ssa_p2p_allreduce_f64(ctypes.c_void_p(data[r][k].data_ptr()), n, ctypes.c_void_p(peers.data_ptr()),
                                                  r, world, ctypes.c_void_p(seqs[r].data_ptr()), slot,
                                                  ctypes.c_void_p(streams[r].cuda_stream)), "ssa_p2p_allreduce_f64")
        torch.cuda.synchronize()
        before = t0.value
        assert timeouts() == before, "a rank gave up waiting for a peer's sequence number"
        for k, n in enumerate(sizes):
            want = sum(torch.arange(n, dtype=torch.float64) * (r + 1) + k for r in range(world))
            for r in range(world):
                assert torch.equal(data[r][k].cpu(), want), (r, k, n)
        assert all(int(s.item()) == len(sizes) + probes for s in seqs)
        # a message larger than a slot is refused, not truncated
        big = torch.zeros(slot + 1, dtype=torch.float64, device="cuda")
        assert L.ssa_p2p_allreduce_f64(ctypes.c_void_p(big.data_ptr()), slot + 1, ctypes.c_void_p(peers.data_ptr()), 0, world,
                                       ctypes.c_void_p(seqs[0].data_ptr()), slot, None) == -2
    finally:
        torch.cuda.synchronize()
        for p, n in bufs:
            L.ssa_p2p_vmm_unmap(p, n)
