"""The peer-to-peer SyncBN exchange's protocol on the CPU: two processes, the host mirror of csrc/p2p.hip
(tests/p2p_emu.py) over shared memory, a gloo process group for the bootstrap only -- as semseg_amd/p2p.py uses
torch.distributed for the handle exchange only.  Sixty collectives of changing sizes with one rank reading slowly every
few calls (its peer then runs one collective ahead and writes the other parity while the slow rank still sums): every
rank must get the rank-ordered sum, bit for bit the same on both."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from p2p_emu import HostP2P
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = None
    try:
        names = ["ssa_p2p_emu_%d_%d" % (port, r) for r in range(world)]
        x = HostP2P(rank, world, 4096, names, create=True)
        dist.barrier()
        x.attach()
        dist.barrier()
        rng = np.random.default_rng(7)
        sizes = rng.integers(1, 4097, size=60)
        digests = []
        for k, n in enumerate(sizes):
            mine = np.arange(n, dtype=np.float64) * (rank + 1) + k
            want = sum(np.arange(n, dtype=np.float64) * (r + 1) + k for r in range(world))
            got = x.all_reduce_sum_(mine.copy(), slow_reader=0.02 if (k % 5 == rank) else 0.0)
            assert np.array_equal(got, want), (rank, k, n)
            digests.append(float(got.sum()))
        q.put((rank, digests))
        dist.barrier()
    finally:
        if x is not None:
            x.close(unlink=True)
        dist.destroy_process_group()


def test_two_ranks_sixty_collectives_with_a_slow_reader():
    world, port = 2, 29000 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out[0] == out[1]
