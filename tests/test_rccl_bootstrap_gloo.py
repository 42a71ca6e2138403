"""semseg_amd.rccl.DirectComm bootstrap on CPU: two gloo ranks and a stand-in for librccl.so
that records the calls.  Checks what can be checked without GPUs: rank 0's 128-byte unique id
(NUL bytes included) reaches every rank intact and is passed BY VALUE to ncclCommInitRank with
the right (nranks, rank); the all-reduce call carries pointer, count, type, op, communicator
and stream in the C order.  (The real library's argument types are set up the same way by
rccl._load(); tests/test_rccl_direct_gpu.py runs it on a GPU.)"""
import ctypes
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UID = bytes([7, 0, 0, 9] + list(range(124)))          # NULs early on: would be cut by a c_char field


class FakeRccl:
    def __init__(self):
        self.calls = []

    def ncclGetUniqueId(self, ref):
        ctypes.memmove(ref, UID, 128)
        return 0

    def ncclCommInitRank(self, comm_ref, nranks, uid, rank):
        self.calls.append(("init", nranks, rank, ctypes.string_at(ctypes.byref(uid), 128)))
        return 0

    def ncclAllReduce(self, send, recv, count, dtype, op, comm, stream):
        self.calls.append(("allreduce", send, recv, count, dtype, op))
        return 0

    def ncclCommDestroy(self, comm):
        self.calls.append(("destroy",))
        return 0

    def ncclGetErrorString(self, rc):
        return b"fake"


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semseg_amd import rccl
    fake = FakeRccl()
    rccl._load = lambda: fake
    torch.cuda.current_device = lambda: 0
    c = rccl.DirectComm()
    q.put((rank, c.rank, c.world, fake.calls[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_unique_id_reaches_every_rank():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, crank, cworld, call in got:
        assert (crank, cworld) == (rank, 2)
        assert call == ("init", 2, rank, UID)


def test_real_library_signatures():
    """rccl._load() finds torch's librccl.so and declares the C signatures (no GPU needed to load it)."""
    sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
    from semseg_amd import rccl
    lib = rccl._load()
    assert ctypes.sizeof(rccl._UniqueId) == 128
    assert lib.ncclCommInitRank.argtypes[2] is rccl._UniqueId            # by value, as rccl.h:220 declares it
    assert len(lib.ncclAllReduce.argtypes) == 7
    assert (rccl.NCCL_FLOAT32, rccl.NCCL_FLOAT64, rccl.NCCL_SUM) == (7, 8, 0)   # rccl.h:448-468
