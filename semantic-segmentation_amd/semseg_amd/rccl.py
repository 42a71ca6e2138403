"""Direct RCCL collectives on the compute stream.

The SyncBN exchange of this path is one all-reduce of a few thousand fp64 values per grouped
BatchNorm level (~300 per training step), the gradient exchange one in-place
all-reduce per gradient-arena chunk.  Through torch.distributed each costs a c10d dispatch, a
stream switch to c10d's communication stream and two event fences -- and c10d's watchdog thread
queries events while the step is being captured in a hipGraph, which aborts the capture
(measured on MI355X, profiles/r02_notes.md).  Here the same `ncclAllReduce` is enqueued
directly on the stream the kernels run on: one library call, no stream switch, a plain kernel
node in the captured graph.  The communicator is RCCL's own (the librccl.so torch already
loaded), bootstrapped over the existing torch.distributed process group.
SSA_RCCL_DIRECT=0 falls back to torch.distributed collectives (eager steps only)."""
import ctypes
import glob
import os

import torch
import torch.distributed as dist

ENABLED = os.environ.get("SSA_RCCL_DIRECT", "1") != "0"

NCCL_FLOAT32, NCCL_FLOAT64, NCCL_SUM, NCCL_AVG = 7, 8, 0, 4      # rccl.h: ncclDataType_t / ncclRedOp_t


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]  # NCCL_UNIQUE_ID_BYTES (opaque; may contain NULs)


def _load():
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + \
        ["/opt/rocm/lib/librccl.so"]
    for path in cands:
        try:
            lib = ctypes.CDLL(path)
        except OSError:
            continue
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        return lib
    raise RuntimeError("librccl.so not found (looked in torch/lib and /opt/rocm/lib)")


class DirectComm:
    """One RCCL communicator over the ranks of a torch.distributed group (default: the world)."""

    def __init__(self, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("DirectComm needs an initialised torch.distributed process group to bootstrap")
        self.lib = _load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        box = [ctypes.string_at(ctypes.byref(uid), 128) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        self.calls = 0                  # collectives issued (bench.py reports them per step)
        self.device = torch.cuda.current_device()
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank),
                    "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_reduce_(self, t, average=False):
        """In-place sum (or mean) over ranks of a dense fp32 / fp64 tensor, enqueued on the current stream."""
        dt = {torch.float32: NCCL_FLOAT32, torch.float64: NCCL_FLOAT64}[t.dtype]
        assert t.is_contiguous() and t.is_cuda
        self.calls += 1
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, NCCL_AVG if average else NCCL_SUM,
                                           self.comm, torch.cuda.current_stream().cuda_stream), "ncclAllReduce")
        return t

    def all_reduce_sum_(self, t):
        return self.all_reduce_(t, False)

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


_COMMS = {}


def usable(t, group=None):
    """Direct RCCL carries this tensor: enabled, default group on the nccl (= RCCL) backend, device memory."""
    return ENABLED and group is None and t.is_cuda and dist.get_backend() == "nccl"


def comm(index=0):
    """Process-wide communicators (created on first use, after init_process_group).  0: the compute stream's
    (SyncBN sums, end-of-step exchanges); 1: the gradient exchange that runs on a communication stream concurrently
    with backward -- collectives of ONE communicator must not be in flight on two streams at once."""
    c = _COMMS.get(index)
    if c is None:
        c = _COMMS[index] = DirectComm()
    return c


def total_calls():
    return sum(c.calls for c in _COMMS.values())


def shutdown():
    for c in _COMMS.values():
        c.destroy()
    _COMMS.clear()
