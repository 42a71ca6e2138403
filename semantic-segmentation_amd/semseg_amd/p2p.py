"""One-shot SyncBN exchange over peer-mapped device memory (opt-in: SSA_SYNCBN_P2P=1).

The SyncBN exchange of the data-parallel step (the reference: apex.parallel.SyncBatchNorm, config.py:216-222,
network/__init__.py:37-39) is ~240 all-reduces of 20-200 KB per training step, every one on the critical path.  Through
RCCL each is a ring / tree collective with its own launch and handshakes (10-20 us on a node); here every rank writes
its partial sums straight into every peer's exchange buffer over xGMI, publishes a sequence number and sums the slots
locally (csrc/p2p.hip: one single-workgroup kernel, a plain node of the captured step -- no communicator inside the
graph for SyncBN at all).

Set-up (once per process group): a fine-grained device buffer per rank, its hipIpc handle gathered over the existing
torch.distributed group, every peer's buffer opened with hipIpcOpenMemHandle (works for two processes on ONE GPU as
well -- which is how the GPU test exercises the kernel, the flags and the parity protocol; what one GPU cannot show is
cross-GPU visibility over xGMI, which rests on the system-scope atomics of the kernel and the fine-grained allocation).
Any failure on the way -- no fine-grained memory, a handle that does not open, a message larger than a slot -- leaves
the exchange with RCCL / torch.distributed (`usable()` is False and parallel._all_reduce_ goes on as before)."""
import ctypes
import os

import torch
import torch.distributed as dist

ENABLED = os.environ.get("SSA_SYNCBN_P2P", "0") == "1"
SLOT_DOUBLES = int(os.environ.get("SSA_SYNCBN_P2P_SLOT", str(32 * 1024)))      # 256 KB per rank and parity

_HIP_MALLOC_FINEGRAINED = 0x1          # hipDeviceMallocFinegrained
_HIP_IPC_LAZY_PEER = 0x1               # hipIpcMemLazyEnablePeerAccess


class _IpcHandle(ctypes.Structure):
    _fields_ = [("reserved", ctypes.c_char * 64)]      # hipIpcMemHandle_t (HIP_IPC_HANDLE_SIZE)


def _hip():
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    lib.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    lib.hipDeviceSynchronize.argtypes = []
    lib.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_IpcHandle), ctypes.c_void_p]
    lib.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _IpcHandle, ctypes.c_uint]
    lib.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
    lib.hipFree.argtypes = [ctypes.c_void_p]
    return lib


class P2PExchange:
    """Exchange buffers of one process group, mapped into this process."""

    def __init__(self, group=None, slot_doubles=SLOT_DOUBLES):
        from ._lib import lib
        self.lib = lib()
        self.hip = _hip()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.slot = int(slot_doubles)
        self.calls = 0
        self.device = torch.cuda.current_device()
        nb = ctypes.c_size_t(0)
        self.lib.ssa_p2p_buffer_bytes(self.world, self.slot, ctypes.byref(nb))
        nbytes = nb.value
        self.mine = ctypes.c_void_p()
        ok = self.hip.hipExtMallocWithFlags(ctypes.byref(self.mine), nbytes, _HIP_MALLOC_FINEGRAINED) == 0
        handle = _IpcHandle()
        if ok:
            ok = self.hip.hipMemset(self.mine, 0, nbytes) == 0 and self.hip.hipDeviceSynchronize() == 0
        if ok:
            ok = self.hip.hipIpcGetMemHandle(ctypes.byref(handle), self.mine) == 0
        # every rank learns whether every rank got this far (a collective: all ranks take the same branch)
        infos = [None] * self.world
        dist.all_gather_object(infos, (bool(ok), bytes(handle.reserved) if ok else b"", os.getpid()), group=group)
        if not all(i[0] for i in infos):
            self._free()
            raise RuntimeError("peer-mapped exchange buffers unavailable on rank(s) %s" % [r for r, i in enumerate(infos) if not i[0]])
        self.opened = []
        ptrs = []
        failed = False
        for r, (_, raw, pid) in enumerate(infos):
            if r == self.rank:
                ptrs.append(self.mine.value)
                continue
            h = _IpcHandle()
            ctypes.memmove(ctypes.byref(h), raw, 64)
            p = ctypes.c_void_p()
            if self.hip.hipIpcOpenMemHandle(ctypes.byref(p), h, _HIP_IPC_LAZY_PEER) != 0 or not p.value:
                failed = True
                ptrs.append(0)
            else:
                self.opened.append(p)
                ptrs.append(p.value)
        flags = [None] * self.world
        dist.all_gather_object(flags, not failed, group=group)
        if not all(flags):
            self._free()
            raise RuntimeError("hipIpcOpenMemHandle failed on rank(s) %s" % [r for r, f in enumerate(flags) if not f])
        self.peers = torch.tensor(ptrs, dtype=torch.int64, device="cuda")
        self.seq = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        dist.barrier(group=group)       # nobody writes a peer before every peer has cleared and mapped its buffer

    def _free(self):
        for p in getattr(self, "opened", []):
            self.hip.hipIpcCloseMemHandle(p)
        self.opened = []
        if self.mine:
            self.hip.hipFree(self.mine)
            self.mine = ctypes.c_void_p()

    def fits(self, t):
        return t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and 0 < t.numel() <= self.slot

    def all_reduce_sum_(self, t):
        """In-place SUM over ranks of a dense fp64 device tensor of at most `slot` elements, on the current stream."""
        from ._lib import check
        self.calls += 1
        check(self.lib.ssa_p2p_allreduce_f64(ctypes.c_void_p(t.data_ptr()), t.numel(), ctypes.c_void_p(self.peers.data_ptr()),
                                             self.rank, self.world, ctypes.c_void_p(self.seq.data_ptr()), self.slot,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ssa_p2p_allreduce_f64")
        return t

    def timeouts(self):
        n = ctypes.c_uint(0)
        self.lib.ssa_p2p_timeouts(ctypes.byref(n))
        return n.value


_EXCHANGES = {}
_FAILED = set()


def exchange(group=None):
    """The process group's exchange (created on first use -- a collective call: every rank reaches it at the same
    point of the program), or None when it could not be set up (remembered: no second attempt)."""
    key = id(group) if group is not None else 0
    if key in _FAILED:
        return None
    x = _EXCHANGES.get(key)
    if x is None:
        try:
            x = _EXCHANGES[key] = P2PExchange(group)
        except Exception as e:      # noqa: BLE001  (every rank raises together: see __init__)
            _FAILED.add(key)
            import warnings
            warnings.warn("SSA_SYNCBN_P2P=1: %s -- the SyncBN exchange stays on RCCL / torch.distributed" % (e,))
            return None
    return x


def usable(t, group=None):
    """This SUM all-reduce goes over the peer-mapped buffers: switched on, device fp64, small enough for a slot."""
    if not ENABLED or not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
        return False
    if t.numel() > SLOT_DOUBLES:
        return False
    x = exchange(group)
    return x is not None and x.fits(t)


def total_calls():
    return sum(x.calls for x in _EXCHANGES.values())
