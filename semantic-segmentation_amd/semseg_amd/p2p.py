"""One-shot SyncBN exchange over peer-mapped device memory (opt-in: SSA_SYNCBN_P2P=1).

The SyncBN exchange of the data-parallel step (the reference: apex.parallel.SyncBatchNorm, config.py:216-222,
network/__init__.py:37-39) is ~240 all-reduces of 20-200 KB per training step, every one on the critical path.  Through
RCCL each is a ring / tree collective with its own launch and handshakes (10-20 us on a node); here every rank writes
its partial sums straight into every peer's exchange buffer over xGMI, publishes a sequence number and sums the slots
locally (csrc/p2p.hip: one single-workgroup kernel, a plain node of the captured step -- no communicator inside the
graph for SyncBN at all).

Set-up (once per process group): an uncached (fine-grained) device buffer per rank, mapped into every peer -- by a
POSIX file descriptor of the allocation sent over a unix socket (HIP's virtual-memory API), or by a hipIpc handle
gathered over the existing torch.distributed group where the container lets hipIpcOpenMemHandle work.  Both work for
two processes on ONE GPU as well -- which is how the GPU test exercises the kernel, the flags and the parity protocol
across address spaces; what one GPU cannot show is cross-GPU visibility over xGMI, which rests on the system-scope
atomics of the kernel and the uncached allocation.
Any failure on the way -- no exportable memory, a handle that does not open, a message larger than a slot -- leaves
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
    """Exchange buffers of one process group, mapped into this process.

    Two ways to get a peer's buffer into this address space, tried in this order (every rank takes the same one: the
    outcome of each attempt is all-gathered):
      "fd"   the virtual-memory API: the owner creates the allocation exportable as a POSIX file descriptor
             (ssa_p2p_vmm_alloc), the descriptor travels over an abstract unix socket (SCM_RIGHTS), the peer maps it
             (ssa_p2p_vmm_import).  Needs no special rights;
      "ipc"  hipIpcGetMemHandle / hipIpcOpenMemHandle on a fine-grained allocation.  In dmabuf mode the runtime fetches
             the exporter's descriptor with pidfd_getfd, which an unprivileged container refuses (seccomp / no
             CAP_SYS_PTRACE: `invalid device pointer`, measured on the round's GPU boxes, profiles/r06_notes.md)."""

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
        self.nbytes = nb.value
        self.mine = ctypes.c_void_p()
        self.mine_mapped = 0            # > 0: `mine` is a virtual-memory mapping of that many bytes
        self.opened = []                # hipIpc mappings of peers
        self.mapped = []                # (ptr, bytes) virtual-memory mappings of peers
        self.route = None
        errors = []
        for route in os.environ.get("SSA_SYNCBN_P2P_ROUTE", "fd,ipc").split(","):
            try:
                ptrs = self._setup_fd() if route == "fd" else self._setup_ipc()
                self.route = route
                break
            except RuntimeError as e:       # raised on EVERY rank together (see _agree)
                errors.append("%s: %s" % (route, e))
                self._free()
        if self.route is None:
            raise RuntimeError("; ".join(errors))
        self.peers = torch.tensor(ptrs, dtype=torch.int64, device="cuda")
        self.seq = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        dist.barrier(group=group)       # nobody writes a peer before every peer has cleared and mapped its buffer

    def _agree(self, ok, what):
        """All ranks learn whether all ranks succeeded; raises on every rank if one did not."""
        flags = [None] * self.world
        dist.all_gather_object(flags, bool(ok), group=self.group)
        if not all(flags):
            raise RuntimeError("%s failed on rank(s) %s" % (what, [r for r, f in enumerate(flags) if not f]))

    # ---- route "fd"
    def _setup_fd(self):
        import socket
        import threading
        fd, mapped = ctypes.c_int(-1), ctypes.c_size_t(0)
        ok = self.lib.ssa_p2p_vmm_alloc(self.nbytes, ctypes.byref(self.mine), ctypes.byref(fd), ctypes.byref(mapped)) == 0
        self.mine_mapped = mapped.value if ok else 0
        token = [os.urandom(8).hex()]
        dist.broadcast_object_list(token, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        name = lambda r: "\0ssa_p2p_%s_%d" % (token[0], r)           # abstract namespace: nothing to unlink
        srv = None
        if ok:
            try:
                srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                srv.bind(name(self.rank))
                srv.listen(self.world)
                srv.settimeout(60)
            except OSError:
                ok = False
        try:
            self._agree(ok, "exportable device allocation (hipMemCreate / hipMemExportToShareableHandle)")
            sizes = [None] * self.world
            dist.all_gather_object(sizes, self.mine_mapped, group=self.group)

            def serve():
                for _ in range(self.world - 1):
                    try:
                        c, _a = srv.accept()
                    except OSError:
                        return
                    with c:
                        socket.send_fds(c, [b"f"], [fd.value])

            th = threading.Thread(target=serve, daemon=True)
            th.start()
            ptrs, failed = [], False
            for r in range(self.world):
                if r == self.rank:
                    ptrs.append(self.mine.value)
                    continue
                p = ctypes.c_void_p()
                try:
                    with socket.socket(socket.AF_UNIX, socket.SOCK_STREAM) as c:
                        c.settimeout(60)
                        c.connect(name(r))
                        _m, fds, _f, _ad = socket.recv_fds(c, 1, 1)
                    rc = self.lib.ssa_p2p_vmm_import(fds[0], sizes[r], ctypes.byref(p))
                    os.close(fds[0])
                    if rc != 0 or not p.value:
                        raise OSError("ssa_p2p_vmm_import: %d" % rc)
                    self.mapped.append((p, sizes[r]))
                    ptrs.append(p.value)
                except (OSError, IndexError):
                    failed = True
                    ptrs.append(0)
            th.join(90)
            self._agree(not failed, "mapping a peer's allocation (hipMemImportFromShareableHandle / hipMemMap)")
            return ptrs
        finally:
            if srv is not None:
                srv.close()
            if fd.value >= 0:
                os.close(fd.value)

    # ---- route "ipc"
    def _setup_ipc(self):
        ok = self.hip.hipExtMallocWithFlags(ctypes.byref(self.mine), self.nbytes, _HIP_MALLOC_FINEGRAINED) == 0
        handle = _IpcHandle()
        if ok:
            ok = self.hip.hipMemset(self.mine, 0, self.nbytes) == 0 and self.hip.hipDeviceSynchronize() == 0
        if ok:
            ok = self.hip.hipIpcGetMemHandle(ctypes.byref(handle), self.mine) == 0
        infos = [None] * self.world
        dist.all_gather_object(infos, (bool(ok), bytes(handle.reserved) if ok else b""), group=self.group)
        if not all(i[0] for i in infos):
            raise RuntimeError("fine-grained allocation / hipIpcGetMemHandle failed on rank(s) %s"
                               % [r for r, i in enumerate(infos) if not i[0]])
        ptrs, failed = [], False
        for r, (_, raw) in enumerate(infos):
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
        self._agree(not failed, "hipIpcOpenMemHandle")
        return ptrs

    def _free(self):
        for p in self.opened:
            self.hip.hipIpcCloseMemHandle(p)
        self.opened = []
        for p, n in self.mapped:
            self.lib.ssa_p2p_vmm_unmap(p, n)
        self.mapped = []
        if self.mine:
            if self.mine_mapped:
                self.lib.ssa_p2p_vmm_unmap(self.mine, self.mine_mapped)
            else:
                self.hip.hipFree(self.mine)
            self.mine = ctypes.c_void_p()
            self.mine_mapped = 0

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
