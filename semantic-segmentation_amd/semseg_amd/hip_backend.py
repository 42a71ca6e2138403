"""HIP backend: autograd.Function wrappers over the C ABI of libsemseg_hip.so.

Every tensor-valued op of the HRNet-OCR-MScale hot path goes through here; the
arithmetic lives in csrc/*.hip.  Activations are NHWC ([B,H,W,C]) bf16, logits
and losses fp32.  PyTorch provides device memory, streams and autograd
bookkeeping only.  Nothing here falls back to ATen math: a missing or stale
library raises in `_lib.lib()`.

Multi-problem execution.  The resolution branches of a HighResolutionModule and the
scale passes of MscaleOCR are independent problems; the network code walks them in
lockstep and hands every depth level to ONE autograd Function here (ConvGroupFn,
BnActGroupFn, BasicBlockGroupFn, ...), which issues the problems' launches inside an
ssa_group_begin/ssa_group_end bracket -- one launch per kernel instantiation instead
of one per problem (csrc/group.h).

Parameter gradients never travel through autograd's AccumulateGrad: weight-gradient
kernels and the BatchNorm backward ADD into slices of a per-step fp32 gradient arena
(cleared by one memset), the weight gradients of a whole module are deferred and
issued as grouped launches, and an end-of-backward callback publishes the slices as
`param.grad` (and hands the arena to the data-parallel wrapper for its all-reduce).

hipGraph-capture safe: no host synchronisation, no `.item()`, allocations come
from torch's caching allocator, kernels are enqueued on the current stream.
"""
import contextlib
import ctypes
import os
import weakref

import torch

from ._lib import lib, check, ConvDesc, PackJob, BnUpdateJob, BnEvalJob, ProfileRec

from ._lib import ACT as _ACT_NAME     # storage format of the activations = the library build (SSA_ACT_DTYPE)
ACT_DTYPE = torch.float16 if _ACT_NAME == "fp16" else torch.bfloat16


_FP16_TRAINING = [False]


def enable_fp16_training(on=True):
    """Called by semseg_amd.amp.attach_scaler: a dynamic loss scaler is in place, the criteria may run backward on the
    fp16-storage build."""
    _FP16_TRAINING[0] = bool(on)


def _no_fp16_training():
    """fp16 storage without loss scaling cannot train: a 1024x1024 step's per-pixel loss gradients (~1e-6) underflow
    fp16 on their way back.  The criteria of the operator surface (ops.HipBackend.cross_entropy / bce_rmi) raise
    rather than train on flushed gradients unless a loss scaler has been attached to the optimizer
    (semseg_amd.amp.initialize / attach_scaler -- what the reference's `amp.initialize(net, optim, opt_level)` does
    under dropin.install(); train.py:380-381)."""
    if _ACT_NAME == "fp16" and not _FP16_TRAINING[0]:
        raise NotImplementedError("SSA_ACT_DTYPE=fp16 trains with dynamic loss scaling only: call "
                                  "semseg_amd.amp.initialize(net, optimizer) (apex.amp.initialize under dropin.install()) "
                                  "and run backward on amp.scale_loss(loss, optimizer)")


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _roundup(a, b):
    return (a + b - 1) // b * b


def _pixels(t):
    """Return (tensor, ld) with `tensor` [B,H,W,C] laid out as dense pixels of
    stride `ld` elements and unit channel stride (a channel slice of a wider
    NHWC buffer qualifies).  Copies only if the layout does not qualify."""
    assert t.dim() == 4
    B, H, W, C = t.shape
    sb, sh, sw, sc = t.stride()
    ok = (sc == 1 or C == 1)
    ld = None
    for size, stride, mult in ((W, sw, 1), (H, sh, W), (B, sb, H * W)):
        if size > 1:
            if stride % mult:
                ok = False
                break
            cand = stride // mult
            if ld is None:
                ld = cand
            elif cand != ld:
                ok = False
                break
    if ld is None:
        ld = C
    if not ok or ld < C:
        t = t.contiguous()
        ld = C
    return t, ld


# --------------------------------------------------------------------------
# grouped launches (csrc/group.h)
# --------------------------------------------------------------------------
@contextlib.contextmanager
def group():
    """Launches of group-aware entry points issued inside the bracket are mutually
    independent problems; they leave as one launch per kernel instantiation."""
    L = lib()
    check(L.ssa_group_begin(), "ssa_group_begin")
    try:
        yield
    except BaseException:
        L.ssa_group_abort()
        raise
    check(L.ssa_group_end(_s()), "ssa_group_end")


_PROFILING = [False]


def profile_begin():
    """bench.py's roofline leg: time every launch of the group-aware kernels with HIP events."""
    check(lib().ssa_profile_begin(), "ssa_profile_begin")
    _PROFILING[0] = True


def profile_end(max_recs=512):
    """-> list of dicts {kernel, launches, jobs, total_us, flops, bytes} per kernel instantiation."""
    _PROFILING[0] = False
    arr = (ProfileRec * max_recs)()
    n = lib().ssa_profile_end(arr, max_recs)
    if n < 0:
        raise RuntimeError("ssa_profile_end failed")
    return [{"kernel": arr[i].kernel.decode(), "launches": arr[i].launches, "jobs": arr[i].jobs,
             "total_us": arr[i].total_us, "flops": arr[i].flops, "bytes": arr[i].bytes} for i in range(min(n, max_recs))]


def _note(flops, nbytes):
    if _PROFILING[0]:
        lib().ssa_profile_note(float(flops), float(nbytes))


# --------------------------------------------------------------------------
# packed filters: persistent bf16 GEMM operands, one per (parameter, form).
# They are re-derived from the fp32 parameters whenever those changed (every
# optimizer step) by ONE batched launch at the start of the step -- captured in
# the step's hipGraph -- instead of 1,276 separate pack launches.
# The cache keys on the parameter's storage address and autograd version counter:
# optimizers that write through `p.data` (no version bump) must call
# invalidate_packed_filters() after their step (FusedSGD bumps the counters itself).
# --------------------------------------------------------------------------
class _Packed:
    __slots__ = ("wref", "out", "Kpad", "version", "job", "shape")


_PACKED = {}
_PACK_TILED = os.environ.get("SSA_PACK_TILED", "1") != "0"


def clear_pack_cache():
    join_pack()
    _PACKED.clear()
    _JOB_TABLES.clear()


def _pack_tiles(jobs, device):
    """Work list of ssa_pack_filters_tiled: every SOURCE tensor's [Cout] x [Cin] plane in 32 x ct tiles; the jobs that
    pack the same tensor (its operand forms) are chained through `elem_begin`, the tiles name the first of them.
    -> (device int32 [ntiles, 4], ntiles, max ct*taps), or None when a filter is too large for that path."""
    L = lib()
    tiles, worst = [], 0
    last = {}                  # (source, shape) -> index of the latest job of that source
    for i, j in enumerate(jobs):
        ct = L.ssa_pack_tile_channels(j.KH, j.KW)
        if ct <= 0:
            return None
        j.elem_begin = 0
        key = (j.w, j.Cout, j.Cin, j.KH, j.KW)
        prev = last.get(key)
        last[key] = i
        if prev is not None:
            jobs[prev].elem_begin = i + 1          # the owner's workgroups go on with this form
            continue
        worst = max(worst, ct * j.KH * j.KW)
        for co0 in range(0, j.Cout, 32):
            for ci0 in range(0, j.Cin, ct):
                tiles.append((i, co0, ci0, ct))
    t = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 4)
    return t.to(device), len(tiles), worst


def invalidate_packed_filters():
    """Force a re-pack of every cached filter at the next step."""
    for e in _PACKED.values():
        e.version = -1


def _make_job(w, out, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows, layout=0):
    return PackJob(w.data_ptr(), out.data_ptr(), 0, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows, layout)


# The re-pack of a training step (288 MB of fp32 parameters -> their 16-bit operand forms, ~0.2 ms, HBM-bound) is the
# first thing of the step and only the stem's filters are needed at once: begin_step packs the first SSA_PACK_EARLY
# filters (registration order = order of first use) on the compute stream and the rest on a stream of its own, next to
# the stem and layer1 (~0.7 ms of 10-30 us launches); the first conv that asks for one of the late filters makes the
# compute stream wait (_packed_filter).  SSA_PACK_EARLY=0 (the default): one launch on the compute stream -- measured
# with 20: 20.19 / 20.46 / 20.38 against 20.41 ms per step, nothing gained (profiles/r06_notes.md call S).
_PACK_EARLY = int(os.environ.get("SSA_PACK_EARLY", "0"))
_PACK_SIDE = {"stream": None, "pending": False, "late": frozenset()}
_JOB_TABLES = {}


def join_pack():
    """The current stream waits for the filters being packed on the side stream."""
    if _PACK_SIDE["pending"]:
        torch.cuda.current_stream().wait_stream(_PACK_SIDE["stream"])
        _PACK_SIDE["pending"] = False
        _PACK_SIDE["late"] = frozenset()


def _pack_launch(stale):
    # a table is valid for exactly these (source, destination) buffers: a model rebuilt at the
    # same parameter addresses has new destination buffers and must not reuse the old table
    tkey = tuple((k, e.out.data_ptr()) for k, e, _ in stale)
    tab = _JOB_TABLES.get(tkey)
    if tab is None:
        if len(_JOB_TABLES) >= 8:
            _JOB_TABLES.clear()
        arr = (PackJob * len(stale))(*[e.job for _, e, _ in stale])      # copies: the chain links are this table's
        dev = stale[0][2].device
        tl = _pack_tiles(arr, dev) if _PACK_TILED else None
        if tl is None:
            for j in arr:
                j.elem_begin = 0
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        tab = _JOB_TABLES[tkey] = dict(dev=host.to(dev), n=len(stale), tiles=tl[0] if tl else None,
                                       ntiles=tl[1] if tl else 0, max_ct_taps=tl[2] if tl else 0)
    if tab["tiles"] is not None:
        check(lib().ssa_pack_filters_tiled(_p(tab["dev"]), _p(tab["tiles"]), tab["ntiles"],
                                           tab["max_ct_taps"], _s()), "ssa_pack_filters_tiled")
    else:
        check(lib().ssa_pack_filters_batched(_p(tab["dev"]), tab["n"], 32, _s()),
              "ssa_pack_filters_batched")


def refresh_packed_filters(overlap=False):
    """Re-pack every registered filter whose parameter changed since it was
    packed.  One launch for all of them; overlap=True (begin_step: host code asks for every filter before its conv is
    launched): two launches, the second on a side stream (above)."""
    join_pack()
    stale = []
    for key, e in list(_PACKED.items()):
        w = e.wref()
        if w is None or w.data_ptr() != key[0]:
            del _PACKED[key]
            continue
        if w._version != e.version:
            stale.append((key, e, w))
    if not stale:
        return
    if overlap and 0 < _PACK_EARLY < len(stale) and stale[0][2].is_cuda:
        early, late = stale[:_PACK_EARLY], stale[_PACK_EARLY:]
        _pack_launch(early)
        if _PACK_SIDE["stream"] is None:
            _PACK_SIDE["stream"] = torch.cuda.Stream()
        side = _PACK_SIDE["stream"]
        side.wait_stream(torch.cuda.current_stream())         # the optimizer's update of the parameters
        with torch.cuda.stream(side):
            _pack_launch(late)
        _PACK_SIDE["pending"] = True
        _PACK_SIDE["late"] = frozenset(k for k, _, _ in late)
    else:
        _pack_launch(stale)
    for _, e, w in stale:
        e.version = w._version


def _packed_filter(weight, mode, cin_pad, cout_pad):
    key = (weight.data_ptr(), mode, cin_pad, cout_pad)
    e = _PACKED.get(key)
    if e is not None and e.wref() is not None and e.version == weight._version and e.shape == tuple(weight.shape):
        if _PACK_SIDE["pending"] and key in _PACK_SIDE["late"]:
            join_pack()
        return e.out, e.Kpad
    Cout, Cin, KH, KW = weight.shape
    layout = 0                  # one fragment order (reserved field of ssa_pack_job)
    api_mode = mode
    if mode >= 4:               # parity class (py, px) of a stride-2 data gradient: (1+py)*(1+px) taps
        rows, kdim = Cin, (1 + ((mode - 4) >> 1)) * (1 + ((mode - 4) & 1)) * cout_pad
    elif mode & 1 == 0:
        rows, kdim = Cout, KH * KW * cin_pad
    else:
        rows, kdim = Cin, KH * KW * cout_pad
    if mode < 2 or mode >= 4:
        Kpad = _roundup(kdim, 32)
    else:                       # MFMA-fragment order (conv_tile.hip): rows padded to 32, K exact
        rows, Kpad = _roundup(rows, 32), kdim
    w = weight.detach()
    direct = w.dtype == torch.float32 and w.is_contiguous()
    if not direct:
        w = w.float().contiguous()
    if e is None or e.wref() is None or e.shape != tuple(weight.shape):
        e = _Packed()
        e.shape = tuple(weight.shape)
        e.out = torch.zeros((rows, Kpad), dtype=ACT_DTYPE, device=weight.device)    # the tiled repack keeps the padding
        e.Kpad = Kpad
        e.wref = weakref.ref(weight)
        e.job = _make_job(w, e.out, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows, layout)
        if direct:              # only parameters packed straight from their own storage are batched
            _PACKED[key] = e
            _JOB_TABLES.clear()              # a new destination buffer: the device tables are stale
    check(lib().ssa_pack_filter(_p(w), _p(e.out), Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, api_mode, _s()),
          "ssa_pack_filter")
    e.version = weight._version
    return e.out, e.Kpad


# --------------------------------------------------------------------------
# fp64 statistics arena: BN (and bias-gradient) partial sums are carved out of
# chunks that are cleared with a single memset instead of one memset per
# BatchNorm call (1,262 per training step).  A chunk that runs full is left alone
# (kernels in flight may still accumulate into it); a fresh one is started.
# --------------------------------------------------------------------------
class _Arena:
    CHUNK = 1 << 22             # doubles (32 MB)

    def __init__(self):
        self.buf = None
        self.cur = 0
        self.spill = []         # full chunks of this step, kept alive until the next reset

    def reset(self, device):
        self.spill = []
        if self.buf is None or self.buf.device != device:
            self.buf = torch.empty((self.CHUNK,), dtype=torch.float64, device=device)
        self.buf.zero_()
        self.cur = 0

    def take(self, n, device):
        n = _roundup(n, 32)
        if self.buf is None or self.buf.device != device:
            self.reset(device)
        if self.cur + n > self.buf.numel():
            self.spill.append(self.buf)
            self.buf = torch.zeros((max(self.CHUNK, n),), dtype=torch.float64, device=device)
            self.cur = 0
        out = self.buf[self.cur:self.cur + n]
        self.cur += n
        return out


_ARENA = _Arena()


# --------------------------------------------------------------------------
# deferred BatchNorm running-statistics updates (see ssa_bn_update_running_batched)
# --------------------------------------------------------------------------
class _BnUpdates:
    def __init__(self):
        self.slots = {}       # id(bn) -> (bn, [persistent pass_stats tensors])
        self.step = {}        # id(bn) -> passes issued in the current step (insertion ordered)
        self.table_key = None
        self.table = None
        self.max_c = 1

    def slot(self, bn):
        """Persistent [2C+1] fp32 buffer for the next training pass over `bn` in this step."""
        k = id(bn)
        ent = self.slots.get(k)
        if ent is None or ent[0] is not bn or ent[1][0].device != bn.running_mean.device:
            ent = self.slots[k] = (bn, [])
        n = self.step.get(k, 0)
        assert n < 8, "more than 8 training passes over one BatchNorm layer in a step"
        while len(ent[1]) <= n:
            ent[1].append(torch.empty((2 * bn.num_features + 1,), dtype=torch.float32,
                                      device=bn.running_mean.device))
        self.step[k] = n + 1
        return ent[1][n]

    def flush(self):
        if not self.step:
            return
        key = tuple(self.step.items())
        if key != self.table_key:
            jobs = []
            for k, n in self.step.items():
                bn, slots = self.slots[k]
                ps = (ctypes.c_void_p * 8)(*[slots[i].data_ptr() for i in range(n)])
                nbt = bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else None
                jobs.append(BnUpdateJob(bn.running_mean.data_ptr(), bn.running_var.data_ptr(), nbt, ps,
                                        bn.num_features, n, 0.1 if bn.momentum is None else bn.momentum, 0))
            arr = (BnUpdateJob * len(jobs))(*jobs)
            dev = self.slots[next(iter(self.step))][0].running_mean.device
            self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self.table_key = key
            self.max_c = max(self.slots[k][0].num_features for k in self.step)
        check(lib().ssa_bn_update_running_batched(_p(self.table), len(self.step), self.max_c, _s()),
              "ssa_bn_update_running_batched")
        self.step = {}


_BN_UPDATES = _BnUpdates()


# --------------------------------------------------------------------------
# evaluation-mode BatchNorm coefficients: constants of a forward, one launch for all layers
# --------------------------------------------------------------------------
class _BnEvalCoefs:
    """scale / shift / mean / invstd of every BatchNorm an inference forward (no autograd) uses, in persistent [4, C]
    buffers.  They used to be one single-workgroup launch per layer AND scale pass (314 per single-scale forward of
    HRNet-OCR, 17 % of its time).  Now: begin_step refreshes every registered layer's buffer with ONE batched launch
    (ssa_bn_finalize_eval_batched: part of a captured forward, so a replay after training reads the current running
    statistics) when the previous forward used the registry; a layer seen for the first time, or a forward that
    follows a training forward, computes its coefficients singly as before and registers.  Validity is a generation
    number bumped by every begin_step -- the running statistics are written by kernels through raw pointers, version
    counters would not notice."""

    def __init__(self):
        self.entries = {}        # key -> [weakref(running_mean), running_var, gamma, beta, coef, generation, C, eps]
        self.gen = 0
        self.uses = 0
        self.table_key = None
        self.table = None
        self.max_c = 1

    def begin(self):
        self.gen += 1
        used, self.uses = self.uses, 0
        if not used or not self.entries:
            return
        dead = [k for k, e in self.entries.items() if e[0]() is None]
        for k in dead:
            del self.entries[k]
        if not self.entries:
            return
        key = tuple(self.entries)
        if key != self.table_key:
            jobs = [BnEvalJob(None if e[2] is None else e[2].data_ptr(), None if e[3] is None else e[3].data_ptr(),
                              e[0]().data_ptr(), e[1].data_ptr(), e[4].data_ptr(), e[6], e[7]) for e in self.entries.values()]
            arr = (BnEvalJob * len(jobs))(*jobs)
            dev = next(iter(self.entries.values()))[4].device
            if any(e[4].device != dev for e in self.entries.values()):
                return                       # layers on several devices: the single launches stay
            self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self.table_key = key
            self.max_c = max(e[6] for e in self.entries.values())
        check(lib().ssa_bn_finalize_eval_batched(_p(self.table), len(self.entries), self.max_c, _s()),
              "ssa_bn_finalize_eval_batched")
        for e in self.entries.values():
            e[5] = self.gen

    def get(self, meta, gamma, beta, C, device):
        """The layer's [4, C] coefficient buffer, current for this forward; None: not cacheable (the caller computes)."""
        rm, rv = meta.running_mean, meta.running_var
        ok = lambda t: t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.device == device)  # noqa: E731
        if rm is None or rv is None or not (ok(rm) and ok(rv) and ok(gamma) and ok(beta)):
            return None
        key = (rm.data_ptr(), rv.data_ptr(), 0 if gamma is None else gamma.data_ptr(), 0 if beta is None else beta.data_ptr(),
               float(meta.eps))
        e = self.entries.get(key)
        if e is None or e[0]() is not rm or e[6] != C:
            e = self.entries[key] = [weakref.ref(rm), rv, gamma, beta,
                                     torch.empty((4, C), dtype=torch.float32, device=device), -1, C, float(meta.eps)]
            self.table_key = None
        self.uses += 1
        if e[5] != self.gen:
            coef = e[4]
            check(lib().ssa_bn_finalize(None, 1.0, C, _p(gamma), _p(beta), _p(rm), _p(rv), float(meta.momentum),
                                        float(meta.eps), 1, _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _s()),
                  "ssa_bn_finalize")
            e[5] = self.gen
        return e[4]


_BN_EVAL = _BnEvalCoefs()


def end_forward():
    """Apply the running-statistics updates of every BatchNorm that ran in training
    mode since begin_step (one launch, passes in issue order)."""
    _BN_UPDATES.flush()


def begin_step(device=None):
    _BN_UPDATES.step = {}
    _PENDING_STATS.clear()
    _BN_EVAL.begin()
    if _GRADS.armed or _WGRAD_Q or _GRADS.slots or _GRADS.chunks:
        # a backward pass that raised never ran its end-of-backward callback (or raised inside it): its queued
        # weight-gradient jobs and its half-filled gradient slices are abandoned here -- otherwise the arena would
        # stay "armed" (or keep stale slots) and no later backward pass would ever publish a gradient again
        del _WGRAD_Q[:]
        _GRADS.abandon()
    refresh_packed_filters(overlap=True)
    if device is not None:
        _ARENA.reset(device)


# --------------------------------------------------------------------------
# gradient arena: parameter gradients are accumulated by the kernels themselves
# --------------------------------------------------------------------------
def _is_param(t):
    return isinstance(t, torch.nn.Parameter) and t.requires_grad and t.dtype == torch.float32


_GRAD_SINK = [None]     # data-parallel wrapper's exchange: object with reduce_range(buf, lo, hi) and finish()


def set_grad_sink(sink):
    """sink.reduce_range(buf, lo, hi): average buf[lo:hi] over the ranks, in place (may run on a communication
    stream); sink.finish(): make the compute stream wait for every exchange issued so far.  With a sink installed the
    queued weight gradients are flushed every SSA_DDP_FLUSH_AT layers DURING backward and the arena range completed by
    each flush is handed to reduce_range at once -- the exchange overlaps the rest of backward
    (apex.parallel.DistributedDataParallel's buckets, network/__init__.py:37-39)."""
    _GRAD_SINK[0] = sink


class _GradArena:
    """fp32 slices, one per parameter, of buffers cleared on allocation.  Weight-gradient
    reduces and the BatchNorm backward ADD into a parameter's slice (every scale pass, every
    flush), so autograd never launches an accumulation; at the end of backward the slices
    become `param.grad` (added to an existing .grad, as autograd would)."""
    FIRST_CHUNK = 1 << 24       # elements, until the total of a step is known

    def __init__(self):
        self.chunks = []        # [tensor, used, elements already handed to the gradient sink]
        self.slots = {}         # id(param) -> (param, view, chunk index, offset)
        self.late = []          # (param, view): contributions that arrived after the parameter's slice was exchanged
        self.total_last = 0
        self.armed = False

    def _arm(self):
        if not self.armed:
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(self.publish)
            self.armed = True

    def slot(self, p):
        e = self.slots.get(id(p))
        if e is not None and e[0] is p:
            self._arm()
            if e[3] >= self.chunks[e[2]][2]:
                return e[1]
            # the slice is already on its way to the other ranks (a parameter used again later in backward):
            # this contribution gets a slice of its own, exchanged with a later range and added at publication
            v = self._take(p)[0]
            self.late.append((p, v))
            return v
        v, ci, off = self._take(p)
        self.slots[id(p)] = (p, v, ci, off)
        self._arm()
        return v

    def _take(self, p):
        n = _roundup(p.numel(), 64)
        if not self.chunks or self.chunks[-1][1] + n > self.chunks[-1][0].numel() or \
                self.chunks[-1][0].device != p.device:
            size = max(n, self.total_last if not self.chunks else self.FIRST_CHUNK, 1)
            if not self.chunks and not self.total_last:
                size = max(n, self.FIRST_CHUNK)
            self.chunks.append([torch.zeros((size,), dtype=torch.float32, device=p.device), 0, 0])
        c = self.chunks[-1]
        off = c[1]
        v = c[0][off:off + p.numel()].view(p.shape)
        c[1] += n
        return v, len(self.chunks) - 1, off

    def reduce_completed(self):
        """Hand every arena range that is final (its kernels are enqueued) to the gradient sink."""
        sink = _GRAD_SINK[0]
        if sink is None:
            return
        for c in self.chunks:
            if c[1] > c[2]:
                sink.reduce_range(c[0], c[2], c[1])
                c[2] = c[1]

    def abandon(self):
        """After a backward pass that raised: forget its half-filled slices AND the bookkeeping the next pass's
        exchange boundaries are derived from (otherwise this rank's first exchange would fire after a different
        number of layers than on the other ranks and the reduce_range sizes would no longer match across ranks)."""
        self.armed = False
        _SINCE_REDUCE[0] = 0
        join_wgrads()                 # the side stream's launches of the abandoned pass are ordered before what follows
        self.chunks = []
        self.slots = {}
        self.late = []

    def publish(self):
        self.armed = False
        try:
            flush_wgrads()
            _SINCE_REDUCE[0] = 0
            if _GRAD_SINK[0] is not None and self.slots:
                with _on_wgrad_stream():
                    self.reduce_completed()
            join_wgrads()
            if not self.slots:
                return
            if _GRAD_SINK[0] is not None:
                _GRAD_SINK[0].finish()
            dst, src = [], []
            for p, v, _, _ in self.slots.values():
                if p.grad is None:
                    p.grad = v
                else:
                    dst.append(p.grad)
                    src.append(v)
            for p, v in self.late:
                dst.append(p.grad)
                src.append(v)
            if dst:
                torch._foreach_add_(dst, src)
            self.total_last = sum(c[1] for c in self.chunks)
        finally:
            # also when the flush / the all-reduce raised: stale slots would make the next backward return early
            # from slot() and never publish again
            del _WGRAD_Q[:]
            self.chunks = []
            self.slots = {}
            self.late = []


_GRADS = _GradArena()


# --------------------------------------------------------------------------
# raw conv launches
# --------------------------------------------------------------------------
def _tile_desc(B, H, W, Cin, ldx, Cout, k, stride, pad, dil, Ho, Wo, out_f32):
    return ConvDesc(B, H, W, Cin, ldx, Ho, Wo, Cout, Cout, k[0], k[1], stride, pad, dil, 0, 0, int(out_f32), -1)


def tile_supported(d):
    return bool(lib().ssa_conv2d_tile_supported(ctypes.byref(d)))


def halo_supported(d):
    return bool(lib().ssa_conv2d_halo_supported(ctypes.byref(d)))


def wide_supported(d):
    """The 256 x 256 GEMM (csrc/conv_gemm_wide.hip) takes this 1x1 problem and is the better choice for it."""
    return bool(lib().ssa_conv2d_gemm_wide_supported(ctypes.byref(d)))


_STAT_REPLICAS = None


def stat_replicas():
    global _STAT_REPLICAS
    if _STAT_REPLICAS is None:
        _STAT_REPLICAS = int(lib().ssa_bn_stat_replicas())
    return _STAT_REPLICAS


def _conv_bytes(P_in, Cin, P_out, Cout, k, out_bytes=2):
    return 2.0 * P_in * Cin + float(out_bytes) * P_out * Cout + 2.0 * Cout * Cin * k[0] * k[1]


def _igemm(x, ldx, geom_in, wp, Kpad, bias, geom_out, Cout, k, stride, pad, dil, transposed, out_f32,
           cfg=-1, stats=None, out=None, affine=None):
    """Raw launch: x viewed as [B,H,W,Cin] (ldx) -> y [B,Ho,Wo,Cout] (`out`: a dense tensor of that size to write)."""
    B, H, W, Cin = geom_in
    Ho, Wo = geom_out
    y = out if out is not None else torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else ACT_DTYPE,
                                                device=x.device)
    assert y.is_contiguous() and y.numel() == B * Ho * Wo * Cout
    d = ConvDesc(B, H, W, Cin, ldx, Ho, Wo, Cout, Cout, k[0], k[1], stride, pad, dil, int(transposed),
                 Kpad, int(out_f32), cfg)
    # algorithmic flops: taps that fall on the stride grid only (transposed) = forward flops
    taps = k[0] * k[1] / (stride * stride if transposed else 1)
    _note(2.0 * B * Ho * Wo * Cout * Cin * taps, _conv_bytes(B * H * W, Cin, B * Ho * Wo, Cout, k, 4 if out_f32 else 2))
    if affine is not None:          # (coef, residual, ldres, relu): the inference BatchNorm as the epilogue
        coef, res, ldres, relu = affine
        check(lib().ssa_conv2d_igemm_affine(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(y), _p(coef), _p(res), ldres,
                                            int(relu), _s()), "ssa_conv2d_igemm_affine")
        return y
    check(lib().ssa_conv2d_igemm_stats(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(y), _p(stats), _s()),
          "ssa_conv2d_igemm")
    return y


def _tile_conv(d, x, wfrag, bias, stats, halo=False, aux=None, ldaux=0, coef=None, mode=0, wide=False):
    """Halo-staged conv launch: conv_tile.hip (small-channel 3x3 convs; aux = fused epilogue tile of
    the data gradient, see ssa_conv2d_tile_aux) or conv_halo_gemm.hip (large-channel 3x3 / 1x1)."""
    y = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=torch.float32 if d.out_f32 else ACT_DTYPE,
                    device=x.device)
    P = d.B * d.Ho * d.Wo
    _note(2.0 * P * d.Cout * d.Cin * d.KH * d.KW,
          _conv_bytes(P, d.Cin, P, d.Cout, (d.KH, d.KW), 4 if d.out_f32 else 2) + (2.0 * P * d.Cout if mode else 0.0))
    L = lib()
    if wide:
        check(L.ssa_conv2d_gemm_wide(ctypes.byref(d), _p(x), _p(wfrag), _p(bias), _p(y), _p(stats), _s()), "ssa_conv2d_gemm_wide")
    elif not halo and bias is None and tile_p_supported(d):
        check(L.ssa_conv2d_tile_p(ctypes.byref(d), _p(x), _p(wfrag), None, _p(y), _p(stats),
                                  _p(aux), ldaux, _p(coef), mode, _s()), "ssa_conv2d_tile_p")
    elif mode:
        check(L.ssa_conv2d_tile_aux(ctypes.byref(d), _p(x), _p(wfrag), _p(bias), _p(y), _p(stats), _p(aux), ldaux,
                                    _p(coef), mode, _s()), "ssa_conv2d_tile_aux")
    elif halo:
        check(L.ssa_conv2d_halo(ctypes.byref(d), _p(x), _p(wfrag), _p(bias), _p(y), _p(stats), _s()), "ssa_conv2d_halo")
    else:
        check(L.ssa_conv2d_tile(ctypes.byref(d), _p(x), _p(wfrag), _p(bias), _p(y), _p(stats), _s()), "ssa_conv2d_tile")
    return y


def _tile_conv_aux(d, x, wfrag, stats, aux, ldaux, coef, mode):
    return _tile_conv(d, x, wfrag, None, stats, aux=aux, ldaux=ldaux, coef=coef, mode=mode)


# ---- persistent halo-tile kernel (csrc/conv_tile_p.hip): the trunk's 48/96/192/384-channel 3x3 convs
_TILE_P_WGS = int(os.environ.get("SSA_TILE_P_WGS", "750"))      # most workgroups a grouped level may launch (< 3 per CU)

def tile_p_supported(d):
    return bool(lib().ssa_conv2d_tile_p_supported(ctypes.byref(d)))


def _tile_p_wgs(d, units):
    """Workgroups ssa_conv2d_tile_p launches for this problem at `units` (one 128-pixel tile x one 48-channel chunk
    of the input x one 32-channel n-block: 27 MFMAs per wave) per workgroup -- mirror of launch_p in
    csrc/conv_tile_p.hip."""
    tiles = d.B * ((d.W + 31) // 32) * ((d.H + 3) // 4)
    nb = (d.Cout + 31) // 32
    nchunk = d.Cin // 48
    tpw = max(1, units // nchunk)
    nstrips = -(-tiles // tpw)
    tpw = -(-tiles // nstrips)
    return -(-tiles // tpw) * nb


@contextlib.contextmanager
def tile_strip(descs):
    """Strip length of the persistent conv kernel for a level whose 3x3 problems are `descs` (those the kernel does
    not take count for nothing): the SHORTEST strips whose workgroups all fit on the chip at once (three per CU;
    a level of more workgroups than slots runs as two rounds: profiles/r03_notes.md call D, r04_notes.md call C).
    Handed to the library for the launches issued inside the bracket (same thread)."""
    ds = [d for d in descs if tile_p_supported(d)]
    if not ds:
        yield
        return
    units = 64
    for u in range(1, 65):
        if sum(_tile_p_wgs(d, u) for d in ds) <= _TILE_P_WGS:
            units = u
            break
    L = lib()
    L.ssa_conv_tile_strip(units)
    try:
        yield
    finally:
        L.ssa_conv_tile_strip(0)


# conv output data_ptr -> (BN partial sums [nrep][2][C], nrep): handed from the conv epilogue to
# the BatchNorm that consumes that output next
_PENDING_STATS = {}
_NO_IGEMM_STATS = bool(os.environ.get("SSA_NO_IGEMM_STATS"))   # debugging switch


def _conv_fwd(x, ldx, weight, b, stride, pad, dil, out_f32, want_stats):
    """One forward convolution launch (possibly queued in the open group bracket).
    x: dense-pixel [B,H,W,Cin] bf16.  Returns (y, stats or None)."""
    B, H, W, Cin = x.shape
    Cout, Cin_real, KH, KW = weight.shape
    assert Cin >= Cin_real and Cin % 8 == 0, (Cin, Cin_real)
    Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    if not out_f32:
        assert Cout % 8 == 0, "bf16 conv outputs need Cout % 8 == 0"
    td = _tile_desc(B, H, W, Cin, ldx, Cout, (KH, KW), stride, pad, dil, Ho, Wo, out_f32)
    al = x.data_ptr() % 16 == 0
    use_tile = al and tile_supported(td)
    use_wide = (not use_tile) and al and wide_supported(td)          # (the library's policy: csrc/conv_gemm_wide.hip)
    use_halo = (not use_tile) and (not use_wide) and al and halo_supported(td)
    stats = None
    if want_stats and not out_f32 and not (_NO_IGEMM_STATS and not (use_tile or use_halo or use_wide)):
        stats = _ARENA.take(stat_replicas() * 2 * Cout, x.device)
    if use_tile or use_halo or use_wide:
        wp, _ = _packed_filter(weight, 2, Cin, 0)
        y = _tile_conv(td, x, wp, b, stats, halo=use_halo, wide=use_wide)
    else:
        wp, Kpad = _packed_filter(weight, 0, Cin, 0)
        y = _igemm(x, ldx, (B, H, W, Cin), wp, Kpad, b, (Ho, Wo), Cout, (KH, KW), stride, pad, dil, False,
                   out_f32, stats=stats)
    if stats is not None:
        _PENDING_STATS[y.data_ptr()] = (stats, stat_replicas())
    return y, stats


def _dgrad_desc(x_shape, weight, dyb, lddy, cout_pad, stride, pad, dil, out_hw):
    B, H, W, Cin = x_shape
    Cout, Cin_real, KH, KW = weight.shape
    Ho, Wo = out_hw
    return _tile_desc(B, Ho, Wo, cout_pad, lddy, Cin, (KH, KW), stride, dil * (KH - 1) - pad, dil, H, W, False)


def _conv_dgrad(x_shape, weight, dyb, lddy, cout_pad, stride, pad, dil, out_hw, aux=None, ldaux=0, coef=None,
                mode=0, stats=None):
    """Data gradient of a forward conv (possibly queued).  mode 1/2: fused epilogue tile
    (only where dgrad_tile_ok())."""
    B, H, W, Cin = x_shape
    Cout, Cin_real, KH, KW = weight.shape
    Ho, Wo = out_hw
    assert Cin == Cin_real
    td = _dgrad_desc(x_shape, weight, dyb, lddy, cout_pad, stride, pad, dil, out_hw)
    al = dyb.data_ptr() % 16 == 0
    use_tile = al and tile_supported(td)
    use_wide = (not use_tile) and al and not mode and wide_supported(td)
    use_halo = (not use_tile) and (not use_wide) and al and halo_supported(td)
    if mode:
        assert use_tile
    if use_tile or use_halo or use_wide:
        wpt, _ = _packed_filter(weight, 3, 0, cout_pad)
        return _tile_conv(td, dyb, wpt, None, stats, halo=use_halo, aux=aux, ldaux=ldaux, coef=coef, mode=mode, wide=use_wide)
    if _DGRAD_S2 and stride == 2 and (KH, KW) == (3, 3) and pad == 1 and dil == 1 and Cin % 8 == 0 and al and \
            lddy % 8 == 0 and (Ho, Wo) == ((H - 1) // 2 + 1, (W - 1) // 2 + 1):
        return _dgrad_s2(x_shape, weight, dyb, lddy, cout_pad, out_hw)
    wpt, Kpad = _packed_filter(weight, 1, 0, cout_pad)
    return _igemm(dyb, lddy, (B, Ho, Wo, cout_pad), wpt, Kpad, None, (H, W), Cin, (KH, KW), stride,
                  dil * (KH - 1) - pad, dil, stride > 1, False)


_DGRAD_S2 = os.environ.get("SSA_DGRAD_S2", "1") != "0"


def _dgrad_s2(x_shape, weight, dyb, lddy, cout_pad, out_hw):
    """Data gradient of a 3x3 stride-2 pad-1 conv by output parity (ssa_conv2d_dgrad_s2): four dense
    sub-problems with 1, 2, 2, 4 taps instead of nine taps over the zero-inserted gradient."""
    B, H, W, Cin = x_shape
    Ho, Wo = out_hw
    Cout = weight.shape[0]
    packs = [_packed_filter(weight, 4 + c, 0, cout_pad) for c in range(4)]
    dx = torch.empty((B, H, W, Cin), dtype=ACT_DTYPE, device=dyb.device)
    wp = (ctypes.c_void_p * 4)(*[t.data_ptr() for t, _ in packs])
    kp = (ctypes.c_int * 4)(*[k for _, k in packs])
    _note(2.0 * B * Ho * Wo * Cout * Cin * 9, _conv_bytes(B * Ho * Wo, cout_pad, B * H * W, Cin, (3, 3)))
    check(lib().ssa_conv2d_dgrad_s2(B, H, W, Cin, Cin, Ho, Wo, cout_pad, lddy, _p(dyb), wp, kp, _p(dx), _s()),
          "ssa_conv2d_dgrad_s2")
    return dx


def dgrad_tile_ok(x_shape, weight, stride, pad, dil, out_hw):
    """True if the data gradient of this conv runs on the halo-tile kernel (whose epilogue can
    fold in the residual gradient / the BatchNorm backward sums)."""
    Cout = weight.shape[0]
    cp = _roundup(Cout, 8)
    td = _dgrad_desc(x_shape, weight, None, cp, cp, stride, pad, dil, out_hw)
    return x_shape[3] == weight.shape[1] and cp == Cout and tile_supported(td)


def _pack_matrix(src, R, C, ld, transpose, rows_out, Kpad):
    out = torch.empty((rows_out, Kpad), dtype=ACT_DTYPE, device=src.device)
    dt = 0 if src.dtype == ACT_DTYPE else 1
    check(lib().ssa_pack_matrix(_p(src), dt, R, C, ld, int(transpose), _p(out), rows_out, Kpad, _s()),
          "ssa_pack_matrix")
    return out


def _add_bf16(a, b):
    """a + b for two dense bf16 tensors of one shape (ssa_sum_act without the ReLU)."""
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(lib().ssa_sum_act(_p(a), _p(b), None, None, _p(out), out.numel(), 0, _s()), "ssa_sum_act")
    return out


# --------------------------------------------------------------------------
# weight gradients: deferred, grouped, accumulated into the gradient arena
# --------------------------------------------------------------------------
_WGRAD_TILE = os.environ.get("SSA_WGRAD_TILE", "1") != "0"       # halo-staged kernel for the trunk 3x3 convs
_WGRAD_STRIP = int(os.environ.get("SSA_WGRAD_STRIP", "8"))       # 128-pixel stages per workgroup in grouped launches
# queued layers that trigger a flush before the end of backward.  Measured (profiles/r02_notes.md, call X): 48 / 96 /
# 192 / end-of-backward-only = 31.0 / 30.0 / 29.4 / 29.1 ms per step -- the more layers a flush carries, the better its
# persistent-workgroup launches fill the chip; the queued (x, dy) pairs of a 1024x1024 step are a few GB of 288.
# On their own stream (below) the flushes overlap the rest of backward instead of delaying it, and early flushes are
# what gives the side stream something to run.  Measured (profiles/r03_notes.md, call S): stream off 23.30 ms; on with a
# flush every 16 / 32 / 64 / 128 / 256 layers / at the end only = 23.93 / 23.39 / 23.11 / 22.94 / 22.75 / 23.32 ms.
_WGRAD_SIDE = os.environ.get("SSA_WGRAD_STREAM", "1") != "0"
_BIAS_SIDE = os.environ.get("SSA_BIAS_GRAD_STREAM", "1") != "0"      # conv bias gradients there too (_bias_grad)
_WGRAD_FLUSH_AT = int(os.environ.get("SSA_WGRAD_FLUSH_AT", "256" if _WGRAD_SIDE else "100000"))
# (Measured and removed, profiles/r03_notes.md calls Y, Z: a decaying flush interval and an early first flush.)
# ... with a gradient sink installed (data parallel): flush every so many queued layers and exchange the completed arena
# range while backward goes on (a step queues ~640 layers: three exchanges, the last one short)
_DDP_FLUSH_AT = int(os.environ.get("SSA_DDP_FLUSH_AT", "256"))
_WGRAD_Q = []


class _WJob:
    __slots__ = ("x", "ldx", "geom_in", "dy", "lddy", "cout_pad", "geom_out", "k", "stride", "pad", "dil",
                 "Cout", "Cin_real", "target", "accumulate", "kind", "nsplit", "ws", "desc")


def _wgrad_job(x, ldx, geom_in, dy, lddy, cout_pad, geom_out, k, stride, pad, dil, Cout, Cin_real, target, accumulate):
    j = _WJob()
    j.x, j.ldx, j.geom_in, j.dy, j.lddy, j.cout_pad, j.geom_out = x, ldx, geom_in, dy, lddy, cout_pad, geom_out
    j.k, j.stride, j.pad, j.dil, j.Cout, j.Cin_real, j.target, j.accumulate = k, stride, pad, dil, Cout, Cin_real, target, accumulate
    return j


_WGRAD_FIT = os.environ.get("SSA_WGRAD_FIT", "1") != "0"
# workgroup slots a weight-gradient launch is fitted to, when the library does not say (ssa_conv2d_wgrad_tile_geometry):
# rounds 2-5 fitted to 512 -- two 64 KB workgroups per CU -- but the 4-wave tile kernel holds 467 registers per lane and
# gets ONE workgroup per CU (kernel-resource-usage report, round 6): a "full round" of 512 was two rounds of 256.
_WGRAD_SLOTS = int(os.environ.get("SSA_WGRAD_SLOTS", "256"))
_WGRAD_GROUP = 32       # layers (problems) per grouped weight-gradient launch: csrc/group.h MAXJOBS of ConvWgradTile


def _fit_tile_strips(jobs, strip):
    """Strip length (128-pixel tiles per workgroup) of the halo-staged weight-gradient launches, per launch: a grouped
    launch carries up to 32 layers of one instantiation (csrc/group.h MAXJOBS) and its workgroups are persistent, so a launch
    of 552 or 640 workgroups runs as a full round on the chip's 512 slots (two 64 KB workgroups per CU) plus a tail
    round of the same length -- 96-105 us where 480 workgroups take 76 (profiles/r04_notes.md).  For every launch
    pick the strip length in [strip, 8*strip] that minimises rounds x (strip + fixed cost); job -> strip."""
    out = {}
    if not _WGRAD_FIT or strip <= 0:
        return out
    groups = {}
    L = lib()
    for j in jobs:
        B, H, W, Cin = j.geom_in
        if j.k != (3, 3) or j.stride != 1 or j.dil != 1 or j.pad != 1 or Cin != j.cout_pad or Cin not in (48, 64, 96, 192, 384):
            continue
        # workgroups per strip, resident workgroups, instantiation: from the library (csrc/conv_wgrad_tile.hip make_plan)
        parts, slots, kind = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        if L.ssa_conv2d_wgrad_tile_geometry(Cin, j.cout_pad, ctypes.byref(parts), ctypes.byref(slots), ctypes.byref(kind)) != 0:
            continue
        if parts.value <= 0 or slots.value <= 0:        # (a stubbed library -- the dry-run harness -- fills nothing in)
            parts.value, slots.value, kind.value = {48: 1, 64: 1, 96: 2, 192: 8, 384: 32}[Cin], _WGRAD_SLOTS, min(Cin, 96)
        tiles = B * ((W + 31) // 32) * ((H + 3) // 4)
        groups.setdefault(kind.value, []).append((j, tiles, parts.value, min(slots.value, _WGRAD_SLOTS)))
    for lst in groups.values():
        for i in range(0, len(lst), _WGRAD_GROUP):
            chunk = lst[i:i + _WGRAD_GROUP]
            slots = chunk[0][3]
            best = None
            # (twice the layers per launch want strips up to twice as long for the same workgroup count: that is the
            # point -- a layer's fp32 partials, one block per workgroup, halve, and so does what WgradReduceK reads)
            for s_ in range(strip, 8 * strip + 1):
                wgs = sum(-(-t // s_) * p for _, t, p, _ in chunk)
                cost = -(-wgs // slots) * (s_ + 2.0)
                if best is None or cost < best[0] - 1e-9:
                    best = (cost, s_)
            for j, _, _, _ in chunk:
                out[id(j)] = best[1]
    return out


def _run_wgrad_jobs(jobs, strip):
    """Plan every job, launch all weight-gradient kernels in one bracket, all reduces in a second."""
    if not jobs:
        return
    L = lib()
    dev = jobs[0].x.device
    by_target = {}
    seq = {}
    for j in jobs:                          # the order the kernels are submitted in below: by parameter, passes together
        seq.setdefault(j.target.data_ptr(), []).append(j)
    fitted = _fit_tile_strips([j for js in seq.values() for j in js], strip)
    for j in jobs:
        B, H, W, Cin = j.geom_in
        Ho, Wo = j.geom_out
        d = ConvDesc(B, H, W, Cin, j.ldx, Ho, Wo, j.Cout, j.Cout, j.k[0], j.k[1], j.stride, j.pad, j.dil, 0, 0, 0, strip)
        nsplit, ws = ctypes.c_int(0), ctypes.c_size_t(0)
        al = j.x.data_ptr() % 16 == 0 and j.dy.data_ptr() % 16 == 0
        d.cfg = -1
        if al and L.ssa_conv2d_wgrad_head_plan(ctypes.byref(d), j.cout_pad, ctypes.byref(nsplit), ctypes.byref(ws)) == 0:
            j.kind = "head"
        else:
            d.cfg = fitted.get(id(j), strip)
            if _WGRAD_TILE and al and j.lddy % 8 == 0 and \
                    L.ssa_conv2d_wgrad_tile_plan(ctypes.byref(d), j.cout_pad, ctypes.byref(nsplit), ctypes.byref(ws)) == 0:
                j.kind = "tile"
            else:
                check(L.ssa_conv2d_wgrad_plan(ctypes.byref(d), j.cout_pad, ctypes.byref(nsplit), ctypes.byref(ws)),
                      "ssa_conv2d_wgrad_plan")
                j.kind = "tr"
        j.nsplit, j.ws, j.desc = nsplit.value, ws.value, d
        by_target.setdefault(j.target.data_ptr(), []).append(j)
    # one partial buffer per parameter: the passes' splits lie behind one another, ONE reduce sums them
    plan = []
    for js in by_target.values():
        partial = torch.empty((sum(j.ws for j in js) // 4,), dtype=torch.float32, device=dev)
        plan.append((js, partial))
    fns = {"head": (L.ssa_conv2d_wgrad_head, "ssa_conv2d_wgrad_head"), "tile": (L.ssa_conv2d_wgrad_tile, "ssa_conv2d_wgrad_tile"),
           "tr": (L.ssa_conv2d_wgrad, "ssa_conv2d_wgrad")}
    with group():
        for js, partial in plan:
            off = 0
            for j in js:
                B, H, W, Cin = j.geom_in
                Ho, Wo = j.geom_out
                _note(2.0 * B * Ho * Wo * j.Cout * j.Cin_real * j.k[0] * j.k[1],
                      2.0 * B * (H * W * Cin + Ho * Wo * j.Cout) + 4.0 * j.Cout * Cin * j.k[0] * j.k[1])
                fn, name = fns[j.kind]
                check(fn(ctypes.byref(j.desc), _p(j.x), _p(j.dy), j.lddy, j.cout_pad, j.nsplit,
                         ctypes.c_void_p(partial.data_ptr() + off), _s()), name)
                off += j.ws
    with group():
        for js, partial in plan:
            j0 = js[0]
            check(L.ssa_conv2d_wgrad_reduce(_p(partial), sum(j.nsplit for j in js), j0.cout_pad, j0.Cout, j0.geom_in[3],
                                            j0.Cin_real, j0.k[0], j0.k[1], _p(j0.target), int(j0.accumulate), _s()),
                  "ssa_conv2d_wgrad_reduce")


# Weight gradients on a stream of their own.  Nothing in backward waits for a weight gradient (they end in the
# gradient arena, read by the optimizer / the gradient exchange), while the chain that backward does wait for --
# BatchNorm reduce -> apply -> data gradient, level after level -- is made of 10-30 us launches that fill a fraction of
# the chip.  Flushed every SSA_WGRAD_FLUSH_AT layers onto a side stream (a parallel branch of the captured step,
# joined before the gradients are published), the 6 ms of weight-gradient kernels run in the chain's shadow.
_SIDE = {"stream": None, "pending": False}
_SINCE_REDUCE = [0]        # layers flushed since the gradient sink was last handed a range


def _side_stream():
    if _SIDE["stream"] is None:
        _SIDE["stream"] = torch.cuda.Stream()
    return _SIDE["stream"]


@contextlib.contextmanager
def _on_wgrad_stream():
    """Work that must be ordered behind the weight gradients issued so far (the gradient exchange of their range)."""
    if _SIDE["pending"]:
        with torch.cuda.stream(_SIDE["stream"]):
            yield
    else:
        yield


# A second forward stream for sub-graphs off the critical path (ops.fork): the upsampling half of an HRNet fuse level
# (1x1 convs -> BatchNorm -> bilinear, ~35 us of 10-20 us launches) next to the chain of stride-2 convs (~110 us) it is
# summed with.  autograd runs a node's backward on the stream of its forward, so the backward halves (bilinear backward
# -> BatchNorm backward -> 1x1 data gradient, ~55 us per level) run in parallel too, with the engine's own cross-stream
# waits.  Branches with a BatchNorm in them stay on the main stream when BatchNorm statistics are exchanged between
# ranks: the collectives of one communicator would otherwise be issued from two streams.  Users (SSA_FORK lists the
# enabled ones): "fuse" above, "loss" = the RMI term next to the three BCE-only terms of the training loss.  Measured
# (profiles/r06_notes.md, calls R, S): none 20.17, fuse 19.98, fuse + loss 19.72 ms per step; the OCR block's auxiliary
# head next to conv3x3_ocr +0.5 ms (two MFMA-bound GEMMs sharing the chip) -- not kept.
_FORK = {"stream": None}
_FORK_TAGS = set(t for t in os.environ.get("SSA_FORK", "fuse,loss,ocr").split(",") if t)      # "" = none


def _tensors_of(out):
    if torch.is_tensor(out):
        yield out
    elif isinstance(out, (list, tuple)):
        for o in out:
            yield from _tensors_of(o)


def fork(thunk, tag="fuse", has_bn=True):
    import torch.distributed as dist
    if tag not in _FORK_TAGS or not torch.cuda.is_available() or \
            (has_bn and dist.is_available() and dist.is_initialized()):
        out = thunk()
        return lambda: out
    main = torch.cuda.current_stream()
    if _FORK["stream"] is None:
        _FORK["stream"] = torch.cuda.Stream()
    side = _FORK["stream"]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = thunk()

    def join():
        cur = torch.cuda.current_stream()
        cur.wait_stream(side)
        for t in _tensors_of(out):
            if t.is_cuda:
                t.record_stream(cur)
        return out
    return join


def join_wgrads():
    """The current stream waits for the weight gradients issued so far."""
    if _SIDE["pending"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["pending"] = False


def flush_wgrads(join=False):
    """Issue the queued weight gradients (grouped) -- at the end of backward, when enough layers
    are queued, and (join=True) before anything reads a gradient slice."""
    if _WGRAD_Q:
        jobs = list(_WGRAD_Q)
        del _WGRAD_Q[:]
        if _WGRAD_SIDE and jobs[0].x.is_cuda:
            side = _side_stream()
            side.wait_stream(torch.cuda.current_stream())     # every operand has been produced
            with torch.cuda.stream(side):
                _run_wgrad_jobs(jobs, _WGRAD_STRIP)
            for j in jobs:                                    # their memory must not be recycled under the side stream
                for t in (j.x, j.dy):
                    t.record_stream(side)
            _SIDE["pending"] = True
        else:
            _run_wgrad_jobs(jobs, _WGRAD_STRIP)
    if join:
        join_wgrads()


def _wgrad(x, ldx, geom_in, dy, lddy, cout_pad, geom_out, k, stride, pad, dil, Cout, Cin_real, weight=None):
    """dW[Cout, Cin_real, KH, KW] fp32 = sum_p dy[p, co] * patch(x)[p, (kh,kw,ci)].
    weight = an nn.Parameter: queued, accumulated into its gradient-arena slice, returns None
    (the gradient is published at the end of backward).  Otherwise computed now and returned."""
    def job(target, accumulate):
        return _wgrad_job(x, ldx, geom_in, dy, lddy, cout_pad, geom_out, k, stride, pad, dil, Cout, Cin_real, target, accumulate)
    if weight is not None and _is_param(weight):
        _WGRAD_Q.append(job(_GRADS.slot(weight), True))
        n = len(_WGRAD_Q)
        exchange = _GRAD_SINK[0] is not None and n + _SINCE_REDUCE[0] >= _DDP_FLUSH_AT
        if n >= _WGRAD_FLUSH_AT or exchange:
            flush_wgrads()
            _SINCE_REDUCE[0] += n
            if exchange:
                with _on_wgrad_stream():
                    _GRADS.reduce_completed()
                _SINCE_REDUCE[0] = 0
        return None
    dw = torch.empty((Cout, Cin_real, k[0], k[1]), dtype=torch.float32, device=x.device)
    _run_wgrad_jobs([job(dw, False)], -1)
    return dw


def _grad_as_bf16(dy, Cout):
    """Incoming gradient -> bf16 [B,H,W,cout_pad] dense pixels."""
    cout_pad = _roundup(Cout, 8)
    if dy.dtype == ACT_DTYPE and cout_pad == Cout:
        dyb, ld = _pixels(dy)
        if ld % 8 or dyb.data_ptr() % 16:
            dyb, ld = dyb.contiguous(), Cout
        return dyb, ld, cout_pad
    dyf = dy.float()
    dyf, ldf = _pixels(dyf)
    B, H, W, _ = dyf.shape
    out = torch.empty((B, H, W, cout_pad), dtype=ACT_DTYPE, device=dy.device)
    check(lib().ssa_pad_cast_f32_bf16(_p(dyf), B * H * W, Cout, ldf, _p(out), cout_pad, _s()),
          "ssa_pad_cast_f32_bf16")
    return out, cout_pad, cout_pad


def _bias_grad(dyb, lddy, cout_pad, Cout, bias=None):
    """db[Cout] = column sums of the incoming gradient.  bias = an nn.Parameter: like a weight gradient nothing in
    backward waits for it -- computed on the weight-gradient stream, added into the parameter's gradient-arena slice,
    returns None (three dependent launches of ~50 us per head conv and scale pass leave the critical path)."""
    B, Ho, Wo, _ = dyb.shape

    def run():
        out = torch.empty((cout_pad,), dtype=torch.float32, device=dyb.device)
        scratch = torch.empty((2 * cout_pad,), dtype=torch.float64, device=dyb.device)
        check(lib().ssa_colsum_bf16(_p(dyb), B * Ho * Wo, cout_pad, lddy, _p(out), _p(scratch), _s()), "ssa_colsum_bf16")
        return out[:Cout]
    if bias is None or not (_is_param(bias) and _WGRAD_SIDE and _BIAS_SIDE and dyb.is_cuda):
        return run()
    slot = _GRADS.slot(bias)
    side = _side_stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        slot.add_(run())
    dyb.record_stream(side)
    _SIDE["pending"] = True
    return None


class ConvGroupFn(torch.autograd.Function):
    """nn.Conv2d forward/backward (groups=1) for N independent problems: one grouped launch per
    kernel instantiation.  spec[i] = (stride, pad, dil, out_f32, want_stats); tensors =
    (x_0, w_0, b_0, x_1, ...), x NHWC bf16, weight OIHW fp32, bias fp32 or None."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        n = len(spec)
        xs, lds, ws, bs = [], [], [], []
        for i in range(n):
            x, ldx = _pixels(tensors[3 * i])
            xs.append(x)
            lds.append(ldx)
            ws.append(tensors[3 * i + 1])
            b = tensors[3 * i + 2]
            if b is not None:
                b = b.detach()
                if b.dtype != torch.float32:
                    b = b.float()
            bs.append(b)
        ys = []
        with group():
            for i in range(n):
                stride, pad, dil, out_f32, want_stats = spec[i]
                y, _ = _conv_fwd(xs[i], lds[i], ws[i], bs[i], stride, pad, dil, out_f32, want_stats)
                ys.append(y)
        ctx.save_for_backward(*(xs + ws + [tensors[3 * i + 2] for i in range(n) if bs[i] is not None]))
        ctx.meta = (spec, lds, [b is not None for b in bs], [tuple(y.shape[1:3]) for y in ys])
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        spec, lds, has_bias, out_hw = ctx.meta
        n = len(spec)
        xs, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:2 * n]
        bias_of = dict(zip([i for i in range(n) if has_bias[i]], ctx.saved_tensors[2 * n:]))
        prep = [None] * n
        for i in range(n):
            if dys[i] is not None:
                prep[i] = _grad_as_bf16(dys[i], ws[i].shape[0])
        dxs = [None] * n
        with group():
            for i in range(n):
                if prep[i] is None or not ctx.needs_input_grad[1 + 3 * i]:
                    continue
                stride, pad, dil = spec[i][:3]
                dyb, lddy, cout_pad = prep[i]
                dxs[i] = _conv_dgrad(tuple(xs[i].shape), ws[i], dyb, lddy, cout_pad, stride, pad, dil, out_hw[i])
        grads = [None]
        for i in range(n):
            dw = db = None
            if prep[i] is not None:
                stride, pad, dil = spec[i][:3]
                dyb, lddy, cout_pad = prep[i]
                x, w = xs[i], ws[i]
                Cout, Cin_real, KH, KW = w.shape
                if ctx.needs_input_grad[2 + 3 * i]:
                    dw = _wgrad(x, lds[i], tuple(x.shape), dyb, lddy, cout_pad, out_hw[i], (KH, KW), stride, pad, dil,
                                Cout, Cin_real, weight=w)
                    if dw is not None and dw.dtype != w.dtype:
                        dw = dw.to(w.dtype)
                if has_bias[i] and ctx.needs_input_grad[3 + 3 * i]:
                    db = _bias_grad(dyb, lddy, cout_pad, Cout, bias_of[i])
            grads += [dxs[i], dw, db]
        return tuple(grads)


class Conv2dFn:
    """Single-problem form of ConvGroupFn (kept for the op-level tests and the heads)."""

    @staticmethod
    def apply(x, weight, bias, stride, pad, dil, out_f32, want_stats=False):
        return ConvGroupFn.apply(((stride, pad, dil, bool(out_f32), bool(want_stats)),), x, weight, bias)[0]


# --------------------------------------------------------------------------
# batch norm (+ residual add + ReLU + per-(b,c) post scale = Dropout2d mask)
# --------------------------------------------------------------------------
def _sync_world(sync):
    from .parallel import sync_world
    return sync_world(sync)


def _allreduce_sums(sums_list):
    """SyncBN exchange for every problem of a level: ONE all-reduce when the partial sums lie
    behind one another in the statistics arena (they do when one Function took them)."""
    from .parallel import allreduce_bn_sums
    spans = []
    for s in sums_list:
        if spans and spans[-1][0].data_ptr() + spans[-1][1] * 8 == s.data_ptr() and \
                spans[-1][0].untyped_storage().data_ptr() == s.untyped_storage().data_ptr():
            spans[-1][1] += s.numel()
        else:
            spans.append([s, s.numel()])
    for s, n in spans:
        flat = s if n == s.numel() else torch.as_strided(s, (n,), (1,))
        allreduce_bn_sums(flat)


class BnMeta:
    """Non-tensor description of one BatchNorm call (built by the operator surface)."""
    __slots__ = ("momentum", "eps", "training", "relu", "sync", "pass_stats", "running_mean", "running_var", "nbt", "out",
                 "infer")

    def __init__(self, momentum, eps, training, relu, sync, pass_stats, running_mean, running_var, nbt=None):
        # training: running_mean/var/nbt given = updated by the normalisation kernel itself (one pass
        # per layer and step only); pass_stats given = deferred to ssa_bn_update_running_batched
        self.momentum, self.eps, self.training, self.relu, self.sync = momentum, eps, training, relu, sync
        self.pass_stats, self.running_mean, self.running_var, self.nbt = pass_stats, running_mean, running_var, nbt
        self.out = None       # where z goes: a [B,H,W,C] channel slice of a wider buffer (ops.cat_slots), else a new tensor
        # built with autograd off (inside Function.forward grad mode is ALWAYS off, and needs_input_grad only says that
        # the parameters could take a gradient): nothing will ask this call for a backward pass
        self.infer = not torch.is_grad_enabled()


def _bn_out(m, shape, device):
    """(z, pixel stride of z) for a BatchNorm call: the caller's slot, or a fresh dense tensor."""
    if m.out is None:
        return torch.empty(shape, dtype=ACT_DTYPE, device=device), shape[3]
    z = m.out
    fits = tuple(z.shape) == tuple(shape) and z.dtype == ACT_DTYPE and z.stride(3) == 1 and z.stride(2) % 8 == 0 and \
        z.stride(1) == shape[2] * z.stride(2) and z.stride(0) == shape[1] * z.stride(1) and z.data_ptr() % 16 == 0
    if not fits:
        # a slot that does not fit the result (another model reusing the placement hint with other shapes): a fresh dense
        # tensor -- ops.cat then finds the operands NOT adjacent (adjacent_slices) and concatenates by copy
        return torch.empty(shape, dtype=ACT_DTYPE, device=device), shape[3]
    return z, z.stride(2)


def adjacent_slices(a, b):
    """a and b are neighbouring channel slices [.., :Ca] / [.., Ca:Ca+Cb] of one NHWC buffer whose pixels are exactly
    Ca + Cb wide (what ops.cat_slots hands out)."""
    if not (a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype and a.shape[:3] == b.shape[:3]):
        return False
    ld = a.shape[3] + b.shape[3]
    try:
        same = a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
    except Exception:       # noqa: BLE001
        return False
    return same and a.stride() == b.stride() and a.stride(3) == 1 and a.stride(2) == ld and \
        a.stride(1) == a.shape[2] * ld and a.stride(0) == a.shape[1] * a.shape[2] * ld and \
        b.storage_offset() == a.storage_offset() + a.shape[3]


class CatViewFn(torch.autograd.Function):
    """torch.cat((a, b), dim=3) for two tensors that already are neighbouring channel slices of one buffer: the dense
    view of that buffer, no copy; backward = the two slices of the gradient.

    READ-ONLY contract: the result ALIASES the storage of both operands (and of the BatchNorm outputs the kernels wrote
    into the slots through raw pointers); autograd does not know -- the version counters are not shared.  The result
    and the slots must therefore never be written in place (no in-place ReLU / dropout on them): that would corrupt
    tensors saved for backward without raising.  Its only consumer is the 1x1 conv of SpatialOCR_Module
    (network/ocr_utils.py:149-158), which reads it; ops.HipBackend.cat falls back to a copying concat for anything else."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.split = a.shape[3]
        B, H, W, Ca = a.shape
        ld = Ca + b.shape[3]
        return a.detach().as_strided((B, H, W, ld), (H * W * ld, W * ld, ld, 1), a.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return g[..., :ctx.split], g[..., ctx.split:]


def _bn_take_stats(jobs):
    """Per job (x, ldx): the batch sums [nrep][2][C] -- from the producing conv's epilogue, or by a
    (grouped) statistics pass.  Returns [(sums, nrep)]."""
    out = [None] * len(jobs)
    todo = []
    for i, (x, ldx) in enumerate(jobs):
        C = x.shape[3]
        pend = _PENDING_STATS.pop(x.data_ptr(), None)
        if pend is not None and pend[0].numel() >= pend[1] * 2 * C:
            out[i] = pend
        else:
            out[i] = (_ARENA.take(2 * C, x.device), 1)
            todo.append(i)
    if todo:
        with group():
            for i in todo:
                x, ldx = jobs[i]
                B, H, W, C = x.shape
                check(lib().ssa_bn_stats(_p(x), B * H * W, C, ldx, _p(out[i][0]), 0, _s()), "ssa_bn_stats")
    return out


_BN_SIGN_MASK = os.environ.get("SSA_BN_SIGN_MASK", "1") != "0"


def _bn_train_fwd(xs, ldxs, metas, gammas, betas, ress, posts, masks=None):
    """Training-mode normalisation of N problems: returns (zs, coefs, counts, worlds).
    masks: a list to receive, per problem, the sign bytes of z ([P, C/8] uint8) where the backward will want a ReLU mask
    it cannot recompute from x (ReLU behind a residual add or a Dropout2d mask), else None."""
    L = lib()
    n = len(xs)
    stats = _bn_take_stats(list(zip(xs, ldxs)))
    worlds = [_sync_world(m.sync) for m in metas]
    counts = [float(x.shape[0] * x.shape[1] * x.shape[2]) for x in xs]
    sync_ids = [i for i in range(n) if worlds[i]]
    if sync_ids:
        _allreduce_sums([stats[i][0] for i in sync_ids])
        for i in sync_ids:
            counts[i] *= worlds[i]
    zs, coefs = [], []
    with group():
        for i in range(n):
            x, m = xs[i], metas[i]
            B, H, W, C = x.shape
            P = B * H * W
            coef = torch.empty((4, C), dtype=torch.float32, device=x.device)  # scale, shift, mean, invstd
            z, ldz = _bn_out(m, (B, H, W, C), x.device)
            res, ldr = ress[i] if ress[i] is not None else (None, 0)
            mask = None
            if masks is not None and _BN_SIGN_MASK and m.relu and (res is not None or posts[i] is not None):
                mask = torch.empty((P, C // 8), dtype=torch.uint8, device=x.device)
            if masks is not None:
                masks.append(mask)
            _note(0.0, 2.0 * P * C * (2 + (1 if res is not None else 0)))
            check(L.ssa_bn_apply_train(_p(x), ldxs[i], _p(res), ldr, _p(z), ldz, P, C, _p(stats[i][0]), stats[i][1],
                                       counts[i], _p(gammas[i]), _p(betas[i]), _p(m.running_mean), _p(m.running_var),
                                       _p(m.nbt), float(m.momentum),
                                       float(m.eps), _p(coef), _p(m.pass_stats), int(m.relu), _p(posts[i]), H * W,
                                       _p(mask), _s()),
                  "ssa_bn_apply_train")
            zs.append(z)
            coefs.append(coef)
    return zs, coefs, counts, worlds


# OFF by default: measured (round 6, calls P / P2, tools/bnbench.py) 49 us for a 7.4 M-element level against 8.3 + 14.0 us
# for the two launches -- the rendezvous of ~930 workgroups across 8 XCDs (one returning atomic per workgroup on one
# address + the polls, all served at the memory-side coherence point) costs more than the second read of (x, dz, mask)
# it saves, which comes from the 256 MB MALL anyway.  SSA_BN_FUSED_BWD=1 switches it on; the kernel and its test stay.
_BN_FUSED_BWD = os.environ.get("SSA_BN_FUSED_BWD", "0") == "1"


def _bn_bwd_fused_ids(todo):
    """ids of the jobs of `todo` that take the one-launch backward: training-mode, no SyncBN exchange, one chunk per
    workgroup, and all of them together within the chip's resident-workgroup capacity (the rendezvous inside the kernel
    waits for every workgroup of a problem)."""
    if not _BN_FUSED_BWD or not todo:
        return set()
    L = lib()
    cap = L.ssa_bn_bwd_fused_capacity()
    total, ids = 0, set()
    for j in todo:
        B, H, W, C = j["x"].shape
        nb = L.ssa_bn_bwd_fused_blocks(B * H * W, C) if (j["training"] and not j["world"]) else 0
        if nb <= 0:
            continue
        if total + nb > cap:
            return set()            # (a bracket that does not fit stays on the two-launch form as a whole)
        total += nb
        ids.add(id(j))
    return ids


def _bn_bwd(jobs):
    """Backward of N BatchNorm(+ReLU/residual/mask) problems.  job: dict with x, ldx, dz, lddz, z,
    coef, g (gamma fp32 or None), gamma_param, beta_param, relu, pst, training, world, count,
    has_res, mask_from_x, sums (already accumulated by a conv epilogue, or None).
    Returns per job (dx, dres, dgamma, dbeta)."""
    L = lib()
    nrep_t = stat_replicas()
    todo = []
    for j in jobs:
        C = j["x"].shape[3]
        j["nrep"] = nrep_t if j["training"] else 1
        if j.get("sums") is None:
            j["sums"] = _ARENA.take(j["nrep"] * 2 * C, j["x"].device)
            todo.append(j)
    # the jobs whose sums nobody has formed yet go through ONE launch (reduce, grid-wide rendezvous, apply: the chunk
    # stays in registers, csrc/bn.hip bn_bwd_fused_body) when every workgroup of the bracket fits on the chip at once
    # and no SyncBN exchange has to happen between the two halves
    fused = _bn_bwd_fused_ids(todo)
    todo = [j for j in todo if id(j) not in fused]
    if todo:
        with group():
            for j in todo:
                x = j["x"]
                B, H, W, C = x.shape
                coef = j["coef"]
                msc, msh = (coef[0], coef[1]) if j["mask_from_x"] else (None, None)
                _note(0.0, 4.0 * B * H * W * C)
                check(L.ssa_bn_bwd_reduce(_p(x), j["ldx"], _p(j["dz"]), j["lddz"], _p(j["z"]), C, B * H * W, C, _p(coef[2]),
                                          _p(coef[3]), int(j["relu"]), _p(j["pst"]), H * W, _p(j["sums"]), j["nrep"], 0,
                                          _p(msc), _p(msh), _p(j.get("mask")), _s()), "ssa_bn_bwd_reduce")
    sync = [j for j in jobs if j["training"] and j["world"]]
    if sync:
        _allreduce_sums([j["sums"] for j in sync])
    out = []
    tails = []
    with group():
        for j in jobs:
            x = j["x"]
            B, H, W, C = x.shape
            P = B * H * W
            dev = x.device
            coef = j["coef"]
            msc, msh = (coef[0], coef[1]) if j["mask_from_x"] else (None, None)
            g = j["g"]
            pscale = 1.0 / j["world"] if (j["training"] and j["world"]) else 1.0
            use_sums = j["sums"]
            pg_g = pg_b = None
            ret_g = ret_b = None
            accumulate = 0
            if g is not None:
                gp, bp = j["gamma_param"], j["beta_param"]
                if gp is not None and _is_param(gp) and bp is not None and _is_param(bp):
                    pg_g, pg_b = _GRADS.slot(gp), _GRADS.slot(bp)
                    accumulate = 1
                else:
                    pg = torch.empty((2, C), dtype=torch.float32, device=dev)
                    pg_g, pg_b = pg[0], pg[1]
                    ret_g, ret_b = pg[0], pg[1]
            fuse_pg = pg_g is not None and j["training"]
            if not j["training"]:
                # eval-mode BN: statistics are constants -> no mean/projection terms, but the
                # parameter gradients still come from the reduced sums
                if pg_g is not None:
                    if accumulate:
                        tmp = torch.empty((2, C), dtype=torch.float32, device=dev)
                        tails.append((j["sums"], C, tmp, pg_g, pg_b))
                    else:
                        tails.append((j["sums"], C, None, pg_g, pg_b))
                use_sums = _ARENA.take(2 * C, dev)
            dx = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dev)
            dres = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dev) if j["has_res"] else None
            _note(0.0, 2.0 * P * C * (3 + (1 if dres is not None else 0)))
            if id(j) in fused:
                ticket = _ARENA.take(1, dev)
                check(L.ssa_bn_bwd_fused(_p(x), j["ldx"], _p(j["dz"]), j["lddz"], _p(j["z"]), C, _p(dx), C, _p(dres), C, P, C,
                                         _p(g), _p(coef[2]), _p(coef[3]), _p(use_sums), j["nrep"], j["count"], int(j["relu"]),
                                         _p(j["pst"]), H * W, _p(pg_g) if fuse_pg else None, _p(pg_b) if fuse_pg else None,
                                         pscale, _p(msc), _p(msh), accumulate if fuse_pg else 0, _p(j.get("mask")),
                                         _p(ticket), _s()), "ssa_bn_bwd_fused")
                out.append((dx, dres, ret_g, ret_b))
                continue
            check(L.ssa_bn_bwd_apply(_p(x), j["ldx"], _p(j["dz"]), j["lddz"], _p(j["z"]), C, _p(dx), C, _p(dres), C, P, C,
                                     _p(g), _p(coef[2]), _p(coef[3]), _p(use_sums), j["nrep"], j["count"], int(j["relu"]),
                                     _p(j["pst"]), H * W, _p(pg_g) if fuse_pg else None, _p(pg_b) if fuse_pg else None,
                                     pscale, _p(msc), _p(msh), accumulate if fuse_pg else 0, _p(j.get("mask")), _s()),
                  "ssa_bn_bwd_apply")
            out.append((dx, dres, ret_g, ret_b))
    for sums, C, tmp, pg_g, pg_b in tails:
        if tmp is None:
            check(L.ssa_bn_param_grads(_p(sums), C, _p(pg_g), _p(pg_b), _s()), "ssa_bn_param_grads")
        else:
            check(L.ssa_bn_param_grads(_p(sums), C, _p(tmp[0]), _p(tmp[1]), _s()), "ssa_bn_param_grads")
            pg_g.add_(tmp[0])
            pg_b.add_(tmp[1])
    return out


def _dz_bf16(dz, C):
    dz, lddz = _pixels(dz if dz.dtype == ACT_DTYPE else dz.to(ACT_DTYPE))
    if lddz % 8 or dz.data_ptr() % 16:
        dz, lddz = dz.contiguous(), C
    return dz, lddz


# --------------------------------------------------------------------------
# inference: conv -> BatchNorm(eval) [+ residual] [-> ReLU] as ONE launch (the trunk's 3x3 convs)
# --------------------------------------------------------------------------
_FUSE_EVAL_BN = os.environ.get("SSA_FUSE_EVAL_BN", "1") != "0"


def conv_bn_infer_group(convs, metas, gammas, betas, xs, ress, relus):
    """Inference (autograd off, BatchNorm in evaluation mode): normalisation, residual add and ReLU as the EPILOGUE of
    the conv kernel, for the problems of a grouped call that run on
      * ssa_conv2d_tile_p (3x3, stride 1, pad 1, no bias, 48 / 96 / 192 / 384 input channels: the HRNet trunk;
        aux_mode 3 / 4), or
      * the implicit-GEMM kernel (whatever the halo / wide kernels do not take: stem, layer1's 1x1 convs, the fuse
        layers' 1x1 and stride-2 convs; ssa_conv2d_igemm_affine).
    Both work on the 16-bit-rounded conv output with ssa_bn_apply's arithmetic: bit for bit the result of the two
    launches, whose conv output is then neither written nor read back.  Returns a list with the result per problem,
    None where the problem is not eligible (the caller runs conv and BatchNorm separately for those)."""
    n = len(xs)
    outs = [None] * n
    if not _FUSE_EVAL_BN:
        return outs
    jobs = []
    for i in range(n):
        conv, m = convs[i], metas[i]
        if m.training or not m.infer or getattr(conv, "groups", 1) != 1 or xs[i].dtype != ACT_DTYPE:
            continue
        x, ldx = _pixels(xs[i])
        B, H, W, Cin = x.shape
        w = conv.weight
        Cout, Cin_real, KH, KW = w.shape
        stride, pad, dil = conv.stride[0], conv.padding[0], conv.dilation[0]
        if Cin < Cin_real or Cin % 8 or Cout % 8 or x.data_ptr() % 16 or w.dtype != torch.float32:
            continue
        Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
        td = _tile_desc(B, H, W, Cin, ldx, Cout, (KH, KW), stride, pad, dil, Ho, Wo, False)
        on_tile_p = conv.bias is None and tile_p_supported(td)
        if not on_tile_p and (tile_supported(td) or wide_supported(td) or halo_supported(td)):
            continue                                    # conv_tile.hip / the head kernels: no such epilogue (yet)
        res, ldr = (None, 0)
        if ress[i] is not None:
            if ress[i].dtype != ACT_DTYPE or tuple(ress[i].shape) != (B, Ho, Wo, Cout):
                continue
            res, ldr = _pixels(ress[i])
            if ldr % 8 or res.data_ptr() % 16:
                continue
        g = gammas[i].detach() if gammas[i] is not None else None
        b = betas[i].detach() if betas[i] is not None else None
        coef = _BN_EVAL.get(m, g, b, Cout, x.device)
        if coef is None:
            continue
        bias = None
        if conv.bias is not None:
            bias = conv.bias.detach()
            if bias.dtype != torch.float32:
                continue
        jobs.append((i, on_tile_p, td, x, ldx, w, bias, res, ldr, coef, bool(relus[i]), (Ho, Wo), (KH, KW), stride, pad, dil))
    if not jobs:
        return outs
    with group(), tile_strip([j[2] for j in jobs if j[1]]):
        for i, on_tile_p, td, x, ldx, w, bias, res, ldr, coef, relu, ohw, k, stride, pad, dil in jobs:
            if on_tile_p:
                wp, _ = _packed_filter(w, 2, td.Cin, 0)
                outs[i] = _tile_conv(td, x, wp, None, None, aux=res, ldaux=ldr, coef=coef, mode=4 if relu else 3)
            else:
                wp, Kpad = _packed_filter(w, 0, td.Cin, 0)
                outs[i] = _igemm(x, ldx, (td.B, td.H, td.W, td.Cin), wp, Kpad, bias, ohw, td.Cout, k, stride, pad, dil,
                                 False, False, affine=(coef, res, ldr, relu))
    return outs


class BnActGroupFn(torch.autograd.Function):
    """z = post * act(bn(x) + residual) for N independent problems.
    metas[i]: BnMeta; tensors = (x_0, gamma_0, beta_0, residual_0, post_0, x_1, ...)."""

    @staticmethod
    def forward(ctx, metas, *tensors):
        L = lib()
        n = len(metas)
        xs, ldxs, gs, bs, ress, posts = [], [], [], [], [], []
        for i in range(n):
            x, gamma, beta, residual, post = tensors[5 * i:5 * i + 5]
            x, ldx = _pixels(x)
            xs.append(x)
            ldxs.append(ldx)
            gs.append(gamma.detach().float() if gamma is not None else None)
            bs.append(beta.detach().float() if beta is not None else None)
            ress.append(_pixels(residual) if residual is not None else None)
            posts.append(post.float().contiguous() if post is not None else None)
        training = [m.training for m in metas]
        assert all(training) or not any(training), "one grouped BatchNorm call mixes training and eval layers"
        masks = []
        if training[0]:
            zs, coefs, counts, worlds = _bn_train_fwd(xs, ldxs, metas, gs, bs, ress, posts, masks)
        else:
            masks = [None] * n
            zs, coefs, counts, worlds = [], [], [], [0] * n
            cached = all(m.infer for m in metas)     # (under autograd the coefficients are saved for a backward pass:
            for i in range(n):                       # a later refresh in place must not reach them)
                C = xs[i].shape[3]
                coef = _BN_EVAL.get(metas[i], gs[i], bs[i], C, xs[i].device) if cached else None
                if coef is None:
                    coef = torch.empty((4, C), dtype=torch.float32, device=xs[i].device)
                    check(L.ssa_bn_finalize(None, 1.0, C, _p(gs[i]), _p(bs[i]), _p(metas[i].running_mean),
                                            _p(metas[i].running_var), float(metas[i].momentum), float(metas[i].eps), 1,
                                            _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _s()), "ssa_bn_finalize")
                coefs.append(coef)
                counts.append(float(xs[i].shape[0] * xs[i].shape[1] * xs[i].shape[2]))
            with group():
                for i in range(n):
                    x = xs[i]
                    B, H, W, C = x.shape
                    z, ldz = _bn_out(metas[i], (B, H, W, C), x.device)
                    res, ldr = ress[i] if ress[i] is not None else (None, 0)
                    check(L.ssa_bn_apply(_p(x), ldxs[i], _p(res), ldr, _p(z), ldz, B * H * W, C, _p(coefs[i][0]),
                                         _p(coefs[i][1]), int(metas[i].relu), _p(posts[i]), H * W, _s()), "ssa_bn_apply")
                    zs.append(z)
        # BN + ReLU without residual / mask: the backward recomputes the ReLU mask from x with the
        # forward's own scale/shift, so z is neither kept for nor read by the backward
        saved, info = [], []
        for i in range(n):
            relu = metas[i].relu
            mask_from_x = relu and ress[i] is None and posts[i] is None
            # (the sign bytes stand in for z where they were written: z is then neither kept for nor read by the backward)
            saved += [xs[i], masks[i] if masks[i] is not None else (zs[i] if (relu and not mask_from_x) else None), gs[i],
                      coefs[i], posts[i]]
            info.append((ldxs[i], relu, metas[i].training, worlds[i], ress[i] is not None, counts[i], mask_from_x))
        ctx.save_for_backward(*saved)
        ctx.info = info
        ctx.params = [(tensors[5 * i + 1], tensors[5 * i + 2]) for i in range(n)]
        return tuple(zs)

    @staticmethod
    def backward(ctx, *dzs):
        info = ctx.info
        n = len(info)
        saved = ctx.saved_tensors
        jobs, idx = [], []
        for i in range(n):
            if dzs[i] is None:
                continue
            x, z, g, coef, pst = saved[5 * i:5 * i + 5]
            ldx, relu, training, world, has_res, count, mask_from_x = info[i]
            dz, lddz = _dz_bf16(dzs[i], x.shape[3])
            mask = z if (z is not None and z.dtype == torch.uint8) else None
            z = None if mask is not None else z
            jobs.append(dict(x=x, ldx=ldx, dz=dz, lddz=lddz, z=z, mask=mask, coef=coef, g=g, gamma_param=ctx.params[i][0],
                             beta_param=ctx.params[i][1], relu=relu, pst=pst, training=training, world=world,
                             count=count, has_res=has_res, mask_from_x=mask_from_x, sums=None))
            idx.append(i)
        res = _bn_bwd(jobs) if jobs else []
        grads = [None] * (1 + 5 * n)
        for (dx, dres, dg, db), i in zip(res, idx):
            grads[1 + 5 * i] = dx
            grads[2 + 5 * i] = dg
            grads[3 + 5 * i] = db
            grads[4 + 5 * i] = dres
        return tuple(grads)


class BatchNormActFn:
    """Single-problem form of BnActGroupFn (argument list of the first version, kept for the
    op-level tests)."""

    @staticmethod
    def apply(x, gamma, beta, residual, post, running_mean, running_var, nbt, momentum, eps, training, relu, sync,
              pass_stats=None):
        m = BnMeta(momentum, eps, training, relu, sync, pass_stats, running_mean, running_var, nbt)
        return BnActGroupFn.apply((m,), x, gamma, beta, residual, post)[0]


# --------------------------------------------------------------------------
# residual BasicBlock (network/hrnetv2.py:37-66) as ONE autograd node for N problems
# --------------------------------------------------------------------------
def _block_descs(x_shape, w):
    """Descriptors of a 3x3 stride-1 'same' conv of a residual block and of its data gradient."""
    B, H, W, Cin = x_shape
    Cout = w.shape[0]
    fwd = _tile_desc(B, H, W, Cin, Cin, Cout, (3, 3), 1, 1, 1, H, W, False)
    bwd = _tile_desc(B, H, W, Cout, Cout, Cin, (3, 3), 1, 1, 1, H, W, False)
    return fwd, bwd


class BasicBlockGroupFn(torch.autograd.Function):
    """out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x), training mode, 3x3 stride-1 convs without
    bias, for N independent problems (branches x scale passes); network/hrnetv2.py:37-66.

    forward  conv1 (+ bn1 statistics in its epilogue), bn1 + ReLU, conv2 (+ bn2 statistics), bn2 + residual + ReLU;
    backward bn2 reduce, bn2 apply (-> dy2 and g, the identity branch's gradient), conv2 data gradient whose epilogue
             accumulates bn1's backward sums, bn1 apply, conv1 data gradient whose epilogue adds g: five grouped launches,
             no bn1 reduce pass, no autograd add for the residual.
    (Round 3 also carried a form with bn1 folded into the convs' operand staging; it measured slower -- the transform
    lengthens latency-bound kernels -- and was removed in round 4.)
    metas[i] = (BnMeta bn1, BnMeta bn2); tensors = (x, w1, g1, b1, w2, g2, b2) per problem."""

    @staticmethod
    def forward(ctx, metas, *tensors):
        n = len(metas)
        T = [tensors[7 * i:7 * i + 7] for i in range(n)]
        xs, ldxs = [], []
        for i in range(n):
            x, ldx = _pixels(T[i][0])
            xs.append(x)
            ldxs.append(ldx)
        f32 = lambda t: t.detach().float()
        descs = [_block_descs(tuple(xs[i].shape), T[i][1])[0] for i in range(n)]
        y1s = []
        with tile_strip(descs), group():
            for i in range(n):
                y1s.append(_conv_fwd(xs[i], ldxs[i], T[i][1], None, 1, 1, 1, False, True)[0])
        a1s, coef1, cnt1, wd1 = _bn_train_fwd(y1s, [y.shape[3] for y in y1s], [m[0] for m in metas],
                                              [f32(T[i][2]) for i in range(n)], [f32(T[i][3]) for i in range(n)],
                                              [None] * n, [None] * n)
        y2s = []
        with tile_strip(descs), group():
            for i in range(n):
                y2s.append(_conv_fwd(a1s[i], a1s[i].shape[3], T[i][4], None, 1, 1, 1, False, True)[0])
        masks2 = []
        outs, coef2, cnt2, wd2 = _bn_train_fwd(y2s, [y.shape[3] for y in y2s], [m[1] for m in metas],
                                               [f32(T[i][5]) for i in range(n)], [f32(T[i][6]) for i in range(n)],
                                               [(xs[i], ldxs[i]) for i in range(n)], [None] * n, masks2)
        saved = []
        for i in range(n):
            # slot 4: the block output's sign bytes (the ReLU mask of bn2's backward), or the output itself
            saved += [xs[i], y1s[i], a1s[i], y2s[i], masks2[i] if masks2[i] is not None else outs[i], coef1[i], coef2[i],
                      T[i][1], T[i][4], f32(T[i][2]), f32(T[i][5])]
        ctx.save_for_backward(*saved)
        ctx.info = (ldxs, cnt1, wd1, cnt2, wd2)
        ctx.params = [(T[i][2], T[i][3], T[i][5], T[i][6]) for i in range(n)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        ldxs, cnt1, wd1, cnt2, wd2 = ctx.info
        n = len(ldxs)
        S = [ctx.saved_tensors[11 * i:11 * i + 11] for i in range(n)]
        act = [i for i in range(n) if douts[i] is not None]
        grads = [None] * (1 + 7 * n)
        if not act:
            return tuple(grads)
        descs_bwd = [_block_descs(tuple(S[i][0].shape), S[i][7])[1] for i in act]
        # ---- bn2 (+ residual add + ReLU)
        jobs2 = []
        for i in act:
            x, y1, a1, y2, out, c1, c2, w1, w2, g1, g2 = S[i]
            C = y2.shape[3]
            dz, lddz = _dz_bf16(douts[i], C)
            mask2 = out if out.dtype == torch.uint8 else None
            jobs2.append(dict(x=y2, ldx=C, dz=dz, lddz=lddz, z=None if mask2 is not None else out, mask=mask2, coef=c2,
                              g=g2, gamma_param=ctx.params[i][2],
                              beta_param=ctx.params[i][3], relu=True, pst=None, training=True, world=wd2[i],
                              count=cnt2[i], has_res=True, mask_from_x=False, sums=None))
        r2 = _bn_bwd(jobs2)          # (dy2, g, None, None)
        # ---- conv2 data gradient; its epilogue accumulates bn1's backward sums where it runs on the tile kernel
        da1 = {}
        sums1 = {}
        with tile_strip(descs_bwd), group():
            for k, i in enumerate(act):
                x, y1, a1, y2, out, c1, c2, w1, w2, g1, g2 = S[i]
                dy2 = r2[k][0]
                C = y2.shape[3]
                shp = tuple(y1.shape)
                if y1.data_ptr() % 16 == 0 and dgrad_tile_ok(shp, w2, 1, 1, 1, tuple(y2.shape[1:3])):
                    sums1[i] = _ARENA.take(stat_replicas() * 2 * shp[3], y1.device)
                    da1[i] = _conv_dgrad(shp, w2, dy2, C, C, 1, 1, 1, tuple(y2.shape[1:3]), aux=y1, ldaux=y1.shape[3],
                                         coef=c1, mode=2, stats=sums1[i])
                else:
                    sums1[i] = None
                    da1[i] = _conv_dgrad(shp, w2, dy2, C, C, 1, 1, 1, tuple(y2.shape[1:3]))
        need_dx = [i for i in act if ctx.needs_input_grad[1 + 7 * i]]
        dxs = {}
        r1 = {}
        # ---- bn1 (+ ReLU, mask recomputed from y1)
        jobs1 = []
        for i in act:
            x, y1, a1, y2, out, c1, c2, w1, w2, g1, g2 = S[i]
            C = y1.shape[3]
            jobs1.append(dict(x=y1, ldx=C, dz=da1[i], lddz=C, z=None, coef=c1, g=g1, gamma_param=ctx.params[i][0],
                              beta_param=ctx.params[i][1], relu=True, pst=None, training=True, world=wd1[i],
                              count=cnt1[i], has_res=False, mask_from_x=True, sums=sums1[i]))
        rb = _bn_bwd(jobs1)          # (dy1, None, dgamma, dbeta)
        dy1s = {}
        for k, i in enumerate(act):
            dy1s[i] = rb[k][0]
            r1[i] = (rb[k][2], rb[k][3])
        # ---- conv1 data gradient + the identity branch's gradient
        late_add = []
        with tile_strip(descs_bwd), group():
            for k, i in enumerate(act):
                if i not in need_dx:
                    continue
                x, y1, a1, y2, out, c1, c2, w1, w2, g1, g2 = S[i]
                dy1, gres = dy1s[i], r2[k][1]
                C = y1.shape[3]
                shp = tuple(x.shape)
                if gres.data_ptr() % 16 == 0 and dgrad_tile_ok(shp, w1, 1, 1, 1, tuple(y1.shape[1:3])):
                    dxs[i] = _conv_dgrad(shp, w1, dy1, C, C, 1, 1, 1, tuple(y1.shape[1:3]), aux=gres, ldaux=gres.shape[3],
                                         mode=1)
                else:
                    dxs[i] = _conv_dgrad(shp, w1, dy1, C, C, 1, 1, 1, tuple(y1.shape[1:3]))
                    late_add.append((i, gres))
        for i, gres in late_add:
            dxs[i] = _add_bf16(dxs[i], gres)
        # ---- weight gradients (queued)
        for k, i in enumerate(act):
            x, y1, a1, y2, out, c1, c2, w1, w2, g1, g2 = S[i]
            dy2 = r2[k][0]
            C2, C1 = y2.shape[3], y1.shape[3]
            dw2 = _wgrad(a1, a1.shape[3], tuple(a1.shape), dy2, C2, C2, tuple(y2.shape[1:3]), (3, 3), 1, 1, 1,
                         w2.shape[0], w2.shape[1], weight=w2)
            dw1 = _wgrad(x, ldxs[i], tuple(x.shape), dy1s[i], C1, C1, tuple(y1.shape[1:3]), (3, 3), 1, 1, 1,
                         w1.shape[0], w1.shape[1], weight=w1)
            base = 1 + 7 * i
            grads[base] = dxs.get(i)
            grads[base + 1] = dw1
            grads[base + 2], grads[base + 3] = r1[i]
            grads[base + 4] = dw2
            grads[base + 5], grads[base + 6] = r2[k][2], r2[k][3]
        return tuple(grads)


class SumActGroupFn(torch.autograd.Function):
    """z_i = relu(sum of up to 4 same-shape bf16 tensors) for N problems: HRNet fuse sums.
    counts[i] = number of terms of problem i; tensors = the terms, problem after problem."""

    @staticmethod
    def forward(ctx, relu, counts, *ts):
        zs, off = [], 0
        dense = [t.contiguous() for t in ts]      # before the bracket: a temporary must outlive the deferred launch
        with group():
            for c in counts:
                assert 1 <= c <= 4
                cs = dense[off:off + c]
                off += c
                z = torch.empty_like(cs[0])
                args = [_p(t) for t in cs] + [None] * (4 - c)
                check(lib().ssa_sum_act(args[0], args[1], args[2], args[3], _p(z), z.numel(), int(relu), _s()), "ssa_sum_act")
                zs.append(z)
        ctx.relu, ctx.counts = relu, counts
        if relu:
            ctx.save_for_backward(*zs)
        return tuple(zs)

    @staticmethod
    def backward(ctx, *dzs):
        gs = []
        if ctx.relu:
            zs = ctx.saved_tensors
            dzc = [None if d is None else d.contiguous() for d in dzs]
            with group():
                for z, dz in zip(zs, dzc):
                    if dz is None:
                        gs.append(None)
                        continue
                    g = torch.empty_like(z)
                    check(lib().ssa_relu_bwd(_p(dz), _p(z), _p(g), z.numel(), _s()), "ssa_relu_bwd")
                    gs.append(g)
        else:
            gs = list(dzs)
        out = [None, None]
        for g, c in zip(gs, ctx.counts):
            out += [g] * c
        return tuple(out)


class FanOutGroupFn(torch.autograd.Function):
    """Tensors with several consumers (a branch output feeds every row of the fuse layers,
    network/hrnetv2.py:236-252): forward hands out counts[i] aliases of tensor i, backward sums the
    consumers' gradients of ALL tensors with one grouped launch -- where autograd's own accumulation
    issues counts[i] - 1 separate adds per tensor (~125 launches per step)."""

    @staticmethod
    def forward(ctx, counts, *ts):
        ctx.counts = counts
        return tuple(t.view_as(t) for t, c in zip(ts, counts) for _ in range(c))

    @staticmethod
    def backward(ctx, *gs):
        out, off, jobs = [None], 0, []
        for c in ctx.counts:
            part = [g for g in gs[off:off + c] if g is not None]
            off += c
            if not part:
                out.append(None)
            elif len(part) == 1:
                out.append(part[0])
            else:
                part = [g if g.dtype == ACT_DTYPE else g.to(ACT_DTYPE) for g in part]
                jobs.append((len(out), [g.contiguous() for g in part]))
                out.append(None)
        while jobs:
            nxt = []
            with group():
                for pos, part in jobs:
                    head, rest = part[:4], part[4:]
                    z = torch.empty_like(head[0])
                    args = [_p(t) for t in head] + [None] * (4 - len(head))
                    check(lib().ssa_sum_act(args[0], args[1], args[2], args[3], _p(z), z.numel(), 0, _s()), "ssa_sum_act")
                    if rest:
                        nxt.append((pos, [z] + rest))
                    else:
                        out[pos] = z
            jobs = nxt
        return tuple(out)


class SumActFn:
    @staticmethod
    def apply(relu, *ts):
        return SumActGroupFn.apply(relu, (len(ts),), *ts)[0]


def _dt(t):
    if t.dtype == ACT_DTYPE:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise TypeError(t.dtype)


def _bilinear_bwd_group(jobs):
    """Backward of N bilinear resizes.  job = (dy_ptr, dy_dtype_code, lddy, B, Hi, Wi, C, Ho, Wo, dx).  Upsampling
    resizes (Ho >= 2 Hi and Wo >= 2 Wi) of 16-bit tensors with C % 8 == 0 -- the trunk's branch upsamples, 2x / 4x /
    8x -- run the separable form: all X passes of the level in one bracket, all Y passes in the next
    (ssa_bilinear_bwd_x / _y; 607 -> 275 us per step).  The others take the one-pass gather: for the 19-channel fp32
    logits the separable form measured SLOWER twice -- with per-element passes (scalar loads: 542 against 470 us per
    step, profiles/r04_notes.md) and with a row pass that stages gradient rows in LDS plus a flat float4 column pass
    (0.80 against 0.28 ms per step, profiles/r05_notes.md call G: 60 KB of LDS per workgroup and twelve candidate
    weights per element cost more than the gather's re-reads, which hit L2).  Round 6: ssa_bilinear_bwd itself routes
    those (fp32, 8..32 channels, upsampling) to ONE LDS-tiled launch -- both separable passes inside a 4 x 16-pixel input
    tile, the row pass with all of a thread's loads in flight (4 rows x 4 / 8 / 12 taps, no branch between them): 73 -> 32
    us at 256^2 <- 1024^2, 87 -> 45 us at 512^2 <- 1024^2 (profiles/r06_bilinbench.txt), -0.27 ms per step; the earlier
    forms lost because their loops ran one load at a time."""
    L = lib()

    def separable(j):
        return j[1] == 0 and j[6] % 8 == 0 and j[2] % 8 == 0 and j[7] >= 2 * j[4] and j[8] >= 2 * j[5]
    sep = [j for j in jobs if separable(j)]
    rest = [j for j in jobs if not separable(j)]
    tmps = []
    with group():
        for dyp, dt, lddy, B, Hi, Wi, C, Ho, Wo, dx in sep:
            tmp = torch.empty((B, Ho, Wi, C), dtype=torch.float32, device=dx.device)
            tmps.append(tmp)
            check(L.ssa_bilinear_bwd_x(dyp, dt, B, Ho, Wo, C, lddy, _p(tmp), Wi, _s()), "ssa_bilinear_bwd_x")
        for dyp, dt, lddy, B, Hi, Wi, C, Ho, Wo, dx in rest:
            check(L.ssa_bilinear_bwd(dyp, dt, B, Ho, Wo, C, lddy, _p(dx), _dt(dx), Hi, Wi, C, _s()), "ssa_bilinear_bwd")
    if sep:
        with group():
            for (dyp, dt, lddy, B, Hi, Wi, C, Ho, Wo, dx), tmp in zip(sep, tmps):
                check(L.ssa_bilinear_bwd_y(_p(tmp), B, Ho, Wi, C, _p(dx), _dt(dx), Hi, C, _s()), "ssa_bilinear_bwd_y")


class BilinearGroupFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=False) of N tensors; spec[i] = (Ho, Wo, out_f32)."""

    @staticmethod
    def forward(ctx, spec, *xs):
        prep = [_pixels(x) for x in xs]
        ys, meta = [], []
        with group():
            for (x, ldx), (Ho, Wo, out_f32) in zip(prep, spec):
                B, Hi, Wi, C = x.shape
                y = torch.empty((B, Ho, Wo, C), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
                check(lib().ssa_bilinear_fwd(_p(x), _dt(x), B, Hi, Wi, C, ldx, _p(y), _dt(y), Ho, Wo, C, _s()),
                      "ssa_bilinear_fwd")
                ys.append(y)
                meta.append((B, Hi, Wi, C, Ho, Wo, x.dtype))
        ctx.meta = meta
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        prep = []
        for dy, (B, Hi, Wi, C, Ho, Wo, in_dtype) in zip(dys, ctx.meta):
            if dy is None:
                prep.append(None)
                continue
            dy, lddy = _pixels(dy)
            if dy.dtype == ACT_DTYPE and (lddy % 8 or dy.data_ptr() % 16):
                dy, lddy = dy.contiguous(), C
            prep.append((dy, lddy))
        dxs = [None]
        jobs = []
        for pr, (B, Hi, Wi, C, Ho, Wo, in_dtype) in zip(prep, ctx.meta):
            if pr is None:
                dxs.append(None)
                continue
            dy, lddy = pr
            dx = torch.empty((B, Hi, Wi, C), dtype=in_dtype, device=dy.device)
            jobs.append((_p(dy), _dt(dy), lddy, B, Hi, Wi, C, Ho, Wo, dx))
            dxs.append(dx)
        _bilinear_bwd_group(jobs)
        return tuple(dxs)


class UpsampleCatGroupFn(torch.autograd.Function):
    """For each group g: cat([y0, up(y1), up(y2), ...], channels) with up = bilinear resize to y0's size
    (align_corners=False).  The resize kernel writes straight into its channel slice of the concatenated buffer
    (output leading dimension = total channels; y0 goes through the same kernel at scale 1, an exact copy) -- no
    upsampled temporaries, no torch.cat pass over the 720-channel tensor (0.25 ms per step at 1024x1024).  Backward:
    the resize's transposed kernel reads its slice of the incoming gradient in place; y0's gradient IS its slice."""

    @staticmethod
    def forward(ctx, counts, *xs):
        outs, meta, k = [], [], 0
        prep = [_pixels(x) for x in xs]
        with group():
            for n in counts:
                g = prep[k:k + n]
                B, Ho, Wo, _ = g[0][0].shape
                Ct = sum(x.shape[3] for x, _ in g)
                assert all(x.dtype == ACT_DTYPE for x, _ in g)
                y = torch.empty((B, Ho, Wo, Ct), dtype=ACT_DTYPE, device=g[0][0].device)
                off = 0
                for x, ldx in g:
                    _, Hi, Wi, C = x.shape
                    assert off % 8 == 0, "channel slices must stay 16-byte aligned"
                    check(lib().ssa_bilinear_fwd(_p(x), _dt(x), B, Hi, Wi, C, ldx, ctypes.c_void_p(y.data_ptr() + 2 * off),
                                                 _dt(y), Ho, Wo, Ct, _s()), "ssa_bilinear_fwd")
                    meta.append((B, Hi, Wi, C, Ho, Wo, off, Ct))
                    off += C
                outs.append(y)
                k += n
        ctx.counts, ctx.meta = counts, meta
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        dxs, k = [None], 0
        prep = []
        for dy in dys:
            if dy is None:
                prep.append(None)
                continue
            dy = dy if dy.dtype == ACT_DTYPE else dy.to(ACT_DTYPE)
            dy, lddy = _pixels(dy)
            if lddy % 8 or dy.data_ptr() % 16:
                dy, lddy = dy.contiguous(), dy.shape[3]
            prep.append((dy, lddy))
        jobs = []
        for gi, n in enumerate(ctx.counts):
            for j in range(n):
                B, Hi, Wi, C, Ho, Wo, off, Ct = ctx.meta[k + j]
                if prep[gi] is None:
                    dxs.append(None)
                    continue
                dy, lddy = prep[gi]
                if j == 0 and (Hi, Wi) == (Ho, Wo):
                    dxs.append(dy[..., off:off + C])      # identity resize: the gradient is the slice itself
                    continue
                dx = torch.empty((B, Hi, Wi, C), dtype=ACT_DTYPE, device=dy.device)
                jobs.append((ctypes.c_void_p(dy.data_ptr() + 2 * off), _dt(dy), lddy, B, Hi, Wi, C, Ho, Wo, dx))
                dxs.append(dx)
            k += n
        _bilinear_bwd_group(jobs)
        return tuple(dxs)


class BilinearFn:
    @staticmethod
    def apply(x, Ho, Wo, out_f32):
        return BilinearGroupFn.apply(((int(Ho), int(Wo), bool(out_f32)),), x)[0]


class MaxPool3x3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, stride=2, padding=1) on NHWC bf16 (ResNet stem)."""

    @staticmethod
    def forward(ctx, x):
        x, ldx = _pixels(x)
        B, H, W, C = x.shape
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        y = torch.empty((B, Ho, Wo, C), dtype=ACT_DTYPE, device=x.device)
        idx = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=x.device)
        check(lib().ssa_maxpool3x3s2_fwd(_p(x), ldx, B, H, W, C, _p(y), _p(idx), Ho, Wo, _s()), "ssa_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (B, H, W, C, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, H, W, C, Ho, Wo = ctx.meta
        dy = (dy if dy.dtype == ACT_DTYPE else dy.to(ACT_DTYPE)).contiguous()
        dx = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dy.device)
        check(lib().ssa_maxpool3x3s2_bwd(_p(dy), _p(idx), B, Ho, Wo, C, _p(dx), H, W, _s()), "ssa_maxpool3x3s2_bwd")
        return dx


class GlobalAvgPoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) on NHWC bf16: [B,H,W,C] -> [B,1,1,C]."""

    @staticmethod
    def forward(ctx, x):
        x, ldx = _pixels(x)
        B, H, W, C = x.shape
        out = torch.empty((B, 1, 1, C), dtype=ACT_DTYPE, device=x.device)
        check(lib().ssa_global_avg_pool_fwd(_p(x), ldx, B, H * W, C, _p(out), _s()), "ssa_global_avg_pool_fwd")
        ctx.meta = (B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, W, C = ctx.meta
        dout = (dout if dout.dtype == ACT_DTYPE else dout.to(ACT_DTYPE)).contiguous()
        dx = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dout.device)
        check(lib().ssa_global_avg_pool_bwd(_p(dout), B, H * W, C, _p(dx), _s()), "ssa_global_avg_pool_bwd")
        return dx


def image_to_nhwc(images, out_hw=None, cpad=16):
    """NCHW fp32 image batch -> NHWC bf16, optionally bilinearly resized."""
    images = images.detach()
    if images.dtype != torch.float32 or not images.is_contiguous():
        images = images.float().contiguous()
    B, C, H, W = images.shape
    Ho, Wo = out_hw if out_hw is not None else (H, W)
    y = torch.empty((B, Ho, Wo, cpad), dtype=ACT_DTYPE, device=images.device)
    check(lib().ssa_image_resize_to_nhwc_bf16(_p(images), B, C, H, W, _p(y), Ho, Wo, cpad, _s()),
          "ssa_image_resize_to_nhwc_bf16")
    return y


# --------------------------------------------------------------------------
# OCR
# --------------------------------------------------------------------------
class OcrGatherFn(torch.autograd.Function):
    """ctx[b,k,c] = sum_p softmax_HW(logits)[b,p,k] * feats[b,p,c].
    feats bf16 [B,H,W,C]; logits fp32 [B,H,W,K]; returns fp32 [B,K,C]."""

    @staticmethod
    def forward(ctx, feats, logits):
        L = lib()
        feats, ldf = _pixels(feats)
        logits, ldl = _pixels(logits.float())
        B, H, W, C = feats.shape
        K = logits.shape[3]
        Kp = _roundup(K, 32)
        HW = H * W
        dev = feats.device
        rowstat = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
        out = torch.empty((B, K, C), dtype=torch.float32, device=dev)
        for b in range(B):
            check(L.ssa_softmax_hw_stats(_p(logits[b]), ldl, HW, K, _p(rowstat[b]), _s()), "ssa_softmax_hw_stats")
            probs = torch.empty((HW, Kp), dtype=ACT_DTYPE, device=dev)
            check(L.ssa_softmax_hw_probs(_p(logits[b]), ldl, HW, K, _p(rowstat[b]), _p(probs), Kp, _s()),
                  "ssa_softmax_hw_probs")
            dw = _wgrad(feats[b], ldf, (1, H, W, C), probs, Kp, Kp, (H, W), (1, 1), 1, 0, 1, K, C)
            out[b] = dw.view(K, C)
        ctx.save_for_backward(feats, logits, rowstat, out)
        ctx.meta = (ldf, ldl)
        return out

    @staticmethod
    def backward(ctx, dctx):
        L = lib()
        feats, logits, rowstat, out = ctx.saved_tensors
        ldf, ldl = ctx.meta
        B, H, W, C = feats.shape
        K = logits.shape[3]
        Kp = _roundup(K, 32)
        HW = H * W
        dev = feats.device
        dctx = dctx.float().contiguous()
        dfeats = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dev) if ctx.needs_input_grad[0] else None
        dlogits = torch.empty((B, H, W, K), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        for b in range(B):
            if dfeats is not None:
                probs = torch.empty((HW, Kp), dtype=ACT_DTYPE, device=dev)
                check(L.ssa_softmax_hw_probs(_p(logits[b]), ldl, HW, K, _p(rowstat[b]), _p(probs), Kp, _s()),
                      "ssa_softmax_hw_probs")
                wp = _pack_matrix(dctx[b], K, C, C, True, C, Kp)          # [C][Kp]: dctx^T
                _igemm(probs, Kp, (1, H, W, Kp), wp, Kp, None, (H, W), C, (1, 1), 1, 0, 1, False, False, out=dfeats[b])
            if dlogits is not None:
                Cp = _roundup(C, 32)
                wp = _pack_matrix(dctx[b], K, C, C, False, K, Cp)         # [K][Cp]
                dprobs = _igemm(feats[b], ldf, (1, H, W, C), wp, Cp, None, (H, W), K, (1, 1), 1, 0, 1, False,
                                True)
                dot = torch.empty((K,), dtype=torch.float32, device=dev)
                check(L.ssa_rowdot_f32(_p(out[b]), _p(dctx[b]), K, C, _p(dot), _s()), "ssa_rowdot_f32")
                check(L.ssa_softmax_hw_bwd(_p(logits[b]), ldl, HW, K, _p(rowstat[b]), _p(dprobs), K, _p(dot),
                                           _p(dlogits[b]), K, 0, _s()), "ssa_softmax_hw_bwd")
        return dfeats, dlogits


_OCR_ATTN_FUSED = os.environ.get("SSA_OCR_ATTN_FUSED", "1") != "0"     # 0: the three-launch form (debugging)


class OcrAttnFn(torch.autograd.Function):
    """out = softmax_k(scale * q k^T) v per image (network/ocr_utils.py:100-113).  q [B,H,W,D] 16 bit; k,v [B,K,D].
    One launch per image (csrc/ocr_attn.hip: sim and probs stay in registers); backward = one launch for dq plus the
    two pixel reductions dv = probs^T dout, dk = dsim^T q on the weight-gradient kernel.  More than 96 object regions
    or D != 256: matmul -> softmax -> matmul on the implicit-GEMM kernel with sim (fp32) and probs in HBM."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        L = lib()
        q, ldq = _pixels(q)
        k = k.contiguous()
        v = v.contiguous()
        B, H, W, D = q.shape
        K = k.shape[1]
        Kp = _roundup(K, 32)
        Dp = _roundup(D, 32)
        dev = q.device
        out = torch.empty((B, H, W, D), dtype=ACT_DTYPE, device=dev)
        fused = _OCR_ATTN_FUSED and bool(L.ssa_ocr_attn_supported(K, D)) and ldq % 8 == 0 and q.data_ptr() % 16 == 0
        if fused:
            k = k if k.dtype == ACT_DTYPE else k.to(ACT_DTYPE)
            v = v if v.dtype == ACT_DTYPE else v.to(ACT_DTYPE)
            for b in range(B):
                _note(4.0 * H * W * K * D, 2.0 * H * W * 2 * D + 4.0 * K * D)
                check(L.ssa_ocr_attn_fwd(_p(q[b]), ldq, _p(k[b]), _p(v[b]), H * W, K, D, float(scale), _p(out[b]), D, _s()),
                      "ssa_ocr_attn_fwd")
            ctx.save_for_backward(q, k, v)
            ctx.meta = (ldq, float(scale), True)
            return out
        sim = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        for b in range(B):
            wk = _pack_matrix(k[b], K, D, D, False, K, Dp)                 # [K][Dp]
            _igemm(q[b], ldq, (1, H, W, D), wk, Dp, None, (H, W), K, (1, 1), 1, 0, 1, False, True, out=sim[b])
            probs = torch.empty((H * W, Kp), dtype=ACT_DTYPE, device=dev)
            check(L.ssa_softmax_lastdim_fwd(_p(sim[b]), K, H * W, K, float(scale), _p(probs), Kp, _s()),
                  "ssa_softmax_lastdim_fwd")
            wv = _pack_matrix(v[b], K, D, D, True, D, Kp)                  # [D][Kp]: v^T
            _igemm(probs, Kp, (1, H, W, Kp), wv, Kp, None, (H, W), D, (1, 1), 1, 0, 1, False, False, out=out[b])
        ctx.save_for_backward(q, k, v, sim)
        ctx.meta = (ldq, float(scale), False)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = lib()
        ldq, scale, fused = ctx.meta
        if fused:
            q, k, v = ctx.saved_tensors
            sim = None
        else:
            q, k, v, sim = ctx.saved_tensors
        B, H, W, D = q.shape
        K = k.shape[1]
        Kp = _roundup(K, 32)
        Dp = _roundup(D, 32)
        HW = H * W
        dev = q.device
        dout, lddo = _pixels(dout if dout.dtype == ACT_DTYPE else dout.to(ACT_DTYPE))
        if lddo % 8 or dout.data_ptr() % 16:
            dout, lddo = dout.contiguous(), D
        dq = torch.empty((B, H, W, D), dtype=ACT_DTYPE, device=dev)
        dk = torch.empty((B, K, D), dtype=torch.float32, device=dev)
        dv = torch.empty((B, K, D), dtype=torch.float32, device=dev)
        for b in range(B):
            probs = torch.empty((HW, Kp), dtype=ACT_DTYPE, device=dev)
            dsim = torch.empty((HW, Kp), dtype=ACT_DTYPE, device=dev)
            if fused:
                _note(8.0 * HW * K * D, 2.0 * HW * (3 * D + 2 * Kp) + 4.0 * K * D)
                check(L.ssa_ocr_attn_bwd(_p(q[b]), ldq, _p(k[b]), _p(v[b]), _p(dout[b]), lddo, HW, K, D, scale,
                                         _p(dq[b]), D, _p(probs), _p(dsim), _s()), "ssa_ocr_attn_bwd")
            else:
                check(L.ssa_softmax_lastdim_fwd(_p(sim[b]), K, HW, K, scale, _p(probs), Kp, _s()),
                      "ssa_softmax_lastdim_fwd")
                wv = _pack_matrix(v[b], K, D, D, False, K, Dp)                 # [K][Dp]
                dprobs = _igemm(dout[b], lddo, (1, H, W, D), wv, Dp, None, (H, W), K, (1, 1), 1, 0, 1, False, True)
                check(L.ssa_softmax_lastdim_bwd(_p(sim[b]), K, HW, K, scale, _p(dprobs), K, _p(dsim), Kp, _s()),
                      "ssa_softmax_lastdim_bwd")
                wk = _pack_matrix(k[b], K, D, D, True, D, Kp)                  # [D][Kp]: k^T
                _igemm(dsim, Kp, (1, H, W, Kp), wk, Kp, None, (H, W), D, (1, 1), 1, 0, 1, False, False, out=dq[b])
            dv[b] = _wgrad(dout[b], lddo, (1, H, W, D), probs, Kp, Kp, (H, W), (1, 1), 1, 0, 1, K, D).view(K, D)
            dk[b] = _wgrad(q[b], ldq, (1, H, W, D), dsim, Kp, Kp, (H, W), (1, 1), 1, 0, 1, K, D).view(K, D)
        return dq, dk.to(k.dtype), dv.to(v.dtype), None


# --------------------------------------------------------------------------
# scale-attention fusion pieces (fp32, [B,H,W,C] with a [B,H,W,1] attention map)
# --------------------------------------------------------------------------
class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.float().contiguous()
        y = torch.empty_like(x)
        check(lib().ssa_sigmoid_fwd(_p(x), _p(y), x.numel(), _s()), "ssa_sigmoid_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = torch.empty_like(y)
        check(lib().ssa_sigmoid_bwd(_p(y), _p(dy), _p(dx), y.numel(), _s()), "ssa_sigmoid_bwd")
        return dx


class BcastMulFn(torch.autograd.Function):
    """out[b,h,w,c] = a[b,h,w,0] * x[b,h,w,c]"""

    @staticmethod
    def forward(ctx, a, x):
        a = a.float().contiguous()
        x = x.float().contiguous()
        out = torch.empty_like(x)
        P, C = a.numel(), x.shape[-1]
        check(lib().ssa_bcast_mul_fwd(_p(a), _p(x), _p(out), P, C, _s()), "ssa_bcast_mul_fwd")
        ctx.save_for_backward(a, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, x = ctx.saved_tensors
        dout = dout.float().contiguous()
        da = torch.empty_like(a)
        dx = torch.empty_like(x)
        check(lib().ssa_bcast_mul_bwd(_p(a), _p(x), _p(dout), _p(da), _p(dx), a.numel(), x.shape[-1], _s()),
              "ssa_bcast_mul_bwd")
        return da, dx


class EwiseFn(torch.autograd.Function):
    """fp32 out = a (+|*|/) b for two same-shape tensors (attention-to-scale heads)."""
    OPS = {"add": 0, "mul": 1, "div": 2}

    @staticmethod
    def forward(ctx, op, a, b):
        a = a.float().contiguous()
        b = b.float().contiguous()
        assert a.shape == b.shape, (a.shape, b.shape)
        out = torch.empty_like(a)
        check(lib().ssa_ewise_f32(EwiseFn.OPS[op], _p(a), _p(b), _p(out), a.numel(), _s()), "ssa_ewise_f32")
        ctx.op = op
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        dout = dout.float().contiguous()
        da = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
        check(lib().ssa_ewise_bwd_f32(EwiseFn.OPS[ctx.op], _p(a), _p(b), _p(dout), _p(da), _p(db), a.numel(), _s()),
              "ssa_ewise_bwd_f32")
        return None, da, db


class AttnBlendFn(torch.autograd.Function):
    """joint = lo + (1 - a) * hi"""

    @staticmethod
    def forward(ctx, lo, a, hi):
        lo = lo.float().contiguous()
        a = a.float().contiguous()
        hi = hi.float().contiguous()
        out = torch.empty_like(hi)
        check(lib().ssa_attn_blend_fwd(_p(lo), _p(a), _p(hi), _p(out), a.numel(), hi.shape[-1], _s()),
              "ssa_attn_blend_fwd")
        ctx.save_for_backward(a, hi)
        return out

    @staticmethod
    def backward(ctx, dj):
        a, hi = ctx.saved_tensors
        dj = dj.float().contiguous()
        da = torch.empty_like(a)
        dhi = torch.empty_like(hi)
        check(lib().ssa_attn_blend_bwd(_p(a), _p(hi), _p(dj), _p(da), _p(dhi), a.numel(), hi.shape[-1], 0, _s()),
              "ssa_attn_blend_bwd")
        return dj, da, dhi


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        logits, ld = _pixels(logits.float())
        B, H, W, C = logits.shape
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        dev = logits.device
        acc = torch.empty((2,), dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad[0]
        dl = torch.empty((B, H, W, C), dtype=torch.float32, device=dev) if need else None
        check(lib().ssa_ce_fwd(_p(logits), ld, _p(labels), B * H * W, C, int(ignore_index), _p(acc), _p(dl), _s()),
              "ssa_ce_fwd")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        check(lib().ssa_loss_finalize(_p(acc), 0.0, _p(loss), _s()), "ssa_loss_finalize")
        ctx.save_for_backward(dl, acc)
        return loss

    @staticmethod
    def backward(ctx, up):
        dl, acc = ctx.saved_tensors
        up = up.float().contiguous()
        g = torch.empty_like(dl)
        check(lib().ssa_scale_grad_to(_p(dl), _p(g), g.numel(), _p(up), 1.0, _p(acc), 0.0, _s()), "ssa_scale_grad_to")
        return g, None, None


class BceRmiFn(torch.autograd.Function):
    """RMILoss.forward_sigmoid: masked BCE, optionally 0.5*bce + 0.5*rmi.  On dense logits the forward saves NO
    gradient: the backward recomputes (sigmoid - onehot) from the logits, already scaled (ssa_bce_bwd, or inside the RMI
    term's kernel) -- an 80 MB write per loss term at 1024 x 1024 x 19 and the pass that read it back are gone."""

    @staticmethod
    def forward(ctx, logits, labels, do_rmi, weight_lambda):
        L = lib()
        logits, ld = _pixels(logits.float())
        B, H, W, C = logits.shape
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.long()
        dev = logits.device
        acc = torch.empty((2,), dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad[0]
        n = B * H * W * C
        recompute = bool(need and ld == C and C >= 4 and n % 4 == 0 and n < (1 << 31) and logits.data_ptr() % 16 == 0)
        dl = torch.empty((B, H, W, C), dtype=torch.float32, device=dev) if (need and not recompute) else None
        check(L.ssa_bce_fwd(_p(logits), ld, _p(labels), B * H * W, C, _p(acc), _p(dl), _s()), "ssa_bce_fwd")
        bce = torch.empty((), dtype=torch.float32, device=dev)
        check(L.ssa_loss_finalize(_p(acc), 1.0, _p(bce), _s()), "ssa_loss_finalize")
        ctx.do_rmi = bool(do_rmi)
        ctx.lam = float(weight_lambda)
        ctx.recompute = recompute
        ctx.ld = ld
        if not do_rmi:
            ctx.save_for_backward(*((logits, labels, acc) if recompute else (dl, acc)))
            return bce
        Hp, Wp = H // 4 + 1, W // 4 + 1
        ppr = torch.empty((B * C, Hp, Wp), dtype=torch.float32, device=dev)
        pla = torch.empty((B * C, Hp, Wp), dtype=torch.float32, device=dev)
        check(L.ssa_rmi_pool(_p(logits), ld, _p(labels), B, H, W, C, _p(ppr), _p(pla), Hp, Wp, _s()), "ssa_rmi_pool")
        gram = torch.empty((B * C, 189), dtype=torch.float64, device=dev)
        check(L.ssa_rmi_gram(_p(ppr), _p(pla), B * C, Hp, Wp, _p(gram), _s()), "ssa_rmi_gram")
        loss_bc = torch.empty((B * C,), dtype=torch.float64, device=dev)
        gmat = torch.empty((B * C, 180), dtype=torch.float64, device=dev)
        check(L.ssa_rmi_solve(_p(gram), B * C, Hp, Wp, _p(loss_bc), _p(gmat), _s()), "ssa_rmi_solve")
        rmi = torch.empty((), dtype=torch.float32, device=dev)
        check(L.ssa_rmi_finalize(_p(loss_bc), B, C, _p(rmi), _s()), "ssa_rmi_finalize")
        ctx.save_for_backward(dl, acc, logits, labels, ppr, pla, gmat)
        return ctx.lam * bce + (1.0 - ctx.lam) * rmi

    @staticmethod
    def backward(ctx, up):
        L = lib()
        up = up.float().contiguous()
        if not ctx.do_rmi:
            if ctx.recompute:
                logits, labels, acc = ctx.saved_tensors
                B, H, W, C = logits.shape
                g = torch.empty_like(logits)
                check(L.ssa_bce_bwd(_p(logits), ctx.ld, _p(labels), B * H * W, C, _p(up), 1.0, _p(acc), 1.0, _p(g), _s()),
                      "ssa_bce_bwd")
                return g, None, None, None
            dl, acc = ctx.saved_tensors
            g = torch.empty_like(dl)
            check(L.ssa_scale_grad_to(_p(dl), _p(g), g.numel(), _p(up), 1.0, _p(acc), 1.0, _s()), "ssa_scale_grad_to")
            return g, None, None, None
        dl, acc, logits, labels, ppr, pla, gmat = ctx.saved_tensors
        B, H, W, C = logits.shape
        Hp, Wp = ppr.shape[1], ppr.shape[2]
        g = torch.empty_like(logits)
        dpool = torch.empty_like(ppr)
        check(L.ssa_rmi_bwd_pooled(_p(ppr), _p(pla), _p(gmat), B * C, Hp, Wp, _p(dpool), _s()), "ssa_rmi_bwd_pooled")
        coef = (1.0 - ctx.lam) / (9.0 * B)
        # lam * d(bce) + (1 - lam) * d(rmi) in ONE pass over the gradient: the BCE half from the saved un-normalised
        # gradient, or (dense logits: none was saved) recomputed from the logit the kernel holds anyway
        check(L.ssa_rmi_bwd_logits_bce(_p(logits), ctx.ld, _p(labels), B, H, W, C, _p(dpool), Hp, Wp, _p(up), coef,
                                       _p(dl), ctx.lam, _p(acc), 1.0, _p(g), _s()), "ssa_rmi_bwd_logits_bce")
        return g, None, None, None
