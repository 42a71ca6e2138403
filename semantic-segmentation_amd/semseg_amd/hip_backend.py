"""HIP backend: autograd.Function wrappers over the C ABI of libsemseg_hip.so.

Every tensor-valued op of the HRNet-OCR-MScale hot path goes through here; the
arithmetic lives in csrc/*.hip.  Activations are NHWC ([B,H,W,C]) bf16, logits
and losses fp32.  PyTorch provides device memory, streams and autograd
bookkeeping only.  Nothing here falls back to ATen math: a missing or stale
library raises in `_lib.lib()`.

hipGraph-capture safe: no host synchronisation, no `.item()`, allocations come
from torch's caching allocator, kernels are enqueued on the current stream.
"""
import ctypes
import math
import os

import torch
import torch.distributed as dist

from ._lib import lib, check, ConvDesc, WgradReduceJob

ACT_DTYPE = torch.bfloat16


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
# packed filters: persistent bf16 GEMM operands, one per (parameter, form).
# They are re-derived from the fp32 parameters whenever those changed (every
# optimizer step) by ONE batched launch at the start of the step -- captured in
# the step's hipGraph -- instead of 1,276 separate pack launches.
# --------------------------------------------------------------------------
import weakref

from ._lib import PackJob


class _Packed:
    __slots__ = ("wref", "out", "Kpad", "version", "job", "shape")


_PACKED = {}
_JOB_TABLE = {"key": None, "dev": None, "n": 0}


def clear_pack_cache():
    _PACKED.clear()
    _JOB_TABLE.update(key=None, dev=None, n=0)


def _make_job(w, out, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows):
    return PackJob(w.data_ptr(), out.data_ptr(), 0, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows, 0)


def refresh_packed_filters():
    """Re-pack every registered filter whose parameter changed since it was
    packed.  One launch for all of them."""
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
    tkey = tuple(k for k, _, _ in stale)
    if _JOB_TABLE["key"] != tkey:
        arr = (PackJob * len(stale))(*[e.job for _, e, _ in stale])
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        _JOB_TABLE.update(key=tkey, dev=host.to(stale[0][2].device), n=len(stale))
    check(lib().ssa_pack_filters_batched(_p(_JOB_TABLE["dev"]), _JOB_TABLE["n"], 32, _s()),
          "ssa_pack_filters_batched")
    for _, e, w in stale:
        e.version = w._version


def _packed_filter(weight, mode, cin_pad, cout_pad):
    key = (weight.data_ptr(), mode, cin_pad, cout_pad)
    e = _PACKED.get(key)
    # same storage + same version counter = same values: also true for the detached "shadow"
    # leaves the 0.5x pass runs on (MscaleOCR._shadow_parameters), which alias the parameter
    if e is not None and e.wref() is not None and e.version == weight._version and e.shape == tuple(weight.shape):
        return e.out, e.Kpad
    Cout, Cin, KH, KW = weight.shape
    if mode & 1 == 0:
        rows, kdim = Cout, KH * KW * cin_pad
    else:
        rows, kdim = Cin, KH * KW * cout_pad
    if mode < 2:
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
        e.out = torch.empty((rows, Kpad), dtype=ACT_DTYPE, device=weight.device)
        e.Kpad = Kpad
        e.wref = weakref.ref(weight)
        e.job = _make_job(w, e.out, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows)
        if direct:              # only parameters packed straight from their own storage are batched
            _PACKED[key] = e
    check(lib().ssa_pack_filter(_p(w), _p(e.out), Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, _s()),
          "ssa_pack_filter")
    e.version = weight._version
    return e.out, e.Kpad


# --------------------------------------------------------------------------
# fp64 statistics arena: BN (and bias-gradient) partial sums are carved out of
# one buffer that is cleared with a single memset per step instead of one
# memset per BatchNorm call (1,262 per training step).
# --------------------------------------------------------------------------
class _Arena:
    def __init__(self):
        self.buf = None
        self.cur = 0

    def reset(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.empty((1 << 22,), dtype=torch.float64, device=device)   # 32 MB
        self.buf.zero_()
        self.cur = 0

    def take(self, n, device):
        n = _roundup(n, 32)
        if self.buf is None or self.buf.device != device or self.cur + n > self.buf.numel():
            self.reset(device)
        out = self.buf[self.cur:self.cur + n]
        self.cur += n
        return out


_ARENA = _Arena()


# --------------------------------------------------------------------------
# deferred BatchNorm running-statistics updates (see ssa_bn_update_running_batched)
# --------------------------------------------------------------------------
from ._lib import BnUpdateJob


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


def end_forward():
    """Apply the running-statistics updates of every BatchNorm that ran in training
    mode since begin_step (one launch, passes in issue order)."""
    _BN_UPDATES.flush()


def begin_step(device=None):
    _BN_UPDATES.step = {}
    del _PENDING_REDUCES[:]                  # (left over only if a backward pass was aborted)
    _REDUCE_CALLBACK_QUEUED[0] = False
    refresh_packed_filters()
    if device is not None:
        _ARENA.reset(device)


def _pack_matrix(src, R, C, ld, transpose, rows_out, Kpad):
    out = torch.empty((rows_out, Kpad), dtype=ACT_DTYPE, device=src.device)
    dt = 0 if src.dtype == ACT_DTYPE else 1
    check(lib().ssa_pack_matrix(_p(src), dt, R, C, ld, int(transpose), _p(out), rows_out, Kpad, _s()),
          "ssa_pack_matrix")
    return out


# Optional per-launch timing (bench.py's roofline leg): a list that receives
# (kind, tile, flops, start_event, end_event) for every GEMM-class launch.
_PROFILE = None


def set_profile(store):
    global _PROFILE
    _PROFILE = store


def _igemm(x, ldx, geom_in, wp, Kpad, bias, geom_out, Cout, k, stride, pad, dil, transposed, out_f32,
           cfg=-1, stats=None):
    """Raw launch: x viewed as [B,H,W,Cin] (ldx) -> y [B,Ho,Wo,Cout]."""
    B, H, W, Cin = geom_in
    Ho, Wo = geom_out
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else ACT_DTYPE, device=x.device)
    d = ConvDesc(B, H, W, Cin, ldx, Ho, Wo, Cout, Cout, k[0], k[1], stride, pad, dil, int(transposed),
                 Kpad, int(out_f32), cfg)
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tile = lib().ssa_conv2d_igemm_tile(ctypes.byref(d))
        e0.record()
    check(lib().ssa_conv2d_igemm_stats(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(y), _p(stats), _s()),
          "ssa_conv2d_igemm")
    if _PROFILE is not None:
        e1.record()
        # algorithmic flops: taps that fall on the stride grid only (transposed) = forward flops
        taps = k[0] * k[1] / (stride * stride if transposed else 1)
        _PROFILE.append(("igemm", tile, 2.0 * B * Ho * Wo * Cout * Cin * taps, e0, e1, (k[0], stride, Cin, Cout, Ho, Wo)))
    return y


def _tile_desc(B, H, W, Cin, ldx, Cout, k, stride, pad, dil, Ho, Wo, out_f32):
    return ConvDesc(B, H, W, Cin, ldx, Ho, Wo, Cout, Cout, k[0], k[1], stride, pad, dil, 0, 0, int(out_f32), -1)


def tile_supported(d):
    return bool(lib().ssa_conv2d_tile_supported(ctypes.byref(d)))


_STAT_REPLICAS = None


def stat_replicas():
    global _STAT_REPLICAS
    if _STAT_REPLICAS is None:
        _STAT_REPLICAS = int(lib().ssa_bn_stat_replicas())
    return _STAT_REPLICAS


def halo_supported(d):
    return bool(lib().ssa_conv2d_halo_supported(ctypes.byref(d)))


def _tile_conv(d, x, wfrag, bias, stats, halo=False):
    """Halo-staged conv launch: conv_tile.hip (small-channel 3x3 convs) or
    conv_halo_gemm.hip (large-channel 3x3 / 1x1 convs), and their data gradients."""
    y = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=torch.float32 if d.out_f32 else ACT_DTYPE,
                    device=x.device)
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = lib().ssa_conv2d_halo if halo else lib().ssa_conv2d_tile
    check(fn(ctypes.byref(d), _p(x), _p(wfrag), _p(bias), _p(y), _p(stats), _s()),
          "ssa_conv2d_halo" if halo else "ssa_conv2d_tile")
    if _PROFILE is not None:
        e1.record()
        _PROFILE.append(("halo" if halo else "tile", 101 if halo else 100,
                         2.0 * d.B * d.Ho * d.Wo * d.Cout * d.Cin * d.KH * d.KW, e0, e1,
                         (d.KH, 1, d.Cin, d.Cout, d.Ho, d.Wo)))
    return y


# --------------------------------------------------------------------------
# Backward fusions (SSA_FUSE_BWD=1, off by default until they have run on hardware):
#   * the backward sums of a BatchNorm+ReLU layer whose output feeds ONE conv are accumulated in
#     the epilogue of that conv's data-gradient kernel (ssa_conv2d_tile_aux mode 2) instead of a
#     bn_bwd_reduce pass over (x, dz);
#   * the gradient of a residual block's identity branch is added in the epilogue of conv1's
#     data-gradient kernel (mode 1) instead of autograd's separate add.
# The modules say which tensors qualify (nn.conv_bn(private_input=, block=)); the links below
# carry the hand-over between the autograd Functions involved.
# --------------------------------------------------------------------------
_FUSE_BWD = os.environ.get("SSA_FUSE_BWD", "0") == "1"


class BnLink:
    """BatchNorm+ReLU layer -> the conv that alone consumes its output."""
    __slots__ = ("x", "ldx", "coef", "C", "sums", "dz_ptr")

    def __init__(self):
        self.x = self.coef = self.sums = self.dz_ptr = None
        self.ldx = self.C = 0


class ResLink:
    """Residual block: conv1 (first consumer of the block input) <-> bn2 (adds the block input)."""
    __slots__ = ("fused", "dres", "ld")

    def __init__(self):
        self.fused = False
        self.dres = None
        self.ld = 0


# side channels from the HipBackend wrappers to the next Function.forward (popped there)
_NEXT_BN_OUT_LINK = [None]      # BatchNormActFn: link to fill for the layer's consumer
_NEXT_CONV_IN_LINK = [None]     # Conv2dFn: link of the BN layer whose output is this conv's private input
_NEXT_CONV_RES_LINK = [None]    # Conv2dFn: residual block this conv is conv1 of
_NEXT_BN_RES_LINK = [None]      # BatchNormActFn: residual block this layer is bn2 of


def _tile_conv_aux(d, x, wfrag, stats, aux, ldaux, coef, mode):
    """conv_tile_aux.hip: the tile data gradient with the fused epilogue tile (see header)."""
    y = torch.empty((d.B, d.Ho, d.Wo, d.Cout), dtype=ACT_DTYPE, device=x.device)
    check(lib().ssa_conv2d_tile_aux(ctypes.byref(d), _p(x), _p(wfrag), None, _p(y), _p(stats), _p(aux), ldaux,
                                    _p(coef), mode, _s()), "ssa_conv2d_tile_aux")
    return y


def _add_bf16(a, b):
    """a + b for two dense bf16 tensors of one shape (ssa_sum_act without the ReLU)."""
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    check(lib().ssa_sum_act(_p(a), _p(b), None, None, _p(out), out.numel(), 0, _s()), "ssa_sum_act")
    return out


# (data_ptr of a conv output, its BN partial sums [nrep][2][C], nrep): handed from the conv
# epilogue to the BatchNorm that consumes that output next (HipBackend.conv_bn_act)
_PENDING_STATS = [None]
_NO_IGEMM_STATS = bool(os.environ.get("SSA_NO_IGEMM_STATS"))   # debugging switch


def _wgrad(x, ldx, geom_in, dy, lddy, cout_pad, geom_out, k, stride, pad, dil, Cout, Cin_real, deferrable=False):
    """dW[Cout, Cin_real, KH, KW] fp32 = sum_p dy[p, co] * patch(x)[p, (kh,kw,ci)].
    deferrable: the result is a parameter gradient that nobody reads before the end of backward
    (Conv2dFn.backward) -- its reduce may be batched with the others (SSA_DEFER_WGRAD_REDUCE)."""
    B, H, W, Cin = geom_in
    Ho, Wo = geom_out
    d = ConvDesc(B, H, W, Cin, ldx, Ho, Wo, Cout, Cout, k[0], k[1], stride, pad, dil, 0, 0, 0, -1)
    nsplit = ctypes.c_int(0)
    ws = ctypes.c_size_t(0)
    L = lib()
    # large-channel head convs: persistent 8-wave kernel (conv_wgrad_head.hip); everything else:
    # the K-pipelined kernel
    head = (x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and
            L.ssa_conv2d_wgrad_head_plan(ctypes.byref(d), cout_pad, ctypes.byref(nsplit), ctypes.byref(ws)) == 0)
    if not head:
        check(L.ssa_conv2d_wgrad_plan(ctypes.byref(d), cout_pad, ctypes.byref(nsplit), ctypes.byref(ws)),
              "ssa_conv2d_wgrad_plan")
    partial = torch.empty((ws.value // 4,), dtype=torch.float32, device=x.device)
    if _PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = L.ssa_conv2d_wgrad_head if head else L.ssa_conv2d_wgrad
    check(fn(ctypes.byref(d), _p(x), _p(dy), lddy, cout_pad, nsplit.value, _p(partial), _s()),
          "ssa_conv2d_wgrad_head" if head else "ssa_conv2d_wgrad")
    if _PROFILE is not None:
        e1.record()
        _PROFILE.append(("wgrad_head" if head else "wgrad", 102 if head else -1,
                         2.0 * B * Ho * Wo * Cout * Cin_real * k[0] * k[1], e0, e1,
                         (k[0], stride, Cin, Cout, Ho, Wo)))
    dw = torch.empty((Cout, Cin_real, k[0], k[1]), dtype=torch.float32, device=x.device)
    if deferrable and _DEFER_WGRAD_REDUCE and _defer_allowed():
        _defer_reduce(partial, dw, WgradReduceJob(partial.data_ptr(), dw.data_ptr(), nsplit.value, cout_pad, Cout,
                                                  Cin, Cin_real, k[0], k[1], 0))
        return dw
    check(L.ssa_conv2d_wgrad_reduce(_p(partial), nsplit.value, cout_pad, Cout, Cin, Cin_real, k[0], k[1],
                                    _p(dw), _s()), "ssa_conv2d_wgrad_reduce")
    return dw


# --------------------------------------------------------------------------
# Deferred weight-gradient reduces (SSA_DEFER_WGRAD_REDUCE=1, off by default until it has run on
# hardware): every conv's backward ends in a ~6 us reduce of its split-K partials, 641 per
# training step.  Nobody reads a weight gradient before backward has finished (autograd only
# stores the tensor), so the reduces are collected and issued at the end of backward in ~9
# launches (ssa_conv2d_wgrad_reduce_batched).  Not under torch.distributed: DDP's hooks read
# the gradients while backward is still running.
# --------------------------------------------------------------------------
_DEFER_WGRAD_REDUCE = os.environ.get("SSA_DEFER_WGRAD_REDUCE", "0") == "1"
_PENDING_REDUCES = []
_REDUCE_CALLBACK_QUEUED = [False]


def _defer_allowed():
    from .parallel import sync_world
    return not sync_world()


def _defer_reduce(partial, dw, job):
    _PENDING_REDUCES.append((partial, dw, job))
    if not _REDUCE_CALLBACK_QUEUED[0]:
        from torch.autograd import Variable
        Variable._execution_engine.queue_callback(flush_wgrad_reduces)
        _REDUCE_CALLBACK_QUEUED[0] = True


def flush_wgrad_reduces(side_streams=()):
    """Issue the deferred reduces on the current stream.  Runs as an end-of-backward callback;
    anything that reads weight gradients from another end-of-backward callback (the shadow
    gradient merge of MscaleOCR) calls it first -- a second call finds nothing to do."""
    _REDUCE_CALLBACK_QUEUED[0] = False
    if not _PENDING_REDUCES:
        return
    jobs = list(_PENDING_REDUCES)
    del _PENDING_REDUCES[:]
    on_gpu = jobs[0][0].is_cuda
    if on_gpu:
        main = torch.cuda.current_stream()
        for st in side_streams or _all_side_streams():
            main.wait_stream(st)             # partials of the other scale pass were produced there
    arr = (WgradReduceJob * len(jobs))(*[j for _, _, j in jobs])
    check(lib().ssa_conv2d_wgrad_reduce_batched(arr, len(jobs), _s()), "ssa_conv2d_wgrad_reduce_batched")
    if on_gpu:
        for partial, dw, _ in jobs:          # allocated on the producing stream, used on this one
            partial.record_stream(main)
            dw.record_stream(main)


def _all_side_streams():
    from . import ops
    return ops.backend().side_streams() if hasattr(ops.backend(), "side_streams") else []


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


class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d forward/backward (groups=1).  x NHWC bf16, weight OIHW fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, out_f32, want_stats=False):
        x, ldx = _pixels(x)
        B, H, W, Cin = x.shape
        Cout, Cin_real, KH, KW = weight.shape
        assert Cin >= Cin_real and Cin % 8 == 0, (Cin, Cin_real)
        Ho = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
        b = None
        if bias is not None:
            b = bias.detach()
            if b.dtype != torch.float32:
                b = b.float()
        if not out_f32:
            assert Cout % 8 == 0, "bf16 conv outputs need Cout % 8 == 0"
        td = _tile_desc(B, H, W, Cin, ldx, Cout, (KH, KW), stride, pad, dil, Ho, Wo, out_f32)
        use_tile = x.data_ptr() % 16 == 0 and tile_supported(td)
        use_halo = (not use_tile) and x.data_ptr() % 16 == 0 and halo_supported(td)
        if use_tile or use_halo:
            wp, _ = _packed_filter(weight, 2, Cin, 0)
            stats = None
            if want_stats and not out_f32:
                stats = _ARENA.take(stat_replicas() * 2 * Cout, x.device)
            y = _tile_conv(td, x, wp, b, stats, halo=use_halo)
            if stats is not None:
                _PENDING_STATS[0] = (y.data_ptr(), stats, stat_replicas())
        else:
            wp, Kpad = _packed_filter(weight, 0, Cin, 0)
            stats = None
            if want_stats and not out_f32 and not _NO_IGEMM_STATS:
                stats = _ARENA.take(stat_replicas() * 2 * Cout, x.device)
            y = _igemm(x, ldx, (B, H, W, Cin), wp, Kpad, b, (Ho, Wo), Cout, (KH, KW), stride, pad, dil, False,
                       out_f32, stats=stats)
            if stats is not None:
                _PENDING_STATS[0] = (y.data_ptr(), stats, stat_replicas())
        ctx.save_for_backward(x, weight)
        ctx.meta = (ldx, stride, pad, dil, bias is not None, (Ho, Wo))
        ctx.bn_link = ctx.res_link = None
        if _FUSE_BWD:
            ctx.bn_link, _NEXT_CONV_IN_LINK[0] = _NEXT_CONV_IN_LINK[0], None
            ctx.res_link, _NEXT_CONV_RES_LINK[0] = _NEXT_CONV_RES_LINK[0], None
            if ctx.res_link is not None:
                # bn2 hands the identity branch's gradient over only if this conv's data gradient
                # will run on the tile kernel (decided here, bn2's forward comes later)
                cp = _roundup(Cout, 8)
                tdg = _tile_desc(B, Ho, Wo, cp, cp, Cin, (KH, KW), stride, dil * (KH - 1) - pad, dil, H, W, False)
                ctx.res_link.fused = bool(Cin == Cin_real and tile_supported(tdg))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        ldx, stride, pad, dil, has_bias, (Ho, Wo) = ctx.meta
        B, H, W, Cin = x.shape
        Cout, Cin_real, KH, KW = weight.shape
        dyb, lddy, cout_pad = _grad_as_bf16(dy, Cout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            assert Cin == Cin_real
            td = _tile_desc(B, Ho, Wo, cout_pad, lddy, Cin, (KH, KW), stride, dil * (KH - 1) - pad, dil, H, W,
                            False)
            use_tile = dyb.data_ptr() % 16 == 0 and tile_supported(td)
            use_halo = (not use_tile) and dyb.data_ptr() % 16 == 0 and halo_supported(td)
            bn_link, res_link = ctx.bn_link, ctx.res_link
            dres = None
            if res_link is not None:
                dres, res_link.dres = res_link.dres, None
            if use_tile or use_halo:
                wpt, _ = _packed_filter(weight, 3, 0, cout_pad)
                if use_tile and bn_link is not None and bn_link.x is not None and dres is None and \
                        tuple(bn_link.x.shape) == (B, H, W, Cin) and bn_link.x.data_ptr() % 16 == 0:
                    sums = _ARENA.take(stat_replicas() * 2 * Cin, x.device)
                    dx = _tile_conv_aux(td, dyb, wpt, sums, bn_link.x, bn_link.ldx, bn_link.coef, 2)
                    bn_link.sums, bn_link.dz_ptr = sums, dx.data_ptr()
                elif use_tile and dres is not None and dres.data_ptr() % 16 == 0:
                    dx = _tile_conv_aux(td, dyb, wpt, None, dres, res_link.ld, None, 1)
                    dres = None
                else:
                    dx = _tile_conv(td, dyb, wpt, None, None, halo=use_halo)
            else:
                wpt, Kpad = _packed_filter(weight, 1, 0, cout_pad)
                dx = _igemm(dyb, lddy, (B, Ho, Wo, cout_pad), wpt, Kpad, None, (H, W), Cin, (KH, KW), stride,
                            dil * (KH - 1) - pad, dil, stride > 1, False)
            if dres is not None:          # handed over but not fused after all: add it here
                dx = _add_bf16(dx, dres)
        if ctx.needs_input_grad[1]:
            dw = _wgrad(x, ldx, (B, H, W, Cin), dyb, lddy, cout_pad, (Ho, Wo), (KH, KW), stride, pad, dil,
                        Cout, Cin_real, deferrable=weight.dtype == torch.float32)
            if dw.dtype != weight.dtype:
                dw = dw.to(weight.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            out = torch.empty((cout_pad,), dtype=torch.float32, device=x.device)
            scratch = torch.empty((2 * cout_pad,), dtype=torch.float64, device=x.device)
            check(lib().ssa_colsum_bf16(_p(dyb), B * Ho * Wo, cout_pad, lddy, _p(out), _p(scratch), _s()),
                  "ssa_colsum_bf16")
            db = out[:Cout]
        return dx, dw, db, None, None, None, None, None


# --------------------------------------------------------------------------
# batch norm (+ residual add + ReLU + per-(b,c) post scale = Dropout2d mask)
# --------------------------------------------------------------------------
def _sync_world(sync):
    from .parallel import sync_world
    return sync_world(sync)


class BatchNormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, post, running_mean, running_var, nbt, momentum, eps, training,
                relu, sync, pass_stats=None):
        L = lib()
        out_link = res_link = None
        if _FUSE_BWD:
            out_link, _NEXT_BN_OUT_LINK[0] = _NEXT_BN_OUT_LINK[0], None
            res_link, _NEXT_BN_RES_LINK[0] = _NEXT_BN_RES_LINK[0], None
        x, ldx = _pixels(x)
        B, H, W, C = x.shape
        P = B * H * W
        dev = x.device
        g = gamma.detach().float() if gamma is not None else None
        bta = beta.detach().float() if beta is not None else None
        coef = torch.empty((4, C), dtype=torch.float32, device=dev)  # scale, shift, mean, invstd
        world = _sync_world(sync) if training else 0
        count = float(P)
        res = ldr = None
        if residual is not None:
            res, ldr = _pixels(residual)
        z = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dev)
        pst = post.float().contiguous() if post is not None else None
        if training:
            pend, _PENDING_STATS[0] = _PENDING_STATS[0], None
            if pend is not None and pend[0] == x.data_ptr() and pend[1].numel() >= pend[2] * 2 * C:
                sums, nrep = pend[1], pend[2]          # accumulated by the producing conv's epilogue
            else:
                sums, nrep = _ARENA.take(2 * C, dev), 1
                check(L.ssa_bn_stats(_p(x), P, C, ldx, _p(sums), 0, _s()), "ssa_bn_stats")
            if world:
                from .parallel import allreduce_bn_sums
                count = allreduce_bn_sums(sums, P)
            check(L.ssa_bn_apply_train(_p(x), ldx, _p(res), ldr or 0, _p(z), C, P, C, _p(sums), nrep, count, _p(g),
                                       _p(bta), _p(running_mean), _p(running_var), _p(nbt), float(momentum),
                                       float(eps), _p(coef), _p(pass_stats), int(relu), _p(pst), H * W, _s()),
                  "ssa_bn_apply_train")
        else:
            check(L.ssa_bn_finalize(None, 1.0, C, _p(g), _p(bta), _p(running_mean), _p(running_var),
                                    float(momentum), float(eps), 1, _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                    _p(coef[3]), _s()), "ssa_bn_finalize")
            check(L.ssa_bn_apply(_p(x), ldx, _p(res), ldr or 0, _p(z), C, P, C, _p(coef[0]), _p(coef[1]),
                                 int(relu), _p(pst), H * W, _s()), "ssa_bn_apply")
        # BN + ReLU without residual / mask: the backward recomputes the ReLU mask from x with the
        # forward's own scale/shift, so z is neither kept for nor read by the backward
        mask_from_x = relu and residual is None and pst is None
        ctx.save_for_backward(x, z if (relu and not mask_from_x) else None, g, coef, pst)
        ctx.meta = (ldx, relu, training, world, residual is not None, count, mask_from_x)
        ctx.out_link = ctx.res_link = None
        if out_link is not None and training and mask_from_x:
            out_link.x, out_link.ldx, out_link.coef, out_link.C = x, ldx, coef, C
            ctx.out_link = out_link
        if res_link is not None and residual is not None and res_link.fused:
            ctx.res_link = res_link
        return z

    @staticmethod
    def backward(ctx, dz):
        L = lib()
        x, z, g, coef, pst = ctx.saved_tensors
        ldx, relu, training, world, has_res, count, mask_from_x = ctx.meta
        msc, msh = (coef[0], coef[1]) if mask_from_x else (None, None)
        B, H, W, C = x.shape
        P = B * H * W
        dev = x.device
        dz, lddz = _pixels(dz if dz.dtype == ACT_DTYPE else dz.to(ACT_DTYPE))
        if lddz % 8 or dz.data_ptr() % 16:
            dz, lddz = dz.contiguous(), C
        nrep = stat_replicas() if training else 1
        sums = None
        link = ctx.out_link
        if link is not None:
            if link.sums is not None and link.dz_ptr == dz.data_ptr() and lddz == C:
                sums = link.sums          # accumulated by the consuming conv's data-gradient epilogue
            link.sums = link.dz_ptr = None
        if sums is None:
            sums = _ARENA.take(nrep * 2 * C, dev)
            check(L.ssa_bn_bwd_reduce(_p(x), ldx, _p(dz), lddz, _p(z), C, P, C, _p(coef[2]), _p(coef[3]),
                                      int(relu), _p(pst), H * W, _p(sums), nrep, 0, _p(msc), _p(msh), _s()),
                  "ssa_bn_bwd_reduce")
        pg = torch.empty((2, C), dtype=torch.float32, device=dev) if g is not None else None
        pscale = 1.0
        use_sums = sums
        if training:
            if world:
                from .parallel import allreduce_bn_sums
                allreduce_bn_sums(sums, P)
                pscale = 1.0 / world
        else:
            # eval-mode BN: statistics are constants -> no mean/projection terms, but
            # the parameter gradients still come from the reduced sums
            if pg is not None:
                check(L.ssa_bn_param_grads(_p(sums), C, _p(pg[0]), _p(pg[1]), _s()), "ssa_bn_param_grads")
            use_sums = _ARENA.take(2 * C, dev)
        dx = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dev)
        dres = torch.empty((B, H, W, C), dtype=ACT_DTYPE, device=dev) if has_res else None
        fuse_pg = pg is not None and training
        check(L.ssa_bn_bwd_apply(_p(x), ldx, _p(dz), lddz, _p(z), C, _p(dx), C, _p(dres), C, P, C, _p(g),
                                 _p(coef[2]), _p(coef[3]), _p(use_sums), nrep, count, int(relu), _p(pst), H * W,
                                 _p(pg[0]) if fuse_pg else None, _p(pg[1]) if fuse_pg else None, pscale,
                                 _p(msc), _p(msh), _s()),
              "ssa_bn_bwd_apply")
        dgamma, dbeta = (pg[0], pg[1]) if pg is not None else (None, None)
        if ctx.res_link is not None and dres is not None:
            # conv1 of this block adds the identity branch's gradient in its data-gradient epilogue
            ctx.res_link.dres, ctx.res_link.ld = dres, C
            dres = None
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None, None, None


class SumActFn(torch.autograd.Function):
    """z = relu(sum of up to 4 same-shape bf16 tensors): HRNet fuse sum."""

    @staticmethod
    def forward(ctx, relu, *ts):
        assert 1 <= len(ts) <= 4
        cs = [t.contiguous() for t in ts]
        z = torch.empty_like(cs[0])
        n = z.numel()
        args = [_p(c) for c in cs] + [None] * (4 - len(cs))
        check(lib().ssa_sum_act(args[0], args[1], args[2], args[3], _p(z), n, int(relu), _s()), "ssa_sum_act")
        ctx.relu = relu
        ctx.n_in = len(ts)
        if relu:
            ctx.save_for_backward(z)
        return z

    @staticmethod
    def backward(ctx, dz):
        dz = dz.contiguous()
        if ctx.relu:
            (z,) = ctx.saved_tensors
            g = torch.empty_like(z)
            check(lib().ssa_relu_bwd(_p(dz), _p(z), _p(g), z.numel(), _s()), "ssa_relu_bwd")
        else:
            g = dz
        return (None,) + (g,) * ctx.n_in


def _dt(t):
    if t.dtype == ACT_DTYPE:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise TypeError(t.dtype)


class BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, out_f32):
        x, ldx = _pixels(x)
        B, Hi, Wi, C = x.shape
        y = torch.empty((B, Ho, Wo, C), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
        check(lib().ssa_bilinear_fwd(_p(x), _dt(x), B, Hi, Wi, C, ldx, _p(y), _dt(y), Ho, Wo, C, _s()),
              "ssa_bilinear_fwd")
        ctx.meta = (B, Hi, Wi, C, Ho, Wo, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Hi, Wi, C, Ho, Wo, in_dtype = ctx.meta
        dy, lddy = _pixels(dy)
        if dy.dtype == ACT_DTYPE and (lddy % 8 or dy.data_ptr() % 16):
            dy, lddy = dy.contiguous(), C
        dx = torch.empty((B, Hi, Wi, C), dtype=in_dtype, device=dy.device)
        check(lib().ssa_bilinear_bwd(_p(dy), _dt(dy), B, Ho, Wo, C, lddy, _p(dx), _dt(dx), Hi, Wi, C, _s()),
              "ssa_bilinear_bwd")
        return dx, None, None, None


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
                dfeats[b] = _igemm(probs, Kp, (1, H, W, Kp), wp, Kp, None, (H, W), C, (1, 1), 1, 0, 1, False,
                                   False)[0]
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


class OcrAttnFn(torch.autograd.Function):
    """out = softmax_k(scale * q k^T) v per image.  q [B,H,W,D] bf16; k,v [B,K,D] bf16."""

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
        sim = torch.empty((B, H, W, K), dtype=torch.float32, device=dev)
        for b in range(B):
            wk = _pack_matrix(k[b], K, D, D, False, K, Dp)                 # [K][Dp]
            sim[b] = _igemm(q[b], ldq, (1, H, W, D), wk, Dp, None, (H, W), K, (1, 1), 1, 0, 1, False, True)[0]
            probs = torch.empty((H * W, Kp), dtype=ACT_DTYPE, device=dev)
            check(L.ssa_softmax_lastdim_fwd(_p(sim[b]), K, H * W, K, float(scale), _p(probs), Kp, _s()),
                  "ssa_softmax_lastdim_fwd")
            wv = _pack_matrix(v[b], K, D, D, True, D, Kp)                  # [D][Kp]: v^T
            out[b] = _igemm(probs, Kp, (1, H, W, Kp), wv, Kp, None, (H, W), D, (1, 1), 1, 0, 1, False, False)[0]
        ctx.save_for_backward(q, k, v, sim)
        ctx.meta = (ldq, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        L = lib()
        q, k, v, sim = ctx.saved_tensors
        ldq, scale = ctx.meta
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
            check(L.ssa_softmax_lastdim_fwd(_p(sim[b]), K, HW, K, scale, _p(probs), Kp, _s()),
                  "ssa_softmax_lastdim_fwd")
            wv = _pack_matrix(v[b], K, D, D, False, K, Dp)                 # [K][Dp]
            dprobs = _igemm(dout[b], lddo, (1, H, W, D), wv, Dp, None, (H, W), K, (1, 1), 1, 0, 1, False, True)
            dv[b] = _wgrad(dout[b], lddo, (1, H, W, D), probs, Kp, Kp, (H, W), (1, 1), 1, 0, 1, K, D).view(K, D)
            dsim = torch.empty((HW, Kp), dtype=ACT_DTYPE, device=dev)
            check(L.ssa_softmax_lastdim_bwd(_p(sim[b]), K, HW, K, scale, _p(dprobs), K, _p(dsim), Kp, _s()),
                  "ssa_softmax_lastdim_bwd")
            wk = _pack_matrix(k[b], K, D, D, True, D, Kp)                  # [D][Kp]: k^T
            dq[b] = _igemm(dsim, Kp, (1, H, W, Kp), wk, Kp, None, (H, W), D, (1, 1), 1, 0, 1, False, False)[0]
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
        g = dl.clone()
        check(lib().ssa_scale_grad(_p(g), g.numel(), _p(up), 1.0, _p(acc), 0.0, _s()), "ssa_scale_grad")
        return g, None, None


class BceRmiFn(torch.autograd.Function):
    """RMILoss.forward_sigmoid: masked BCE, optionally 0.5*bce + 0.5*rmi."""

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
        dl = torch.empty((B, H, W, C), dtype=torch.float32, device=dev) if need else None
        check(L.ssa_bce_fwd(_p(logits), ld, _p(labels), B * H * W, C, _p(acc), _p(dl), _s()), "ssa_bce_fwd")
        bce = torch.empty((), dtype=torch.float32, device=dev)
        check(L.ssa_loss_finalize(_p(acc), 1.0, _p(bce), _s()), "ssa_loss_finalize")
        ctx.do_rmi = bool(do_rmi)
        ctx.lam = float(weight_lambda)
        if not do_rmi:
            ctx.save_for_backward(dl, acc)
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
        ctx.ld = ld
        return ctx.lam * bce + (1.0 - ctx.lam) * rmi

    @staticmethod
    def backward(ctx, up):
        L = lib()
        up = up.float().contiguous()
        if not ctx.do_rmi:
            dl, acc = ctx.saved_tensors
            g = dl.clone()
            check(L.ssa_scale_grad(_p(g), g.numel(), _p(up), 1.0, _p(acc), 1.0, _s()), "ssa_scale_grad")
            return g, None, None, None
        dl, acc, logits, labels, ppr, pla, gmat = ctx.saved_tensors
        B, H, W, C = logits.shape
        Hp, Wp = ppr.shape[1], ppr.shape[2]
        g = dl.clone()
        check(L.ssa_scale_grad(_p(g), g.numel(), _p(up), ctx.lam, _p(acc), 1.0, _s()), "ssa_scale_grad")
        dpool = torch.empty_like(ppr)
        check(L.ssa_rmi_bwd_pooled(_p(ppr), _p(pla), _p(gmat), B * C, Hp, Wp, _p(dpool), _s()), "ssa_rmi_bwd_pooled")
        coef = (1.0 - ctx.lam) / (9.0 * B)
        check(L.ssa_rmi_bwd_logits(_p(logits), ctx.ld, _p(labels), B, H, W, C, _p(dpool), Hp, Wp, _p(up), coef,
                                   _p(g), 1, _s()), "ssa_rmi_bwd_logits")
        return g, None, None, None
