"""Operator surface used by semseg_amd.network / semseg_amd.loss.

All functions take and return NHWC tensors ([B,H,W,C]); activations in the
backend's `act_dtype` (bf16 on the HIP backend), logits/losses fp32.  The
default -- and only shipped -- backend is the HIP one (hip_backend.py over
libsemseg_hip.so); it is created on first use and raises if the library is
missing.  `_set_backend_for_tests` exists so the CPU test-suite can check the
module wiring against the reference with the oracle's operators; product code
never calls it.

Multi-problem calls.  The hot ops (conv, conv+BN, BasicBlock, fuse sum, bilinear)
accept a LIST of independent problems wherever they accept a tensor -- the resolution
branches of a HighResolutionModule times the scale passes of MscaleOCR -- and return a
list.  `BackendBase` maps such a call over the single-problem primitives; the HIP
backend overrides them with grouped launches (one launch per kernel instantiation for
the whole list, csrc/group.h).
"""
import torch

_BACKEND = None


def _is_list(v):
    return isinstance(v, (list, tuple))


def _lst(v, n):
    return list(v) if _is_list(v) else [v] * n


class BackendBase:
    """List-aware front of the operator surface over single-problem primitives
    (`_conv2d`, `_conv_bn_act`, `_batch_norm_act`, `_sum_act`, `_bilinear`)."""

    def group(self):
        import contextlib
        return contextlib.nullcontext()

    def conv2d(self, x, weight, bias, stride, padding, dilation, out_f32=False, want_stats=False):
        if not _is_list(x):
            return self._conv2d(x, weight, bias, stride, padding, dilation, out_f32)
        n = len(x)
        return [self._conv2d(xi, w, b, stride, padding, dilation, out_f32)
                for xi, w, b in zip(x, _lst(weight, n), _lst(bias, n))]

    def cat_slots(self, like, widths):
        """Output placement for a later `cat`: None here (the result is concatenated by a copy); the HIP backend hands
        out channel slices of ONE buffer per problem of `like`, which `conv_bn_act(..., out=)` writes and `cat` then
        recognises -- the concatenation costs nothing (SpatialOCR_Module: 2 x 134 MB per step at 1024 x 1024)."""
        return None

    def conv_bn_act(self, conv, bn, x, residual=None, relu=False, post=None, out=None):
        assert out is None, "output placement is the HIP backend's (cat_slots returns None here)"
        if not _is_list(x):
            return self._conv_bn_act(conv, bn, x, residual, relu, post)
        n = len(x)
        return [self._conv_bn_act(c, b, xi, r, rl, po) for c, b, xi, r, rl, po in
                zip(_lst(conv, n), _lst(bn, n), x, _lst(residual, n), _lst(relu, n), _lst(post, n))]

    def batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
        if not _is_list(x):
            return self._batch_norm_act(x, bn, residual, relu, post)
        n = len(x)
        return [self._batch_norm_act(xi, b, r, rl, po) for xi, b, r, rl, po in
                zip(x, _lst(bn, n), _lst(residual, n), _lst(relu, n), _lst(post, n))]

    def basic_block(self, blocks, xs):
        """conv3x3-BN-ReLU-conv3x3-BN-(+x)-ReLU (network/hrnetv2.py:37-66, no downsample branch)
        of blocks[i] on xs[i]."""
        mid = [self._conv_bn_act(b.conv1, b.bn1, x, None, True, None) for b, x in zip(blocks, xs)]
        return [self._conv_bn_act(b.conv2, b.bn2, m, x, True, None) for b, m, x in zip(blocks, mid, xs)]

    def sum_act(self, tensors, relu=True):
        if tensors and _is_list(tensors[0]):
            return [self._sum_act(t, relu) for t in tensors]
        return self._sum_act(tensors, relu)

    def bilinear(self, x, size, out_f32=False):
        if not _is_list(x):
            return self._bilinear(x, size, out_f32)
        sizes = size if _is_list(size[0]) or hasattr(size[0], "__len__") else [size] * len(x)
        return [self._bilinear(xi, s, out_f32) for xi, s in zip(x, sizes)]

    def upsample_cat(self, groups):
        """groups[p] = [y0, y1, ...] (NHWC): y1.. resized to y0's size (bilinear, align_corners=False) and all of them
        concatenated on channels -- the head input of HighResolutionNet.forward (network/hrnetv2.py:418-431)."""
        sizes = [tuple(g[0].shape[1:3]) for g in groups]
        low = [(p, i) for p, g in enumerate(groups) for i in range(1, len(g))]
        ups = self.bilinear([groups[p][i] for p, i in low], [sizes[p] for p, i in low]) if low else []
        up = dict(zip(low, ups))
        return [self.cat([g[0]] + [up[(p, i)] for i in range(1, len(g))]) for p, g in enumerate(groups)]

    def fan_out(self, tensors, counts):
        """tensors[i] is about to be consumed counts[i] times: returns counts[i] handles per tensor (the
        HIP backend sums their gradients with one grouped launch instead of autograd's adds)."""
        return [[t] * c for t, c in zip(tensors, counts)]

    def fork(self, thunk, tag="fuse", has_bn=True):
        """A sub-graph nothing on the caller's path needs until `join()`: returns join, a callable that hands back
        thunk's result.  Here it simply ran; the HIP backend issues it on a second stream (a parallel branch of the
        captured step) and `join()` makes the current stream wait for it."""
        out = thunk()
        return lambda: out

    @staticmethod
    def parallel(thunks, level=1):
        """Independent sub-graphs, issued one after the other (thunks[1:] first, thunks[0] last:
        the reference runs the low-scale pass before the 1.0x pass, and BatchNorm running
        statistics are updated in issue order)."""
        outs = [None] * len(thunks)
        for i in range(1, len(thunks)):
            outs[i] = thunks[i]()
        outs[0] = thunks[0]()
        return outs


class HipBackend(BackendBase):
    name = "hip"

    def __init__(self):
        from . import hip_backend as hb
        hb.lib()   # fail loudly now if the extension is absent
        self.hb = hb
        self.act_dtype = hb.ACT_DTYPE

    def begin_step(self, device=None):
        self.hb.begin_step(device)

    def group(self):
        return self.hb.group()

    def image_to_nhwc(self, images, out_hw=None):
        return self.hb.image_to_nhwc(images, out_hw)

    # -- convolution -----------------------------------------------------------------
    def conv2d(self, x, weight, bias, stride, padding, dilation, out_f32=False, want_stats=False):
        multi = _is_list(x)
        xs = list(x) if multi else [x]
        n = len(xs)
        ws, bs = _lst(weight, n), _lst(bias, n)
        spec = tuple((stride, padding, dilation, bool(out_f32), bool(want_stats)) for _ in range(n))
        flat = []
        for xi, w, b in zip(xs, ws, bs):
            flat += [xi, w, b]
        ys = self.hb.ConvGroupFn.apply(spec, *flat)
        return list(ys) if multi else ys[0]

    def _bn_meta(self, bn, relu):
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        # training: the running statistics are updated by end_forward() (deferred, one launch, in
        # issue order: the passes over one layer are problems of the same grouped launch)
        return self.hb.BnMeta(0.1 if bn.momentum is None else bn.momentum, bn.eps, bool(bn.training), bool(relu),
                              getattr(bn, "sync", False), self.hb._BN_UPDATES.slot(bn) if track else None,
                              None if bn.training else bn.running_mean, None if bn.training else bn.running_var)

    def batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
        multi = _is_list(x)
        xs = list(x) if multi else [x]
        n = len(xs)
        zs = self._bn_group(xs, _lst(bn, n), _lst(residual, n), _lst(relu, n), _lst(post, n))
        return zs if multi else zs[0]

    def _bn_group(self, xs, bns, ress, relus, posts, outs=None):
        metas = tuple(self._bn_meta(b, r) for b, r in zip(bns, relus))
        if outs is not None:
            for m, o in zip(metas, outs):
                m.out = o
        flat = []
        for xi, b, r, po in zip(xs, bns, ress, posts):
            flat += [xi, b.weight, b.bias, r, po]
        return list(self.hb.BnActGroupFn.apply(metas, *flat))

    def conv_bn_act(self, conv, bn, x, residual=None, relu=False, post=None, out=None):
        """conv -> BatchNorm (+residual, ReLU, mask).  In training the conv epilogue accumulates
        the batch statistics, and the normalisation picks them up instead of re-reading the
        conv output.  Lists = independent problems (grouped launches).  out: where the result goes (a `cat_slots`
        channel slice per problem) instead of a fresh tensor."""
        multi = _is_list(x)
        xs = list(x) if multi else [x]
        outs = None if out is None else (list(out) if multi else [out])
        zs = self._conv_bn_group(_lst(conv, len(xs)), _lst(bn, len(xs)), xs, residual, relu, post, outs)
        return zs if multi else zs[0]

    def cat_slots(self, like, widths):
        multi = _is_list(like)
        slots = []
        for t in (like if multi else [like]):
            B, H, W = t.shape[:3]
            buf = torch.empty((B, H, W, sum(widths)), dtype=self.act_dtype, device=t.device)
            views, off = [], 0
            for w in widths:
                views.append(buf[..., off:off + w])
                off += w
            slots.append(views)
        return slots if multi else slots[0]

    def _conv_bn_group(self, convs, bns, xs, residual=None, relu=False, post=None, outs=None):
        n = len(xs)
        ress, relus, posts = _lst(residual, n), _lst(relu, n), _lst(post, n)
        if not torch.is_grad_enabled() and outs is None and not any(b.training for b in bns) and \
                all(po is None for po in posts):
            # inference: the trunk's 3x3 convs carry their BatchNorm (+ residual, ReLU) as the epilogue -- one launch
            # instead of two and no round trip of the conv output through HBM (hip_backend.conv_bn_infer_group)
            metas = [self._bn_meta(b, r) for b, r in zip(bns, relus)]
            zs = self.hb.conv_bn_infer_group(convs, metas, [b.weight for b in bns], [b.bias for b in bns], xs, ress, relus)
            rest = [i for i in range(n) if zs[i] is None]
            if rest and len(rest) < n:
                sub = self._conv_bn_unfused([convs[i] for i in rest], [bns[i] for i in rest], [xs[i] for i in rest],
                                            [ress[i] for i in rest], [relus[i] for i in rest], [None] * len(rest), None)
                for i, z in zip(rest, sub):
                    zs[i] = z
            if not rest or len(rest) < n:
                return zs
        return self._conv_bn_unfused(convs, bns, xs, ress, relus, posts, outs)

    def _conv_bn_unfused(self, convs, bns, xs, ress, relus, posts, outs):
        spec = tuple((c.stride[0], c.padding[0], c.dilation[0], False, bool(b.training)) for c, b in zip(convs, bns))
        flat = []
        for xi, c in zip(xs, convs):
            flat += [xi, c.weight, c.bias]
        ys = self.hb.ConvGroupFn.apply(spec, *flat)
        return self._bn_group(list(ys), bns, ress, relus, posts, outs)

    def basic_block(self, blocks, xs):
        ok = torch.is_grad_enabled() and all(
            b.bn1.training and b.bn2.training and b.conv1.bias is None and b.conv2.bias is None and
            b.conv1.stride[0] == 1 and b.conv1.kernel_size[0] == 3 and b.conv1.padding[0] == 1 and
            b.conv1.dilation[0] == 1 and b.conv1.in_channels == b.conv2.out_channels and
            b.conv1.weight.requires_grad for b in blocks)
        if not ok:
            mid = self._conv_bn_group([b.conv1 for b in blocks], [b.bn1 for b in blocks], list(xs), relu=True)
            return self._conv_bn_group([b.conv2 for b in blocks], [b.bn2 for b in blocks], mid, residual=list(xs), relu=True)
        metas = tuple((self._bn_meta(b.bn1, True), self._bn_meta(b.bn2, True)) for b in blocks)
        flat = []
        for b, x in zip(blocks, xs):
            flat += [x, b.conv1.weight, b.bn1.weight, b.bn1.bias, b.conv2.weight, b.bn2.weight, b.bn2.bias]
        return list(self.hb.BasicBlockGroupFn.apply(metas, *flat))

    def fork(self, thunk, tag="fuse", has_bn=True):
        return self.hb.fork(thunk, tag, has_bn)

    def fan_out(self, tensors, counts):
        if not torch.is_grad_enabled() or not any(t.requires_grad for t in tensors) or max(counts) < 2 or \
                any(t.dtype != self.act_dtype or t.numel() % 8 for t in tensors):
            return BackendBase.fan_out(self, tensors, counts)
        flat = self.hb.FanOutGroupFn.apply(tuple(counts), *tensors)
        out, off = [], 0
        for c in counts:
            out.append(list(flat[off:off + c]))
            off += c
        return out

    def end_forward(self):
        self.hb.end_forward()

    def flush_backward(self):
        """Deferred backward work whose results another end-of-backward callback is about to read."""
        self.hb.flush_wgrads(join=True)

    def sum_act(self, tensors, relu=True):
        multi = bool(tensors) and _is_list(tensors[0])
        probs = [list(t) for t in tensors] if multi else [list(tensors)]
        flat = [t for p in probs for t in p]
        zs = self.hb.SumActGroupFn.apply(bool(relu), tuple(len(p) for p in probs), *flat)
        return list(zs) if multi else zs[0]

    def bilinear(self, x, size, out_f32=False):
        multi = _is_list(x)
        xs = list(x) if multi else [x]
        sizes = [tuple(s) for s in size] if (multi and hasattr(size[0], "__len__")) else [tuple(size)] * len(xs)
        outs = [None] * len(xs)
        todo, spec = [], []
        for i, (t, s) in enumerate(zip(xs, sizes)):
            if tuple(t.shape[1:3]) == s and (t.dtype == torch.float32) == bool(out_f32 or t.dtype == torch.float32):
                outs[i] = t
            else:
                todo.append(i)
                spec.append((int(s[0]), int(s[1]), bool(out_f32)))
        if todo:
            ys = self.hb.BilinearGroupFn.apply(tuple(spec), *[xs[i] for i in todo])
            for i, y in zip(todo, ys):
                outs[i] = y
        return outs if multi else outs[0]

    def upsample_cat(self, groups):
        counts = tuple(len(g) for g in groups)
        return list(self.hb.UpsampleCatGroupFn.apply(counts, *[t for g in groups for t in g]))

    def max_pool3x3s2(self, x):
        return self.hb.MaxPool3x3s2Fn.apply(x)

    def global_avg_pool(self, x):
        return self.hb.GlobalAvgPoolFn.apply(x)

    def cat(self, tensors):
        if len(tensors) == 2 and self.hb.adjacent_slices(tensors[0], tensors[1]):
            return self.hb.CatViewFn.apply(tensors[0], tensors[1])      # both already lie in one buffer (cat_slots)
        return torch.cat(tensors, dim=3)      # pure data movement

    def to_act(self, x):
        return x.to(self.act_dtype)           # dtype cast only

    def ocr_gather(self, feats, logits):
        return self.hb.OcrGatherFn.apply(feats, logits)

    def ocr_attention(self, q, k, v, scale):
        return self.hb.OcrAttnFn.apply(q, k, v, scale)

    def sigmoid(self, x):
        return self.hb.SigmoidFn.apply(x)

    def bcast_mul(self, a, x):
        return self.hb.BcastMulFn.apply(a, x)

    def attn_blend(self, lo, a, hi):
        return self.hb.AttnBlendFn.apply(lo, a, hi)

    def ewise(self, op, a, b):
        """fp32 a (+|*|/) b, same shapes ('add', 'mul', 'div')."""
        return self.hb.EwiseFn.apply(op, a, b)

    def relu(self, x):
        return self.sum_act([x], relu=True)

    def cross_entropy(self, logits, labels, ignore_index):
        if logits.requires_grad and torch.is_grad_enabled():
            self.hb._no_fp16_training()
        return self.hb.CrossEntropyFn.apply(logits, labels, ignore_index)

    def bce_rmi(self, logits, labels, do_rmi, weight_lambda=0.5):
        if logits.requires_grad and torch.is_grad_enabled():
            self.hb._no_fp16_training()
        return self.hb.BceRmiFn.apply(logits, labels, bool(do_rmi), weight_lambda)


def backend():
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def _set_backend_for_tests(b):
    """Test-suite hook (tests/test_wiring_cpu.py).  Not used by product code."""
    global _BACKEND
    _BACKEND = b
