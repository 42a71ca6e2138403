"""Operator surface used by semseg_amd.network / semseg_amd.loss.

All functions take and return NHWC tensors ([B,H,W,C]); activations in the
backend's `act_dtype` (bf16 on the HIP backend), logits/losses fp32.  The
default -- and only shipped -- backend is the HIP one (hip_backend.py over
libsemseg_hip.so); it is created on first use and raises if the library is
missing.  `_set_backend_for_tests` exists so the CPU test-suite can check the
module wiring against the reference with the oracle's operators; product code
never calls it.
"""
import os

import torch

_BACKEND = None


class HipBackend:
    name = "hip"

    def __init__(self):
        from . import hip_backend as hb
        hb.lib()   # fail loudly now if the extension is absent
        self.hb = hb
        self.act_dtype = hb.ACT_DTYPE

    def begin_step(self, device=None):
        self.hb.begin_step(device)

    # -- independent sub-graphs on concurrent HIP streams ------------------
    # The two scale passes of MscaleOCR and the 2-4 resolution branches of every
    # HighResolutionModule are independent chains of small kernels (B=1: most
    # launches cannot fill 256 CUs).  Each chain gets its own stream; fork/join
    # are events, so a captured hipGraph keeps them as parallel branches.
    # autograd replays every op's backward on the stream of its forward, so the
    # backward pass is concurrent in the same way.
    # bit 0: the two scale passes, bit 1: the HRNet branches (SSA_CONCURRENCY=0 disables both).
    # Measured on MI355X, 1024x1024 crop, hipGraph replay: sequential 108.2 ms/step,
    # scale passes concurrent 77.1, scale passes + branches 101.8 (the per-module
    # fork/join events cost more than the overlap buys at this size; at 256x256 the
    # branches alone give 70.3 -> 60.1) -- hence the default of 1.
    concurrency = int(os.environ.get("SSA_CONCURRENCY", "1"))
    # The 0.5x pass runs on detached aliases of the parameters and its gradients are added to the
    # 1.0x pass's by ONE multi-tensor add at the end of backward, instead of autograd's 955
    # per-parameter `add` launches (MscaleOCR._shadow_parameters).  Off under torch.distributed:
    # DDP's per-parameter hooks must see the complete gradient.
    shadow_lo_pass = os.environ.get("SSA_SHADOW", "1") != "0"

    def use_shadow_pass(self):
        from .parallel import sync_world
        return self.shadow_lo_pass and not sync_world()

    def side_streams(self):
        return list(self._streams.values())
    _streams = {}
    _side_handles = set()

    @staticmethod
    def _sequential(thunks):
        outs = [None] * len(thunks)          # same issue order as the concurrent path
        for i in range(1, len(thunks)):
            outs[i] = thunks[i]()
        outs[0] = thunks[0]()
        return outs

    def parallel(self, thunks, level=1):
        if not (self.concurrency & level) or len(thunks) < 2 or not torch.cuda.is_available():
            return self._sequential(thunks)
        main = torch.cuda.current_stream()
        if main.cuda_stream in self._side_handles:
            # nested fork (a fork from an already forked stream): hipStreamEndCapture
            # segfaults on such graphs (ROCm 7.2), so only the root stream forks
            return self._sequential(thunks)
        side = []
        for i in range(1, len(thunks)):
            key = (main.cuda_stream, level, i)     # nested forks never share a stream
            st = self._streams.get(key)
            if st is None:
                st = self._streams[key] = torch.cuda.Stream()
                self._side_handles.add(st.cuda_stream)
            st.wait_stream(main)                 # fork: everything enqueued on `main` so far
            side.append(st)
        outs = [None] * len(thunks)
        for i in range(1, len(thunks)):
            with torch.cuda.stream(side[i - 1]):
                outs[i] = thunks[i]()
        outs[0] = thunks[0]()
        for i, st in enumerate(side):
            main.wait_stream(st)                 # join
            _record_stream(outs[i + 1], main)
        return outs

    def image_to_nhwc(self, images, out_hw=None):
        return self.hb.image_to_nhwc(images, out_hw)

    def conv2d(self, x, weight, bias, stride, padding, dilation, out_f32=False, want_stats=False):
        return self.hb.Conv2dFn.apply(x, weight, bias, stride, padding, dilation, out_f32, want_stats)

    def conv_bn_act(self, conv, bn, x, residual=None, relu=False, post=None, private_input=False, block=None):
        """conv -> BatchNorm (+residual, ReLU, mask).  In training the conv epilogue
        accumulates the batch statistics where the kernel supports it, and the
        normalisation picks them up instead of re-reading the conv output.
        private_input: `x` is the output of a BatchNorm+ReLU layer and this conv is its ONLY
        consumer; block: the residual_link() of the residual block this call belongs to (its conv1
        call has residual=None and the block input as `x`, its last call has the block input as
        `residual`).  Both only matter under SSA_FUSE_BWD (hip_backend.py)."""
        hb = self.hb
        if hb._FUSE_BWD and torch.is_grad_enabled():
            if private_input:
                hb._NEXT_CONV_IN_LINK[0] = getattr(x, "_ssa_bn_link", None)
            if block is not None:
                if residual is None:
                    hb._NEXT_CONV_RES_LINK[0] = block
                else:
                    hb._NEXT_BN_RES_LINK[0] = block
        y = self.conv2d(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], conv.dilation[0], False,
                        bool(bn.training))
        return self.batch_norm_act(y, bn, residual, relu, post)

    def residual_link(self):
        """Hand-over object for one residual block (None unless SSA_FUSE_BWD)."""
        return self.hb.ResLink() if (self.hb._FUSE_BWD and torch.is_grad_enabled()) else None

    def batch_norm_act(self, x, bn, residual=None, relu=False, post=None):
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        link = None
        if self.hb._FUSE_BWD and bn.training and relu and residual is None and post is None and \
                torch.is_grad_enabled():
            link = self.hb._NEXT_BN_OUT_LINK[0] = self.hb.BnLink()
        # training: the running statistics are updated by end_forward() (deferred, in
        # issue order) because passes over the same layer run on concurrent streams
        z = self.hb.BatchNormActFn.apply(
            x, bn.weight, bn.bias, residual, post,
            None if bn.training else bn.running_mean, None if bn.training else bn.running_var, None,
            0.1 if bn.momentum is None else bn.momentum, bn.eps, bn.training, relu,
            getattr(bn, "sync", False), self.hb._BN_UPDATES.slot(bn) if track else None)
        if link is not None and link.x is not None:
            z._ssa_bn_link = link          # picked up by the conv that consumes z alone (private_input=True)
        return z

    def end_forward(self):
        self.hb.end_forward()

    def flush_backward(self):
        """Deferred backward work whose results another end-of-backward callback is about to read."""
        self.hb.flush_wgrad_reduces()

    def sum_act(self, tensors, relu=True):
        return self.hb.SumActFn.apply(relu, *tensors)

    def bilinear(self, x, size, out_f32=False):
        if tuple(x.shape[1:3]) == tuple(size) and (x.dtype == torch.float32) == bool(out_f32 or x.dtype == torch.float32):
            return x
        return self.hb.BilinearFn.apply(x, int(size[0]), int(size[1]), bool(out_f32))

    def max_pool3x3s2(self, x):
        return self.hb.MaxPool3x3s2Fn.apply(x)

    def global_avg_pool(self, x):
        return self.hb.GlobalAvgPoolFn.apply(x)

    def cat(self, tensors):
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

    def cross_entropy(self, logits, labels, ignore_index):
        return self.hb.CrossEntropyFn.apply(logits, labels, ignore_index)

    def bce_rmi(self, logits, labels, do_rmi, weight_lambda=0.5):
        return self.hb.BceRmiFn.apply(logits, labels, bool(do_rmi), weight_lambda)


def _record_stream(obj, stream):
    """Tell the caching allocator that tensors produced on a side stream are
    consumed on `stream` (so their memory is not recycled under that use)."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


def backend():
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def _set_backend_for_tests(b):
    """Test-suite hook (tests/test_wiring_cpu.py).  Not used by product code."""
    global _BACKEND
    _BACKEND = b
