"""Layer classes that keep torch's parameter containers (so state_dict keys,
shapes and initialisers are those of nn.Conv2d / nn.BatchNorm2d, as the
reference's checkpoints expect) but run on the HIP operator surface."""
import torch
from torch import nn

from . import ops
from .config import cfg


class Conv2d(nn.Conv2d):
    """nn.Conv2d (groups=1, square kernel/stride/padding/dilation) on NHWC input."""

    def forward(self, x, out_f32=False):
        """x: a tensor, or a list of tensors (independent problems through the same filter)."""
        assert self.groups == 1 and self.padding_mode == "zeros"
        return ops.backend().conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0],
                                    self.dilation[0], out_f32)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with the following add / ReLU / Dropout2d mask fused into
    the normalisation pass: z = post * relu(bn(x) + residual)."""
    sync = False

    def forward(self, x, residual=None, relu=False, post=None):
        # running statistics and num_batches_tracked are updated by the backend
        # (inside the fused normalisation kernel on the HIP path)
        return ops.backend().batch_norm_act(x, self, residual, relu, post)


class SyncBatchNorm(BatchNorm2d):
    """Cross-rank batch statistics: the per-channel fp64 (sum, sum of squares)
    -- and in backward (sum dy, sum dy*xhat) -- are all-reduced over RCCL, which
    equals BatchNorm over the concatenated global batch.  Stands in for
    apex.parallel.SyncBatchNorm (config.py:216-222)."""
    sync = True


def Norm2d(channels, **kwargs):
    """cfg.MODEL.BNFUNC factory, network/mynn.py:18-24."""
    layer = cfg.MODEL.BNFUNC or BatchNorm2d
    return layer(channels, **kwargs)


def norm_types():
    return (BatchNorm2d,) + ((cfg.MODEL.BNFUNC,) if cfg.MODEL.BNFUNC else ())


def BNReLU(ch):
    """Same container shape as network/utils.py:314-317 (keys '<n>.0.weight')."""
    return nn.Sequential(Norm2d(ch), nn.ReLU())


def conv_bn(conv, bn, x, residual=None, relu=False, post=None, out=None):
    """conv -> norm (+ residual add, ReLU, Dropout2d mask) as one backend call, so
    the backend may fuse the batch statistics into the conv epilogue.
    `x` may be a LIST of independent problems (scale passes, resolution branches); `conv`, `bn`,
    `residual`, `relu`, `post` are then one value for all of them or lists of the same length,
    and a list is returned (ops.BackendBase)."""
    for c in (conv if isinstance(conv, (list, tuple)) else (conv,)):
        assert c.groups == 1 and c.padding_mode == "zeros"
    if out is None:
        return ops.backend().conv_bn_act(conv, bn, x, residual, relu, post)
    return ops.backend().conv_bn_act(conv, bn, x, residual, relu, post, out=out)      # (ops.cat_slots placement)


def initialize_weights(*models):
    """network/mynn.py:27-39."""
    for model in models:
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
