"""Resampling helpers with the reference's names (network/mynn.py:42-114), on
NHWC tensors.  Upsample() returns fp32 like the reference's
@amp.float_function wrappers."""
import math

from .. import ops
from ..nn import Norm2d, initialize_weights  # noqa: F401


def Upsample(x, size):
    """x / size: a tensor and its target size, or lists of them (independent problems)."""
    return ops.backend().bilinear(x, size, out_f32=True)


def scale_as(x, y):
    return ops.backend().bilinear(x, y.shape[1:3], out_f32=(x.dtype.is_floating_point and x.dtype.itemsize == 4))


def resized_hw(h, w, scale_factor):
    """Output size of F.interpolate(scale_factor=s, recompute_scale_factor=True)."""
    return int(math.floor(h * scale_factor)), int(math.floor(w * scale_factor))
