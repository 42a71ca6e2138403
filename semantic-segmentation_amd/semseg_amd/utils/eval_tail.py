"""Evaluation tail on the GPU: argmax + confusion matrix without the trip of the full
[B,C,H,W] fp32 prediction to the host (utils/trnval_utils.py:173-196, utils/misc.py:50-67)."""
import ctypes

import torch

from .._lib import lib, check


def confusion_matrix(pred, gts, num_classes, hist=None, return_predictions=False):
    """pred: the model's 'pred' output, [B,C,H,W] fp32 (the zero-copy NCHW view of an NHWC
    buffer that semseg_amd.network returns, or any layout); gts: [B,H,W] int64.
    Returns (hist int64 [C,C] on the device, accumulated into `hist` if given
    [, predictions uint8 [B,H,W]])."""
    assert pred.is_cuda and pred.dim() == 4 and pred.shape[1] == num_classes
    B, C, H, W = pred.shape
    nhwc = pred.permute(0, 2, 3, 1)
    if nhwc.dtype != torch.float32 or not nhwc.is_contiguous():
        nhwc = nhwc.float().contiguous()
    g = gts.to(device=pred.device, dtype=torch.int64).contiguous()
    assert tuple(g.shape) == (B, H, W)
    if hist is None:
        hist = torch.zeros((C, C), dtype=torch.int64, device=pred.device)
    out = torch.empty((B, H, W), dtype=torch.uint8, device=pred.device) if return_predictions else None
    P = ctypes.c_void_p
    check(lib().ssa_confusion_matrix(P(nhwc.data_ptr()), C, P(g.data_ptr()), B * H * W, C,
                                     P(out.data_ptr()) if out is not None else None, P(hist.data_ptr()),
                                     P(torch.cuda.current_stream().cuda_stream)), "ssa_confusion_matrix")
    return (hist, out) if return_predictions else hist


def fast_hist(pred, gtruth, num_classes):
    """Name-compatible with utils/misc.py:50 for device tensors: pred = class ids [N]
    (any integer dtype), gtruth [N]; returns int64 [C,C] on the device."""
    p = pred.reshape(-1).to(torch.int64)
    g = gtruth.reshape(-1).to(torch.int64)
    mask = (g >= 0) & (g < num_classes)
    return torch.bincount(num_classes * g[mask] + p[mask], minlength=num_classes ** 2).reshape(num_classes, num_classes)
