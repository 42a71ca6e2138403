"""Criteria with the reference's call contract
`criterion(logits [B,C,H,W], targets [B,H,W], do_rmi=None)`
(loss/utils.py:40-67,121-134; loss/rmi.py:34-134), computed by the fused HIP
loss kernels on channels-last fp32 logits."""
import torch
from torch import nn

from .. import ops
from ..config import cfg


def _nhwc_f32(logits):
    """[B,C,H,W] -> [B,H,W,C] fp32; zero-copy when the tensor is the permuted
    view of an NHWC buffer (what semseg_amd.network hands over)."""
    x = logits.permute(0, 2, 3, 1)
    if x.dtype not in (torch.float32, torch.float64):
        x = x.float()
    return x


class CrossEntropyLoss2d(nn.Module):
    """loss/utils.py:121-134"""

    def __init__(self, weight=None, ignore_index=255, reduction="mean"):
        super().__init__()
        assert weight is None and reduction == "mean"
        self.ignore_index = ignore_index

    def forward(self, inputs, targets, do_rmi=None):
        return ops.backend().cross_entropy(_nhwc_f32(inputs), targets, self.ignore_index)


class RMILoss(nn.Module):
    """loss/rmi.py:34-134 with the reference's fixed hyper-parameters
    (radius 3, avg-pool 4/4, lambda 0.5, lambda_way 1)."""

    def __init__(self, num_classes=21, rmi_radius=3, rmi_pool_way=1, rmi_pool_size=4, rmi_pool_stride=4,
                 loss_weight_lambda=0.5, lambda_way=1, ignore_index=255):
        super().__init__()
        assert rmi_radius == 3 and rmi_pool_way == 1 and rmi_pool_size == 4 and rmi_pool_stride == 4
        assert lambda_way == 1
        self.num_classes = num_classes
        self.weight_lambda = loss_weight_lambda
        self.ignore_index = ignore_index

    def forward(self, logits_4D, labels_4D, do_rmi=True):
        x = _nhwc_f32(logits_4D)
        assert x.shape[3] == self.num_classes
        return ops.backend().bce_rmi(x, labels_4D, bool(do_rmi), self.weight_lambda)


def get_loss(args):
    """loss/utils.py:40-67 (the two criteria the hot-path recipes use)."""
    if getattr(args, "rmi_loss", False):
        criterion = RMILoss(num_classes=cfg.DATASET.NUM_CLASSES, ignore_index=cfg.DATASET.IGNORE_LABEL).cuda()
    elif getattr(args, "img_wt_loss", False) or getattr(args, "jointwtborder", False):
        raise NotImplementedError("only --rmi_loss and plain cross entropy are on the accelerated path")
    else:
        criterion = CrossEntropyLoss2d(ignore_index=cfg.DATASET.IGNORE_LABEL).cuda()
    criterion_val = CrossEntropyLoss2d(weight=None, ignore_index=cfg.DATASET.IGNORE_LABEL).cuda()
    return criterion, criterion_val
