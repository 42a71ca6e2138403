"""ORACLE -- test infrastructure only.  Never imported by the product package.

CPU restatement (numpy, fp32 arithmetic step by step) of the optimizer step the
reference runs at train.py:509: torch.optim.SGD built by loss/optimizer.py:47-53
(momentum, weight decay, nesterov=False, dampening 0) -- torch/optim/sgd.py
`_single_tensor_sgd` of the pinned torch 2.10 -- and of the LR multipliers of
loss/optimizer.py:67-92.  Pinned against torch.optim.SGD itself and against LR
sequences generated from the real reference (tests/golden/make_golden_optim.py)
in tests/test_optim_cpu.py."""
import math

import numpy as np


def sgd_step(params, grads, bufs, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
    """In-place on lists of fp32 numpy arrays; bufs[i] may be None before the first
    step (torch clones the first d_p into the buffer).  Returns the buffers."""
    f = np.float32
    out = []
    for p, g, b in zip(params, grads, bufs):
        d = g.astype(np.float32)
        if weight_decay != 0:
            d = d + f(weight_decay) * p                 # grad.add(param, alpha=weight_decay)
        if momentum != 0:
            b = d.copy() if b is None else f(momentum) * b + d   # buf.mul_(momentum).add_(d_p, alpha=1-0)
            d = d + f(momentum) * b if nesterov else b
        p -= f(lr) * d                                   # param.add_(d_p, alpha=-lr)
        out.append(b)
    return out


def lr_multiplier(schedule, epoch, max_epoch, poly_exp, poly_step=None, rescale=None, repoly=None,
                  reduce_border_epoch=-1):
    """loss/optimizer.py:67-92"""
    if schedule == "poly":
        return math.pow(1 - epoch / max_epoch, poly_exp)
    if schedule == "poly2":
        e = poly_exp if epoch < poly_step else 2 * poly_exp
        return math.pow(1 - epoch / max_epoch, e)
    if schedule == "scl-poly":
        t = reduce_border_epoch
        if epoch < t:
            return math.pow(1 - epoch / max_epoch, poly_exp)
        return rescale * math.pow(1 - (epoch - t) / (max_epoch - t), repoly)
    raise ValueError(schedule)
