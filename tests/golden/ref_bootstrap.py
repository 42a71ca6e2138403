"""Import the REAL reference (/root/reference) on CPU for golden-vector
generation.  Build-container only: /root/reference does not exist on the GPU
box, so nothing under tests/ imports this at test time -- only make_golden.py.

Stubs follow SURVEY.md Appendix B: apex / runx / cv2 / torchvision are absent
from the image; they are replaced by inert shims (never in the product path).
"""
import contextlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Logx:
    def __getattr__(self, name):
        return lambda *a, **k: None


def bootstrap(num_classes=19):
    sys.dont_write_bytecode = True
    if not hasattr(np, "int"):
        np.int = int
    amp = _mod("apex.amp", float_function=lambda f: f, half_function=lambda f: f,
               disable_casts=contextlib.nullcontext, initialize=lambda *a, **k: a,
               scale_loss=None)
    par = _mod("apex.parallel", SyncBatchNorm=torch.nn.SyncBatchNorm,
               DistributedDataParallel=lambda m, **k: m)
    _mod("apex", amp=amp, parallel=par)
    logx = _Logx()
    _mod("runx", logx=_mod("runx.logx", logx=logx))
    _mod("cv2")
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    tv.utils = _mod("torchvision.utils")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from config import cfg
    ckpt = os.path.join(tempfile.gettempdir(), "empty_hrnet_ckpt.pth")
    torch.save({}, ckpt)
    cfg.MODEL.HRNET_CHECKPOINT = ckpt
    cfg.MODEL.BNFUNC = torch.nn.BatchNorm2d
    cfg.DATASET.NUM_CLASSES = num_classes
    cfg.OPTIONS.TORCH_VERSION = 2.1
    torch.cuda.DoubleTensor = torch.DoubleTensor  # loss/rmi.py:171-172 on CPU
    return cfg
