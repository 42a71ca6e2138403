#!/usr/bin/env python
"""Which torch (aten) kernels does one training step of the north-star configuration still launch?

Runs the step of bench.py (ocrnet.HRNet_Mscale + RMI loss, SGD) on CPU tensors with every launching entry point of
libsemseg_hip.so replaced by a stand-in (the dry-run harness of tests/test_hip_glue_dryrun_cpu.py), under a
TorchDispatchMode that records each aten op that would launch a device kernel, with its element count and the innermost
semseg_amd source line that issued it.  Values are garbage; shapes, strides and the autograd wiring are real.
usage: python tools/torch_ops_audit.py [crop]      (element counts scale with crop^2; default 256)"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

VIEW_OPS = ("view", "reshape", "_unsafe_view", "permute", "transpose", "t", "slice", "select", "expand", "as_strided", "detach",
            "alias", "unsqueeze", "squeeze", "empty", "empty_like", "empty_strided", "new_empty", "unbind", "split", "chunk",
            "narrow", "size", "stride", "is_", "sym_", "lift_fresh", "_local_scalar_dense", "item", "unfold", "set_", "result_type",
            "_to_copy_meta", "new_empty_strided", "resolve_", "_has_", "is_same_size", "prim", "scalar_tensor", "contiguous_view")


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if not name.startswith(VIEW_OPS):
            n = 0
            for t in (out if isinstance(out, (list, tuple)) else [out]):
                if isinstance(t, torch.Tensor):
                    n = max(n, t.numel())
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "semseg_amd" in fr.filename and "torch_ops_audit" not in fr.filename:
                    where = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT).replace("semantic-segmentation_amd/", ""), fr.lineno, fr.name)
                    break
            if where == "?":           # issued by the autograd engine (gradient accumulation): say what it was summing
                shp = [tuple(t.shape) for t in (out if isinstance(out, (list, tuple)) else [out]) if isinstance(t, torch.Tensor)]
                where = "autograd engine, output %s %s" % (shp[0] if shp else "", str(out.dtype).replace("torch.", "") if isinstance(out, torch.Tensor) else "")
            r = self.rows[(name, where)]
            r[0] += 1
            r[1] += n
        return out


def main():
    crop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    import contextlib
    from test_hip_glue_dryrun_cpu import DryLib
    from semseg_amd import _lib, hip_backend, ops
    from semseg_amd.config import cfg
    from semseg_amd.loss import RMILoss
    from semseg_amd.loss import optimizer as sopt
    from semseg_amd.network import get_model
    d = DryLib(_lib.lib())
    _lib._LIB = d
    hip_backend._s = lambda: None
    sopt._on_gpu = lambda p: True
    sopt._launch_scope = lambda device: contextlib.nullcontext((None, False))
    ops._set_backend_for_tests(ops.HipBackend())
    cfg.MODEL.N_SCALES = None
    net = get_model("ocrnet.HRNet_Mscale", 19, RMILoss(num_classes=19, ignore_index=255)).train()
    g = torch.Generator().manual_seed(0)
    inputs = {"images": torch.randn(1, 3, crop, crop, generator=g), "gts": torch.randint(0, 19, (1, crop, crop), generator=g)}
    optim = sopt.FusedSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4) if hasattr(sopt, "FusedSGD") else None
    # one untraced step first: caches (packed filters, job tables) fill outside the audit, as in the captured graph
    for traced in (False, True):
        ctx = Audit() if traced else contextlib.nullcontext()
        with ctx as a:
            hip_backend.begin_step(torch.device("cpu")) if hasattr(hip_backend, "begin_step") else None
            loss = net(inputs)
            loss = loss["pred"] if isinstance(loss, dict) else loss
            loss.backward()
            if optim is not None:
                optim.step()
                optim.zero_grad(set_to_none=True)
    rows = sorted(a.rows.items(), key=lambda kv: -kv[1][1])
    print("aten kernels of one training step at %dx%d (count, total output elements, op, issued from)" % (crop, crop))
    for (name, where), (cnt, n) in rows:
        if n >= 1:
            print("%5d %12d  %-28s %s" % (cnt, n, name, where))
    print("total aten launches: %d" % sum(v[0] for v in a.rows.values()))


if __name__ == "__main__":
    main()
