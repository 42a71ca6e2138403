"""Drop-in installation under the reference's own entry points.

    import semseg_amd.dropin as dropin; dropin.install()      # before train.py imports network/loss

registers this package's modules under the names the reference resolves at run
time, so `train.py`, `config.py` and `scripts/*.yml` run unchanged:

  network.ocrnet / network.deepv3 / network.mscale / network.mscale2 / network.attnscale / network.hrnetv2 /
  network.ocr_utils / network.utils / network.mynn
        -> semseg_amd.network.*      (importlib target of --arch, network/__init__.py:45-54)
  loss.utils (get_loss, CrossEntropyLoss2d), loss.rmi (RMILoss), loss.optimizer (get_optimizer,
  restore_opt, restore_net, ...)
        -> semseg_amd.loss.*         (train.py:45-46,341,378)
  apex.parallel.SyncBatchNorm / DistributedDataParallel, apex.amp
        -> semseg_amd.nn.SyncBatchNorm / semseg_amd.parallel.DistributedDataParallel /
           semseg_amd.amp behind apex.amp's names: `--fp16` selects the fp16-storage build of the library and
           amp.initialize / amp.scale_loss carry apex's dynamic loss scaling (on the device, inside the captured step);
           on the bf16 build they are the identity     (config.py:218-220, network/__init__.py:37-39, train.py:381,504)

Configuration: the model factories (`--arch` targets), `get_loss` and `get_optimizer` registered by
`install()` re-read the reference's global `cfg` every time they are called (`sync_config()`), so the
values `assert_and_infer_cfg(args)` and `datasets.setup_loaders(args)` (NUM_CLASSES, train.py:338-341)
put there are the ones the modules are built with -- train.py needs no extra call.
"""
import contextlib
import functools
import inspect
import sys
import types


def _amp_shim():
    amp = types.ModuleType("apex.amp")
    amp.float_function = lambda f: f
    amp.half_function = lambda f: f
    amp.disable_casts = contextlib.nullcontext

    def initialize(model, optimizers=None, opt_level="O1", **kw):
        # train.py:381 hands (net, optim) through here before it wraps the net for data parallelism.  On a GPU the
        # hipGraph proxies (semseg_amd/graphed.py) go back: the unmodified loop then replays one captured step per
        # iteration (4x the eager loop's speed at batch 1; a refused capture falls back to the eager step, logged
        # once).  SSA_GRAPHED_STEP=0: the plain objects.
        import os
        import torch
        from . import amp as samp
        on_gpu = any(p.is_cuda for p in model.parameters())
        # the fp16-storage build (`--fp16` under install(), or SSA_ACT_DTYPE=fp16): apex's dynamic loss scaler, on the
        # device, inside the optimizer step (semseg_amd/amp.py); bf16 storage needs none
        samp.initialize(model, optimizers, opt_level)
        # The captured step goes back only where one process drives one GPU: with several visible GPUs and no process
        # group the reference wraps the net in torch.nn.DataParallel (network/__init__.py:28), whose replicas would share
        # ONE captured step on cuda:0 (advisor, round 4).
        one_gpu_per_process = torch.cuda.device_count() == 1 or \
            (torch.distributed.is_available() and torch.distributed.is_initialized())
        if os.environ.get("SSA_GRAPHED_STEP", "1") != "0" and on_gpu and one_gpu_per_process and \
                optimizers is not None and not isinstance(optimizers, (list, tuple)):
            from .graphed import graph_training
            return graph_training(model, optimizers)
        return model, optimizers
    amp.initialize = initialize

    def scale_loss(loss, optimizers, **kw):
        from . import amp as samp
        return samp.scale_loss(loss, optimizers, **kw)      # identity on the bf16 build; loss * S on the fp16 build
    amp.scale_loss = scale_loss
    return amp


def _select_storage_from_argv():
    """`--fp16` (train.py:200, scripts/train_*.yml `fp16: true`) selects the fp16-storage build of the library, which
    is a property of the process: it has to be decided before semseg_amd._lib is imported."""
    import os
    if "SSA_ACT_DTYPE" not in os.environ and "--fp16" in sys.argv and "semseg_amd._lib" not in sys.modules:
        os.environ["SSA_ACT_DTYPE"] = "fp16"


def install(replace_apex=True):
    _select_storage_from_argv()
    from . import nn as snn, parallel, network, loss
    from .network import ocrnet, hrnetv2, ocr_utils, utils as nutils, mynn, deepv3, mscale, mscale2, attnscale
    from .loss import criteria, optimizer
    if replace_apex:
        apex = types.ModuleType("apex")
        par = types.ModuleType("apex.parallel")
        par.SyncBatchNorm = snn.SyncBatchNorm
        par.DistributedDataParallel = parallel.DistributedDataParallel
        apex.parallel = par
        apex.amp = _amp_shim()
        sys.modules["apex"] = apex
        sys.modules["apex.parallel"] = par
        sys.modules["apex.amp"] = apex.amp
    for mod in (ocrnet, deepv3, mscale, mscale2, attnscale):
        for fname, fn in list(vars(mod).items()):
            if inspect.isfunction(fn) and fn.__module__ == mod.__name__ and not fname.startswith("_") and \
                    "num_classes" in inspect.signature(fn).parameters and not getattr(fn, "_ssa_synced", False):
                setattr(mod, fname, _synced(fn))
    for mod, names in ((criteria, ("get_loss",)), (optimizer, ("get_optimizer",))):
        for fname in names:
            fn = getattr(mod, fname)
            if not getattr(fn, "_ssa_synced", False):
                setattr(mod, fname, _synced(fn))
    loss.get_loss, loss.get_optimizer = criteria.get_loss, optimizer.get_optimizer
    for name, mod in (("network.ocrnet", ocrnet), ("network.hrnetv2", hrnetv2),
                      ("network.ocr_utils", ocr_utils), ("network.utils", nutils),
                      ("network.mynn", mynn), ("network.deepv3", deepv3), ("network.mscale", mscale),
                      ("network.mscale2", mscale2), ("network.attnscale", attnscale),
                      ("loss.utils", criteria), ("loss.rmi", criteria), ("loss.optimizer", optimizer)):
        sys.modules[name] = mod
    return network, loss


def _synced(fn):
    """fn with our cfg refreshed from the reference's first (when the reference's config is loaded)."""
    @functools.wraps(fn)
    def wrapper(*a, **k):
        if "config" in sys.modules and hasattr(sys.modules["config"], "assert_and_infer_cfg"):
            sync_config()
        return fn(*a, **k)
    wrapper._ssa_synced = True
    return wrapper


def sync_config():
    """Mirror the reference's frozen cfg (after assert_and_infer_cfg) and pick
    the norm layer the way config.py:216-225 does."""
    from config import cfg as ref_cfg          # the reference's config.py on sys.path
    from .config import cfg, sync_from_reference
    from . import nn as snn
    sync_from_reference(ref_cfg)
    cfg.MODEL.BNFUNC = snn.SyncBatchNorm if getattr(ref_cfg.MODEL, "BN", "") == "syncnorm" or \
        "SyncBatchNorm" in getattr(ref_cfg.MODEL.BNFUNC, "__name__", "") else snn.BatchNorm2d
