"""semseg_amd.dropin.install(): the names the reference resolves at run time
(`--arch` via importlib in network/__init__.py:45-54, `from loss.utils import
get_loss`, `from loss.optimizer import get_optimizer, restore_opt, restore_net`
at train.py:45-46, `apex.parallel.*`, `apex.amp`) resolve to this package.  The
second test drives the REAL reference's own `network.get_model` / `config.cfg`
when /root/reference is present (build container); it is skipped elsewhere."""
import argparse
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _run(code):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "semantic-segmentation_amd"), ROOT]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_install_registers_the_references_names():
    out = _run("""
import importlib, sys
import semseg_amd.dropin as dropin
dropin.install()
import semseg_amd
for name in ("network.ocrnet", "network.deepv3", "network.mscale", "network.mscale2", "network.hrnetv2",
             "network.ocr_utils", "network.utils", "network.mynn", "loss.utils", "loss.rmi", "loss.optimizer",
             "apex", "apex.parallel", "apex.amp"):
    assert name in sys.modules, name
from semseg_amd.network import ocrnet, mscale, mscale2, deepv3
assert importlib.import_module("network.ocrnet").HRNet_Mscale is ocrnet.HRNet_Mscale
assert importlib.import_module("network.mscale").HRNet is mscale.HRNet
assert importlib.import_module("network.mscale2").DeepV3R50 is mscale2.DeepV3R50
assert importlib.import_module("network.deepv3").DeepV3PlusR50 is deepv3.DeepV3PlusR50
from semseg_amd.loss import get_loss, RMILoss, get_optimizer
import semseg_amd.loss.optimizer as opt
assert sys.modules["loss.utils"].get_loss is get_loss and sys.modules["loss.rmi"].RMILoss is RMILoss
assert sys.modules["loss.optimizer"].get_optimizer is get_optimizer
for fn in ("restore_opt", "restore_net", "forgiving_state_restore", "load_weights", "restore_snapshot"):
    assert hasattr(sys.modules["loss.optimizer"], fn), fn
from apex.parallel import SyncBatchNorm, DistributedDataParallel
from apex import amp
import semseg_amd.nn, semseg_amd.parallel
assert SyncBatchNorm is semseg_amd.nn.SyncBatchNorm
assert DistributedDataParallel is semseg_amd.parallel.DistributedDataParallel
with amp.scale_loss(3.0, None) as l:
    assert l == 3.0
assert amp.float_function(len) is len
print("ok")
""")
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only in the build container")
def test_reference_factory_builds_our_modules():
    """The reference's own network.get_model / loss resolution, with the drop-in installed."""
    out = _run("""
import sys, types, contextlib
import numpy as np, torch
sys.dont_write_bytecode = True
for name in ("runx", "runx.logx", "cv2", "torchvision", "torchvision.transforms", "torchvision.utils"):
    sys.modules[name] = types.ModuleType(name)           # absent third-party modules (SURVEY.md 8c)
class _L:
    def __getattr__(self, k): return lambda *a, **kw: None
sys.modules["runx.logx"].logx = _L()
sys.modules["runx"].logx = sys.modules["runx.logx"]
import semseg_amd.dropin as dropin
dropin.install()
sys.path.insert(0, %r)
from config import cfg                                      # the reference's config.py
cfg.DATASET.NUM_CLASSES = 19
cfg.MODEL.HRNET_CHECKPOINT = ""       # no ImageNet checkpoint in the container: random init
import network                                              # the reference's network/__init__.py
from loss.utils import get_loss                             # resolves to ours
from loss.optimizer import get_optimizer, restore_opt, restore_net
import semseg_amd.network.ocrnet as ours, semseg_amd.network.mscale as ours_ms, semseg_amd.loss as ours_loss
assert get_loss is ours_loss.get_loss and get_optimizer is ours_loss.get_optimizer
dropin.sync_config()
from semseg_amd.loss import CrossEntropyLoss2d
crit = CrossEntropyLoss2d(ignore_index=255)
net = network.get_model("network.ocrnet.HRNet_Mscale", 19, crit)
assert type(net) is ours.MscaleOCR and len(net.state_dict()) == 1903
net = network.get_model("network.mscale.HRNet", 19, crit)
assert type(net) is ours_ms.MscaleBasic
net = network.get_model("network.deepv3.DeepV3PlusR50", 19, crit)
assert len(net.state_dict()) == 363
import argparse
args = argparse.Namespace(optimizer="sgd", lr=0.01, weight_decay=1e-4, momentum=0.9, lr_schedule="poly",
                          max_epoch=10, poly_exp=2.0, poly_step=110, rescale=1.0, repoly=1.5, amsgrad=False)
opt, sch = get_optimizer(args, net)
assert type(opt).__name__ == "FusedSGD" and abs(opt.param_groups[0]["lr"] - 0.01) < 1e-12
print("ok")
""" % REF)
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is only in the build container")
@pytest.mark.parametrize("arch", ["ocrnet.HRNet_Mscale", "deepv3.DeepV3PlusR50"])
def test_reference_train_py_runs_through_the_dropin(arch, tmp_path):
    """The reference's OWN train.py (train.py:324-597), unmodified, for two epochs of two iterations
    plus validation on its nullloader (BASELINE configs[0] plumbing): argparse -> assert_and_infer_cfg ->
    setup_loaders -> get_loss -> get_net -> get_optimizer -> wrap_network_in_dataparallel -> train() ->
    validate() -> eval_metrics, everything model-, loss- and optimizer-side resolving to this package
    (tests/ref_train_driver.py: real host glue, kernel launches replaced by ctypes-signature checks)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_train_driver.py"), arch, str(tmp_path)],
                       capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "DRIVER OK" in r.stdout
    assert r.stdout.count("[train main loss") == 4            # 2 epochs x 2 iterations went through train()
    assert "sgd steps: 4" in r.stdout and "bn updates: 4" in r.stdout
    assert r.stdout.count("mean_iu") >= 2                      # validate() + eval_metrics ran after each epoch
