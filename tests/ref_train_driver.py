"""TEST-ONLY driver: runs the REFERENCE's own train.py (/root/reference, build container only) for a few
iterations on CPU with the drop-in installed -- argparse -> assert_and_infer_cfg -> datasets.setup_loaders
(nullloader) -> get_loss -> network.get_net -> get_optimizer -> amp.initialize -> wrap_network_in_dataparallel
-> train() -> validate() (train.py:324-597) -- with libsemseg_hip.so loaded and its launching entry points
replaced by ctypes-signature checks (tests/test_hip_glue_dryrun_cpu.py), the third-party modules the image
lacks (runx, cv2, torchvision) replaced by inert stand-ins, and `.cuda()` made a no-op.
Usage: python tests/ref_train_driver.py <arch> <result_dir>"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(ROOT, "semantic-segmentation_amd"), ROOT, os.path.join(ROOT, "tests")]


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Logx:
    rank0 = True
    logdir = None

    def initialize(self, logdir=None, **kw):
        self.logdir = logdir
        os.makedirs(logdir, exist_ok=True)

    def msg(self, s):
        print("logx:", s)

    def __getattr__(self, name):
        return lambda *a, **k: None


class _Compose:
    def __init__(self, ts):
        self.ts = list(ts)

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class _Identity:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


def main(arch, result_dir):
    if not hasattr(np, "int"):
        np.int = int
    logx = _Logx()
    _mod("runx", logx=_mod("runx.logx", logx=logx))
    _mod("cv2")
    sk = _mod("skimage")
    sk.filters = _mod("skimage.filters", gaussian=lambda *a, **k: None)
    sk.restoration = _mod("skimage.restoration", denoise_bilateral=lambda *a, **k: None)
    sk.segmentation = _mod("skimage.segmentation", find_boundaries=lambda *a, **k: None)
    _mod("tensorboardX", SummaryWriter=object)
    _mod("coolname", generate_slug=lambda *a, **k: "test-run")
    tv = _mod("torchvision")
    class _Transforms(types.ModuleType):          # any transform the image dumper asks for: the identity
        Compose = _Compose

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Identity
    tv.transforms = sys.modules["torchvision.transforms"] = _Transforms("torchvision.transforms")
    tv.utils = _mod("torchvision.utils", make_grid=lambda *a, **k: None, save_image=lambda *a, **k: None)
    import semseg_amd.dropin as dropin
    dropin.install()
    # the HIP glue runs for real; launches are replaced by signature checks (no GPU here)
    from test_hip_glue_dryrun_cpu import DryLib
    from semseg_amd import _lib, hip_backend
    from semseg_amd.loss import optimizer as sopt
    import contextlib
    dry = DryLib(_lib.lib())
    _lib._LIB = dry
    hip_backend._s = lambda: None
    sopt._on_gpu = lambda p: True
    sopt._launch_scope = lambda device: contextlib.nullcontext((None, False))
    # no device in this container: .cuda() is the identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1         # train.py: args.ngpu -> batch size of the non-apex loader
    torch.cuda.DoubleTensor = torch.DoubleTensor
    sys.path.insert(0, REF)
    os.chdir(REF)
    import datasets.nullloader as nl
    nl.Loader.__len__ = lambda self: 2          # two iterations per epoch
    from config import cfg
    cfg.MODEL.HRNET_CHECKPOINT = ""
    cfg.ASSETS_PATH = result_dir
    sys.argv = ["train.py", "--dataset", "nullloader", "--arch", arch, "--crop_size", "64,96", "--bs_trn", "1",
                "--bs_val", "1", "--max_epoch", "2", "--result_dir", result_dir, "--rmi_loss", "--lr", "0.01",
                "--poly_exp", "2.0", "--num_workers", "0", "--val_freq", "1", "--supervised_mscale_loss_wt", "0.05",
                "--n_scales", "0.5,1.0", "--class_uniform_pct", "0"]
    import runpy
    # the validation image dumper (utils/misc.py:247-420) is cv2/torchvision/PIL plumbing outside the
    # hot path; with those libraries stubbed it cannot render -- everything around it runs
    import utils.misc as ref_misc
    ref_misc.ImageDumper.dump = lambda self, *a, **k: None
    ref_misc.ImageDumper.write_summaries = lambda self, *a, **k: None
    runpy.run_path(os.path.join(REF, "train.py"), run_name="__main__")
    print("DRIVER launches:", sum(dry.calls.values()), "sgd steps:", dry.calls["ssa_sgd_momentum_step"],
          "bn updates:", dry.calls["ssa_bn_update_running_batched"])
    print("DRIVER OK")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
