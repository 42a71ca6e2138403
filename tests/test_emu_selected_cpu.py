"""A fast selection of the GPU-marked kernel tests, run on the CPU EMULATION of the kernels.

`tools/emu` compiles the unchanged kernel sources for the host (workgroups as cooperative fibers; MFMA, the transposing
LDS read and the LDS DMA emulated with the lane maps the probe tests pin on the device), and `SSA_EMU=1` makes the
`-m gpu` kernel tests run against that library with CPU tensors as device memory (tests/conftest.py, tests/emu_util.py).
The whole emulated suite takes hours (a 130x131x192->200 conv is four minutes); this module runs, in a subprocess, the
tests that take seconds and cover the element-wise / loss / resampling kernels and the small conv dispatch classes --
so that index arithmetic, LDS layouts and barrier structure of those kernels are checked by the default CPU suite, not
only on the GPU box.  Test infrastructure: the product never loads the emulation library."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (file, -k expression): each entry a few seconds under emulation
SELECTION = [
    ("tests/test_kernels_gpu.py", "bce_rmi or scale_fusion or cross_entropy or sigmoid or softmax"),
    ("tests/test_kernels_gpu.py", "probe or bn_train or bn_eval or bn_deferred or bilinear or maxpool or conv_channel_slice"),
    ("tests/test_kernels_gpu.py", "test_conv_fwd_bwd and (case1] or case10] or case19] or case32] or case33])"),
    ("tests/test_group_gpu.py", "upsample_cat or cat_slots"),
    # round 5: the fused object attention (19 / 65 / 96 regions, ragged pixel counts, the three-launch form), a grouped
    # weight-gradient launch of more than 16 layers
    ("tests/test_kernels_gpu.py", "ocr_attention or twenty"),
]


SELECTION_ENV = []     # (file, -k expression, extra environment)


def test_selected_kernel_tests_pass_on_the_emulated_kernels():
    env = dict(os.environ, SSA_EMU="1")
    env.pop("PYTEST_CURRENT_TEST", None)
    for path, expr, extra in SELECTION_ENV:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, path), "-q", "-x", "-m", "gpu", "-k", expr,
                            "-p", "no:cacheprovider"], cwd=ROOT, env=dict(env, **extra), capture_output=True, text=True, timeout=900)
        tail = "\n".join(r.stdout.splitlines()[-15:])
        assert r.returncode == 0, "%s -k %r under SSA_EMU=1 %r:\n%s\n%s" % (path, expr, extra, tail, r.stderr[-2000:])
        assert " passed" in tail and "failed" not in tail, tail
    for path, expr in SELECTION:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, path), "-q", "-x", "-m", "gpu", "-k", expr,
                            "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        tail = "\n".join(r.stdout.splitlines()[-15:])
        assert r.returncode == 0, "%s -k %r under SSA_EMU=1:\n%s\n%s" % (path, expr, tail, r.stderr[-2000:])
        assert " passed" in tail and "failed" not in tail, tail
