"""The fp16-storage instantiation of the library (lib/libsemseg_hip_f16.so, -DSSA_ELEM_F16; SSA_ACT_DTYPE=fp16) -- the
reference's own reduced precision (--fp16 / apex O1: train.py:381, scripts/eval_mapillary.yml:10,14; BASELINE.json
configs[4]).

The storage format is a property of the process (one library build per process), so the SAME test files run again in a
child process with SSA_ACT_DTYPE=fp16: tests/util.py rounds inputs / references to the product's format, the
storage-emulation backend (tests/bf16_emu_backend.py) follows it, and the end-to-end evaluation test additionally
asserts the fp16 bound (eval `pred` relative error <= 0.03: 1.5 x the ~0.013-0.02 floor fp16 storage itself gives on
this network, against ~0.17 for bf16).  Training on this build (dynamic loss scaling): tests/test_amp_fp16_gpu.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, log, timeout):
    env = dict(os.environ, SSA_ACT_DTYPE="fp16")
    env.pop("PYTEST_CURRENT_TEST", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-s"] + args,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", log), "w") as f:
        f.write(r.stdout[-200000:])
        f.write(r.stderr[-20000:])
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, "%s under SSA_ACT_DTYPE=fp16:\n%s\n%s" % (args, tail, r.stderr[-2000:])
    assert " passed" in tail and " failed" not in tail, tail
    return r.stdout


@pytest.mark.gpu
def test_kernels_on_the_fp16_build():
    """Every op-level kernel test (forward, data and weight gradients, BatchNorm, resampling, OCR, losses) against the
    oracle with fp16-rounded inputs, and the grouped / fused-backward tests."""
    _run(["tests/test_kernels_gpu.py", "tests/test_group_gpu.py", "tests/test_fuse_bwd_gpu.py"], "fp16_kernels.log", 900)


@pytest.mark.gpu
def test_eval_end_to_end_on_the_fp16_build():
    """HRNet-OCR-MScale evaluation (two-scale and hierarchical {0.5, 1, 2}) end to end against the fp32 oracle: within
    1.5 x the fp16-storage emulation at every tensor the operator surface returns, and `pred` within 0.03 relative."""
    out = _run(["tests/test_e2e_gpu.py", "-k", "eval_op_by_op or eval_nscale"], "fp16_e2e_eval.log", 900)
    assert "fp16 storage: eval pred rel err" in out, out[-2000:]


@pytest.mark.gpu
def test_eval_teacher_forced_on_the_fp16_build():
    """Teacher-forced evaluation, op by op at one-rounding tolerance, on the fp16 build: the hierarchical {0.5, 1, 2}
    evaluation at 128 x 192 (135 ops, 859 comparisons) and BASELINE configs[1] at 1 x 3 x 1024 x 2048 (277 comparisons).
    Round 4 left the full-size case out: `ocr_attention` showed isolated elements 3e-2 off.  That was the TEST's regime,
    not the kernel (round-4 review): uncalibrated BatchNorm buffers drove q.k^T to 1e8, the softmax was one-hot at every
    pixel and fp32 summation order picked the row of V.  With calibrated buffers (calibrate_eval_bn) the harness asserts
    the operand ranges and the op holds the one-rounding bound."""
    _run(["tests/test_parity_eval_gpu.py", "-k", "three_scales_small or single_scale_1024x2048"], "fp16_eval_parity.log", 900)


@pytest.mark.gpu
def test_eval_mapillary_teacher_forced_on_the_fp16_build():
    """BASELINE configs[4] in the format it names (fp16): the recipe's four-scale chain {0.25, 0.5, 1.0, 2.0} at
    896 x 1152 and the three-scale case at 1152 x 1536 (2.0x pass 2304 x 3072), 65 classes, teacher-forced op by op at
    one-rounding tolerance on libsemseg_hip_f16.so (scripts/eval_mapillary.yml:10-18, network/ocrnet.py:185-262)."""
    _run(["tests/test_parity_eval_gpu.py", "-k", "mapillary_65_classes"], "fp16_eval_mapillary.log", 1500)
