"""Self-test of the teacher-forcing harness (tests/teacher_backend.py) without a GPU: with the oracle's
emulation backend standing in for the HIP side, the whole training step of HRNet-OCR-MScale is traced
op by op, forward and backward, and every comparison must be exact -- so a failure of the GPU test is
a property of the kernels, not of the harness."""
import copy

import torch


def test_teacher_harness_is_exact_against_itself():
    from semseg_amd import ops
    from teacher_backend import TeacherBackend
    from test_parity_1024_gpu import _build, _bench_batch
    images, gts = _bench_batch(128)
    cpu_net, sd = _build()
    mirror = copy.deepcopy(cpu_net).train()
    tb = TeacherBackend(cpu_net, mirror, device="cpu")
    prev = ops._BACKEND
    ops._set_backend_for_tests(tb)
    try:
        loss = cpu_net({"images": images, "gts": gts})
        n_fwd = tb.rec.n_ops
        loss.backward()
    finally:
        ops._set_backend_for_tests(prev)
    print(tb.rec.summary(5))
    assert n_fwd > 100
    assert sum(1 for r in tb.rec.rows if r[2].startswith("dparam")) > 900       # every parameter's gradient was compared
    assert sum(1 for r in tb.rec.rows if r[2].startswith("din")) > 500
    assert not tb.rec.failures()
    assert max(r[4] for r in tb.rec.rows) < 1e-6
