"""The loss-scaling kernels of fp16 training (csrc/optim.hip: ssa_amp_check_grads, the amp_state argument of
ssa_sgd_momentum_step, ssa_amp_update) on the CPU EMULATION of the kernel sources (tools/emu; test infrastructure, the
product never loads it), through the C ABI with CPU tensors as device memory: un-scaled updates equal a plain
momentum-SGD recursion, an inf / nan anywhere skips the WHOLE step and halves the scale, clean steps grow it, the bounds
hold.  The same properties are checked on the device through FusedSGD in tests/test_amp_fp16_gpu.py."""
import ctypes

import pytest
import torch


@pytest.fixture(scope="module")
def L():
    from emu_util import emu_lib
    return emu_lib()


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _nums(ts):
    return (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])


SIZES = (7, 4096, 4097, 100003, 33)


def _step(L, P, B, grads, state, interval=3, lr=0.05, m=0.9, wd=1e-4, lo=1.0, hi=2.0 ** 24):
    assert L.ssa_amp_check_grads(_ptrs(grads), _nums(grads), len(grads), state.data_ptr(), None) == 0
    assert L.ssa_sgd_momentum_step(_ptrs(P), _ptrs(grads), _ptrs(B), _nums(P), len(P), lr, None, m, wd, 0,
                                   state.data_ptr(), None) == 0
    assert L.ssa_amp_update(state.data_ptr(), interval, 2.0, 0.5, lo, hi, None) == 0


def test_scaled_steps_equal_the_unscaled_recursion_and_the_scale_grows(L):
    g = torch.Generator().manual_seed(0)
    P = [torch.randn(n, generator=g) for n in SIZES]
    R = [p.clone() for p in P]
    B, RB = [torch.zeros(n) for n in SIZES], [torch.zeros(n) for n in SIZES]
    state = torch.tensor([1024.0, 0.0, 0.0, 1.0 / 1024.0])
    scales = []
    for it in range(7):
        S = float(state[0])
        scales.append(S)
        true = [torch.randn(n, generator=g) for n in SIZES]
        _step(L, P, B, [t * S for t in true], state)
        for p, gr, b in zip(R, true, RB):                    # d = g + wd p;  buf = m buf + d;  p -= lr buf
            b.mul_(0.9).add_(gr + 1e-4 * p)
            p.sub_(0.05 * b)
        for p, r in zip(P, R):
            assert torch.allclose(p, r, rtol=2e-6, atol=2e-7), it
    assert scales == [1024.0, 1024.0, 1024.0, 2048.0, 2048.0, 2048.0, 4096.0]
    assert float(state[1]) == 0.0 and abs(float(state[3]) * float(state[0]) - 1.0) < 1e-6


@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
def test_one_bad_element_skips_everything(L, bad):
    g = torch.Generator().manual_seed(1)
    P = [torch.randn(n, generator=g) for n in SIZES]
    B = [torch.randn(n, generator=g) for n in SIZES]
    state = torch.tensor([512.0, 0.0, 2.0, 1.0 / 512.0])
    before, bufs = [p.clone() for p in P], [b.clone() for b in B]
    grads = [torch.randn(n, generator=g) * 512.0 for n in SIZES]
    grads[3][77777] = bad                                    # in the scalar tail of a later chunk of the fourth tensor
    _step(L, P, B, grads, state)
    assert all(torch.equal(p, q) for p, q in zip(P, before)) and all(torch.equal(p, q) for p, q in zip(B, bufs))
    assert state.tolist()[:3] == [256.0, 0.0, 0.0] and float(state[3]) == 1.0 / 256.0


def test_bounds(L):
    P, B = [torch.zeros(100)], [torch.zeros(100)]
    state = torch.tensor([2.0, 0.0, 0.0, 0.5])
    for _ in range(4):
        _step(L, P, B, [torch.ones(100)], state, interval=1, lo=1.0, hi=4.0)
    assert float(state[0]) == 4.0
    for _ in range(5):
        _step(L, P, B, [torch.full((100,), float("inf"))], state, interval=1, lo=1.0, hi=4.0)
    assert float(state[0]) == 1.0
    # invalid arguments are refused, not launched
    assert L.ssa_amp_update(None, 1, 2.0, 0.5, 1.0, 4.0, None) != 0
    assert L.ssa_amp_update(state.data_ptr(), 0, 2.0, 0.5, 1.0, 4.0, None) != 0
    assert L.ssa_amp_check_grads(None, None, 1, state.data_ptr(), None) != 0
