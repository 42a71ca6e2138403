"""CPU emulation harness for the HIP kernels (TEST INFRASTRUCTURE).

`tools/emu/build.sh` compiles the unchanged kernel sources of semantic-segmentation_amd/csrc for the host
against a shim of the HIP device language (workgroups as cooperative fibers; MFMA, ds_read_tr16 and the LDS DMA
emulated with the lane maps pinned on the device by the probe tests).  `emu_backend()` binds the host glue of
`semseg_amd.hip_backend` to that library for the duration of a test, with CPU tensors as "device memory": the
kernels' index arithmetic, LDS layouts, barrier structure and the host-side launch planning are then checked
against the oracle without a GPU.  The product never loads this library (semseg_amd/_lib.py opens
lib/libsemseg_hip.so and nothing else); the binding below is installed by monkey-patching from the tests.
"""
import contextlib
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_F16 = os.environ.get("SSA_ACT_DTYPE", "bf16").lower() in ("fp16", "f16", "float16", "half")
EMU_LIB = os.path.join(ROOT, "tools", "emu", "build_f16" if _F16 else "build", "libsemseg_emu.so")

_HANDLE = None


def build_emu():
    subprocess.check_call(["sh", os.path.join(ROOT, "tools", "emu", "build.sh")] + (["f16"] if _F16 else []),
                          stdout=subprocess.DEVNULL)
    return EMU_LIB


def emu_lib():
    """The emulation library with the C-ABI signatures of semseg_amd._lib bound (built on first use)."""
    global _HANDLE
    if _HANDLE is None:
        from semseg_amd import _lib
        build_emu()
        h = ctypes.CDLL(EMU_LIB)
        for name, (argtypes, restype) in _lib._SIGS.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _HANDLE = h
    return _HANDLE


@contextlib.contextmanager
def emu_backend():
    """semseg_amd.hip_backend running on the emulation library and CPU tensors."""
    from semseg_amd import _lib, hip_backend as hb
    h = emu_lib()
    saved = (_lib._LIB, hb._s)
    _lib._LIB = h
    hb._s = lambda: None
    hb.clear_pack_cache()
    hb._ARENA.buf = None
    try:
        yield hb
    finally:
        hb.clear_pack_cache()
        hb._ARENA.buf = None
        _lib._LIB, hb._s = saved
