"""The C-ABI library loads on a GPU-less host and exports every function that
include/semseg_hip.h declares; the ctypes table binds exactly that set."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "semseg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(ssa_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from semseg_amd import _lib
    names = _header_functions()
    assert len(names) >= 45
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    assert handle.ssa_version() >= 1


def test_ctypes_table_matches_header():
    from semseg_amd import _lib
    assert _lib.declared_symbols() == _header_functions()
    _lib.lib()   # binds argtypes; raises on a stale build


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    import ctypes
    from semseg_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc(1, 8, 8, 12, 12, 8, 8, 16, 16, 3, 3, 1, 1, 1, 0, 128, 0, -1)   # Cin % 8 != 0
    assert L.ssa_conv2d_igemm(ctypes.byref(d), None, None, None, None, None) == -1
    assert L.ssa_bn_stats(None, 10, 48, 48, None, 1, None) == -1
    assert L.ssa_pack_filter(None, None, 1, 1, 1, 1, 8, 8, 32, 0, None) == -1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "semantic-segmentation_amd", "semseg_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
