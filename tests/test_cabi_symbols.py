"""The C-ABI library loads on a GPU-less host and exports every function that
include/semseg_hip.h declares; the ctypes table binds exactly that set."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "semseg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long|const char\*)\s+(ssa_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from semseg_amd import _lib
    names = _header_functions()
    assert len(names) >= 45
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    assert handle.ssa_version() >= 1


def test_both_storage_builds_export_the_same_abi_and_carry_the_source_hash():
    """libsemseg_hip.so (bf16) and libsemseg_hip_f16.so (fp16): same symbols, the element type each was built for, and
    the hash of the kernel sources that lie next to them (a stale binary with the same ABI would otherwise pass)."""
    import ctypes
    from semseg_amd import _lib
    names = _header_functions()
    here = _lib.source_sha()
    assert here is not None and len(here) == 16
    libdir = os.path.dirname(_lib.LIB_PATH)
    for fname, elem in (("libsemseg_hip.so", 0), ("libsemseg_hip_f16.so", 1)):
        h = ctypes.CDLL(os.path.join(libdir, fname))
        for n in names:
            assert hasattr(h, n), (fname, n)
        assert h.ssa_elem_type() == elem, fname
        h.ssa_source_sha.restype = ctypes.c_char_p
        assert h.ssa_source_sha().decode() == here, (fname, h.ssa_source_sha(), here)


def test_stale_library_is_refused(monkeypatch):
    from semseg_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "source_sha", lambda: "0123456789abcdef")
    import pytest
    with pytest.raises(RuntimeError, match="built from other sources"):
        _lib.lib()


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
