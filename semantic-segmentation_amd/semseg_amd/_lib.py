"""ctypes binding of libsemseg_hip.so (the C ABI declared in include/semseg_hip.h).

The library is the product: there is no Python/ATen fallback.  If the shared
object is missing the loader raises, and every wrapper raises RuntimeError on a
non-zero return code (the reference's own convention is plain Python
exceptions, e.g. network/ocrnet.py:105).
"""
import ctypes
import os
from ctypes import c_int, c_long, c_float, c_double, c_void_p, c_size_t, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# Storage format of the 16-bit activations = which build of the library the process runs on (csrc/common.h):
#   SSA_ACT_DTYPE=bf16 (default)  lib/libsemseg_hip.so
#   SSA_ACT_DTYPE=fp16            lib/libsemseg_hip_f16.so -- the reference's own reduced precision (--fp16 / apex O1)
ACT = os.environ.get("SSA_ACT_DTYPE", "bf16").lower()
if ACT in ("f16", "float16", "half"):
    ACT = "fp16"
if ACT not in ("bf16", "fp16"):
    raise RuntimeError("SSA_ACT_DTYPE must be bf16 or fp16, not %r" % ACT)
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "lib", "libsemseg_hip.so" if ACT == "bf16" else "libsemseg_hip_f16.so"))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))

_LIB = None


def source_sha():
    """First 16 hex digits of sha256 over the kernel sources next to the library -- what csrc/Makefile embeds in the
    binary (ssa_source_sha).  None if the sources are not there (a binary-only installation)."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "group.h"),
                                                              os.path.normpath(os.path.join(_HERE, "..", "..", "include", "semseg_hip.h"))]
    if not files or not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class PackJob(ctypes.Structure):
    """Mirror of ssa_pack_job (64 bytes)."""
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("elem_begin", c_long)] + [(n, c_int) for n in (
        "Cout", "Cin", "KH", "KW", "cin_pad", "cout_pad", "Kpad", "mode", "rows", "layout")]


class BnEvalJob(ctypes.Structure):
    """Mirror of ssa_bn_eval_job (48 bytes)."""
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("coef", c_void_p), ("C", c_int), ("eps", c_float)]


class BnUpdateJob(ctypes.Structure):
    """Mirror of ssa_bn_update_job (104 bytes)."""
    _fields_ = [("running_mean", c_void_p), ("running_var", c_void_p), ("num_batches_tracked", c_void_p),
                ("pass_stats", c_void_p * 8), ("C", c_int), ("npass", c_int), ("momentum", c_float),
                ("pad_", c_int)]


class ProfileRec(ctypes.Structure):
    """Mirror of ssa_profile_rec."""
    _fields_ = [("kernel", ctypes.c_char * 120), ("launches", c_long), ("jobs", c_long),
                ("total_us", c_double), ("flops", c_double), ("bytes", c_double)]


class ConvDesc(ctypes.Structure):
    """Mirror of ssa_conv_desc."""
    _fields_ = [(n, c_int) for n in (
        "B", "H", "W", "Cin", "ldx", "Ho", "Wo", "Cout", "ldy", "KH", "KW",
        "stride", "pad", "dil", "transposed", "Kpad", "out_f32", "cfg")]


_P = c_void_p
_SIGS = {
    "ssa_version": ([], c_int),
    "ssa_elem_type": ([], c_int),
    "ssa_source_sha": ([], ctypes.c_char_p),
    "ssa_group_begin": ([], c_int),
    "ssa_group_end": ([_P], c_int),
    "ssa_group_abort": ([], c_int),
    "ssa_launch_count": ([c_int], c_long),
    "ssa_profile_begin": ([], c_int),
    "ssa_profile_note": ([c_double, c_double], c_int),
    "ssa_profile_end": ([POINTER(ProfileRec), c_int], c_int),
    "ssa_conv2d_igemm": ([POINTER(ConvDesc), _P, _P, _P, _P, _P], c_int),
    "ssa_conv2d_igemm_stats": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P], c_int),
    "ssa_conv2d_igemm_tile": ([POINTER(ConvDesc)], c_int),
    "ssa_conv2d_tile_supported": ([POINTER(ConvDesc)], c_int),
    "ssa_conv2d_tile": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P], c_int),
    "ssa_conv2d_tile_aux": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, c_int, _P, c_int, _P], c_int),
    "ssa_bn_stat_replicas": ([], c_int),
    "ssa_conv2d_tile_p_supported": ([POINTER(ConvDesc)], c_int),
    "ssa_conv_tile_strip": ([c_int], c_int),
    "ssa_conv2d_tile_p": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, c_int, _P, c_int, _P], c_int),
    "ssa_conv2d_halo_supported": ([POINTER(ConvDesc)], c_int),
    "ssa_conv2d_halo": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P], c_int),
    "ssa_conv2d_gemm_wide_supported": ([POINTER(ConvDesc)], c_int),
    "ssa_conv2d_gemm_wide": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P], c_int),
    "ssa_conv2d_halo_reg_supported": ([POINTER(ConvDesc)], c_int),
    "ssa_conv2d_halo_reg": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P], c_int),
    "ssa_pack_filter": ([_P, _P] + [c_int] * 8 + [_P], c_int),
    "ssa_conv2d_wgrad_plan": ([POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_size_t)], c_int),
    "ssa_conv2d_wgrad": ([POINTER(ConvDesc), _P, _P, c_int, c_int, c_int, _P, _P], c_int),
    "ssa_conv2d_wgrad_tile_plan": ([POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_size_t)], c_int),
    "ssa_conv2d_wgrad_tile": ([POINTER(ConvDesc), _P, _P, c_int, c_int, c_int, _P, _P], c_int),
    "ssa_conv2d_wgrad_tile_geometry": ([c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)], c_int),
    "ssa_conv2d_wgrad_head_plan": ([POINTER(ConvDesc), c_int, POINTER(c_int), POINTER(c_size_t)], c_int),
    "ssa_conv2d_wgrad_head": ([POINTER(ConvDesc), _P, _P, c_int, c_int, c_int, _P, _P], c_int),
    "ssa_conv2d_wgrad_reduce": ([_P] + [c_int] * 7 + [_P, c_int, _P], c_int),
    "ssa_colsum_bf16": ([_P, c_long, c_int, c_int, _P, _P, _P], c_int),
    "ssa_pad_cast_f32_bf16": ([_P, c_long, c_int, c_int, _P, c_int, _P], c_int),
    "ssa_bn_stats": ([_P, c_long, c_int, c_int, _P, c_int, _P], c_int),
    "ssa_bn_apply_train": ([_P, c_int, _P, c_int, _P, c_int, c_long, c_int, _P, c_int, c_double, _P, _P, _P, _P,
                            _P, c_float, c_float, _P, _P, c_int, _P, c_long, _P, _P], c_int),
    "ssa_bn_update_running_batched": ([_P, c_int, c_int, _P], c_int),
    "ssa_pack_filters_batched": ([_P, c_int, c_int, _P], c_int),
    "ssa_pack_tile_channels": ([c_int, c_int], c_int),
    "ssa_conv2d_dgrad_s2": ([c_int] * 9 + [_P, _P, _P, _P, _P], c_int),
    "ssa_pack_filters_tiled": ([_P, _P, c_int, c_int, _P], c_int),
    "ssa_conv2d_igemm_affine": ([POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, c_int, c_int, _P], c_int),
    "ssa_bn_finalize_eval_batched": ([_P, c_int, c_int, _P], c_int),
    "ssa_bn_finalize": ([_P, c_double, c_int, _P, _P, _P, _P, c_float, c_float, c_int,
                         _P, _P, _P, _P, _P], c_int),
    "ssa_bn_apply": ([_P, c_int, _P, c_int, _P, c_int, c_long, c_int, _P, _P, c_int, _P,
                      c_long, _P], c_int),
    "ssa_bn_bwd_reduce": ([_P, c_int, _P, c_int, _P, c_int, c_long, c_int, _P, _P, c_int, _P,
                           c_long, _P, c_int, c_int, _P, _P, _P, _P], c_int),
    "ssa_bn_bwd_apply": ([_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_long, c_int,
                          _P, _P, _P, _P, c_int, c_double, c_int, _P, c_long, _P, _P, c_float, _P, _P, c_int, _P, _P], c_int),
    "ssa_bn_param_grads": ([_P, c_int, _P, _P, _P], c_int),
    "ssa_p2p_buffer_bytes": ([c_int, c_long, POINTER(c_size_t)], c_int),
    "ssa_p2p_allreduce_f64": ([_P, c_long, _P, c_int, c_int, _P, c_long, _P], c_int),
    "ssa_p2p_timeouts": ([_P], c_int),
    "ssa_p2p_vmm_alloc": ([c_size_t, POINTER(c_void_p), POINTER(c_int), POINTER(c_size_t)], c_int),
    "ssa_p2p_vmm_import": ([c_int, c_size_t, POINTER(c_void_p)], c_int),
    "ssa_p2p_vmm_unmap": ([_P, c_size_t], c_int),
    "ssa_bn_bwd_fused_blocks": ([c_long, c_int], c_int),
    "ssa_bn_bwd_fused_capacity": ([], c_int),
    "ssa_bn_bwd_fused_timeouts": ([_P], c_int),
    "ssa_bn_bwd_fused": ([_P, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, c_long, c_int,
                          _P, _P, _P, _P, c_int, c_double, c_int, _P, c_long, _P, _P, c_float, _P, _P, c_int, _P, _P, _P], c_int),
    "ssa_sum_act": ([_P, _P, _P, _P, _P, c_long, c_int, _P], c_int),
    "ssa_relu_bwd": ([_P, _P, _P, c_long, _P], c_int),
    "ssa_nchw_f32_to_nhwc_bf16": ([_P, _P, c_int, c_int, c_int, c_int, c_int, _P], c_int),
    "ssa_image_resize_to_nhwc_bf16": ([_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P], c_int),
    "ssa_bilinear_fwd": ([_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int,
                          c_int, _P], c_int),
    "ssa_bilinear_bwd": ([_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int,
                          c_int, _P], c_int),
    "ssa_bilinear_bwd_x": ([_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P], c_int),
    "ssa_bilinear_bwd_y": ([_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P], c_int),
    "ssa_maxpool3x3s2_fwd": ([_P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P], c_int),
    "ssa_maxpool3x3s2_bwd": ([_P, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P], c_int),
    "ssa_global_avg_pool_fwd": ([_P, c_int, c_int, c_long, c_int, _P, _P], c_int),
    "ssa_global_avg_pool_bwd": ([_P, c_int, c_long, c_int, _P, _P], c_int),
    "ssa_resize_nearest_u8": ([_P, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, _P], c_int),
    "ssa_softmax_hw_stats": ([_P, c_int, c_long, c_int, _P, _P], c_int),
    "ssa_softmax_hw_probs": ([_P, c_int, c_long, c_int, _P, _P, c_int, _P], c_int),
    "ssa_rowdot_f32": ([_P, _P, c_int, c_int, _P, _P], c_int),
    "ssa_softmax_hw_bwd": ([_P, c_int, c_long, c_int, _P, _P, c_int, _P, _P, c_int, c_int, _P], c_int),
    "ssa_softmax_lastdim_fwd": ([_P, c_int, c_long, c_int, c_float, _P, c_int, _P], c_int),
    "ssa_softmax_lastdim_bwd": ([_P, c_int, c_long, c_int, c_float, _P, c_int, _P, c_int, _P], c_int),
    "ssa_pack_matrix": ([_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P], c_int),
    "ssa_ocr_attn_supported": ([c_int, c_int], c_int),
    "ssa_ocr_attn_fwd": ([_P, c_int, _P, _P, c_long, c_int, c_int, c_float, _P, c_int, _P], c_int),
    "ssa_ocr_attn_bwd": ([_P, c_int, _P, _P, _P, c_int, c_long, c_int, c_int, c_float, _P, c_int, _P, _P, _P], c_int),
    "ssa_sigmoid_fwd": ([_P, _P, c_long, _P], c_int),
    "ssa_sigmoid_bwd": ([_P, _P, _P, c_long, _P], c_int),
    "ssa_bcast_mul_fwd": ([_P, _P, _P, c_long, c_int, _P], c_int),
    "ssa_bcast_mul_bwd": ([_P, _P, _P, _P, _P, c_long, c_int, _P], c_int),
    "ssa_attn_blend_fwd": ([_P, _P, _P, _P, c_long, c_int, _P], c_int),
    "ssa_attn_blend_bwd": ([_P, _P, _P, _P, _P, c_long, c_int, c_int, _P], c_int),
    "ssa_ce_fwd": ([_P, c_int, _P, c_long, c_int, c_int, _P, _P, _P], c_int),
    "ssa_bce_fwd": ([_P, c_int, _P, c_long, c_int, _P, _P, _P], c_int),
    "ssa_loss_finalize": ([_P, c_double, _P, _P], c_int),
    "ssa_bce_bwd": ([_P, c_int, _P, c_long, c_int, _P, c_double, _P, c_double, _P, _P], c_int),
    "ssa_scale_grad": ([_P, c_long, _P, c_double, _P, c_double, _P], c_int),
    "ssa_scale_grad_to": ([_P, _P, c_long, _P, c_double, _P, c_double, _P], c_int),
    "ssa_rmi_pool": ([_P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P], c_int),
    "ssa_rmi_gram": ([_P, _P, c_int, c_int, c_int, _P, _P], c_int),
    "ssa_rmi_solve": ([_P, c_int, c_int, c_int, _P, _P, _P], c_int),
    "ssa_rmi_finalize": ([_P, c_int, c_int, _P, _P], c_int),
    "ssa_rmi_bwd_pooled": ([_P, _P, _P, c_int, c_int, c_int, _P, _P], c_int),
    "ssa_rmi_bwd_logits": ([_P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P,
                            c_double, _P, c_int, _P], c_int),
    "ssa_rmi_bwd_logits_bce": ([_P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P, c_double, _P, c_double, _P,
                                c_double, _P, _P], c_int),
    "ssa_image_u8_crop_flip_normalize": ([_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P],
                                         c_int),
    "ssa_resample_u8": ([_P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P], c_int),
    "ssa_label_u8_crop_flip": ([_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P], c_int),
    "ssa_confusion_matrix": ([_P, c_int, _P, c_long, c_int, _P, _P, _P], c_int),
    "ssa_sgd_momentum_step": ([_P, _P, _P, _P, c_int, c_float, _P, c_float, c_float, c_int, _P, _P], c_int),
    "ssa_amp_check_grads": ([_P, _P, c_int, _P, _P], c_int),
    "ssa_amp_update": ([_P, c_int, c_float, c_float, c_float, c_float, _P], c_int),
    "ssa_amp_update_counted": ([_P, _P, c_int, c_float, c_float, c_float, c_float, _P], c_int),
    "ssa_ewise_f32": ([c_int, _P, _P, _P, c_long, _P], c_int),
    "ssa_ewise_bwd_f32": ([c_int, _P, _P, _P, _P, _P, c_long, _P], c_int),
    "ssa_axpy_f32": ([_P, c_float, _P, c_long, c_int, _P], c_int),
    "ssa_probe_mfma32": ([_P, _P, _P, _P], c_int),
    "ssa_probe_mfma16": ([_P, _P, _P, _P], c_int),
    "ssa_probe_swap16": ([_P, _P], c_int),
    "ssa_probe_tr16": ([_P, c_int, _P], c_int),
}


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load libsemseg_hip.so (once).  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libsemseg_hip.so not found at %s -- the HIP extension is required "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no "
                "fallback path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        missing = []
        for name, (argtypes, restype) in _SIGS.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = restype
        if missing:
            raise RuntimeError("%s is stale, missing symbols: %s" % (os.path.basename(LIB_PATH), ", ".join(missing)))
        if handle.ssa_elem_type() != (1 if ACT == "fp16" else 0):
            raise RuntimeError("%s was built for the other storage format (SSA_ACT_DTYPE=%s)" % (LIB_PATH, ACT))
        # the binary is git-ignored and travels as a file: tie it to the sources it claims to be built from
        built, here = handle.ssa_source_sha().decode(), source_sha()
        if here is not None and built != here and os.environ.get("SSA_ALLOW_STALE_LIB", "0") != "1":
            raise RuntimeError("%s was built from other sources (embedded %s, csrc/ now %s): rebuild with "
                               "`make -C semantic-segmentation_amd/csrc`" % (os.path.basename(LIB_PATH), built, here))
        _LIB = handle
    return _LIB


def built_sha():
    return lib().ssa_source_sha().decode()


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(rc, "hipError %d" % rc)
        raise RuntimeError("%s failed: %s" % (what, kind))
