"""Label-map resize on the GPU, bit-identical to `mask.resize(size, Image.NEAREST)`
(transforms/joint_transforms.py:193,267,290,319,339,364,466 of the reference).

Pillow's nearest resize walks a double-precision accumulator: the source
coordinate of destination column 0 is scale*0.5 and every further column adds
`scale` to the running sum (ImagingScaleAffine), truncating toward zero.  The
closed form floor((x+0.5)*scale) differs from that in the last bit for some
sizes, so the tables below reproduce the running sum (np.cumsum adds
sequentially, in the same order)."""
import ctypes

import numpy as np
import torch


def nearest_index_table(n_dst, n_src):
    """int32[n_dst]: source index of every destination index (Pillow's rule)."""
    scale = np.float64(n_src) / np.float64(n_dst)
    steps = np.full(n_dst, scale, dtype=np.float64)
    steps[0] = scale * np.float64(0.5)
    pos = np.cumsum(steps)                      # pos[k] = ((scale/2 + scale) + scale) + ...
    idx = np.where(pos < 0, -1, pos.astype(np.int64))
    return np.clip(idx, 0, n_src - 1).astype(np.int32)


_TABLES = {}


def _table(n_dst, n_src, device):
    key = (n_dst, n_src, str(device))
    t = _TABLES.get(key)
    if t is None:
        t = _TABLES[key] = torch.from_numpy(nearest_index_table(n_dst, n_src)).to(device)
    return t


def resize_labels_nearest(mask, size):
    """mask: uint8 CUDA tensor [H,W] or [B,H,W]; size = (Hd, Wd) -> same rank, uint8."""
    from .._lib import lib, check
    assert mask.dtype == torch.uint8 and mask.is_cuda, "label maps are uint8 device tensors"
    squeeze = mask.dim() == 2
    m = mask.unsqueeze(0) if squeeze else mask
    m = m.contiguous()
    B, Hs, Ws = m.shape
    Hd, Wd = int(size[0]), int(size[1])
    out = torch.empty((B, Hd, Wd), dtype=torch.uint8, device=m.device)
    iy, ix = _table(Hd, Hs, m.device), _table(Wd, Ws, m.device)
    check(lib().ssa_resize_nearest_u8(ctypes.c_void_p(m.data_ptr()), B, Hs, Ws, ctypes.c_void_p(out.data_ptr()),
                                      Hd, Wd, ctypes.c_void_p(iy.data_ptr()), ctypes.c_void_p(ix.data_ptr()),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "ssa_resize_nearest_u8")
    return out[0] if squeeze else out


# ---------------------------------------------------------------------------------------------
# Tail of the image pipeline on the device: crop window + horizontal flip of
# the (image, labels) pair, ToTensor + Normalize on the image, MaskToTensor on the labels
# (datasets/base_loader.py:120-150, transforms/joint_transforms.py:276-281).  The host sends the
# raw uint8 buffers once; what comes back is what `net({'images': ..., 'gts': ...})` consumes.
# ---------------------------------------------------------------------------------------------
MEAN_STD = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])      # config.py:96-97 (cfg.DATASET.MEAN / STD)


def crop_flip_normalize(img_u8, labels_u8, window, flip, mean_std=MEAN_STD):
    """img_u8: uint8 CUDA [H,W,3] (RGB, as np.array(PIL image)); labels_u8: uint8 CUDA [H,W] or None;
    window = (x0, y0, w, h) as PIL's crop box origin + size; flip: mirror the cropped pair.
    Returns (image [1,h,w,16] bf16 NHWC -- hand it to the network's trunk --, labels [1,h,w] int64)."""
    from .._lib import lib, check
    assert img_u8.dtype == torch.uint8 and img_u8.is_cuda and img_u8.dim() == 3 and img_u8.shape[2] == 3
    img_u8 = img_u8.contiguous()
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    x0, y0, cw, ch = (int(v) for v in window)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    mean = (ctypes.c_float * 3)(*mean_std[0])
    std = (ctypes.c_float * 3)(*mean_std[1])
    out = torch.empty((1, ch, cw, 16), dtype=torch.bfloat16, device=img_u8.device)
    check(lib().ssa_image_u8_crop_flip_normalize(ctypes.c_void_p(img_u8.data_ptr()), H, W, x0, y0, cw, ch, int(bool(flip)),
                                                 mean, std, ctypes.c_void_p(out.data_ptr()), 16, stream),
          "ssa_image_u8_crop_flip_normalize")
    gts = None
    if labels_u8 is not None:
        assert labels_u8.dtype == torch.uint8 and labels_u8.is_cuda and tuple(labels_u8.shape) == (H, W)
        labels_u8 = labels_u8.contiguous()
        gts = torch.empty((1, ch, cw), dtype=torch.int64, device=img_u8.device)
        check(lib().ssa_label_u8_crop_flip(ctypes.c_void_p(labels_u8.data_ptr()), H, W, x0, y0, cw, ch,
                                           int(bool(flip)), ctypes.c_void_p(gts.data_ptr()), stream),
              "ssa_label_u8_crop_flip")
    return out, gts
