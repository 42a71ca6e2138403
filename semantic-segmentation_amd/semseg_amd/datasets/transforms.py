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
