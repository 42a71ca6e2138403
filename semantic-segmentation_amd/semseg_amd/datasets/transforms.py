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
    from ..hip_backend import ACT_DTYPE
    out = torch.empty((1, ch, cw, 16), dtype=ACT_DTYPE, device=img_u8.device)
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


# ---------------------------------------------------------------------------------------------
# The scale step on the device: `img.resize((w, h), Image.BICUBIC)` (transforms/joint_transforms.py:
# 433-471 and the Scale / ResizeHeight transforms), bit-identical to Pillow.  Pillow resamples 8-bit
# images in two passes with 22-bit fixed-point taps (libImaging/Resample.c); the tap tables are
# derived here in double precision, operation by operation as precompute_coeffs /
# normalize_coeffs_8bpc do, and the two integer passes run on the GPU (ssa_resample_u8).
# ---------------------------------------------------------------------------------------------
def bicubic_tables(n_dst, n_src):
    """-> (ksize, bounds int32 [n_dst, 2] = (first, count), coefs int32 [n_dst, ksize])."""
    scale = np.float64(n_src) / np.float64(n_dst)
    fscale = scale if scale >= 1.0 else np.float64(1.0)
    support = np.float64(2.0) * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = np.float64(0.0) + (np.arange(n_dst, dtype=np.float64) + 0.5) * scale
    ss = np.float64(1.0) / fscale
    first = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncates; values >= -0.5 here
    first = np.where(center - support + 0.5 < 0, 0, first)
    last = np.minimum((center + support + 0.5).astype(np.int64), n_src)
    count = last - first
    j = np.arange(ksize, dtype=np.int64)[None, :]
    x = (j + first[:, None] - center[:, None] + 0.5) * ss
    ax = np.abs(x)
    a = np.float64(-0.5)
    w = np.where(ax < 1.0, ((a + 2.0) * ax - (a + 3.0)) * ax * ax + 1,
                 np.where(ax < 2.0, (((ax - 5) * ax + 8) * ax - 4) * a, 0.0))
    w = np.where(j < count[:, None], w, 0.0)
    ww = np.zeros(n_dst, dtype=np.float64)
    for k in range(ksize):                       # the C loop adds the taps in order
        ww = ww + w[:, k]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(w < 0, (-0.5 + w * float(1 << 22)).astype(np.int64), (0.5 + w * float(1 << 22)).astype(np.int64))
    fixed = np.where(j < count[:, None], fixed, 0)
    return ksize, np.stack([first, count], 1).astype(np.int32), fixed.astype(np.int32)


_BICUBIC = {}


def _bicubic(n_dst, n_src, device):
    key = (n_dst, n_src, str(device))
    t = _BICUBIC.get(key)
    if t is None:
        ksize, bounds, coefs = bicubic_tables(n_dst, n_src)
        t = _BICUBIC[key] = (ksize, torch.from_numpy(bounds).to(device), torch.from_numpy(coefs).to(device))
    return t


def resize_image_bicubic(img_u8, size):
    """img_u8: uint8 CUDA [H,W,C] (as np.array(PIL image)); size = (Hd, Wd) -> uint8 [Hd,Wd,C], bit-identical to
    `Image.fromarray(img).resize((Wd, Hd), Image.BICUBIC)`."""
    from .._lib import lib, check
    assert img_u8.dtype == torch.uint8 and img_u8.is_cuda and img_u8.dim() == 3
    cur = img_u8.contiguous()
    Hd, Wd = int(size[0]), int(size[1])
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for axis, n_out in ((1, Wd), (0, Hd)):       # horizontal pass first, then vertical (ImagingResampleInner)
        Hs, Ws, C = (int(v) for v in cur.shape)
        if n_out == (Ws if axis else Hs):
            continue
        ksize, bounds, coefs = _bicubic(n_out, Ws if axis else Hs, cur.device)
        out = torch.empty((Hs, n_out, C) if axis else (n_out, Ws, C), dtype=torch.uint8, device=cur.device)
        check(lib().ssa_resample_u8(ctypes.c_void_p(cur.data_ptr()), Hs, Ws, C, axis, ctypes.c_void_p(out.data_ptr()),
                                    n_out, ctypes.c_void_p(bounds.data_ptr()), ctypes.c_void_p(coefs.data_ptr()), ksize,
                                    stream), "ssa_resample_u8")
        cur = out
    return cur


class DevicePrefetcher:
    """H2D overlap for the input pipeline (datasets/base_loader.py:120-150 hands CPU tensors to
    train.py:487, which copies them synchronously): wraps any iterable of (images, gts, ...) batches, keeps
    ONE batch in flight -- pinned host staging + non-blocking copies on a side stream -- and yields device
    tensors whose readiness the consumer stream waits on.  Non-tensor entries pass through."""

    def __init__(self, loader, device="cuda"):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        if self.stream is None:
            return batch, None
        with torch.cuda.stream(self.stream):
            out = tuple(t.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))       # the next batch's copy runs under this batch's compute
            except StopIteration:
                nxt = None
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for t in cur:
                    if torch.is_tensor(t):
                        t.record_stream(torch.cuda.current_stream(self.device))
            yield cur
