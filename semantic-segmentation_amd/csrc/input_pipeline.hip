// Tail of the input pipeline on the device: what the reference's
// DataLoader workers do on the CPU after the (PIL) scale step --
//   joint crop (transforms/joint_transforms.py RandomSizeAndCrop/RandomCrop: a window of the
//   image and of the label map), RandomHorizontallyFlip (joint_transforms.py:276-281:
//   Image.FLIP_LEFT_RIGHT of the cropped pair), then ToTensor + Normalize(mean, std) on the
//   image (datasets/base_loader.py:141-142, config.py:96-97) and MaskToTensor on the labels
// -- applied to the uint8 buffers after ONE host-to-device copy of the raw crop source.
// HBM-bound byte work: 3 B read + 32 B written per pixel (the trunk's first conv reads NHWC
// bf16 with the 3 channels padded to 16), 1 B read + 8 B written per label.
// Arithmetic: bf16(((float)u8 / 255 - mean[c]) / std[c]) with IEEE fp32 division, i.e. exactly
// torch's ToTensor().div(255) followed by Normalize's sub/div, then a round-to-nearest-even
// cast -- bit-identical to the CPU pipeline followed by `.to(bfloat16)`.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

struct Norm3 { float mean[3], stdv[3]; };

__global__ __launch_bounds__(256) void image_crop_flip_normalize_kernel(
    const unsigned char* __restrict__ img, int W, int x0, int y0, int cw, int ch, int flip, Norm3 nm,
    bf16_t* __restrict__ out, int cpad) {
  const long n = (long)cw * ch;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / cw), x = (int)(i - (long)y * cw);
    const int sx = x0 + (flip ? cw - 1 - x : x), sy = y0 + y;
    const unsigned char* p = img + ((long)sy * W + sx) * 3;
    bf16_t* o = out + i * cpad;
    float f[8];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p[c], 255.f), nm.mean[c]), nm.stdv[c]);
#pragma unroll
    for (int c = 3; c < 8; ++c) f[c] = 0.f;
    *reinterpret_cast<uint4*>(o) = pack8(f);
    for (int c0 = 8; c0 < cpad; c0 += 8) *reinterpret_cast<uint4*>(o + c0) = make_uint4(0, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void label_crop_flip_kernel(const unsigned char* __restrict__ lab, int W,
                                                              int x0, int y0, int cw, int ch, int flip,
                                                              long* __restrict__ out) {
  const long n = (long)cw * ch;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / cw), x = (int)(i - (long)y * cw);
    out[i] = (long)lab[(long)(y0 + y) * W + x0 + (flip ? cw - 1 - x : x)];
  }
}

bool window_ok(int H, int W, int x0, int y0, int cw, int ch) {
  return H > 0 && W > 0 && cw > 0 && ch > 0 && x0 >= 0 && y0 >= 0 && (long)x0 + cw <= W && (long)y0 + ch <= H;
}

// One pass of Pillow's 8-bit resampling (libImaging/Resample.c ImagingResampleHorizontal_8bpc /
// Vertical_8bpc) over an interleaved uint8 [H][W][C] image: out = clip8((2^21 + sum_k in * kk) >> 22)
// with the per-output (first tap, tap count) bounds and the 22-bit fixed-point coefficients the host
// derived exactly as precompute_coeffs + normalize_coeffs_8bpc do.  Integer arithmetic: bit-identical
// to `img.resize(size, Image.BICUBIC)` (transforms/joint_transforms.py:433-471) pass by pass.
__global__ __launch_bounds__(256) void resample_u8_kernel(
    const unsigned char* __restrict__ src, int Hs, int Ws, int C, int axis, unsigned char* __restrict__ dst,
    int Hd, int Wd, const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize) {
  const long n = (long)Hd * Wd * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long t = i / C;
    const int x = (int)(t % Wd), y = (int)(t / Wd);
    const int o = axis ? x : y;
    const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int* k = coefs + (long)o * ksize;
    int ss = 1 << 21;
    if (axis) {
      const unsigned char* p = src + ((long)y * Ws + first) * C + c;
      for (int j = 0; j < cnt; ++j) ss += (int)p[(long)j * C] * k[j];
    } else {
      const unsigned char* p = src + ((long)first * Ws + x) * C + c;
      for (int j = 0; j < cnt; ++j) ss += (int)p[(long)j * Ws * C] * k[j];
    }
    ss >>= 22;
    dst[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

}  // namespace

extern "C" {

int ssa_image_u8_crop_flip_normalize(const unsigned char* img_hwc, int H, int W, int x0, int y0, int cw,
                                     int ch, int flip, const float* mean3, const float* std3,
                                     void* out_nhwc_bf16, int cpad, void* stream) {
  if (!img_hwc || !out_nhwc_bf16 || !mean3 || !std3 || !window_ok(H, W, x0, y0, cw, ch)) return SSA_EINVAL;
  if (cpad < 8 || cpad % 8 || (reinterpret_cast<uintptr_t>(out_nhwc_bf16) & 15u)) return SSA_EINVAL;
  Norm3 nm;
  for (int c = 0; c < 3; ++c) {
    if (!(std3[c] > 0.f)) return SSA_EINVAL;
    nm.mean[c] = mean3[c];
    nm.stdv[c] = std3[c];
  }
  const long n = (long)cw * ch;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(image_crop_flip_normalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img_hwc,
                     W, x0, y0, cw, ch, flip ? 1 : 0, nm, (bf16_t*)out_nhwc_bf16, cpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_label_u8_crop_flip(const unsigned char* lab_hw, int H, int W, int x0, int y0, int cw, int ch,
                           int flip, int64_t* out, void* stream) {
  if (!lab_hw || !out || !window_ok(H, W, x0, y0, cw, ch)) return SSA_EINVAL;
  const long n = (long)cw * ch;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(label_crop_flip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, lab_hw, W, x0, y0,
                     cw, ch, flip ? 1 : 0, (long*)out);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_resample_u8(const unsigned char* src, int Hs, int Ws, int C, int axis, unsigned char* dst, int n_out,
                    const int* bounds, const int* coefs, int ksize, void* stream) {
  if (!src || !dst || !bounds || !coefs || Hs <= 0 || Ws <= 0 || C <= 0 || n_out <= 0 || ksize <= 0 ||
      (axis != 0 && axis != 1))
    return SSA_EINVAL;
  const int Hd = axis ? Hs : n_out, Wd = axis ? n_out : Ws;
  const long n = (long)Hd * Wd * C;
  const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
  hipLaunchKernelGGL(resample_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, C, axis, dst,
                     Hd, Wd, bounds, coefs, ksize);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
