// Pooling ops of the DeepLabV3+/ResNet-50 path on NHWC bf16 activations (gfx950):
//   nn.MaxPool2d(3, stride 2, padding 1)   network/Resnet.py:147 (ResNet stem)
//   nn.AdaptiveAvgPool2d(1)                network/utils.py:201  (ASPP image pooling)
// Both are HBM-bound streaming kernels: one thread owns one 16-byte channel group.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

// forward: y = max over the 3x3 window (padding never wins: -inf), idx = winning tap
// (first maximum in (kh,kw) scan order, NaN propagates -- PyTorch's rule, so ties
// among post-ReLU zeros route the gradient like the reference does)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_t* __restrict__ x, int ldx, int B, int H,
                                                          int W, int C, bf16_t* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int Ho, int Wo) {
  const int VC = C >> 3;
  const long n = (long)B * Ho * Wo * VC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % VC);
    long p = i / VC;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float best[8];
    int arg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; arg[j] = -1; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iy = oy * 2 - 1 + kh, ix = ox * 2 - 1 + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + ((long)(b * H + iy) * W + ix) * ldx + cg * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (arg[j] < 0) arg[j] = kh * 3 + kw;                 // first in-bounds tap
          if (f[j] > best[j] || f[j] != f[j]) { best[j] = f[j]; arg[j] = kh * 3 + kw; }
        }
      }
    const long o = ((long)(b * Ho + oy) * Wo + ox) * C + cg * 8;
    *reinterpret_cast<uint4*>(y + o) = pack8(best);
    uint2 a;
    a.x = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
    a.y = (unsigned)arg[4] | ((unsigned)arg[5] << 8) | ((unsigned)arg[6] << 16) | ((unsigned)arg[7] << 24);
    *reinterpret_cast<uint2*>(idx + o) = a;
  }
}

// backward as a gather (deterministic, no atomics): an input pixel collects dy from the
// (at most four) windows that cover it and whose stored argmax is this pixel
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_t* __restrict__ dy,
                                                          const unsigned char* __restrict__ idx, int B,
                                                          int Ho, int Wo, int C, bf16_t* __restrict__ dx,
                                                          int H, int W) {
  const int VC = C >> 3;
  const long n = (long)B * H * W * VC;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % VC);
    long p = i / VC;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const int b = (int)(p / H);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    const int oy0 = iy >> 1, ox0 = ix >> 1;          // windows oy with 2oy-1 <= iy <= 2oy+1
    for (int oy = oy0; oy <= (iy + 1) >> 1; ++oy)
      for (int ox = ox0; ox <= (ix + 1) >> 1; ++ox) {
        if (oy >= Ho || ox >= Wo) continue;
        const int tap = (iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1));
        const long o = ((long)(b * Ho + oy) * Wo + ox) * C + cg * 8;
        const uint2 a = *reinterpret_cast<const uint2*>(idx + o);
        float d[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + o), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned t = ((j < 4 ? a.x : a.y) >> (8 * (j & 3))) & 0xffu;
          if ((int)t == tap) g[j] += d[j];
        }
      }
    *reinterpret_cast<uint4*>(dx + ((long)(b * H + iy) * W + ix) * C + cg * 8) = pack8(g);
  }
}

// out[b, c] = mean over the image's pixels; one workgroup per (image, 8-channel group)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const bf16_t* __restrict__ x, int ldx, long HW, int C,
                                                      bf16_t* __restrict__ out) {
  __shared__ float sh[4][8];
  const int cg = blockIdx.x, b = blockIdx.y;
  const bf16_t* xb = x + (long)b * HW * ldx + cg * 8;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  for (long p = threadIdx.x; p < HW; p += 256) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(xb + p * ldx), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = wave_sum(s[j]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[wave][j] = s[j];
  __syncthreads();
  if (threadIdx.x < 8) {
    const float v = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    out[(long)b * C + cg * 8 + threadIdx.x] = f2bf(v / (float)HW);
  }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const bf16_t* __restrict__ dout, long HW, int C,
                                                      bf16_t* __restrict__ dx, int B) {
  const int VC = C >> 3;
  const long n = (long)B * HW * VC;
  const float inv = 1.f / (float)HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % VC);
    const long p = i / VC;
    const int b = (int)(p / HW);
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(dout + (long)b * C + cg * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= inv;
    *reinterpret_cast<uint4*>(dx + p * C + cg * 8) = pack8(f);
  }
}

int grid_for(long n) { return (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256); }

}  // namespace

extern "C" {

int ssa_maxpool3x3s2_fwd(const void* x, int ldx, int B, int H, int W, int C, void* y, unsigned char* idx,
                         int Ho, int Wo, void* stream) {
  if (!x || !y || !idx || C % 8 || ldx % 8 || B < 1 || Ho != (H + 1) / 2 || Wo != (W + 1) / 2) return SSA_EINVAL;
  const long n = (long)B * Ho * Wo * (C >> 3);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, ldx, B, H, W, C, (bf16_t*)y, idx, Ho, Wo);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_maxpool3x3s2_bwd(const void* dy, const unsigned char* idx, int B, int Ho, int Wo, int C, void* dx,
                         int H, int W, void* stream) {
  if (!dy || !idx || !dx || C % 8 || B < 1) return SSA_EINVAL;
  const long n = (long)B * H * W * (C >> 3);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, idx, B, Ho, Wo, C, (bf16_t*)dx, H, W);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_global_avg_pool_fwd(const void* x, int ldx, int B, long HW, int C, void* out, void* stream) {
  if (!x || !out || C % 8 || ldx % 8 || B < 1 || HW < 1) return SSA_EINVAL;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(C >> 3, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     ldx, HW, C, (bf16_t*)out);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_global_avg_pool_bwd(const void* dout, int B, long HW, int C, void* dx, void* stream) {
  if (!dout || !dx || C % 8 || B < 1 || HW < 1) return SSA_EINVAL;
  const long n = (long)B * HW * (C >> 3);
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dout, HW, C, (bf16_t*)dx, B);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
