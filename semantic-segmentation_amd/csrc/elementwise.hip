// Streaming elementwise kernels of the HRNet-OCR-MScale path (HBM-bound):
//   * fuse-sum + ReLU of HighResolutionModule.forward (network/hrnetv2.py:236-252)
//   * NCHW fp32 image -> NHWC bf16 (train.py:487 hands NCHW fp32 to the module)
//   * sigmoid of the scale-attention logit (network/utils.py:363)
//   * attention-weighted two-scale fusion (network/ocrnet.py:289-298)
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"

namespace {

__device__ __forceinline__ void sum_act_kernel_body(const uint4* __restrict__ a, const uint4* __restrict__ b,
                               const uint4* __restrict__ c, const uint4* __restrict__ d,
                               uint4* __restrict__ z, long nv, int relu, const int bx, const int gx) {
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gx * blockDim.x) {
    float f[8], g[8];
    unpack8(a[i], f);
    if (b) { unpack8(b[i], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j]; }
    if (c) { unpack8(c[i], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j]; }
    if (d) { unpack8(d[i], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j]; }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
    }
    z[i] = pack8(f);
  }
}

__device__ __forceinline__ void relu_bwd_kernel_body(const uint4* __restrict__ dz, const uint4* __restrict__ z,
                                uint4* __restrict__ g, long nv, const int bx, const int gx) {
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gx * blockDim.x) {
    float f[8], zz[8];
    unpack8(dz[i], f);
    unpack8(z[i], zz);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = zz[j] > 0.f ? f[j] : 0.f;
    g[i] = pack8(f);
  }
}


struct SumActK {
  struct Args { const uint4* a; const uint4* b; const uint4* c; const uint4* d; uint4* z; long nv; int relu; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    sum_act_kernel_body(a.a, a.b, a.c, a.d, a.z, a.nv, a.relu, bx, gx);
  }
};
struct ReluBwdK {
  struct Args { const uint4* dz; const uint4* z; uint4* g; long nv; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    relu_bwd_kernel_body(a.dz, a.z, a.g, a.nv, bx, gx);
  }
};

// one thread per output pixel-group: reads C planes (coalesced along W), writes
// one cpad-wide NHWC pixel.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int B,
                                    int C, long HW, int cpad) {
  const long n = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    bf16_t* dst = y + i * cpad;
    for (int c0 = 0; c0 < cpad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        f[j] = c < C ? x[(b * C + c) * HW + p] : 0.f;
      }
      *reinterpret_cast<uint4*>(dst + c0) = pack8(f);
    }
  }
}

__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = 1.f / (1.f + __expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                   float* __restrict__ dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float s = y[i];
    dx[i] = dy[i] * s * (1.f - s);
  }
}

__global__ void bcast_mul_fwd_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                     float* __restrict__ out, long n, int C) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = a[i / C] * x[i];
}
// one wave handles pixels; lanes cover channels (C <= 64)
__global__ void bcast_mul_bwd_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                     const float* __restrict__ dout, float* __restrict__ da,
                                     float* __restrict__ dx, long P, int C) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long p = wave; p < P; p += nwaves) {
    float acc = 0.f;
    const float av = a[p];
    for (int c = lane; c < C; c += 64) {
      const float g = dout[p * C + c];
      acc += g * x[p * C + c];
      dx[p * C + c] = g * av;
    }
    acc = wave_sum(acc);
    if (lane == 0) da[p] = acc;
  }
}

__global__ void attn_blend_fwd_kernel(const float* __restrict__ lo, const float* __restrict__ a,
                                      const float* __restrict__ hi, float* __restrict__ joint,
                                      long n, int C) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    joint[i] = lo[i] + (1.f - a[i / C]) * hi[i];
}
__global__ void attn_blend_bwd_kernel(const float* __restrict__ a, const float* __restrict__ hi,
                                      const float* __restrict__ dj, float* __restrict__ da,
                                      float* __restrict__ dhi, long P, int C, int accumulate_da) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long p = wave; p < P; p += nwaves) {
    float acc = 0.f;
    const float om = 1.f - a[p];
    for (int c = lane; c < C; c += 64) {
      const float g = dj[p * C + c];
      acc -= g * hi[p * C + c];
      if (dhi) dhi[p * C + c] = g * om;
    }
    acc = wave_sum(acc);
    if (lane == 0) da[p] = accumulate_da ? da[p] + acc : acc;
  }
}

// The same for few classes (C <= 64: a wave per pixel leaves 64 - C lanes idle and pays a 6-step shuffle reduction per
// pixel -- 187 us at 1024x1024x19): a block owns 256 consecutive pixels, streams their dj / hi rows as float4 (coalesced,
// dhi written in the same pass), parks the products in LDS and lets thread t sum pixel t's row (stride C floats).
__global__ __launch_bounds__(256) void attn_blend_bwd_tile_kernel(const float* __restrict__ a, const float* __restrict__ hi,
                                                                  const float* __restrict__ dj, float* __restrict__ da,
                                                                  float* __restrict__ dhi, long P, int C, int accumulate_da) {
  SSA_DYN_LDS(float, prod);                    // [256 * C] g * hi, then [256] 1 - a
  float* om = prod + 256 * C;
  const int tid = threadIdx.x;
  for (long base = (long)blockIdx.x * 256; base < P; base += (long)gridDim.x * 256) {
    const int npx = (int)min(256L, P - base);
    const int n = npx * C;
    if (tid < npx) om[tid] = 1.f - a[base + tid];
    __syncthreads();
    const float* djb = dj + base * C;
    const float* hib = hi + base * C;
    float* dhb = dhi ? dhi + base * C : nullptr;
    const int n4 = (n % 4 == 0) ? n / 4 : 0;   // the last, ragged block goes element by element
    for (int g = tid; g < n4; g += 256) {
      const float4 gv = reinterpret_cast<const float4*>(djb)[g];
      const float4 hv = reinterpret_cast<const float4*>(hib)[g];
      reinterpret_cast<float4*>(prod)[g] = make_float4(gv.x * hv.x, gv.y * hv.y, gv.z * hv.z, gv.w * hv.w);
      if (dhb) {
        const int i0 = g * 4;
        int px = i0 / C, c = i0 - px * C;
        float o[4];
        const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (c == C) { c = 0; ++px; }
          o[k] = gg[k] * om[px];
          ++c;
        }
        reinterpret_cast<float4*>(dhb)[g] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    for (int i = n4 * 4 + tid; i < n; i += 256) {
      const float gsc = djb[i];
      prod[i] = gsc * hib[i];
      if (dhb) dhb[i] = gsc * om[i / C];
    }
    __syncthreads();
    if (tid < npx) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc -= prod[tid * C + c];
      da[base + tid] = accumulate_da ? da[base + tid] + acc : acc;
    }
    __syncthreads();
  }
}

__global__ void axpy_kernel(const float* __restrict__ x, float alpha, float* __restrict__ y, long n,
                            int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = accumulate ? y[i] + alpha * x[i] : alpha * x[i];
}

// fp32 elementwise add / mul / div of two same-shape tensors and their backward: the attention
// normalisation and the weighted sum over scales of network/attnscale.py:153-166,330-352.
__global__ void ewise_fwd_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                 float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = a[i], y = b[i];
    out[i] = op == 0 ? x + y : (op == 1 ? x * y : x / y);
  }
}
__global__ void ewise_bwd_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                 const float* __restrict__ dout, float* __restrict__ da,
                                 float* __restrict__ db, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = dout[i];
    float ga, gb;
    if (op == 0) { ga = g; gb = g; }
    else if (op == 1) { ga = g * b[i]; gb = g * a[i]; }
    else { const float r = 1.f / b[i]; ga = g * r; gb = -g * a[i] * r * r; }
    if (da) da[i] = ga;
    if (db) db[i] = gb;
  }
}

inline int grid_for(long n, int per_block = 256, int cap = 4096) {
  long b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int ssa_sum_act(const void* a, const void* b, const void* c, const void* d, void* z, long n,
                int relu, void* stream) {
  if (!a || !z || n <= 0 || n % 8) return SSA_EINVAL;
  const long nv = n / 8;
  SumActK::Args k{(const uint4*)a, (const uint4*)b, (const uint4*)c, (const uint4*)d, (uint4*)z, nv, relu};
  return ssa::submit<SumActK>(k, grid_for(nv), 1, 0, (hipStream_t)stream);
}

int ssa_relu_bwd(const void* dz, const void* z, void* g, long n, void* stream) {
  if (!dz || !z || !g || n <= 0 || n % 8) return SSA_EINVAL;
  const long nv = n / 8;
  ReluBwdK::Args k{(const uint4*)dz, (const uint4*)z, (uint4*)g, nv};
  return ssa::submit<ReluBwdK>(k, grid_for(nv), 1, 0, (hipStream_t)stream);
}

int ssa_nchw_f32_to_nhwc_bf16(const float* x, void* y, int B, int C, int H, int W, int cpad,
                              void* stream) {
  if (!x || !y || cpad % 8 || cpad < C) return SSA_EINVAL;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0,
                     (hipStream_t)stream, x, (bf16_t*)y, B, C, HW, cpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_sigmoid_fwd(const float* x, float* y, long n, void* stream) {
  if (!x || !y || n <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
int ssa_sigmoid_bwd(const float* y, const float* dy, float* dx, long n, void* stream) {
  if (!y || !dy || !dx || n <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_bcast_mul_fwd(const float* a, const float* x, float* out, long P, int C, void* stream) {
  if (!a || !x || !out || P <= 0 || C <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(bcast_mul_fwd_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream,
                     a, x, out, P * C, C);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
int ssa_bcast_mul_bwd(const float* a, const float* x, const float* dout, float* da, float* dx,
                      long P, int C, void* stream) {
  if (!a || !x || !dout || !da || !dx || P <= 0 || C <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(bcast_mul_bwd_kernel, dim3(grid_for(P, 4)), dim3(256), 0, (hipStream_t)stream,
                     a, x, dout, da, dx, P, C);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_attn_blend_fwd(const float* lo, const float* a, const float* hi, float* joint, long P,
                       int C, void* stream) {
  if (!lo || !a || !hi || !joint || P <= 0 || C <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(attn_blend_fwd_kernel, dim3(grid_for(P * C)), dim3(256), 0,
                     (hipStream_t)stream, lo, a, hi, joint, P * C, C);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
int ssa_attn_blend_bwd(const float* a, const float* hi, const float* djoint, float* da, float* dhi,
                       long P, int C, int accumulate_da, void* stream) {
  if (!a || !hi || !djoint || !da || P <= 0 || C <= 0) return SSA_EINVAL;
  if (C <= 56 && ((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(djoint) | reinterpret_cast<uintptr_t>(dhi)) & 15u) == 0) {
    hipLaunchKernelGGL(attn_blend_bwd_tile_kernel, dim3(grid_for(P, 256, 2048)), dim3(256), (size_t)(256 * C + 256) * sizeof(float),
                       (hipStream_t)stream, a, hi, djoint, da, dhi, P, C, accumulate_da);
    SSA_LAUNCH_CHECK();
    return SSA_OK;
  }
  hipLaunchKernelGGL(attn_blend_bwd_kernel, dim3(grid_for(P, 4)), dim3(256), 0, (hipStream_t)stream,
                     a, hi, djoint, da, dhi, P, C, accumulate_da);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_ewise_f32(int op, const float* a, const float* b, float* out, long n, void* stream) {
  if (!a || !b || !out || n <= 0 || op < 0 || op > 2) return SSA_EINVAL;
  hipLaunchKernelGGL(ewise_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, out, n);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_ewise_bwd_f32(int op, const float* a, const float* b, const float* dout, float* da, float* db, long n,
                      void* stream) {
  if (!a || !b || !dout || n <= 0 || op < 0 || op > 2) return SSA_EINVAL;
  hipLaunchKernelGGL(ewise_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, dout, da, db, n);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_axpy_f32(const float* x, float alpha, float* y, long n, int accumulate, void* stream) {
  if (!x || !y || n <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, alpha, y,
                     n, accumulate);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
