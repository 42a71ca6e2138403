// Softmax pieces of the OCR head.  The matrix products of SpatialGather
// (network/ocr_utils.py:39-45, K9) and ObjectAttentionBlock
// (network/ocr_utils.py:100-113, K10) run on the MFMA implicit-GEMM kernels of
// conv_igemm.hip (probabilities / keys / values packed as GEMM operands by
// ssa_pack_matrix); only the two softmaxes and their backward live here:
//   * softmax over HW (65,536-long rows) of the aux logits, per (image, class)
//   * softmax over the K=19 object regions, per pixel
#include "common.h"
#include "../../include/semseg_hip.h"
#include <float.h>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  // monotone int mapping trick
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void hw_init_kernel(float* __restrict__ rowstat, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) { rowstat[2 * k] = -FLT_MAX; rowstat[2 * k + 1] = 0.f; }
}

// pass 0: per-class max over pixels; pass 1: per-class sum of exp(x - max)
template <int PASS>
__global__ __launch_bounds__(NT) void hw_reduce_kernel(const float* __restrict__ logits, int ld,
                                                       long HW, int K, float* __restrict__ rowstat,
                                                       long pix_per_block) {
  __shared__ float red[NT];
  const int t = threadIdx.x;
  const int RP = NT / K, NA = RP * K;
  const bool active = t < NA;
  const int k = t % K, pr = t / K;
  float acc = PASS == 0 ? -FLT_MAX : 0.f;
  const float mx = PASS == 1 ? rowstat[2 * k] : 0.f;
  if (active) {
    const long p0 = blockIdx.x * pix_per_block, p1 = min(HW, p0 + pix_per_block);
    for (long p = p0 + pr; p < p1; p += RP) {
      const float v = logits[p * ld + k];
      if (PASS == 0) acc = fmaxf(acc, v);
      else acc += __expf(v - mx);
    }
  }
  red[t] = acc;
  __syncthreads();
  if (t < K) {
    float r = red[t];
    for (int j = 1; j < RP; ++j) r = PASS == 0 ? fmaxf(r, red[t + j * K]) : r + red[t + j * K];
    if (PASS == 0) atomic_max_f32(&rowstat[2 * t], r);
    else atomicAdd(&rowstat[2 * t + 1], r);
  }
}

// probs[p, k] = exp(x - max_k) / sum_k  -> bf16 [HW, Kpad], zero padded
__global__ void hw_probs_kernel(const float* __restrict__ logits, int ld, long HW, int K,
                                const float* __restrict__ rowstat, bf16_t* __restrict__ probs,
                                int Kpad) {
  const long n = HW * Kpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long p = i / Kpad;
    const int k = (int)(i - p * Kpad);
    float v = 0.f;
    if (k < K) v = __expf(logits[p * ld + k] - rowstat[2 * k]) / rowstat[2 * k + 1];
    probs[i] = f2bf(v);
  }
}

// dot[k] = sum_c ctx[k,c] * dctx[k,c]
__global__ void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b, int C,
                              float* __restrict__ out) {
  const int k = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < C; c += 64) acc += a[(long)k * C + c] * b[(long)k * C + c];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[k] = acc;
}

// dlogits[p,k] = probs[p,k] * (dprobs[p,k] - dot[k])
__global__ void hw_softmax_bwd_kernel(const float* __restrict__ logits, int ld, long HW, int K,
                                      const float* __restrict__ rowstat,
                                      const float* __restrict__ dprobs, int lddp,
                                      const float* __restrict__ dot, float* __restrict__ dlogits,
                                      int lddl, int accumulate) {
  const long n = HW * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long p = i / K;
    const int k = (int)(i - p * K);
    const float pr = __expf(logits[p * ld + k] - rowstat[2 * k]) / rowstat[2 * k + 1];
    const float g = pr * (dprobs[p * lddp + k] - dot[k]);
    float* dst = dlogits + p * lddl + k;
    *dst = accumulate ? *dst + g : g;
  }
}

// per-pixel softmax over K (<= 64) logits: one thread per pixel
__global__ void lastdim_softmax_fwd_kernel(const float* __restrict__ sim, int ld, long P, int K,
                                           float scale, bf16_t* __restrict__ probs, int Kpad) {
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const float* s = sim + p * ld;
    float mx = -FLT_MAX;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s[k] * scale);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(s[k] * scale - mx);
    const float inv = 1.f / sum;
    bf16_t* o = probs + p * Kpad;
    for (int k = 0; k < Kpad; ++k) o[k] = f2bf(k < K ? __expf(s[k] * scale - mx) * inv : 0.f);
  }
}
// dsim[p,k] = scale * pr[k] * (dpr[k] - sum_j pr[j] dpr[j])   -> bf16 [P, Kpad]
__global__ void lastdim_softmax_bwd_kernel(const float* __restrict__ sim, int ld, long P, int K,
                                           float scale, const float* __restrict__ dprobs, int lddp,
                                           bf16_t* __restrict__ dsim, int Kpad) {
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const float* s = sim + p * ld;
    const float* dp = dprobs + p * lddp;
    float mx = -FLT_MAX;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s[k] * scale);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(s[k] * scale - mx);
    const float inv = 1.f / sum;
    float dot = 0.f;
    for (int k = 0; k < K; ++k) dot += __expf(s[k] * scale - mx) * inv * dp[k];
    bf16_t* o = dsim + p * Kpad;
    for (int k = 0; k < Kpad; ++k) {
      float g = 0.f;
      if (k < K) g = scale * __expf(s[k] * scale - mx) * inv * (dp[k] - dot);
      o[k] = f2bf(g);
    }
  }
}

// dst[r, c] (bf16, [Rpad? rows][Kpad]) = src[r, c] or src[c, r]; zero padded.
template <typename T>
__global__ void pack_matrix_kernel(const T* __restrict__ src, int R, int C, int ld, int transpose,
                                   bf16_t* __restrict__ dst, int rows_out, int Kpad) {
  const long n = (long)rows_out * Kpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Kpad), c = (int)(i - (long)r * Kpad);
    float v = 0.f;
    if (!transpose) { if (r < R && c < C) v = ld_as_f32(src + (long)r * ld + c); }
    else { if (c < R && r < C) v = ld_as_f32(src + (long)c * ld + r); }
    dst[i] = f2bf(v);
  }
}

inline int grid_for(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int ssa_softmax_hw_stats(const float* logits, int ld, long HW, int K, float* rowstat,
                         void* stream) {
  if (!logits || !rowstat || K <= 0 || K > NT || HW <= 0) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(hw_init_kernel, dim3(1), dim3(NT), 0, s, rowstat, K);
  SSA_LAUNCH_CHECK();
  const int RP = NT / K;
  long ppb = (long)RP * 16;
  long blocks = (HW + ppb - 1) / ppb;
  if (blocks > 1024) { blocks = 1024; ppb = (HW + blocks - 1) / blocks; blocks = (HW + ppb - 1) / ppb; }
  hipLaunchKernelGGL(hw_reduce_kernel<0>, dim3((int)blocks), dim3(NT), 0, s, logits, ld, HW, K,
                     rowstat, ppb);
  SSA_LAUNCH_CHECK();
  hipLaunchKernelGGL(hw_reduce_kernel<1>, dim3((int)blocks), dim3(NT), 0, s, logits, ld, HW, K,
                     rowstat, ppb);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_softmax_hw_probs(const float* logits, int ld, long HW, int K, const float* rowstat,
                         void* probs, int Kpad, void* stream) {
  if (!logits || !rowstat || !probs || Kpad < K || Kpad % 8) return SSA_EINVAL;
  hipLaunchKernelGGL(hw_probs_kernel, dim3(grid_for(HW * Kpad)), dim3(256), 0, (hipStream_t)stream,
                     logits, ld, HW, K, rowstat, (bf16_t*)probs, Kpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rowdot_f32(const float* a, const float* b, int K, int C, float* out, void* stream) {
  if (!a || !b || !out || K <= 0 || C <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(rowdot_kernel, dim3(K), dim3(64), 0, (hipStream_t)stream, a, b, C, out);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_softmax_hw_bwd(const float* logits, int ld, long HW, int K, const float* rowstat,
                       const float* dprobs, int lddp, const float* dot, float* dlogits, int lddl,
                       int accumulate, void* stream) {
  if (!logits || !rowstat || !dprobs || !dot || !dlogits) return SSA_EINVAL;
  hipLaunchKernelGGL(hw_softmax_bwd_kernel, dim3(grid_for(HW * K)), dim3(256), 0,
                     (hipStream_t)stream, logits, ld, HW, K, rowstat, dprobs, lddp, dot, dlogits,
                     lddl, accumulate);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_softmax_lastdim_fwd(const float* sim, int ld, long P, int K, float scale, void* probs,
                            int Kpad, void* stream) {
  if (!sim || !probs || K <= 0 || Kpad < K || Kpad % 8) return SSA_EINVAL;
  hipLaunchKernelGGL(lastdim_softmax_fwd_kernel, dim3(grid_for(P)), dim3(256), 0,
                     (hipStream_t)stream, sim, ld, P, K, scale, (bf16_t*)probs, Kpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_softmax_lastdim_bwd(const float* sim, int ld, long P, int K, float scale,
                            const float* dprobs, int lddp, void* dsim, int Kpad, void* stream) {
  if (!sim || !dprobs || !dsim || K <= 0 || Kpad < K || Kpad % 8) return SSA_EINVAL;
  hipLaunchKernelGGL(lastdim_softmax_bwd_kernel, dim3(grid_for(P)), dim3(256), 0,
                     (hipStream_t)stream, sim, ld, P, K, scale, dprobs, lddp, (bf16_t*)dsim, Kpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_pack_matrix(const void* src, int src_dtype, int R, int C, int ld, int transpose, void* dst,
                    int rows_out, int Kpad, void* stream) {
  if (!src || !dst || Kpad % 32 || rows_out <= 0) return SSA_EINVAL;
  if ((transpose ? R : C) > Kpad || (transpose ? C : R) > rows_out) return SSA_EINVAL;
  const long n = (long)rows_out * Kpad;
  if (src_dtype == 0)
    hipLaunchKernelGGL((pack_matrix_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)src, R, C, ld, transpose, (bf16_t*)dst,
                       rows_out, Kpad);
  else
    hipLaunchKernelGGL((pack_matrix_kernel<float>), dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)src, R, C, ld, transpose, (bf16_t*)dst,
                       rows_out, Kpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
