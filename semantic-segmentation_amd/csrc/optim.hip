// SGD with momentum and weight decay over every parameter tensor of the net in a
// handful of launches (train.py:509 `optim.step()` on the torch.optim.SGD built by
// loss/optimizer.py:47-53: nesterov=False, dampening 0, one param group).
//
// HBM-bound streaming update: per element read p, g, buf and write p, buf (20 B),
// one pass.  Up to 96 tensors ride in one launch: their pointers and chunk
// prefix sums are KERNEL ARGUMENTS (3.4 KB of the 4 KB limit), so nothing has to
// be uploaded and a captured hipGraph holds the whole update.  The learning rate
// can come from device memory so that a captured step follows the LR schedule.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

constexpr int kTensors = 96;          // tensors per launch
constexpr int kThreads = 256;
constexpr int kChunk = kThreads * 16; // elements per workgroup

struct SgdBatch {
  float* p[kTensors];
  const float* g[kTensors];
  float* buf[kTensors];               // null: no momentum
  long numel[kTensors];
  int chunk_start[kTensors + 1];      // prefix sum of ceil(numel / kChunk)
  int n;
};

struct SgdHyper {
  float lr;
  const float* lr_dev;                // when set, overrides lr
  float momentum, weight_decay;
  int nesterov;
};

// the rounding sequence of torch's SGD: d = g + wd*p; buf = m*buf + d; p = p - lr*buf
__device__ __forceinline__ void sgd_update(float& p, float g, float& b, bool has_buf, float lr, float m,
                                           float wd, int nesterov) {
  float d = wd != 0.f ? __fmaf_rn(wd, p, g) : g;
  if (has_buf) {
    b = __fadd_rn(__fmul_rn(m, b), d);
    d = nesterov ? __fmaf_rn(m, b, d) : b;
  }
  p = __fmaf_rn(-lr, d, p);
}

__global__ __launch_bounds__(kThreads) void sgd_momentum_kernel(const SgdBatch tb, const SgdHyper hp) {
  // which tensor does this workgroup's chunk belong to
  int lo = 0, hi = tb.n;
  const int blk = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tb.chunk_start[mid] <= blk) lo = mid; else hi = mid;
  }
  const int t = lo;
  float* __restrict__ p = tb.p[t];
  const float* __restrict__ g = tb.g[t];
  float* __restrict__ buf = tb.buf[t];
  const long n = tb.numel[t];
  const long base = (long)(blk - tb.chunk_start[t]) * kChunk;
  const long end = base + kChunk < n ? base + kChunk : n;
  const float lr = hp.lr_dev ? *hp.lr_dev : hp.lr;
  const float m = hp.momentum, wd = hp.weight_decay;
  const bool has_buf = buf != nullptr;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)buf)) & 15) == 0;
  if (vec) {
    const long vend = base + ((end - base) & ~3L);
    for (long i = base + threadIdx.x * 4L; i < vend; i += kThreads * 4L) {
      float4 pv = *(const float4*)(p + i);
      const float4 gv = *(const float4*)(g + i);
      float4 bv = has_buf ? *(const float4*)(buf + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      sgd_update(pv.x, gv.x, bv.x, has_buf, lr, m, wd, hp.nesterov);
      sgd_update(pv.y, gv.y, bv.y, has_buf, lr, m, wd, hp.nesterov);
      sgd_update(pv.z, gv.z, bv.z, has_buf, lr, m, wd, hp.nesterov);
      sgd_update(pv.w, gv.w, bv.w, has_buf, lr, m, wd, hp.nesterov);
      *(float4*)(p + i) = pv;
      if (has_buf) *(float4*)(buf + i) = bv;
    }
    for (long i = vend + threadIdx.x; i < end; i += kThreads) {
      float pv = p[i], bv = has_buf ? buf[i] : 0.f;
      sgd_update(pv, g[i], bv, has_buf, lr, m, wd, hp.nesterov);
      p[i] = pv;
      if (has_buf) buf[i] = bv;
    }
  } else {
    for (long i = base + threadIdx.x; i < end; i += kThreads) {
      float pv = p[i], bv = has_buf ? buf[i] : 0.f;
      sgd_update(pv, g[i], bv, has_buf, lr, m, wd, hp.nesterov);
      p[i] = pv;
      if (has_buf) buf[i] = bv;
    }
  }
}

}  // namespace

extern "C" int ssa_sgd_momentum_step(void* const* params, const void* const* grads, void* const* bufs,
                                     const int64_t* numel, int n_tensors, float lr, const float* lr_dev,
                                     float momentum, float weight_decay, int nesterov, void* stream) {
  if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !numel))) return SSA_EINVAL;
  if (nesterov && (momentum <= 0.f || !bufs)) return SSA_EINVAL;
  SgdHyper hp{lr, lr_dev, momentum, weight_decay, nesterov};
  int i = 0;
  while (i < n_tensors) {
    SgdBatch tb;
    tb.n = 0;
    tb.chunk_start[0] = 0;
    // at most kTensors tensors and 2^20 chunks (4 G elements) per launch
    while (i < n_tensors && tb.n < kTensors && tb.chunk_start[tb.n] < (1 << 20)) {
      const int64_t n = numel[i];
      if (n < 0 || !params[i] || !grads[i]) return SSA_EINVAL;
      if (n == 0) { ++i; continue; }
      const int64_t chunks = (n + kChunk - 1) / kChunk;
      if (chunks > (1 << 30)) return SSA_EUNSUPPORTED;
      tb.p[tb.n] = (float*)params[i];
      tb.g[tb.n] = (const float*)grads[i];
      tb.buf[tb.n] = bufs ? (float*)bufs[i] : nullptr;
      tb.numel[tb.n] = n;
      tb.chunk_start[tb.n + 1] = tb.chunk_start[tb.n] + (int)chunks;
      ++tb.n;
      ++i;
    }
    if (tb.n == 0) continue;
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(tb.chunk_start[tb.n]), dim3(kThreads), 0,
                       (hipStream_t)stream, tb, hp);
    SSA_LAUNCH_CHECK();
  }
  return SSA_OK;
}
