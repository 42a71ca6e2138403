// SGD with momentum and weight decay over every parameter tensor of the net in a
// handful of launches (train.py:509 `optim.step()` on the torch.optim.SGD built by
// loss/optimizer.py:47-53: nesterov=False, dampening 0, one param group).
//
// HBM-bound streaming update: per element read p, g, buf and write p, buf (20 B),
// one pass.  Up to 96 tensors ride in one launch: their pointers and chunk
// prefix sums are KERNEL ARGUMENTS (3.4 KB of the 4 KB limit), so nothing has to
// be uploaded and a captured hipGraph holds the whole update.  The learning rate
// can come from device memory so that a captured step follows the LR schedule.
//
// Dynamic loss scaling of fp16 training (apex.amp O1/O2, train.py:381,503-505: `amp.scale_loss`) lives here too,
// as three capturable pieces around a 4-float device record  state = {scale, found_inf, clean steps, 1 / scale}:
//   ssa_amp_check_grads   one pass over the gradients: found_inf = 1 if any element is inf / nan
//   ssa_sgd_momentum_step (amp_state given) multiplies every gradient by 1 / scale and does NOTHING when found_inf
//   ssa_amp_update        overflow: scale *= backoff, the step was skipped; else after `growth_interval` clean steps
//                         scale *= growth  -- apex's LossScaler (2^16, x2 every 2000, /2)
// The loss is multiplied by state[0] on the device (a 0-dim tensor product), so the whole scaled step replays as a
// hipGraph and the scale it uses is the one the previous replay left.
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
  const float* amp;                   // loss-scaling record {scale, found_inf, clean steps, 1 / scale} or null
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
  float gs = 1.f;                     // gradient un-scaling (fp16 training)
  if (hp.amp) {
    if (hp.amp[1] != 0.f) return;     // an overflowed step is skipped: parameters and momentum stay
    gs = hp.amp[3];
  }
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)buf)) & 15) == 0;
  if (vec) {
    const long vend = base + ((end - base) & ~3L);
    for (long i = base + threadIdx.x * 4L; i < vend; i += kThreads * 4L) {
      float4 pv = *(const float4*)(p + i);
      float4 gv = *(const float4*)(g + i);
      if (hp.amp) { gv.x *= gs; gv.y *= gs; gv.z *= gs; gv.w *= gs; }
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
      sgd_update(pv, hp.amp ? g[i] * gs : g[i], bv, has_buf, lr, m, wd, hp.nesterov);
      p[i] = pv;
      if (has_buf) buf[i] = bv;
    }
  } else {
    for (long i = base + threadIdx.x; i < end; i += kThreads) {
      float pv = p[i], bv = has_buf ? buf[i] : 0.f;
      sgd_update(pv, hp.amp ? g[i] * gs : g[i], bv, has_buf, lr, m, wd, hp.nesterov);
      p[i] = pv;
      if (has_buf) buf[i] = bv;
    }
  }
}

// found_inf = 1 if any gradient element is not finite (same chunking as the update)
__global__ __launch_bounds__(kThreads) void amp_check_kernel(const SgdBatch tb, float* __restrict__ state) {
  int lo = 0, hi = tb.n;
  const int blk = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tb.chunk_start[mid] <= blk) lo = mid; else hi = mid;
  }
  const float* __restrict__ g = tb.g[lo];
  const long n = tb.numel[lo];
  const long base = (long)(blk - tb.chunk_start[lo]) * kChunk;
  const long end = base + kChunk < n ? base + kChunk : n;
  bool bad = false;
  // (x - x is 0 for finite x, nan for inf / nan; the exponent test reads the same without arithmetic)
  if ((((uintptr_t)g) & 15) == 0) {
    const long vend = base + ((end - base) & ~3L);
    for (long i = base + threadIdx.x * 4L; i < vend; i += kThreads * 4L) {
      const uint4 v = *(const uint4*)(g + i);
      bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) |
             ((v.z & 0x7f800000u) == 0x7f800000u) | ((v.w & 0x7f800000u) == 0x7f800000u);
    }
    for (long i = vend + threadIdx.x; i < end; i += kThreads) bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
  } else {
    for (long i = base + threadIdx.x; i < end; i += kThreads) bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
  }
  if (bad) state[1] = 1.f;            // (every writer writes the same value)
}

__global__ void amp_update_kernel(float* __restrict__ state, int growth_interval, float growth, float backoff,
                                  float min_scale, float max_scale, float* __restrict__ counters) {
  if (threadIdx.x || blockIdx.x) return;
  float scale = state[0], good = state[2];
  if (counters) {                     // {skipped steps in total, skipped steps in a row}: what a host-side log reads
    if (state[1] != 0.f) { counters[0] += 1.f; counters[1] += 1.f; } else { counters[1] = 0.f; }
  }
  if (state[1] != 0.f) {
    scale = fmaxf(scale * backoff, min_scale);
    good = 0.f;
  } else {
    good += 1.f;
    if (good >= (float)growth_interval) {
      scale = fminf(scale * growth, max_scale);
      good = 0.f;
    }
  }
  state[0] = scale;
  state[1] = 0.f;
  state[2] = good;
  state[3] = 1.f / scale;
}

}  // namespace

extern "C" int ssa_sgd_momentum_step(void* const* params, const void* const* grads, void* const* bufs,
                                     const int64_t* numel, int n_tensors, float lr, const float* lr_dev,
                                     float momentum, float weight_decay, int nesterov, const float* amp_state,
                                     void* stream) {
  if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !numel))) return SSA_EINVAL;
  if (nesterov && (momentum <= 0.f || !bufs)) return SSA_EINVAL;
  SgdHyper hp{lr, lr_dev, momentum, weight_decay, nesterov, amp_state};
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

extern "C" int ssa_amp_check_grads(const void* const* grads, const int64_t* numel, int n_tensors, float* amp_state,
                                   void* stream) {
  if (n_tensors < 0 || !amp_state || (n_tensors > 0 && (!grads || !numel))) return SSA_EINVAL;
  int i = 0;
  while (i < n_tensors) {
    SgdBatch tb;
    tb.n = 0;
    tb.chunk_start[0] = 0;
    while (i < n_tensors && tb.n < kTensors && tb.chunk_start[tb.n] < (1 << 20)) {
      const int64_t n = numel[i];
      if (n < 0 || !grads[i]) return SSA_EINVAL;
      if (n == 0) { ++i; continue; }
      const int64_t chunks = (n + kChunk - 1) / kChunk;
      if (chunks > (1 << 30)) return SSA_EUNSUPPORTED;
      tb.p[tb.n] = nullptr;
      tb.g[tb.n] = (const float*)grads[i];
      tb.buf[tb.n] = nullptr;
      tb.numel[tb.n] = n;
      tb.chunk_start[tb.n + 1] = tb.chunk_start[tb.n] + (int)chunks;
      ++tb.n;
      ++i;
    }
    if (tb.n == 0) continue;
    hipLaunchKernelGGL(amp_check_kernel, dim3(tb.chunk_start[tb.n]), dim3(kThreads), 0, (hipStream_t)stream, tb,
                       amp_state);
    SSA_LAUNCH_CHECK();
  }
  return SSA_OK;
}

extern "C" int ssa_amp_update_counted(float* amp_state, float* counters, int growth_interval, float growth, float backoff,
                                      float min_scale, float max_scale, void* stream) {
  if (!amp_state || growth_interval < 1 || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) ||
      !(min_scale > 0.f) || !(max_scale >= min_scale))
    return SSA_EINVAL;
  hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, amp_state, growth_interval, growth,
                     backoff, min_scale, max_scale, counters);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

extern "C" int ssa_amp_update(float* amp_state, int growth_interval, float growth, float backoff, float min_scale,
                              float max_scale, void* stream) {
  return ssa_amp_update_counted(amp_state, nullptr, growth_interval, growth, backoff, min_scale, max_scale, stream);
}
