// All split-K weight-gradient partials of a backward pass reduced in a handful of
// launches (641 conv layers per training step each end in a `wgrad_reduce` of ~6 us:
// SURVEY.md K1-K6 weight gradients).  The host glue defers the reduces to the end of
// backward and hands over the whole job list; up to 72 jobs ride in one launch as
// KERNEL ARGUMENTS (no table upload; a captured hipGraph owns the pointers).
//
// Per job the arithmetic is that of wgrad_reduce_kernel (conv_igemm.hip), in the same
// summation order, so the result is bit-identical to the per-layer path:
// workgroup (co, 64-column chunk): 4 split lanes, each summing every 4th split into two
// alternating accumulators, combined as (l0 + l1) + (l2 + l3); k = (kh,kw,ci) -> OIHW.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

constexpr int kJobs = 72;

struct ReduceBatch {
  ssa_wgrad_reduce_job job[kJobs];     // 48 bytes each
  int block_start[kJobs + 1];          // prefix sum of Cout * ceil(Kflat / 64)
  int n;
};
static_assert(sizeof(ssa_wgrad_reduce_job) == 48, "job layout");
static_assert(sizeof(ReduceBatch) <= 4096 - 64, "kernel argument space");

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const ReduceBatch tb) {
  __shared__ float sh[4][64];
  int lo = 0, hi = tb.n;
  const int blk = blockIdx.x;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tb.block_start[mid] <= blk) lo = mid; else hi = mid;
  }
  const ssa_wgrad_reduce_job& jb = tb.job[lo];
  const int taps = jb.KH * jb.KW;
  const int Kflat = taps * jb.Cin_pad;
  const int nchunks = (Kflat + 63) / 64;
  const int lb = blk - tb.block_start[lo];
  const int co = lb / nchunks, chunk = lb - co * nchunks;
  const int kk = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int k = chunk * 64 + kk;
  const long split_stride = (long)jb.cout_pad * Kflat;
  const int nsplit = jb.nsplit;
  float s0 = 0.f, s1 = 0.f;
  if (k < Kflat) {
    const float* src = jb.partial + (long)co * Kflat + k;
    int sp = sl;
    for (; sp + 4 < nsplit; sp += 8) {
      s0 += src[(long)sp * split_stride];
      s1 += src[(long)(sp + 4) * split_stride];
    }
    if (sp < nsplit) s0 += src[(long)sp * split_stride];
  }
  sh[sl][kk] = s0 + s1;
  __syncthreads();
  if (sl == 0 && k < Kflat) {
    const float v = (sh[0][kk] + sh[1][kk]) + (sh[2][kk] + sh[3][kk]);
    const int tap = k / jb.Cin_pad, ci = k - tap * jb.Cin_pad;
    if (ci < jb.Cin) jb.dw[((long)co * jb.Cin + ci) * taps + tap] = v;
  }
}

}  // namespace

extern "C" int ssa_conv2d_wgrad_reduce_batched(const ssa_wgrad_reduce_job* jobs, int njobs, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return SSA_EINVAL;
  int i = 0;
  while (i < njobs) {
    ReduceBatch tb;
    tb.n = 0;
    tb.block_start[0] = 0;
    while (i < njobs && tb.n < kJobs && tb.block_start[tb.n] < (1 << 24)) {
      const ssa_wgrad_reduce_job& j = jobs[i];
      if (!j.partial || !j.dw || j.nsplit < 1 || j.Cout < 1 || j.Cout > j.cout_pad || j.Cin > j.Cin_pad ||
          j.KH < 1 || j.KW < 1)
        return SSA_EINVAL;
      const long Kflat = (long)j.KH * j.KW * j.Cin_pad;
      const long blocks = (long)j.Cout * ((Kflat + 63) / 64);
      if (blocks > (1 << 24)) return SSA_EUNSUPPORTED;
      tb.job[tb.n] = j;
      tb.block_start[tb.n + 1] = tb.block_start[tb.n] + (int)blocks;
      ++tb.n;
      ++i;
    }
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(tb.block_start[tb.n]), dim3(256), 0,
                       (hipStream_t)stream, tb);
    SSA_LAUNCH_CHECK();
  }
  return SSA_OK;
}
