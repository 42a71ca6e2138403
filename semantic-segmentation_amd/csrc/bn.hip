// BatchNorm2d (training + eval) on NHWC bf16 activations, fp32 math, fp64
// statistics.  Replaces cfg.MODEL.BNFUNC = nn.BatchNorm2d / apex SyncBatchNorm
// (config.py:216-225, network/mynn.py:18-24).
//
// The statistics are exposed as raw fp64 sums so that SyncBN is "all-reduce the
// 2C sums, pass the global count to ssa_bn_finalize" -- numerically identical to
// BatchNorm over the concatenated global batch.
//
// All passes are HBM-bound streaming kernels: each thread owns one 16-byte
// channel group (8 channels) and walks pixels with that group fixed, so the
// per-channel accumulators live in registers and every wave-level access is a
// contiguous run of 16-byte pieces.
#include "common.h"
#include <stdlib.h>
#include "group.h"
#include "../../include/semseg_hip.h"

namespace {

constexpr int NT = 256;

// number of threads of a block that do work: the largest multiple of VC <= NT
__host__ __device__ inline int active_threads(int VC) { return (NT / VC) * VC; }

// Accumulate per-channel sums for this block into fp64 global sums.
// v0/v1 hold 8 channels each (this thread's channel group cg).  The threads' partials meet in LDS
// without atomics: every thread stores its 16 values transposed ([value][thread]: conflict free), then
// thread c < 2C sums the RP = NA / VC threads that own channel c's group and issues ONE fp64 atomic.
// (The previous LDS-atomic version serialised RP-way on every channel: with 4 pixel rows per thread
// the epilogue cost more than the streaming.)
__device__ __forceinline__ void block_reduce_2x8(const float* v0, const float* v1, int cg, int C,
                                                 bool active, double* gsums, float* sh, int bx, int nrep = 1) {
  gsums += (long)(bx % nrep) * 2 * C;   // [nrep][2][C]: spread the same-address atomics
  // sh: [16][NT + 1] floats (odd row stride: the 8 channels of a group fall in 8 banks)
  constexpr int LDR = NT + 1;
  const int t = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sh[j * LDR + t] = active ? v0[j] : 0.f;
    sh[(8 + j) * LDR + t] = active ? v1[j] : 0.f;
  }
  __syncthreads();
  const int VC = C >> 3, RP = active_threads(VC) / VC;
  for (int i = t; i < 2 * C; i += NT) {
    const int which = i / C, c = i - which * C;
    const int g = c >> 3, j = c & 7;
    const float* src = sh + (which * 8 + j) * LDR + g;
    float a0 = 0.f, a1 = 0.f;
    int r = 0;
    for (; r + 1 < RP; r += 2) { a0 += src[r * VC]; a1 += src[(r + 1) * VC]; }
    if (r < RP) a0 += src[r * VC];
    atomicAdd(&gsums[i], (double)(a0 + a1));
  }
}

// Sum of the statistics replicas of channel c ([nrep][2][C] fp64), in replica order.  The (up to 8,
// kStatReplicas of the conv epilogues) replica loads are issued TOGETHER: a rolled loop over a run-time
// nrep waits for each pair of loads in turn -- eight L2 round trips in the prologue of every workgroup
// of every apply / backward-apply launch (11-19 us for a TWO-workgroup launch before this).
__device__ __forceinline__ void replica_sums(const double* __restrict__ sums, int nrep, int C, int c,
                                             double* s1, double* s2) {
  double v1[8], v2[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const bool ok = r < nrep;
    v1[r] = ok ? sums[(long)r * 2 * C + c] : 0.0;
    v2[r] = ok ? sums[(long)r * 2 * C + C + c] : 0.0;
  }
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int r = 0; r < 8; ++r) { a += v1[r]; b += v2[r]; }      // adding +0.0 for r >= nrep changes nothing
  for (int r = 8; r < nrep; ++r) { a += sums[(long)r * 2 * C + c]; b += sums[(long)r * 2 * C + C + c]; }
  *s1 = a;
  *s2 = b;
}

// ---- the streaming passes -------------------------------------------------------------------------------------
// Every pass gives a thread ROWS pixel rows of its channel group and issues ALL of their loads (up to 3 tensors x
// ROWS x 16 bytes) before anything else -- also before the per-workgroup coefficient prologue, whose own (L2) loads
// then fly together with the data: one memory round trip per workgroup instead of the three of round 3 (prologue,
// then two batches of four rows behind `#pragma unroll 4`).  Rows past the end of the workgroup's range load the
// range's first pixel (always valid) and are masked, so the loads are unconditional and issue back to back.  A level
// of the trunk (7.4 M elements) is in flight in its entirety.
template <int ROWS>
struct RowSet {
  unsigned ok;              // bit u: row u lies inside [p0, p1)
  int off[ROWS];            // pixel offset of row u from p0 (0 for masked rows)
  __device__ __forceinline__ void init(long p0, long p1, int pr, int RP, bool active) {
    ok = 0;
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      const long q = (long)pr + (long)u * RP;
      const bool o = active && p0 + q < p1;
      ok |= (o ? 1u : 0u) << u;
      off[u] = o ? (int)q : 0;
    }
  }
  __device__ __forceinline__ void load(const bf16_t* base, int ld, uint4 (&v)[ROWS]) const {
#pragma unroll
    for (int u = 0; u < ROWS; ++u) v[u] = *reinterpret_cast<const uint4*>(base + (long)off[u] * ld);
  }
};

template <int ROWS>
__device__ __forceinline__ void bn_stats_body(const bf16_t* __restrict__ x, long P, int C,
                                              int ld, double* __restrict__ sums,
                                              long pix_per_block, const int bx, const bool squares) {
  SSA_DYN_LDS(float, sh);
  const int VC = C >> 3, NA = active_threads(VC), RP = NA / VC;
  const int t = threadIdx.x;
  const bool active = t < NA;
  const int cg = active ? t % VC : 0, pr = active ? t / VC : 0;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  const long pb = bx * pix_per_block;
  const long pe = min(P, pb + pix_per_block);
  for (long p0 = pb; p0 < pe; p0 += (long)RP * ROWS) {
    RowSet<ROWS> rs;
    rs.init(p0, pe, pr, RP, active);
    uint4 v[ROWS];
    rs.load(x + p0 * ld + cg * 8, ld, v);
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      const float keep = ((rs.ok >> u) & 1u) ? 1.f : 0.f;
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float fk = f[j] * keep; s[j] += fk; if (squares) q[j] += fk * f[j]; }
    }
  }
  block_reduce_2x8(s, q, cg, C, active, sums, sh, bx);
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float momentum, float eps, int use_running,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (use_running) {
    mean = running_mean[c];
    var = running_var[c];
  } else {
    mean = sums[c] / count;
    var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? (double)gamma[c] : 1.0;
  const double b = beta ? (double)beta[c] : 0.0;
  scale[c] = (float)(g * invstd);
  shift[c] = (float)(b - mean * g * invstd);
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = (float)invstd;
}

// bit j of the result = element j of a packed 8-element piece is > 0 (as stored: after the rounding to 16 bits)
__device__ __forceinline__ unsigned positive_bits(const uint4& pk) {
  const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned lo = w[i] & 0xffffu, hi = w[i] >> 16;
    m |= (((lo & 0x7fffu) != 0 && !(lo & 0x8000u)) ? 1u : 0u) << (2 * i);
    m |= (((hi & 0x7fffu) != 0 && !(hi & 0x8000u)) ? 1u : 0u) << (2 * i + 1);
  }
  return m;
}

// z = post * act(scale * x + shift + residual): the arithmetic shared by the eval and the training apply.
// mrow (training, ReLU behind a residual add): one byte per pixel and 8-channel group, bit j = z_j > 0 -- the
// backward passes read this byte instead of the 16 bytes of z (the mask is all they want of it).
template <int ROWS>
__device__ __forceinline__ void bn_apply_rows(const RowSet<ROWS>& rs, const uint4 (&v)[ROWS], const uint4 (&rv)[ROWS],
                                              bool has_res, const float (&a)[8], const float (&b)[8], int relu,
                                              const float* __restrict__ post, long pix_per_img, long p0, int C, int cg,
                                              bf16_t* __restrict__ zb, int ldz, unsigned char* __restrict__ mrow = nullptr) {
#pragma unroll
  for (int u = 0; u < ROWS; ++u) {
    float f[8];
    unpack8(v[u], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] * a[j] + b[j];
    if (has_res) {
      float r[8];
      unpack8(rv[u], r);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += r[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
    }
    if (post) {
      const float* pp = post + (unsigned)((unsigned)(p0 + rs.off[u]) / (unsigned)pix_per_img) * C + cg * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] *= pp[j];
    }
    if ((rs.ok >> u) & 1u) {
      const uint4 pk = pack8(f);
      *reinterpret_cast<uint4*>(zb + (long)rs.off[u] * ldz) = pk;
      if (mrow) mrow[(long)rs.off[u] * (C >> 3)] = (unsigned char)positive_bits(pk);
    }
#ifndef SSA_EMU
    // one row at a time: left alone the scheduler unpacks all ROWS rows side by side (16 more registers per row) and the
    // training apply lands on 129 registers -- one over the 128 that let FOUR workgroups share a CU, i.e. a trunk
    // level's ~930 workgroups all be resident at once instead of 768 + a second round (profiles/r06_notes.md)
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

template <int ROWS>
__device__ __forceinline__ void bn_apply_body(
    const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ res, int ldr,
    bf16_t* __restrict__ z, int ldz, long P, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, int relu, const float* __restrict__ post, long pix_per_img,
    long pix_per_block, const int bx) {
  const int VC = C >> 3, NA = active_threads(VC), RP = NA / VC;
  const int t = threadIdx.x;
  if (t >= NA) return;
  const int cg = t % VC, pr = t / VC;
  const long pb = bx * pix_per_block;
  const long pe = min(P, pb + pix_per_block);
  float a[8], b[8];
  bool have_coef = false;
  for (long p0 = pb; p0 < pe; p0 += (long)RP * ROWS) {
    RowSet<ROWS> rs;
    rs.init(p0, pe, pr, RP, true);
    uint4 v[ROWS], rv[ROWS];
    rs.load(x + p0 * ldx + cg * 8, ldx, v);
    if (res) rs.load(res + p0 * ldr + cg * 8, ldr, rv);
    if (!have_coef) {
      const float4 a0 = *reinterpret_cast<const float4*>(scale + cg * 8), a1 = *reinterpret_cast<const float4*>(scale + cg * 8 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(shift + cg * 8), b1 = *reinterpret_cast<const float4*>(shift + cg * 8 + 4);
      a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
      b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
      have_coef = true;
    }
    bn_apply_rows<ROWS>(rs, v, rv, res != nullptr, a, b, relu, post, pix_per_img, p0, C, cg, z + p0 * ldz + cg * 8, ldz);
  }
}

// Training-mode apply with the finalize step fused in: thread c derives scale/shift of channel c from the fp64
// sums (the replica loads of all channels issue together, BEHIND the workgroup's data loads), block 0 additionally
// publishes mean/invstd/scale/shift for the backward pass, updates the running statistics and bumps
// num_batches_tracked.
template <int ROWS>
__device__ __forceinline__ void bn_apply_train_body(
    const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ res, int ldr,
    bf16_t* __restrict__ z, int ldz, long P, int C, const double* __restrict__ sums, int nrep,
    double count, double inv_count, double unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var,
    long* __restrict__ num_batches_tracked, float momentum, float eps, float* __restrict__ coef,
    float* __restrict__ pass_stats, int relu, const float* __restrict__ post, long pix_per_img,
    long pix_per_block, unsigned char* __restrict__ mask, const int bx) {
  SSA_DYN_LDS(float, sh);                 // [2C]: scale, shift for this launch
  const int VC = C >> 3, NA = active_threads(VC), RP = NA / VC;
  const int t = threadIdx.x;
  const bool active = t < NA;
  const int cg = active ? t % VC : 0, pr = active ? t / VC : 0;
  const long pb = bx * pix_per_block;
  const long pe = min(P, pb + pix_per_block);
  // ---- the first chunk's loads, then the coefficient prologue while they are in flight
  RowSet<ROWS> rs;
  rs.init(pb, pe, pr, RP, active);
  uint4 v[ROWS], rv[ROWS];
  rs.load(x + pb * ldx + cg * 8, ldx, v);
  if (res) rs.load(res + pb * ldr + cg * 8, ldr, rv);
  for (int c = t; c < C; c += NT) {
    double s1 = 0.0, s2 = 0.0;
    replica_sums(sums, nrep, C, c, &s1, &s2);
    // This prologue sits on the critical path of EVERY workgroup (the launch is one round of ~900 workgroups: its
    // duration is a workgroup's lifetime), so it holds no fp64 division or square root -- software sequences of a few
    // hundred dependent instructions each: the same pass with the coefficients given ran 6.5 us against 12.8 with them
    // (tools/bnbench.py, profiles/r06_notes.md).  1 / count and count / (count - 1) come from the host;
    // 1 / sqrt(var + eps) is v_rsq_f32 plus one Newton step (the result is stored as fp32 anyway: <= 1 ulp of it).
    const double mean = s1 * inv_count;
    double var = s2 * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float ve = (float)(var + (double)eps);
    float rs = ssa_rsqrt(ve);
    rs = rs * (1.5f - 0.5f * ve * rs * rs);
    const double invstd = (double)rs;
    const double g = gamma ? (double)gamma[c] : 1.0;
    const double b = beta ? (double)beta[c] : 0.0;
    const float sc = (float)(g * invstd), sf = (float)(b - mean * g * invstd);
    sh[c] = sc;
    sh[C + c] = sf;
    if (bx == 0) {
      coef[c] = sc;
      coef[C + c] = sf;
      coef[2 * C + c] = (float)mean;
      coef[3 * C + c] = (float)invstd;
      if (running_mean) {
        const double unbiased = var * unbias;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
      }
      if (pass_stats) {
        pass_stats[c] = (float)mean;
        pass_stats[C + c] = (float)var;
      }
    }
  }
  if (bx == 0 && t == 0) {
    if (num_batches_tracked) *num_batches_tracked += 1;
    if (pass_stats) pass_stats[2 * C] = (float)count;
  }
  __syncthreads();
  if (!active) return;
  float a[8], b[8];
  {
    const float4 a0 = *reinterpret_cast<const float4*>(sh + cg * 8), a1 = *reinterpret_cast<const float4*>(sh + cg * 8 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(sh + C + cg * 8), b1 = *reinterpret_cast<const float4*>(sh + C + cg * 8 + 4);
    a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w; a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  }
  for (long p0 = pb;;) {
    bn_apply_rows<ROWS>(rs, v, rv, res != nullptr, a, b, relu, post, pix_per_img, p0, C, cg, z + p0 * ldz + cg * 8, ldz,
                        mask ? mask + p0 * VC + cg : nullptr);
    p0 += (long)RP * ROWS;
    if (p0 >= pe) break;
    rs.init(p0, pe, pr, RP, true);
    rs.load(x + p0 * ldx + cg * 8, ldx, v);
    if (res) rs.load(res + p0 * ldr + cg * 8, ldr, rv);
  }
}

// masked gradient g = post * dz where the ReLU let it through (mask from z, or recomputed from x)
__device__ __forceinline__ void bn_bwd_mask(float (&g)[8], const float (&xv)[8], const uint4& zraw, bool use_z, int relu,
                                            bool mask_from_x, const float (&ma)[8], const float (&mb)[8],
                                            const float* __restrict__ pp, bool use_bits = false, unsigned bits = 0) {
  if (pp) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= pp[j];
  }
  if (relu && mask_from_x) {          // z = relu(scale*x + shift), z not read
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = (xv[j] * ma[j] + mb[j]) > 0.f ? g[j] : 0.f;
  } else if (relu && use_bits) {      // the forward's sign byte (bn_apply_rows)
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = ((bits >> j) & 1u) ? g[j] : 0.f;
  } else if (relu && use_z) {
    float zv[8];
    unpack8(zraw, zv);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = zv[j] > 0.f ? g[j] : 0.f;
  }
}

// the sign bytes of a thread's rows (masked rows read the chunk's first pixel, like the data loads)
template <int ROWS>
__device__ __forceinline__ void load_sign_bytes(const RowSet<ROWS>& rs, const unsigned char* __restrict__ base, int VC,
                                                unsigned (&mk)[ROWS]) {
#pragma unroll
  for (int u = 0; u < ROWS; ++u) mk[u] = base[(long)rs.off[u] * VC];
}

// MODE as in bn_bwd_apply_body below (0: sign bytes / no ReLU, 1: mask recomputed from x, 2: mask read off z).  The loop
// accumulates sum g and sum g x per thread and centres ONCE at the end -- sum g xhat = invstd (sum g x - mean sum g), per
// thread, over its <= ROWS x chunks pixels, before the block reduction -- so mean / invstd are not live in it: with the
// mask source compile-time and the rows kept apart the 8-row form fits 128 registers (it ran at 246 = two workgroups per
// CU: a head-sized launch of 1,280-3,072 workgroups took three to six rounds).
template <int ROWS, int MODE>
__device__ __forceinline__ void bn_bwd_reduce_body(
    const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ dz, int lddz,
    const bf16_t* __restrict__ z, int ldz, long P, int C, const float* __restrict__ mean,
    const float* __restrict__ invstd, int relu, const float* __restrict__ post, long pix_per_img,
    double* __restrict__ sums, int nrep, long pix_per_block, const float* __restrict__ mscale,
    const float* __restrict__ mshift, const unsigned char* __restrict__ mask, const int bx) {
  SSA_DYN_LDS(float, sh);
  constexpr bool ZMASK = MODE == 2;
  const int VC = C >> 3, NA = active_threads(VC), RP = NA / VC;
  const int t = threadIdx.x;
  const bool active = t < NA;
  const int cg = active ? t % VC : 0, pr = active ? t / VC : 0;
  const bool use_bits = MODE == 0 && relu && mask != nullptr;
  const bool from_x = MODE == 1 && relu;
  float sg[8], sgx[8], ma[8], mb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sg[j] = 0.f; sgx[j] = 0.f; ma[j] = 0.f; mb[j] = 0.f; }
  const long pb = bx * pix_per_block;
  const long pe = min(P, pb + pix_per_block);
  for (long p0 = pb; p0 < pe; p0 += (long)RP * ROWS) {
    RowSet<ROWS> rs;
    rs.init(p0, pe, pr, RP, active);
    uint4 gv[ROWS], xr[ROWS], zr[ZMASK ? ROWS : 1];
    unsigned mk[MODE == 0 ? ROWS : 1];
    rs.load(dz + p0 * lddz + cg * 8, lddz, gv);
    rs.load(x + p0 * ldx + cg * 8, ldx, xr);
    if constexpr (ZMASK) rs.load(z + p0 * ldz + cg * 8, ldz, zr);
    if constexpr (MODE == 0) { if (use_bits) load_sign_bytes<ROWS>(rs, mask + p0 * VC + cg, VC, mk); }
    if constexpr (MODE == 1) {      // (behind the data loads; L1 hits from the second chunk on)
#pragma unroll
      for (int j = 0; j < 8; ++j) { ma[j] = mscale[cg * 8 + j]; mb[j] = mshift[cg * 8 + j]; }
    }
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      float g[8], xv[8];
      unpack8(gv[u], g);
      unpack8(xr[u], xv);
      const float* pp = post ? post + (unsigned)((unsigned)(p0 + rs.off[u]) / (unsigned)pix_per_img) * C + cg * 8 : nullptr;
      bn_bwd_mask(g, xv, zr[ZMASK ? u : 0], ZMASK && relu, relu, from_x, ma, mb, pp, use_bits,
                  (MODE == 0 && use_bits) ? mk[MODE == 0 ? u : 0] : 0u);
      const bool keep = (rs.ok >> u) & 1u;                      // masked rows re-read the chunk's first pixel
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gk = keep ? g[j] : 0.f;
        sg[j] += gk;
        sgx[j] += gk * xv[j];
      }
#ifndef SSA_EMU
      __builtin_amdgcn_sched_barrier(0);        // one row at a time (register pressure: see bn_apply_rows)
#endif
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sgx[j] = (sgx[j] - mean[cg * 8 + j] * sg[j]) * invstd[cg * 8 + j];
  block_reduce_2x8(sg, sgx, cg, C, active, sums, sh, bx, nrep);
}

// MODE: where the ReLU mask comes from -- 0: the forward's sign bytes (or there is no ReLU), 1: recomputed from x
// (mask scale / shift), 2: read off z.  Compile-time, so that each form holds only its own operands in registers.
template <int ROWS, int MODE>
__device__ __forceinline__ void bn_bwd_apply_body(
    const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ dz, int lddz,
    const bf16_t* __restrict__ z, int ldz, bf16_t* __restrict__ dx, int lddx,
    bf16_t* __restrict__ dres, int lddres, long P, int C, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const double* __restrict__ sums, int nrep, double inv_count, int relu,
    const float* __restrict__ post, long pix_per_img, long pix_per_block,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float param_grad_scale,
    const float* __restrict__ mscale, const float* __restrict__ mshift, const int accumulate_pg,
    const unsigned char* __restrict__ mask, const int bx) {
  // dx = a (g - c1 - xhat c2),  xhat = (x - mean) invstd,  a = gamma invstd,  c1 = sum_g / N,  c2 = sum_g_xhat / N
  //    = A g + (Bx x + D)   with   A = a,  Bx = -a c2 invstd,  D = a (c2 invstd mean - c1):
  // three per-channel constants in registers instead of five (this pass ran at 173 registers = TWO workgroups per CU;
  // a trunk level is ~930 workgroups, i.e. two rounds -- with <= 128 registers all of it is resident at once)
  SSA_DYN_LDS(float, sh);                 // [5C]: A, Bx, D, mask scale, mask shift
  const int VC = C >> 3, NA = active_threads(VC), RP = NA / VC;
  const int t = threadIdx.x;
  const bool active = t < NA;
  const int cg = active ? t % VC : 0, pr = active ? t / VC : 0;
  constexpr bool ZMASK = MODE == 2;
  const bool use_bits = MODE == 0 && relu && mask != nullptr;
  const bool from_x = MODE == 1 && relu;
  const long pb = bx * pix_per_block;
  const long pe = min(P, pb + pix_per_block);
  // ---- the first chunk's loads, then the coefficient prologue while they are in flight
  RowSet<ROWS> rs;
  rs.init(pb, pe, pr, RP, active);
  uint4 gv[ROWS], xr[ROWS], zr[ZMASK ? ROWS : 1];
  unsigned mk[MODE == 0 ? ROWS : 1];
  rs.load(dz + pb * lddz + cg * 8, lddz, gv);
  rs.load(x + pb * ldx + cg * 8, ldx, xr);
  if constexpr (ZMASK) rs.load(z + pb * ldz + cg * 8, ldz, zr);
  if constexpr (MODE == 0) { if (use_bits) load_sign_bytes<ROWS>(rs, mask + pb * VC + cg, VC, mk); }
  for (int c = t; c < C; c += NT) {
    double s1 = 0.0, s2 = 0.0;
    replica_sums(sums, nrep, C, c, &s1, &s2);
    const float is_ = invstd[c], mu_ = mean[c];
    const float a_ = (gamma ? gamma[c] : 1.f) * is_;
    const float c1_ = (float)(s1 * inv_count);    // (no fp64 division on every workgroup's critical path: see the apply)
    const float c2_ = (float)(s2 * inv_count);
    sh[c] = a_;
    sh[C + c] = -a_ * c2_ * is_;
    sh[2 * C + c] = a_ * (c2_ * is_ * mu_ - c1_);
    sh[3 * C + c] = mscale ? mscale[c] : 0.f;
    sh[4 * C + c] = mscale ? mshift[c] : 0.f;
    if (bx == 0) {
      // accumulate_pg: the gradient buffer is shared by every pass over this layer (cleared once
      // per step); passes grouped into one launch add concurrently, hence the atomics
      if (accumulate_pg) {
        if (dbeta) unsafeAtomicAdd(&dbeta[c], (float)(s1 * param_grad_scale));
        if (dgamma) unsafeAtomicAdd(&dgamma[c], (float)(s2 * param_grad_scale));
      } else {
        if (dbeta) dbeta[c] = (float)(s1 * param_grad_scale);
        if (dgamma) dgamma[c] = (float)(s2 * param_grad_scale);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  float A[8], Bx[8], D[8], ma[8], mb[8];
  {
    auto ld8 = [&](int row, float (&o)[8]) {
      const float4 v0 = *reinterpret_cast<const float4*>(sh + row * C + cg * 8), v1 = *reinterpret_cast<const float4*>(sh + row * C + cg * 8 + 4);
      o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
    };
    ld8(0, A); ld8(1, Bx); ld8(2, D);
    if constexpr (MODE == 1) { ld8(3, ma); ld8(4, mb); }
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { ma[j] = 0.f; mb[j] = 0.f; }
    }
  }
  for (long p0 = pb;;) {
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      float g[8], xv[8];
      unpack8(gv[u], g);
      unpack8(xr[u], xv);
      const float* pp = post ? post + (unsigned)((unsigned)(p0 + rs.off[u]) / (unsigned)pix_per_img) * C + cg * 8 : nullptr;
      bn_bwd_mask(g, xv, zr[ZMASK ? u : 0], ZMASK && relu, relu, from_x, ma, mb, pp, use_bits,
                  (MODE == 0 && use_bits) ? mk[MODE == 0 ? u : 0] : 0u);
      const bool ok = (rs.ok >> u) & 1u;
      if (dres && ok) *reinterpret_cast<uint4*>(dres + (p0 + rs.off[u]) * lddres + cg * 8) = pack8(g);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = A[j] * g[j] + (Bx[j] * xv[j] + D[j]);
      if (ok) *reinterpret_cast<uint4*>(dx + (p0 + rs.off[u]) * lddx + cg * 8) = pack8(o);
#ifndef SSA_EMU
      __builtin_amdgcn_sched_barrier(0);        // one row at a time (register pressure: see bn_apply_rows)
#endif
    }
    p0 += (long)RP * ROWS;
    if (p0 >= pe) break;
    rs.init(p0, pe, pr, RP, true);
    rs.load(dz + p0 * lddz + cg * 8, lddz, gv);
    rs.load(x + p0 * ldx + cg * 8, ldx, xr);
    if constexpr (ZMASK) rs.load(z + p0 * ldz + cg * 8, ldz, zr);
    if constexpr (MODE == 0) { if (use_bits) load_sign_bytes<ROWS>(rs, mask + p0 * VC + cg, VC, mk); }
  }
}

// ---- backward reduce + apply as ONE launch (round 6) ---------------------------------------------------------
// The two-launch form reads (x, dz, mask) twice: once to form sum g and sum g xhat, once more to apply them.  A trunk
// level is ~930 workgroups of one chunk each, all resident at once (<= 128 registers, 4 workgroups per CU): here every
// workgroup keeps its chunk IN REGISTERS across a grid-wide rendezvous -- phase 1 is the reduce (per-workgroup partial
// sums -> fp64 atomics), then one atomic ticket per workgroup and a spin on the ticket counter, then phase 2 derives the
// coefficients from the completed sums and writes dx (and the residual gradient).  The sums are read with agent-scope
// atomic loads: they were accumulated by atomics at the coherence point, and another XCD's L2 owes a plain load nothing
// before the kernel boundary.
// The rendezvous needs every workgroup of the PROBLEM resident (or able to become resident while others spin): the
// host only routes a bracket here whose workgroups all fit on the chip at once (ssa_bn_bwd_fused_blocks), kernels of
// other streams that hold CUs (the weight-gradient stream) finish on their own and free them.  A workgroup that has
// spun ~1 s gives up, counts itself in g_bn_fused_timeouts (ssa_bn_bwd_fused_timeouts) and applies what sums there are:
// a wrong gradient the tests and bench.py detect, never a hung GPU.
#ifndef SSA_BN_SPIN_SLEEP      // 64-clock units between two polls of the ticket counter (experiment builds)
#define SSA_BN_SPIN_SLEEP 8
#endif
__device__ unsigned g_bn_fused_timeouts;

__device__ __forceinline__ double coherent_load(const double* p) {
#ifdef SSA_EMU
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

template <int ROWS, int MODE>
__device__ __forceinline__ void bn_bwd_fused_body(
    const bf16_t* __restrict__ x, int ldx, const bf16_t* __restrict__ dz, int lddz,
    const bf16_t* __restrict__ z, int ldz, bf16_t* __restrict__ dx, int lddx,
    bf16_t* __restrict__ dres, int lddres, long P, int C, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    double* __restrict__ sums, int nrep, double inv_count, int relu,
    const float* __restrict__ post, long pix_per_img, long pix_per_block,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float param_grad_scale,
    const float* __restrict__ mscale, const float* __restrict__ mshift, const int accumulate_pg,
    const unsigned char* __restrict__ mask, unsigned* __restrict__ ticket, const int bx, const int nblocks) {
  SSA_DYN_LDS(float, sh);                 // phase 1: [16][NT + 1] transposed partials; phase 2: [5C] coefficients
  const int VC = C >> 3, NA = active_threads(VC), RP = NA / VC;
  const int t = threadIdx.x;
  const bool active = t < NA;
  const int cg = active ? t % VC : 0, pr = active ? t / VC : 0;
  constexpr bool ZMASK = MODE == 2;
  const bool use_bits = MODE == 0 && relu && mask != nullptr;
  const bool from_x = MODE == 1 && relu;
  const long pb = bx * pix_per_block;
  const long pe = min(P, pb + pix_per_block);   // ONE chunk: pix_per_block == RP * ROWS (the launcher's contract)
  RowSet<ROWS> rs;
  rs.init(pb, pe, pr, RP, active);
  uint4 gv[ROWS], xr[ROWS], zr[ZMASK ? ROWS : 1];
  unsigned mk[MODE == 0 ? ROWS : 1];
  rs.load(dz + pb * lddz + cg * 8, lddz, gv);
  rs.load(x + pb * ldx + cg * 8, ldx, xr);
  if constexpr (ZMASK) rs.load(z + pb * ldz + cg * 8, ldz, zr);
  if constexpr (MODE == 0) { if (use_bits) load_sign_bytes<ROWS>(rs, mask + pb * VC + cg, VC, mk); }
  // ---- phase 1: this workgroup's share of sum g and sum g xhat
  {
    float sg[8], sgx[8], ma[8], mb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sg[j] = 0.f; sgx[j] = 0.f; ma[j] = 0.f; mb[j] = 0.f; }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { ma[j] = mscale[cg * 8 + j]; mb[j] = mshift[cg * 8 + j]; }
    }
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      float g[8], xv[8];
      unpack8(gv[u], g);
      unpack8(xr[u], xv);
      const float* pp = post ? post + (unsigned)((unsigned)(pb + rs.off[u]) / (unsigned)pix_per_img) * C + cg * 8 : nullptr;
      bn_bwd_mask(g, xv, zr[ZMASK ? u : 0], ZMASK && relu, relu, from_x, ma, mb, pp, use_bits,
                  (MODE == 0 && use_bits) ? mk[MODE == 0 ? u : 0] : 0u);
      const bool keep = (rs.ok >> u) & 1u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gk = keep ? g[j] : 0.f;
        sg[j] += gk;
        sgx[j] += gk * xv[j];
      }
#ifndef SSA_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sgx[j] = (sgx[j] - mean[cg * 8 + j] * sg[j]) * invstd[cg * 8 + j];
    block_reduce_2x8(sg, sgx, cg, C, active, sums, sh, bx, nrep);
  }
  // ---- rendezvous: every workgroup's atomics are performed before its ticket is; the last ticket releases everybody
  // (no __threadfence(): an agent-scope release fence writes the XCD's whole L2 back, an acquire load invalidates it --
  // ~100 us per launch with 930 workgroups doing both, call P.  Nothing here needs either: the sums only ever see atomics,
  // which are performed at the coherence point, and the barrier below waits vmcnt(0) -- every atomic of this workgroup has
  // been acknowledged before thread 0 takes the ticket; the polls and the phase-2 loads of the sums are relaxed
  // agent-scope atomics, which bypass the caches; everything else phase 2 reads was written by earlier kernels)
  __syncthreads();
  if (t == 0) {
    unsigned seen = atomicAdd(ticket, 1u) + 1u;
    unsigned spins = 0;
    while (seen < (unsigned)nblocks) {
#ifndef SSA_EMU
      __builtin_amdgcn_s_sleep(SSA_BN_SPIN_SLEEP);
      seen = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
      seen = *ticket;
#endif
      if (++spins > (1u << 22)) { atomicAdd(&g_bn_fused_timeouts, 1u); break; }
    }
  }
  __syncthreads();
  // ---- phase 2: coefficients from the completed sums, then dx from the chunk still in registers
  for (int c = t; c < C; c += NT) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < nrep; ++r) {
      s1 += coherent_load(sums + (long)r * 2 * C + c);
      s2 += coherent_load(sums + (long)r * 2 * C + C + c);
    }
    const float is_ = invstd[c], mu_ = mean[c];
    const float a_ = (gamma ? gamma[c] : 1.f) * is_;
    const float c1_ = (float)(s1 * inv_count);
    const float c2_ = (float)(s2 * inv_count);
    sh[c] = a_;
    sh[C + c] = -a_ * c2_ * is_;
    sh[2 * C + c] = a_ * (c2_ * is_ * mu_ - c1_);
    sh[3 * C + c] = mscale ? mscale[c] : 0.f;
    sh[4 * C + c] = mscale ? mshift[c] : 0.f;
    if (bx == 0) {
      if (accumulate_pg) {
        if (dbeta) unsafeAtomicAdd(&dbeta[c], (float)(s1 * param_grad_scale));
        if (dgamma) unsafeAtomicAdd(&dgamma[c], (float)(s2 * param_grad_scale));
      } else {
        if (dbeta) dbeta[c] = (float)(s1 * param_grad_scale);
        if (dgamma) dgamma[c] = (float)(s2 * param_grad_scale);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  float A[8], Bx[8], D[8], ma[8], mb[8];
  {
    auto ld8 = [&](int row, float (&o)[8]) {
      const float4 v0 = *reinterpret_cast<const float4*>(sh + row * C + cg * 8), v1 = *reinterpret_cast<const float4*>(sh + row * C + cg * 8 + 4);
      o[0] = v0.x; o[1] = v0.y; o[2] = v0.z; o[3] = v0.w; o[4] = v1.x; o[5] = v1.y; o[6] = v1.z; o[7] = v1.w;
    };
    ld8(0, A); ld8(1, Bx); ld8(2, D);
    if constexpr (MODE == 1) { ld8(3, ma); ld8(4, mb); }
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) { ma[j] = 0.f; mb[j] = 0.f; }
    }
  }
#pragma unroll
  for (int u = 0; u < ROWS; ++u) {
    float g[8], xv[8];
    unpack8(gv[u], g);
    unpack8(xr[u], xv);
    const float* pp = post ? post + (unsigned)((unsigned)(pb + rs.off[u]) / (unsigned)pix_per_img) * C + cg * 8 : nullptr;
    bn_bwd_mask(g, xv, zr[ZMASK ? u : 0], ZMASK && relu, relu, from_x, ma, mb, pp, use_bits,
                (MODE == 0 && use_bits) ? mk[MODE == 0 ? u : 0] : 0u);
    const bool ok = (rs.ok >> u) & 1u;
    if (dres && ok) *reinterpret_cast<uint4*>(dres + (pb + rs.off[u]) * lddres + cg * 8) = pack8(g);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = A[j] * g[j] + (Bx[j] * xv[j] + D[j]);
    if (ok) *reinterpret_cast<uint4*>(dx + (pb + rs.off[u]) * lddx + cg * 8) = pack8(o);
#ifndef SSA_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

__global__ void bn_param_grads_kernel(const double* __restrict__ sums, int C,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = (float)sums[c];
  if (dgamma) dgamma[c] = (float)sums[C + c];
}


// ---- group-aware wrappers (group.h): one launch for all the BatchNorm calls of a depth level.  ROWS = pixel rows
// per thread and chunk (one instantiation per pass and build, so that a level's problems share a launch)
template <int ROWS>
struct BnStatsK {
  struct Args { const bf16_t* x; double* sums; long P, ppb; int C, ld; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int) {
    bn_stats_body<ROWS>(a.x, a.P, a.C, a.ld, a.sums, a.ppb, bx, true);
  }
};
template <int ROWS>
struct ColsumK {
  struct Args { const bf16_t* x; double* sums; long P, ppb; int C, ld; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int) {
    bn_stats_body<ROWS>(a.x, a.P, a.C, a.ld, a.sums, a.ppb, bx, false);
  }
};
template <int ROWS>
struct BnApplyK {
  struct Args { const bf16_t* x; const bf16_t* res; bf16_t* z; const float* scale; const float* shift;
                const float* post; long P, pix_per_img, ppb; int ldx, ldr, ldz, C, relu; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int) {
    bn_apply_body<ROWS>(a.x, a.ldx, a.res, a.ldr, a.z, a.ldz, a.P, a.C, a.scale, a.shift, a.relu, a.post,
                        a.pix_per_img, a.ppb, bx);
  }
};
template <int ROWS>
struct BnApplyTrainK {
  static constexpr int WPE = ROWS <= 4 ? 4 : 2;      // <= 128 registers: four workgroups per CU (a trunk level resident at once)
  struct Args { const bf16_t* x; const bf16_t* res; bf16_t* z; const double* sums; const float* gamma;
                const float* beta; float* running_mean; float* running_var; long* nbt; float* coef;
                float* pass_stats; const float* post; unsigned char* mask; double count, inv_count, unbias;
                long P, pix_per_img, ppb; int ldx, ldr, ldz, C, nrep, relu; float momentum, eps; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int) {
    bn_apply_train_body<ROWS>(a.x, a.ldx, a.res, a.ldr, a.z, a.ldz, a.P, a.C, a.sums, a.nrep, a.count, a.inv_count, a.unbias, a.gamma,
                              a.beta, a.running_mean, a.running_var, a.nbt, a.momentum, a.eps, a.coef,
                              a.pass_stats, a.relu, a.post, a.pix_per_img, a.ppb, a.mask, bx);
  }
};
template <int ROWS, int MODE>
struct BnBwdReduceKM {
  static constexpr int WPE = ROWS <= 4 ? 4 : 2;      // 4 rows: <= 128 registers (see bn_bwd_reduce_body)
  struct Args { const bf16_t* x; const bf16_t* dz; const bf16_t* z; const float* mean; const float* invstd;
                const float* post; double* sums; const float* mscale; const float* mshift; const unsigned char* mask;
                long P, pix_per_img, ppb; int ldx, lddz, ldz, C, relu, nrep; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int) {
    bn_bwd_reduce_body<ROWS, MODE>(a.x, a.ldx, a.dz, a.lddz, a.z, a.ldz, a.P, a.C, a.mean, a.invstd, a.relu, a.post,
                             a.pix_per_img, a.sums, a.nrep, a.ppb, a.mscale, a.mshift, a.mask, bx);
  }
};
template <int ROWS> struct BnBwdReduceK : BnBwdReduceKM<ROWS, 0> {};
template <int ROWS> struct BnBwdReduceXK : BnBwdReduceKM<ROWS, 1> {};
template <int ROWS> struct BnBwdReduceZK : BnBwdReduceKM<ROWS, 2> {};

template <int ROWS, int MODE>
struct BnBwdApplyKM {
  static constexpr int WPE = ROWS <= 4 ? 4 : 2;  // <= 128 registers: four workgroups per CU (see bn_bwd_apply_body)
  struct Args { const bf16_t* x; const bf16_t* dz; const bf16_t* z; bf16_t* dx; bf16_t* dres;
                const float* gamma; const float* mean; const float* invstd; const double* sums;
                const float* post; float* dgamma; float* dbeta; const float* mscale; const float* mshift;
                const unsigned char* mask; double inv_count; long P, pix_per_img, ppb; int ldx, lddz, ldz, lddx, lddres, C, nrep, relu;
                float param_grad_scale; int accumulate_pg; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int) {
    bn_bwd_apply_body<ROWS, MODE>(a.x, a.ldx, a.dz, a.lddz, a.z, a.ldz, a.dx, a.lddx, a.dres, a.lddres, a.P, a.C,
                            a.gamma, a.mean, a.invstd, a.sums, a.nrep, a.inv_count, a.relu, a.post, a.pix_per_img,
                            a.ppb, a.dgamma, a.dbeta, a.param_grad_scale, a.mscale, a.mshift, a.accumulate_pg, a.mask, bx);
  }
};
// (BnBwdApplyK is the name the profiles know: the sign-byte / no-ReLU form -- bn2 of a block; XK recomputes the mask
// from x -- bn1; ZK reads z)
template <int ROWS> struct BnBwdApplyK : BnBwdApplyKM<ROWS, 0> {};
template <int ROWS> struct BnBwdApplyXK : BnBwdApplyKM<ROWS, 1> {};
template <int ROWS> struct BnBwdApplyZK : BnBwdApplyKM<ROWS, 2> {};

template <int ROWS, int MODE>
struct BnBwdFusedKM {
  static constexpr int WPE = 4;                  // <= 128 registers: four workgroups per CU -- what the residency count assumes
  struct Args { const bf16_t* x; const bf16_t* dz; const bf16_t* z; bf16_t* dx; bf16_t* dres;
                const float* gamma; const float* mean; const float* invstd; double* sums;
                const float* post; float* dgamma; float* dbeta; const float* mscale; const float* mshift;
                const unsigned char* mask; unsigned* ticket; double inv_count; long P, pix_per_img, ppb;
                int ldx, lddz, ldz, lddx, lddres, C, nrep, relu; float param_grad_scale; int accumulate_pg; };
  static constexpr int NT = ::NT;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    bn_bwd_fused_body<ROWS, MODE>(a.x, a.ldx, a.dz, a.lddz, a.z, a.ldz, a.dx, a.lddx, a.dres, a.lddres, a.P, a.C,
                                  a.gamma, a.mean, a.invstd, a.sums, a.nrep, a.inv_count, a.relu, a.post, a.pix_per_img,
                                  a.ppb, a.dgamma, a.dbeta, a.param_grad_scale, a.mscale, a.mshift, a.accumulate_pg,
                                  a.mask, a.ticket, bx, gx);
  }
};
struct BnBwdFusedK : BnBwdFusedKM<4, 0> {};
struct BnBwdFusedXK : BnBwdFusedKM<4, 1> {};
struct BnBwdFusedZK : BnBwdFusedKM<4, 2> {};


__global__ void d2f_kernel(const double* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// Deferred running-statistics update: one launch for every BatchNorm layer of the
// network, each layer's passes (the 0.5x and the 1.0x pass run on concurrent
// streams) applied in issue order -- same result as the reference's sequential
// in-place updates (momentum, unbiased variance), no race between the streams.
struct BnUpdateJob {      // mirror of ssa_bn_update_job (include/semseg_hip.h), 96 bytes
  float* running_mean;
  float* running_var;
  long* num_batches_tracked;
  const float* pass_stats[8];   // each: mean[C], biased var[C], count
  int C, npass;
  float momentum;
  int pad_;
};

struct BnEvalJob {          // mirror of ssa_bn_eval_job (include/semseg_hip.h)
  const float* gamma; const float* beta; const float* running_mean; const float* running_var;
  float* coef;              // [4][C]: scale, shift, mean, invstd
  int C; float eps;
};

__global__ __launch_bounds__(128) void bn_finalize_eval_batched_kernel(const BnEvalJob* __restrict__ jobs) {
  const BnEvalJob j = jobs[blockIdx.y];
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c >= j.C) return;
  // the arithmetic of bn_finalize_kernel(use_running = 1)
  const double mean = j.running_mean[c], var = j.running_var[c];
  const double invstd = 1.0 / sqrt(var + (double)j.eps);
  const double g = j.gamma ? (double)j.gamma[c] : 1.0;
  const double b = j.beta ? (double)j.beta[c] : 0.0;
  j.coef[c] = (float)(g * invstd);
  j.coef[j.C + c] = (float)(b - mean * g * invstd);
  j.coef[2 * j.C + c] = (float)mean;
  j.coef[3 * j.C + c] = (float)invstd;
}

__global__ __launch_bounds__(128) void bn_update_running_kernel(const BnUpdateJob* __restrict__ jobs) {
  const BnUpdateJob j = jobs[blockIdx.y];
  const int c = blockIdx.x * 128 + threadIdx.x;
  if (c == 0 && j.num_batches_tracked) *j.num_batches_tracked += j.npass;
  if (c >= j.C) return;
  double rm = j.running_mean[c], rv = j.running_var[c];
  for (int p = 0; p < j.npass; ++p) {
    const float* ps = j.pass_stats[p];
    const double mean = ps[c], var = ps[j.C + c], count = ps[2 * j.C];
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rm = (float)((1.0 - j.momentum) * rm + j.momentum * mean);
    rv = (float)((1.0 - j.momentum) * rv + j.momentum * unbiased);
  }
  j.running_mean[c] = (float)rm;
  j.running_var[c] = (float)rv;
}

struct Grid { int blocks; long ppb; int rows; };
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
int norm_rows(int r) { return r <= 2 ? 2 : (r <= 4 ? 4 : 8); }
// `rows` pixel rows per thread (2, 4 or 8: the instantiation) in one chunk per workgroup; at most max_blocks
// workgroups, beyond which a workgroup walks several chunks (the reducing kernels end in 2C fp64 atomics per
// workgroup, so they get a lower cap).
Grid plan_grid(long P, int C, int rows, long max_blocks = 16384, int chunks = 1) {
  const int VC = C >> 3;
  const int RP = active_threads(VC) / VC;
  const long chunk = (long)RP * rows;
  long ppb = chunk;
  long blocks = (P + ppb - 1) / ppb;
  // `chunks` > 1 (the passes with a per-workgroup coefficient prologue on wide layers): a workgroup walks that many
  // chunks as long as the launch still fills every workgroup slot of the chip (4 per CU)
  if (chunks > 1 && blocks > 1024) {
    long want = (blocks + chunks - 1) / chunks;
    if (want < 1024) want = 1024;
    if (want < max_blocks) max_blocks = want;
  }
  if (blocks > max_blocks) {
    ppb = (P + max_blocks - 1) / max_blocks;
    ppb = (ppb + chunk - 1) / chunk * chunk;
    blocks = (P + ppb - 1) / ppb;
  }
  if (blocks < 1) blocks = 1;
  return {(int)blocks, ppb, rows};
}
// Rows per thread of the three pass families (SSA_BN_ROWS_*: 2 / 4 / 8; sweep in profiles/r04_notes.md)
// The training apply and the backward apply derive their per-channel coefficients from the [nrep][2][C] fp64 sums in
// EVERY workgroup (128 B per channel: 92 KB at the aux head's 720 channels -- against the 11.5 KB of data of that layer's
// 8-pixel chunk: a 10,240-workgroup launch read 0.94 GB of sums for 0.24 GB of activations, 102 us).  From 256 channels
// on a workgroup therefore walks several chunks per prologue (SSA_BN_WIDE_CHUNKS, 0 = one chunk as before).
int prologue_chunks(int C) {
  static const int n = env_int("SSA_BN_WIDE_CHUNKS", 6);
  return (C >= 256 && n > 1) ? n : 1;
}
int apply_rows() { static const int r = norm_rows(env_int("SSA_BN_ROWS_APPLY", 4)); return r; }
int bwd_rows() { static const int r = norm_rows(env_int("SSA_BN_ROWS_BWD", 4)); return r; }
int reduce_rows() { static const int r = norm_rows(env_int("SSA_BN_ROWS_REDUCE", 8)); return r; }
Grid plan_reduce_grid(long P, int C) {
  static const int cap = env_int("SSA_BN_REDUCE_BLOCKS", 2048);
  return plan_grid(P, C, reduce_rows(), cap);
}
// submit<K<ROWS>> for the planned row count
#define SSA_BN_SUBMIT(K, g, a, lds, s)                                              \
  ((g).rows == 2 ? ssa::submit<K<2>>(K<2>::Args a, (g).blocks, 1, lds, s)            \
   : (g).rows == 4 ? ssa::submit<K<4>>(K<4>::Args a, (g).blocks, 1, lds, s)          \
                   : ssa::submit<K<8>>(K<8>::Args a, (g).blocks, 1, lds, s))

bool ok_c(int C) { return C > 0 && C % 8 == 0 && C <= 2048; }
bool ok_p(long P) { return P > 0 && P < (1L << 31); }     // pixel indices are 32-bit inside a chunk

}  // namespace

extern "C" {

int ssa_bn_stats(const void* x, long P, int C, int ld, double* sums, int zero_sums, void* stream) {
  if (!x || !sums || !ok_c(C) || !ok_p(P) || ld % 8) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (zero_sums) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, s);
    if (e != hipSuccess) return (int)e;
  }
  const Grid g = plan_reduce_grid(P, C);
  return SSA_BN_SUBMIT(BnStatsK, g, ({(const bf16_t*)x, sums, P, g.ppb, C, ld}), 16 * (NT + 1) * sizeof(float), s);
}

int ssa_bn_finalize(const double* sums, double count, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    int use_running, float* scale, float* shift, float* mean, float* invstd,
                    void* stream) {
  if (!scale || !shift || C <= 0) return SSA_EINVAL;
  if (use_running ? (!running_mean || !running_var) : (!sums || count <= 0)) return SSA_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                     sums, count, C, gamma, beta, running_mean, running_var, momentum, eps,
                     use_running, scale, shift, mean, invstd);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_bn_apply(const void* x, int ldx, const void* residual, int ldr, void* z, int ldz, long P,
                 int C, const float* scale, const float* shift, int relu, const float* post,
                 long pix_per_img, void* stream) {
  if (!x || !z || !scale || !shift || !ok_c(C) || !ok_p(P) || ldx % 8 || ldz % 8 || (residual && ldr % 8))
    return SSA_EINVAL;
  const Grid g = plan_grid(P, C, apply_rows());
  return SSA_BN_SUBMIT(BnApplyK, g, ({(const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)z, scale, shift, post, P,
                                      pix_per_img, g.ppb, ldx, ldr, ldz, C, relu}), 0, (hipStream_t)stream);
}

int ssa_bn_apply_train(const void* x, int ldx, const void* residual, int ldr, void* z, int ldz,
                       long P, int C, const double* sums, int nrep, double count, const float* gamma,
                       const float* beta, float* running_mean, float* running_var,
                       long* num_batches_tracked, float momentum, float eps, float* coef,
                       float* pass_stats, int relu, const float* post, long pix_per_img,
                       void* sign_mask, void* stream) {
  if (!x || !z || !sums || !coef || !ok_c(C) || !ok_p(P) || ldx % 8 || ldz % 8 || (residual && ldr % 8) ||
      count <= 0 || nrep < 1 || (running_mean && !running_var))
    return SSA_EINVAL;
  const Grid g = plan_grid(P, C, apply_rows(), 16384, prologue_chunks(C));
  return SSA_BN_SUBMIT(BnApplyTrainK, g, ({(const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)z, sums, gamma, beta,
                                           running_mean, running_var, num_batches_tracked, coef, pass_stats, post,
                                           (unsigned char*)sign_mask, count, 1.0 / count,
                                           count > 1.0 ? count / (count - 1.0) : 1.0,
                                           P, pix_per_img, g.ppb, ldx, ldr, ldz, C, nrep, relu, momentum, eps}),
                       2 * C * sizeof(float), (hipStream_t)stream);
}

// Evaluation-mode coefficients of MANY BatchNorm layers in one launch: scale / shift (and mean / invstd) from the running
// statistics -- constants of an evaluation forward that were one single-workgroup launch per layer and scale pass
// (314 per single-scale forward of HRNet-OCR: 17 % of its time, profiles/r06_notes.md call EV).
int ssa_bn_finalize_eval_batched(const void* jobs_dev, int njobs, int max_channels, void* stream) {
  if (!jobs_dev || njobs < 1 || max_channels < 1) return SSA_EINVAL;
  static_assert(sizeof(BnEvalJob) == 48, "ssa_bn_eval_job layout");
  hipLaunchKernelGGL(bn_finalize_eval_batched_kernel, dim3((max_channels + 127) / 128, njobs), dim3(128), 0,
                     (hipStream_t)stream, (const BnEvalJob*)jobs_dev);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_bn_update_running_batched(const void* jobs_dev, int njobs, int max_channels, void* stream) {
  if (!jobs_dev || njobs < 1 || max_channels < 1) return SSA_EINVAL;
  static_assert(sizeof(BnUpdateJob) == 104, "ssa_bn_update_job layout");
  hipLaunchKernelGGL(bn_update_running_kernel, dim3((max_channels + 127) / 128, njobs), dim3(128), 0,
                     (hipStream_t)stream, (const BnUpdateJob*)jobs_dev);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_bn_bwd_reduce(const void* x, int ldx, const void* dz, int lddz, const void* z, int ldz,
                      long P, int C, const float* mean, const float* invstd, int relu,
                      const float* post, long pix_per_img, double* sums, int nrep, int zero_sums,
                      const float* mask_scale, const float* mask_shift, const void* sign_mask, void* stream) {
  if (!x || !dz || !sums || !mean || !invstd || !ok_c(C) || !ok_p(P) || (relu && !z && !mask_scale && !sign_mask) || nrep < 1 ||
      (mask_scale && !mask_shift))
    return SSA_EINVAL;
  if (ldx % 8 || lddz % 8 || (z && ldz % 8)) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (zero_sums) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C * nrep, s);
    if (e != hipSuccess) return (int)e;
  }
  const Grid g = plan_reduce_grid(P, C);
#define SSA_BN_BWD_REDUCE(K)                                                                                                  \
  SSA_BN_SUBMIT(K, g, ({(const bf16_t*)x, (const bf16_t*)dz, (const bf16_t*)z, mean, invstd, post, sums, mask_scale, mask_shift, \
                        (const unsigned char*)sign_mask, P, pix_per_img, g.ppb, ldx, lddz, ldz, C, relu, nrep}),               \
                16 * (NT + 1) * sizeof(float), s)
  if (relu && mask_scale) return SSA_BN_BWD_REDUCE(BnBwdReduceXK);
  if (relu && !sign_mask) return SSA_BN_BWD_REDUCE(BnBwdReduceZK);
  return SSA_BN_BWD_REDUCE(BnBwdReduceK);
#undef SSA_BN_BWD_REDUCE
}

int ssa_bn_bwd_apply(const void* x, int ldx, const void* dz, int lddz, const void* z, int ldz,
                     void* dx, int lddx, void* dres, int lddres, long P, int C, const float* gamma,
                     const float* mean, const float* invstd, const double* sums, int nrep,
                     double count, int relu, const float* post, long pix_per_img, float* dgamma,
                     float* dbeta, float param_grad_scale, const float* mask_scale,
                     const float* mask_shift, int accumulate_param_grads, const void* sign_mask, void* stream) {
  if (!x || !dz || !dx || !sums || !mean || !invstd || !ok_c(C) || !ok_p(P) || (relu && !z && !mask_scale && !sign_mask) || nrep < 1 ||
      (mask_scale && !mask_shift))
    return SSA_EINVAL;
  if (ldx % 8 || lddz % 8 || lddx % 8 || (z && ldz % 8) || (dres && lddres % 8)) return SSA_EINVAL;
  const Grid g = plan_grid(P, C, bwd_rows(), 16384, prologue_chunks(C));
  // one instantiation per source of the ReLU mask (bn_bwd_apply_body's MODE)
#define SSA_BN_BWD_APPLY(K)                                                                                              \
  SSA_BN_SUBMIT(K, g, ({(const bf16_t*)x, (const bf16_t*)dz, (const bf16_t*)z, (bf16_t*)dx, (bf16_t*)dres, gamma, mean, invstd, \
                        sums, post, dgamma, dbeta, mask_scale, mask_shift, (const unsigned char*)sign_mask, 1.0 / count, P,    \
                        pix_per_img, g.ppb, ldx, lddz, ldz, lddx, lddres, C, nrep, relu, param_grad_scale,                     \
                        accumulate_param_grads}), 5 * C * sizeof(float), (hipStream_t)stream)
  if (relu && mask_scale) return SSA_BN_BWD_APPLY(BnBwdApplyXK);
  if (relu && !sign_mask) return SSA_BN_BWD_APPLY(BnBwdApplyZK);
  return SSA_BN_BWD_APPLY(BnBwdApplyK);
#undef SSA_BN_BWD_APPLY
}

// One-launch backward (bn_bwd_fused_body).  ssa_bn_bwd_fused_blocks: the workgroups the problem takes (one 4-row chunk
// each) -- the caller sums them over the bracket and uses this entry point only while the total stays within
// ssa_bn_bwd_fused_capacity() (every workgroup of the launch resident at once); 0: not supported (emulation build,
// odd shapes).  `ticket`: one zeroed 32-bit word per call.
int ssa_bn_bwd_fused_blocks(long P, int C) {
#ifdef SSA_EMU
  return 0;                                      // workgroups run one after another on the host: no rendezvous
#else
  if (!ok_c(C) || !ok_p(P)) return 0;
  const Grid g = plan_grid(P, C, 4);
  const int VC = C >> 3;
  if (g.ppb != (long)(active_threads(VC) / VC) * 4) return 0;      // more than one chunk per workgroup
  return g.blocks;
#endif
}

int ssa_bn_bwd_fused_capacity(void) {
#ifdef SSA_EMU
  return 0;
#else
  static int cap = -1;
  if (cap < 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 0;
    cap = 4 * cus;                               // WPE = 4, 40 KB of LDS at most: four workgroups per CU
  }
  return cap;
#endif
}

int ssa_bn_bwd_fused_timeouts(unsigned* out) {
  if (!out) return SSA_EINVAL;
#ifdef SSA_EMU
  *out = g_bn_fused_timeouts;
  return SSA_OK;
#else
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bn_fused_timeouts), sizeof(unsigned));
  return e == hipSuccess ? SSA_OK : (int)e;
#endif
}

int ssa_bn_bwd_fused(const void* x, int ldx, const void* dz, int lddz, const void* z, int ldz,
                     void* dx, int lddx, void* dres, int lddres, long P, int C, const float* gamma,
                     const float* mean, const float* invstd, double* sums, int nrep,
                     double count, int relu, const float* post, long pix_per_img, float* dgamma,
                     float* dbeta, float param_grad_scale, const float* mask_scale,
                     const float* mask_shift, int accumulate_param_grads, const void* sign_mask, void* ticket,
                     void* stream) {
  if (!x || !dz || !dx || !sums || !mean || !invstd || !ticket || !ok_c(C) || !ok_p(P) ||
      (relu && !z && !mask_scale && !sign_mask) || nrep < 1 || (mask_scale && !mask_shift))
    return SSA_EINVAL;
  if (ldx % 8 || lddz % 8 || lddx % 8 || (z && ldz % 8) || (dres && lddres % 8)) return SSA_EINVAL;
  if (ssa_bn_bwd_fused_blocks(P, C) <= 0) return SSA_EUNSUPPORTED;
  const Grid g = plan_grid(P, C, 4);
  const size_t lds_red = 16 * (NT + 1) * sizeof(float), lds_coef = 5 * (size_t)C * sizeof(float);
  const size_t lds = lds_red > lds_coef ? lds_red : lds_coef;
#define SSA_BN_BWD_FUSED(K)                                                                                               \
  ssa::submit<K>(K::Args{(const bf16_t*)x, (const bf16_t*)dz, (const bf16_t*)z, (bf16_t*)dx, (bf16_t*)dres, gamma, mean, invstd, \
                         sums, post, dgamma, dbeta, mask_scale, mask_shift, (const unsigned char*)sign_mask,                  \
                         (unsigned*)ticket, 1.0 / count, P, pix_per_img, g.ppb, ldx, lddz, ldz, lddx, lddres, C, nrep, relu,   \
                         param_grad_scale, accumulate_param_grads}, g.blocks, 1, lds, (hipStream_t)stream)
  if (relu && mask_scale) return SSA_BN_BWD_FUSED(BnBwdFusedXK);
  if (relu && !sign_mask) return SSA_BN_BWD_FUSED(BnBwdFusedZK);
  return SSA_BN_BWD_FUSED(BnBwdFusedK);
#undef SSA_BN_BWD_FUSED
}

int ssa_bn_param_grads(const double* sums, int C, float* dgamma, float* dbeta, void* stream) {
  if (!sums || C <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 127) / 128), dim3(128), 0,
                     (hipStream_t)stream, sums, C, dgamma, dbeta);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_colsum_bf16(const void* x, long P, int C, int ld, float* out, double* scratch2c,
                    void* stream) {
  if (!x || !out || !scratch2c || !ok_c(C) || !ok_p(P) || ld % 8) return SSA_EINVAL;
  if (ssa::group_state().depth > 0) return SSA_EINVAL;   // three dependent launches: not inside a group bracket
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(scratch2c, 0, sizeof(double) * 2 * C, s);
  if (e != hipSuccess) return (int)e;
  // one replica of the sums: every workgroup ends in C same-address fp64 atomics -- 2,048 workgroups on a
  // 65,536 x 512 gradient ran 59 us for 67 MB (1.1 TB/s), the atomics' queue, not the reads; 256 workgroups here
  const Grid g = plan_grid(P, C, reduce_rows(), 256);
  if (int rc = SSA_BN_SUBMIT(ColsumK, g, ({(const bf16_t*)x, scratch2c, P, g.ppb, C, ld}), 16 * (NT + 1) * sizeof(float), s))
    return rc;
  hipLaunchKernelGGL(d2f_kernel, dim3((C + 127) / 128), dim3(128), 0, s, scratch2c, out, C);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
