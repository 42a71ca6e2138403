// Per-pixel losses of the path, on fp32 NHWC logits [P, C] + int64 labels [P].
//   * CrossEntropyLoss2d            loss/utils.py:121-134        (K13)
//   * RMILoss part I  (masked BCE)  loss/rmi.py:91-116           (K14)
//   * RMILoss part II (RMI bound)   loss/rmi.py:139-215,
//                                   loss/rmi_utils.py:15-56,95-107 (K15)
// Forward kernels also emit the un-normalised gradient so the backward is one
// scaling pass.  RMI never materialises the [B,C,9,65025] fp64 neighbourhood
// stack: Gram matrices are accumulated straight from the pooled maps.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"
#include <float.h>

namespace {

constexpr int NT = 256;
constexpr float kClipMin = 1e-6f;   // loss/rmi.py:24  _CLIP_MIN
constexpr double kPosAlpha = 5e-4;  // loss/rmi.py:26  _POS_ALPHA

__device__ __forceinline__ void block_acc2(double a, double b, double* acc) {
  __shared__ double red[2 * (NT / 64)];
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
    for (int i = 0; i < NT / 64; ++i) { sa += red[2 * i]; sb += red[2 * i + 1]; }
    atomicAdd(&acc[0], sa);
    atomicAdd(&acc[1], sb);
  }
}

// MODE 0: softmax cross entropy;  MODE 1: masked sigmoid BCE (sum over classes)
template <int MODE>
__global__ __launch_bounds__(NT) void pixel_loss_kernel(const float* __restrict__ logits, int ld,
                                                        const int64_t* __restrict__ labels, long P,
                                                        int C, int ignore_index,
                                                        double* __restrict__ acc,
                                                        float* __restrict__ dlogits) {
  SSA_DYN_LDS(float, tile);  // [NT][C]
  double lsum = 0.0, lcnt = 0.0;
  for (long base = (long)blockIdx.x * NT; base < P; base += (long)gridDim.x * NT) {
    const long npx = min((long)NT, P - base);
    // coalesced stage-in
    if (ld == C) {
      const float* src = logits + base * C;
      for (long i = threadIdx.x; i < npx * C; i += NT) tile[i] = src[i];
    } else {
      for (long i = threadIdx.x; i < npx * C; i += NT) tile[i] = logits[(base + i / C) * ld + i % C];
    }
    __syncthreads();
    if (threadIdx.x < npx) {
      float* x = tile + threadIdx.x * C;
      const long lab = labels[base + threadIdx.x];
      if (MODE == 0) {
        const bool valid = lab != ignore_index && lab >= 0 && lab < C;
        float mx = -FLT_MAX;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, x[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        if (valid) { lsum += (double)(lse - x[lab]); lcnt += 1.0; }
        if (dlogits) {
          const float inv = 1.f / se;
          for (int c = 0; c < C; ++c) {
            const float p = expf(x[c] - mx) * inv;
            x[c] = valid ? (p - (c == lab ? 1.f : 0.f)) : 0.f;
          }
        }
      } else {
        const bool valid = lab >= 0 && lab < C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
          const float v = x[c], t = (c == lab) ? 1.f : 0.f;
          // stable BCE-with-logits: max(v,0) - v*t + log1p(exp(-|v|))
          s += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
          if (dlogits) x[c] = valid ? (1.f / (1.f + expf(-v)) - t) : 0.f;
        }
        if (valid) { lsum += (double)s; lcnt += 1.0; }
      }
    }
    __syncthreads();
    if (dlogits) {
      float* dst = dlogits + base * C;  // gradient buffer is dense [P, C]
      for (long i = threadIdx.x; i < npx * C; i += NT) dst[i] = tile[i];
    }
    __syncthreads();
  }
  block_acc2(lsum, lcnt, acc);
}

// Masked sigmoid BCE, dense logits: nothing couples the classes of a pixel, so the tile kernel's stage-in / per-pixel
// pass / stage-out with three barriers is not needed -- every thread streams four consecutive elements (a float4,
// possibly across a pixel boundary) and the label of their pixel(s).  The tile kernel spends its time in two expf, a
// log1pf and a division per element (library routines of 20-40 instructions each: 102 us at 1024x1024x19, no faster
// when merely streamed); here ONE exponential  e = exp(-|v|)  serves both terms,
//     log(1 + exp(-|v|)) = log(1 + e)         sigmoid(v) = v >= 0 ? 1 / (1 + e) : e / (1 + e)
// with the hardware exp2 / log2 / rcp (1 ulp-class; the loss changes in the 7th digit, tests/test_kernels_gpu.py's
// tolerance for this op is 1e-4).  fp32 partial sum per float4, fp64 across them.
__global__ __launch_bounds__(NT) void bce_stream_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                        unsigned n4, int C, double* __restrict__ acc,
                                                        float* __restrict__ dlogits) {
  double lsum = 0.0, lcnt = 0.0;
  const unsigned last_pix = (n4 * 4u) / (unsigned)C - 1u;
  // one float4 and the labels of its (at most two, C >= 4) pixels
  auto element = [&](unsigned g, const float4 v4, unsigned pix, long lab, const long lab_next) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float d[4];
    const unsigned pix0 = pix;
    int c = (int)(g * 4u - pix * (unsigned)C);
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c == C) { c = 0; ++pix; lab = (pix == pix0 + 1u) ? lab_next : labels[pix]; }
      const bool valid = lab >= 0 && lab < C;
      const float t = (c == lab) ? 1.f : 0.f;
      const float e = __expf(-fabsf(v[k]));
      const float r = __frcp_rn(1.f + e);
      const float s = fmaxf(v[k], 0.f) - v[k] * t + __logf(1.f + e);
      if (valid) { part += s; if (c == 0) lcnt += 1.0; }
      d[k] = valid ? ((v[k] >= 0.f ? r : e * r) - t) : 0.f;
      ++c;
    }
    lsum += (double)part;
    if (dlogits) reinterpret_cast<float4*>(dlogits)[g] = make_float4(d[0], d[1], d[2], d[3]);
  };
  // four elements per trip, their loads issued before the arithmetic of any: with one float4 per thread in flight
  // (262 K threads x 16 B = 4 MB) the 160 MB of a 1024 x 1024 x 19 call took 70 us (2.3 TB/s).  512 workgroups: every
  // one ends in two same-address fp64 atomics, which queue at ~25 ns each (ColsumK, profiles/r06_notes.md call V).
  const unsigned stride = gridDim.x * NT;
  for (unsigned g0 = blockIdx.x * NT + threadIdx.x; g0 < n4; g0 += 4u * stride) {
    float4 v[4];
    unsigned gi[4], pi[4];
    long l0[4], l1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gi[u] = g0 + u * stride < n4 ? g0 + u * stride : g0;
      v[u] = reinterpret_cast<const float4*>(logits)[gi[u]];
      pi[u] = (gi[u] * 4u) / (unsigned)C;
      l0[u] = labels[pi[u]];
      l1[u] = labels[pi[u] < last_pix ? pi[u] + 1u : last_pix];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u == 0 || g0 + u * stride < n4) element(gi[u], v[u], pi[u], l0[u], l1[u]);
  }
  block_acc2(lsum, lcnt, acc);
}

// Backward of the dense masked BCE without a saved gradient: d = (sigmoid(v) - onehot) * upstream * coef / count for
// the valid pixels, recomputed from the logits with the forward's arithmetic (one exponential, hardware rcp) -- the
// forward then writes no gradient at all (80 MB per loss term at 1024 x 1024 x 19) and the scaling pass that read it
// back (ssa_scale_grad_to) is this kernel.
__global__ __launch_bounds__(NT) void bce_bwd_stream_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                            unsigned n4, int C, const float* __restrict__ upstream, double coef,
                                                            const double* __restrict__ acc, double denom_add,
                                                            float* __restrict__ dlogits) {
  const float s = (float)((double)upstream[0] * coef / (acc[1] + denom_add));
  const unsigned last_pix = (n4 * 4u) / (unsigned)C - 1u;
  auto element = [&](unsigned g, const float4 v4, unsigned pix, long lab, const long lab_next) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float d[4];
    const unsigned pix0 = pix;
    int c = (int)(g * 4u - pix * (unsigned)C);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c == C) { c = 0; ++pix; lab = (pix == pix0 + 1u) ? lab_next : labels[pix]; }
      const bool valid = lab >= 0 && lab < C;
      const float t = (c == lab) ? 1.f : 0.f;
      const float e = __expf(-fabsf(v[k]));
      const float r = __frcp_rn(1.f + e);
      d[k] = valid ? ((v[k] >= 0.f ? r : e * r) - t) * s : 0.f;
      ++c;
    }
    reinterpret_cast<float4*>(dlogits)[g] = make_float4(d[0], d[1], d[2], d[3]);
  };
  const unsigned stride = gridDim.x * NT;
  for (unsigned g = blockIdx.x * NT + threadIdx.x; g < n4; g += 2u * stride) {
    const bool two = g + stride < n4;
    const unsigned g2 = two ? g + stride : g;
    const float4 va = reinterpret_cast<const float4*>(logits)[g];
    const float4 vb = reinterpret_cast<const float4*>(logits)[g2];
    const unsigned pa = (g * 4u) / (unsigned)C, pb = (g2 * 4u) / (unsigned)C;
    const long la = labels[pa], la1 = labels[pa < last_pix ? pa + 1u : last_pix];
    const long lb = labels[pb], lb1 = labels[pb < last_pix ? pb + 1u : last_pix];
    element(g, va, pa, la, la1);
    if (two) element(g2, vb, pb, lb, lb1);
  }
}

__global__ void loss_finalize_kernel(const double* __restrict__ acc, double denom_add,
                                     float* __restrict__ loss) {
  loss[0] = (float)(acc[0] / (acc[1] + denom_add));
}

// dst = src * upstream * coef / (acc[1] + denom_add)     (src may be dst; otherwise the two do not overlap)
__global__ void scale_grad_kernel(const float* src, float* dst, long n, const float* __restrict__ upstream,
                                  double coef, const double* __restrict__ acc, double denom_add) {
  const float s = (float)((double)upstream[0] * coef / (acc[1] + denom_add));
  const long n4 = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0 ? n >> 2 : 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(src)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    reinterpret_cast<float4*>(dst)[i] = v;
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[i] * s;
}

// ------------------------------------------------------------------- RMI
// pooled_pr[b,c,py,px] = avg over the 4x4 cell (zero padded, /16) of
//   sigmoid(logit)*mask + 1e-6 ; pooled_la likewise of onehot*mask.
// One block = one pooled row, 32 pooled columns: stages 4 x 128 input pixels.
constexpr int CELLS = 32;
__global__ __launch_bounds__(NT) void rmi_pool_kernel(const float* __restrict__ logits, int ld,
                                                      const int64_t* __restrict__ labels, int H,
                                                      int W, int C, float* __restrict__ ppr,
                                                      float* __restrict__ pla, int Hp, int Wp) {
  SSA_DYN_LDS(float, sm);  // [4][CELLS*4][C] probs, then [4][CELLS*4] labels (as float)
  float* pr = sm;
  float* lb = sm + 4 * CELLS * 4 * C;
  const int b = blockIdx.z, py = blockIdx.y, px0 = blockIdx.x * CELLS;
  const int y0 = py * 4 - 2, x0 = px0 * 4 - 2;
  const int NPX = CELLS * 4;
  // stage labels (as float; -1 = outside / invalid)
  for (int i = threadIdx.x; i < 4 * NPX; i += NT) {
    const int r = i / NPX, xx = i - r * NPX;
    const int y = y0 + r, x = x0 + xx;
    float l = -1.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const long lab = labels[((long)b * H + y) * W + x];
      l = (lab >= 0 && lab < C) ? (float)lab : -1.f;
    }
    lb[i] = l;
  }
  __syncthreads();
  // four elements per trip, their loads issued before the first exponential (pixels outside the image re-read the
  // image's first logit and store 0): 38 single-load trips per thread at three waves per SIMD ran 65 us for 80 MB
  const int total = 4 * NPX * C;
  for (int i0 = threadIdx.x; i0 < total; i0 += 4 * NT) {
    float lg[4];
    bool inside[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT < total ? i0 + u * NT : i0;
      const int c = i % C, pxl = i / C;
      const int r = pxl / NPX, xx = pxl - r * NPX;
      const int y = y0 + r, x = x0 + xx;
      inside[u] = y >= 0 && y < H && x >= 0 && x < W;
      lg[u] = logits[inside[u] ? (((long)b * H + y) * W + x) * ld + c : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * NT;
      if (i >= total) break;
      const float m = lb[i / C] >= 0.f ? 1.f : 0.f;
      pr[i] = inside[u] ? m / (1.f + expf(-lg[u])) + kClipMin : 0.f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CELLS * C; i += NT) {
    const int c = i / CELLS, cell = i - c * CELLS;
    const int px = px0 + cell;
    if (px >= Wp) continue;
    float sp = 0.f, sl = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pxl = r * NPX + cell * 4 + q;
        sp += pr[pxl * C + c];
        sl += (lb[pxl] == (float)c) ? 1.f : 0.f;
      }
    const long o = (((long)b * C + c) * Hp + py) * Wp + px;
    ppr[o] = sp * (1.f / 16.f);
    pla[o] = sl * (1.f / 16.f);
  }
}

// Gram entries, per (b,c): 45 la-la (upper), 45 pr-pr (upper), 81 la-pr,
// 9 sum la, 9 sum pr = 189 doubles.
constexpr int NG = 189;
constexpr int GROWS = 4;  // position rows per block
__device__ __forceinline__ void tri_index(int e, int* i, int* j) {
  int r = 0, rem = e;
  while (rem >= 9 - r) { rem -= 9 - r; ++r; }
  *i = r; *j = r + rem;
}
__host__ __device__ inline int gram_stride(int Wp) { return Wp + ((35 - Wp % 32) % 32); }

__global__ __launch_bounds__(NT) void rmi_gram_kernel(const float* __restrict__ ppr,
                                                      const float* __restrict__ pla, int Hp, int Wp,
                                                      double* __restrict__ gram) {
  SSA_DYN_LDS(double, smd);  // la tile [GROWS+2][S], pr tile [GROWS+2][S], converted to fp64 ONCE while staged
  // row stride S = 3 (mod 32): at every step the 189 threads read the 9 neighbourhood offsets {0,1,2,S,S+1,S+2,2S,..}
  // of the two tiles -- with S = Wp = 257 (1 mod 32) three of them share a bank.  The tiles hold doubles: the inner
  // loop is then one ds_read_b64 pair + one fp64 FMA per product instead of two float reads, two conversions and the
  // FMA (fp64 runs at half rate; the kernel is 233 M products per call).  A float converts exactly, a product of two
  // floats is exact in fp64: the sums are bit-identical to the float-tile version.
  const int S = gram_stride(Wp);
  const int bc = blockIdx.y;
  const int Hn = Hp - 2, Wn = Wp - 2;
  const int yb = blockIdx.x * GROWS;
  const int nrows = min(GROWS, Hn - yb);
  double* la = smd;
  double* pr = smd + (GROWS + 2) * S;
  const float* gla = pla + (long)bc * Hp * Wp + (long)yb * Wp;
  const float* gpr = ppr + (long)bc * Hp * Wp + (long)yb * Wp;
  for (int i = threadIdx.x; i < (nrows + 2) * Wp; i += NT) {
    const int r = i / Wp, c = i - r * Wp;
    la[r * S + c] = (double)gla[i];
    pr[r * S + c] = (double)gpr[i];
  }
  __syncthreads();
  const int e = threadIdx.x;
  if (e >= NG) return;
  int ti, tj, kind;  // kind 0: a*b, 1: a only
  const double *ta, *tb;
  if (e < 45) { tri_index(e, &ti, &tj); ta = la; tb = la; kind = 0; }
  else if (e < 90) { tri_index(e - 45, &ti, &tj); ta = pr; tb = pr; kind = 0; }
  else if (e < 171) { ti = (e - 90) / 9; tj = (e - 90) % 9; ta = la; tb = pr; kind = 0; }
  else if (e < 180) { ti = e - 171; tj = 0; ta = la; tb = la; kind = 1; }
  else { ti = e - 180; tj = 0; ta = pr; tb = pr; kind = 1; }
  const int oa = (ti / 3) * S + (ti % 3), ob = (tj / 3) * S + (tj % 3);
  double acc = 0.0;
  for (int y = 0; y < nrows; ++y) {
    const double* ra = ta + y * S + oa;
    const double* rb = tb + y * S + ob;
    if (kind == 0) { for (int x = 0; x < Wn; ++x) acc += ra[x] * rb[x]; }
    else { for (int x = 0; x < Wn; ++x) acc += ra[x]; }
  }
  atomicAdd(&gram[(long)bc * NG + e], acc);
}

// 9x9 helpers (row-major double[81]), executed by a single thread
__device__ void mat_mul9(const double* A, const double* B, double* C, bool tA, bool tB) {
  for (int i = 0; i < 9; ++i)
    for (int j = 0; j < 9; ++j) {
      double s = 0.0;
      for (int k = 0; k < 9; ++k) s += (tA ? A[k * 9 + i] : A[i * 9 + k]) * (tB ? B[j * 9 + k] : B[k * 9 + j]);
      C[i * 9 + j] = s;
    }
}
// in-place Gauss-Jordan inverse with partial pivoting; returns false if singular
__device__ bool mat_inv9(const double* A, double* X, double* W /*81 scratch*/) {
  for (int i = 0; i < 81; ++i) { W[i] = A[i]; X[i] = 0.0; }
  for (int i = 0; i < 9; ++i) X[i * 9 + i] = 1.0;
  for (int col = 0; col < 9; ++col) {
    int piv = col;
    double best = fabs(W[col * 9 + col]);
    for (int r = col + 1; r < 9; ++r) { const double v = fabs(W[r * 9 + col]); if (v > best) { best = v; piv = r; } }
    if (best == 0.0) return false;
    if (piv != col)
      for (int k = 0; k < 9; ++k) {
        double t = W[col * 9 + k]; W[col * 9 + k] = W[piv * 9 + k]; W[piv * 9 + k] = t;
        t = X[col * 9 + k]; X[col * 9 + k] = X[piv * 9 + k]; X[piv * 9 + k] = t;
      }
    const double inv = 1.0 / W[col * 9 + col];
    for (int k = 0; k < 9; ++k) { W[col * 9 + k] *= inv; X[col * 9 + k] *= inv; }
    for (int r = 0; r < 9; ++r) {
      if (r == col) continue;
      const double f = W[r * 9 + col];
      if (f == 0.0) continue;
      for (int k = 0; k < 9; ++k) { W[r * 9 + k] -= f * W[col * 9 + k]; X[r * 9 + k] -= f * X[col * 9 + k]; }
    }
  }
  return true;
}
// lower Cholesky; returns false if not positive definite
__device__ bool chol9(const double* A, double* L) {
  for (int i = 0; i < 81; ++i) L[i] = 0.0;
  for (int j = 0; j < 9; ++j) {
    double d = A[j * 9 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 9 + k] * L[j * 9 + k];
    if (!(d > 0.0)) return false;
    const double ljj = sqrt(d);
    L[j * 9 + j] = ljj;
    for (int i = j + 1; i < 9; ++i) {
      double s = A[i * 9 + j];
      for (int k = 0; k < j; ++k) s -= L[i * 9 + k] * L[j * 9 + k];
      L[i * 9 + j] = s / ljj;
    }
  }
  return true;
}
// inverse of lower-triangular L
__device__ void tri_inv9(const double* L, double* Li) {
  for (int i = 0; i < 81; ++i) Li[i] = 0.0;
  for (int j = 0; j < 9; ++j) {
    Li[j * 9 + j] = 1.0 / L[j * 9 + j];
    for (int i = j + 1; i < 9; ++i) {
      double s = 0.0;
      for (int k = j; k < i; ++k) s += L[i * 9 + k] * Li[k * 9 + j];
      Li[i * 9 + j] = -s / L[i * 9 + i];
    }
  }
}

// The same helpers for a workgroup of >= 81 threads, thread t owning element (t / 9, t % 9): every element is
// computed by the same operations in the same order as in the single-thread versions above (bit-identical results);
// the serial kernel spent 180-290 us of the step's critical path in dependent LDS round trips of ONE lane.
__device__ __forceinline__ void pmat_mul9(const double* A, const double* B, double* C, bool tA, bool tB, int t) {
  if (t < 81) {
    const int i = t / 9, j = t - 9 * i;
    double s = 0.0;
    for (int k = 0; k < 9; ++k) s += (tA ? A[k * 9 + i] : A[i * 9 + k]) * (tB ? B[j * 9 + k] : B[k * 9 + j]);
    C[t] = s;
  }
  __syncthreads();
}
__device__ bool pmat_inv9(const double* A, double* X, double* W, int* flag, int t) {
  const int r = t / 9, k = t - 9 * r;
  if (t < 81) { W[t] = A[t]; X[t] = (r == k) ? 1.0 : 0.0; }
  __syncthreads();
  for (int col = 0; col < 9; ++col) {
    if (t == 0) {
      int piv = col;
      double best = fabs(W[col * 9 + col]);
      for (int q = col + 1; q < 9; ++q) { const double v = fabs(W[q * 9 + col]); if (v > best) { best = v; piv = q; } }
      flag[0] = best == 0.0 ? -1 : piv;
    }
    __syncthreads();
    const int piv = flag[0];
    if (piv < 0) return false;
    if (piv != col && t < 9) {
      double u = W[col * 9 + t]; W[col * 9 + t] = W[piv * 9 + t]; W[piv * 9 + t] = u;
      u = X[col * 9 + t]; X[col * 9 + t] = X[piv * 9 + t]; X[piv * 9 + t] = u;
    }
    __syncthreads();
    const double inv = 1.0 / W[col * 9 + col];
    __syncthreads();
    if (t < 9) { W[col * 9 + t] *= inv; X[col * 9 + t] *= inv; }
    __syncthreads();
    const double f = t < 81 ? W[r * 9 + col] : 0.0;
    __syncthreads();
    if (t < 81 && r != col && f != 0.0) { W[t] -= f * W[col * 9 + k]; X[t] -= f * X[col * 9 + k]; }
    __syncthreads();
  }
  return true;
}

__global__ __launch_bounds__(128) void rmi_solve_kernel(const double* __restrict__ gram, int Hp, int Wp,
                                                        double* __restrict__ loss_bc, double* __restrict__ gmat) {
  __shared__ double S[11 * 81];
  __shared__ int flag[2];
  const int t = threadIdx.x;
  const int bc = blockIdx.x;
  const double* g = gram + (long)bc * NG;
  const double n = (double)(Hp - 2) * (double)(Wp - 2);
  double* Cll = S; double* Cpp = S + 81; double* Clp = S + 162; double* X = S + 243;
  double* T1 = S + 324; double* T2 = S + 405; double* M = S + 486; double* Lc = S + 567;
  double* Li = S + 648; double* GM = S + 729; double* T3 = S + 810;
  const double* sla = g + 171; const double* spr = g + 180;
  double* out = gmat + (long)bc * (2 * 81 + 18);
  if (t < 81) {
    const int i = t / 9, j = t - 9 * i;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int e = lo * 9 - lo * (lo - 1) / 2 + (hi - lo);          // index of (lo, hi) in the packed upper triangle
    Cll[t] = g[e] - sla[lo] * sla[hi] / n;
    Cpp[t] = g[45 + e] - spr[lo] * spr[hi] / n;
    Clp[t] = g[90 + t] - sla[i] * spr[j] / n;
  }
  if (t < 9) { out[162 + t] = sla[t] / n; out[171 + t] = spr[t] / n; }
  __syncthreads();
  // X = (Cpp + alpha I)^-1
  if (t < 81) T1[t] = Cpp[t] + ((t / 9 == t % 9) ? kPosAlpha : 0.0);
  __syncthreads();
  bool ok = pmat_inv9(T1, X, T2, flag, t);
  __syncthreads();
  // A = Cll - Clp X Clp^T ; M = A + alpha I
  pmat_mul9(Clp, X, T1, false, false, t);        // T1 = Clp X
  pmat_mul9(T1, Clp, T2, false, true, t);        // T2 = Clp X Clp^T
  if (t < 81) M[t] = (Cll[t] - T2[t]) + ((t / 9 == t % 9) ? kPosAlpha : 0.0);
  __syncthreads();
  if (t == 0) {
    const bool pd = ok && chol9(M, Lc);
    flag[1] = pd ? 1 : 0;
    if (pd) {
      double loss = 0.0;
      for (int i = 0; i < 9; ++i) loss += log(Lc[i * 9 + i] + 1e-8);
      loss_bc[bc] = loss;  // = 0.5 * 2 * sum log(diag + 1e-8)
      tri_inv9(Lc, Li);
    } else {
      loss_bc[bc] = __longlong_as_double(0x7ff8000000000000LL);
    }
  }
  __syncthreads();
  if (!flag[1]) {
    for (int i = t; i < 162; i += blockDim.x) out[i] = 0.0;
    return;
  }
  // G_M = 0.5 * Li^T diag(w) Li, w_i = L_ii / (L_ii + 1e-8)
  if (t < 81) {
    const int i = t / 9, j = t - 9 * i;
    double s = 0.0;
    for (int k = 0; k < 9; ++k) {
      const double w = Lc[k * 9 + k] / (Lc[k * 9 + k] + 1e-8);
      s += Li[k * 9 + i] * w * Li[k * 9 + j];
    }
    GM[t] = 0.5 * s;
  }
  __syncthreads();
  // G_lp = -2 GM Clp X = -2 GM T1
  pmat_mul9(GM, T1, T2, false, false, t);
  if (t < 81) out[t] = -2.0 * T2[t];
  // G_pp = X Clp^T GM Clp X = T1^T GM T1     (T2 = GM T1 already)
  pmat_mul9(T1, T2, T3, true, false, t);
  if (t < 81) out[81 + t] = T3[t];
}

__global__ void rmi_finalize_kernel(const double* __restrict__ loss_bc, int B, int C,
                                    float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int c = 0; c < C; ++c) {
    double m = 0.0;
    for (int b = 0; b < B; ++b) m += loss_bc[b * C + c];
    s += m / (double)B / 9.0;
  }
  out[0] = (float)s;
}

// dP[bc,Y,X] = sum_j dpr_j(Y-dy_j, X-dx_j),
// dpr_j(q) = sum_i Glp[i][j] (la_i(q)-mla_i) + 2 sum_i Gpp[j][i] (pr_i(q)-mpr_i)
__global__ __launch_bounds__(NT) void rmi_bwd_pooled_kernel(const float* __restrict__ ppr,
                                                            const float* __restrict__ pla,
                                                            const double* __restrict__ gmat, int Hp,
                                                            int Wp, float* __restrict__ dpooled) {
  __shared__ double G[2 * 81 + 18];
  const int bc = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * 81 + 18; i += NT) G[i] = gmat[(long)bc * (2 * 81 + 18) + i];
  __syncthreads();
  const int Hn = Hp - 2, Wn = Wp - 2;
  const long cell = (long)blockIdx.x * NT + threadIdx.x;
  if (cell >= (long)Hp * Wp) return;
  const int Y = (int)(cell / Wp), X = (int)(cell - (long)Y * Wp);
  const float* L = pla + (long)bc * Hp * Wp;
  const float* P = ppr + (long)bc * Hp * Wp;
  double acc = 0.0;
  for (int j = 0; j < 9; ++j) {
    const int y = Y - j / 3, x = X - j % 3;
    if (y < 0 || y >= Hn || x < 0 || x >= Wn) continue;
    double s = 0.0;
    for (int i = 0; i < 9; ++i) {
      const long o = (long)(y + i / 3) * Wp + (x + i % 3);
      s += G[i * 9 + j] * ((double)L[o] - G[162 + i]);
      s += 2.0 * G[81 + j * 9 + i] * ((double)P[o] - G[171 + i]);
    }
    acc += s;
  }
  dpooled[(long)bc * Hp * Wp + cell] = (float)acc;
}

__global__ void rmi_bwd_logits_kernel(const float* __restrict__ logits, int ld,
                                      const int64_t* __restrict__ labels, int B, int H, int W, int C,
                                      const float* __restrict__ dpooled, int Hp, int Wp,
                                      const float* __restrict__ upstream, double coef,
                                      float* __restrict__ dlogits, int accumulate,
                                      const float* __restrict__ bce_src, double bce_coef,
                                      const double* __restrict__ bce_acc, double bce_denom_add) {
  const long n = (long)B * H * W * C;
  const float k = (float)((double)upstream[0] * coef / 16.0);
  // bce_src: the un-normalised BCE gradient of the same logits (ssa_bce_fwd) -- scaled here as ssa_scale_grad_to would,
  // instead of a pass of its own over the 80 MB that this kernel then reads back
  // bce_acc without bce_src: the BCE half is recomputed here from the logit this thread holds anyway
  const float kb = bce_acc ? (float)((double)upstream[0] * bce_coef / (bce_acc[1] + bce_denom_add)) : 0.f;
  // every load of an element is issued whatever its label says (none of the addresses depends on it) and four
  // elements go per trip: with the label -> branch -> logit -> pooled-gradient chain of dependent loads, one element
  // at a time, the 160 MB of a 1024 x 1024 x 19 call took 103 us
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
    float lg[4], dp[4], old[4];
    long lab[4];
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = i0 + u * stride < n ? i0 + u * stride : i0;
      const int c = (int)(i % C);
      cc[u] = c;
      const long p = i / C;
      const int x = (int)(p % W);
      const long t = p / W;
      const int y = (int)(t % H), b = (int)(t / H);
      const int py = (y + 2) >> 2, px = (x + 2) >> 2;
      lab[u] = labels[p];
      lg[u] = logits[p * ld + c];
      dp[u] = dpooled[(((long)b * C + c) * Hp + py) * Wp + px];
      old[u] = bce_src ? bce_src[i] * kb : (accumulate ? dlogits[i] : 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = i0 + u * stride;
      if (i >= n) break;
      float g = 0.f;
      if (lab[u] >= 0 && lab[u] < C) {
        const float s = 1.f / (1.f + expf(-lg[u]));
        g = k * dp[u] * s * (1.f - s);
        if (bce_acc && !bce_src) {          // (sigmoid - onehot) * kb, with the BCE forward's arithmetic
          const float e = __expf(-fabsf(lg[u]));
          const float r = __frcp_rn(1.f + e);
          g += ((lg[u] >= 0.f ? r : e * r) - (cc[u] == (int)lab[u] ? 1.f : 0.f)) * kb;
        }
      }
      dlogits[i] = old[u] + g;
    }
  }
}

inline int grid_for(long n, int cap = 8192) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

int ssa_ce_fwd(const float* logits, int ld, const int64_t* labels, long P, int C, int ignore_index,
               double* acc, float* dlogits, void* stream) {
  if (!logits || !labels || !acc || P <= 0 || C <= 0 || C > 128) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  {   // the [NT][C] fp32 tile exceeds the default 64 KB of dynamic LDS from 65 classes (Mapillary) on
    static ssa::LdsLimit lds_limit;         // per device (group.h)
    if (int rc = ssa::raise_lds_limit((const void*)pixel_loss_kernel<0>, (size_t)NT * C * sizeof(float), 64 * 1024, &lds_limit))
      return rc;
  }
  hipLaunchKernelGGL(pixel_loss_kernel<0>, dim3(grid_for(P, 2048)), dim3(NT), NT * C * sizeof(float),
                     s, logits, ld, labels, P, C, ignore_index, acc, dlogits);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_bce_fwd(const float* logits, int ld, const int64_t* labels, long P, int C, double* acc,
                float* dlogits, void* stream) {
  if (!logits || !labels || !acc || P <= 0 || C <= 0 || C > 128) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(double), s);
  if (e != hipSuccess) return (int)e;
  const long n = P * C;
  if (ld == C && n % 4 == 0 && n < (1L << 31) && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15u) == 0) {
    // 512 blocks = 2 per CU: every block ends in two fp64 atomics on the same two addresses, which serialise
    hipLaunchKernelGGL(bce_stream_kernel, dim3(grid_for(n / 4, 512)), dim3(NT), 0, s, logits, labels, (unsigned)(n / 4), C,
                       acc, dlogits);
    SSA_LAUNCH_CHECK();
    return SSA_OK;
  }
  {   // the [NT][C] fp32 tile exceeds the default 64 KB of dynamic LDS from 65 classes (Mapillary) on
    static ssa::LdsLimit lds_limit;         // per device (group.h)
    if (int rc = ssa::raise_lds_limit((const void*)pixel_loss_kernel<1>, (size_t)NT * C * sizeof(float), 64 * 1024, &lds_limit))
      return rc;
  }
  hipLaunchKernelGGL(pixel_loss_kernel<1>, dim3(grid_for(P, 2048)), dim3(NT), NT * C * sizeof(float),
                     s, logits, ld, labels, P, C, 0, acc, dlogits);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_loss_finalize(const double* acc, double denom_add, float* loss, void* stream) {
  if (!acc || !loss) return SSA_EINVAL;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, denom_add, loss);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_scale_grad_to(const float* src, float* dst, long n, const float* upstream, double coef, const double* acc,
                      double denom_add, void* stream) {
  if (!src || !dst || !upstream || !acc || n <= 0) return SSA_EINVAL;
  hipLaunchKernelGGL(scale_grad_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, dst, n,
                     upstream, coef, acc, denom_add);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_bce_bwd(const float* logits, int ld, const int64_t* labels, long P, int C, const float* upstream, double coef,
                const double* acc, double denom_add, float* dlogits, void* stream) {
  if (!logits || !labels || !upstream || !acc || !dlogits || P <= 0 || C < 4 || C > 128) return SSA_EINVAL;
  const long n = P * C;
  if (ld != C || n % 4 || n >= (1L << 31) ||
      ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15u) != 0)
    return SSA_EUNSUPPORTED;                     // (the caller keeps the saved gradient of ssa_bce_fwd for those)
  hipLaunchKernelGGL(bce_bwd_stream_kernel, dim3(grid_for(n / 4, 2048)), dim3(NT), 0, (hipStream_t)stream, logits, labels,
                     (unsigned)(n / 4), C, upstream, coef, acc, denom_add, dlogits);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_scale_grad(float* g, long n, const float* upstream, double coef, const double* acc,
                   double denom_add, void* stream) {
  return ssa_scale_grad_to(g, g, n, upstream, coef, acc, denom_add, stream);
}

int ssa_rmi_pool(const float* logits, int ld, const int64_t* labels, int B, int H, int W, int C,
                 float* pooled_pr, float* pooled_la, int Hp, int Wp, void* stream) {
  if (!logits || !labels || !pooled_pr || !pooled_la) return SSA_EINVAL;
  if (Hp != H / 4 + 1 || Wp != W / 4 + 1) return SSA_EINVAL;
  const size_t lds = (size_t)(4 * CELLS * 4 * C + 4 * CELLS * 4) * sizeof(float);
  if (lds > 64000) return SSA_EUNSUPPORTED;
  hipLaunchKernelGGL(rmi_pool_kernel, dim3((Wp + CELLS - 1) / CELLS, Hp, B), dim3(NT), lds,
                     (hipStream_t)stream, logits, ld, labels, H, W, C, pooled_pr, pooled_la, Hp, Wp);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rmi_gram(const float* pooled_pr, const float* pooled_la, int BC, int Hp, int Wp,
                 double* gram, void* stream) {
  if (!pooled_pr || !pooled_la || !gram || Hp < 3 || Wp < 3) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(gram, 0, sizeof(double) * NG * BC, s);
  if (e != hipSuccess) return (int)e;
  const size_t lds = (size_t)2 * (GROWS + 2) * gram_stride(Wp) * sizeof(double);
  // wide crops (pooled width beyond ~620: crops wider than ~2,480 pixels at pool stride 4) need more than the default
  // 64 KB of dynamic LDS for the fp64 tiles: raise the kernel's limit, up to the CU's 160 KB (pooled width ~1,700)
  static ssa::LdsLimit lds_limit;           // per device (group.h)
  if (lds > 160 * 1024) return SSA_EUNSUPPORTED;
  if (int rc = ssa::raise_lds_limit((const void*)rmi_gram_kernel, lds, 60000, &lds_limit)) return rc;
  hipLaunchKernelGGL(rmi_gram_kernel, dim3((Hp - 2 + GROWS - 1) / GROWS, BC), dim3(NT), lds, s,
                     pooled_pr, pooled_la, Hp, Wp, gram);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rmi_solve(const double* gram, int BC, int Hp, int Wp, double* loss_bc, double* gmat,
                  void* stream) {
  if (!gram || !loss_bc || !gmat) return SSA_EINVAL;
  hipLaunchKernelGGL(rmi_solve_kernel, dim3(BC), dim3(128), 0, (hipStream_t)stream, gram, Hp, Wp,
                     loss_bc, gmat);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rmi_finalize(const double* loss_bc, int B, int C, float* out, void* stream) {
  if (!loss_bc || !out) return SSA_EINVAL;
  hipLaunchKernelGGL(rmi_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_bc, B, C, out);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rmi_bwd_pooled(const float* pooled_pr, const float* pooled_la, const double* gmat, int BC,
                       int Hp, int Wp, float* dpooled, void* stream) {
  if (!pooled_pr || !pooled_la || !gmat || !dpooled) return SSA_EINVAL;
  hipLaunchKernelGGL(rmi_bwd_pooled_kernel, dim3(((long)Hp * Wp + NT - 1) / NT, BC), dim3(NT), 0,
                     (hipStream_t)stream, pooled_pr, pooled_la, gmat, Hp, Wp, dpooled);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rmi_bwd_logits(const float* logits, int ld, const int64_t* labels, int B, int H, int W,
                       int C, const float* dpooled, int Hp, int Wp, const float* upstream,
                       double coef, float* dlogits, int accumulate, void* stream) {
  if (!logits || !labels || !dpooled || !upstream || !dlogits) return SSA_EINVAL;
  hipLaunchKernelGGL(rmi_bwd_logits_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0,
                     (hipStream_t)stream, logits, ld, labels, B, H, W, C, dpooled, Hp, Wp, upstream,
                     coef, dlogits, accumulate, (const float*)nullptr, 0.0, (const double*)nullptr, 0.0);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_rmi_bwd_logits_bce(const float* logits, int ld, const int64_t* labels, int B, int H, int W,
                           int C, const float* dpooled, int Hp, int Wp, const float* upstream,
                           double coef, const float* bce_grad, double bce_coef, const double* bce_acc,
                           double bce_denom_add, float* dlogits, void* stream) {
  if (!logits || !labels || !dpooled || !upstream || !dlogits || !bce_acc) return SSA_EINVAL;   // bce_grad NULL: recomputed
  hipLaunchKernelGGL(rmi_bwd_logits_kernel, dim3(grid_for((long)B * H * W * C)), dim3(256), 0,
                     (hipStream_t)stream, logits, ld, labels, B, H, W, C, dpooled, Hp, Wp, upstream,
                     coef, dlogits, 0, bce_grad, bce_coef, bce_acc, bce_denom_add);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Device-side evaluation tail: per-pixel argmax over
// the class logits + confusion matrix, replacing the host path of
// utils/trnval_utils.py:173-196 (softmax -> .cpu() -> max(1)) and utils/misc.py:50-67
// (fast_hist: np.bincount(C*gt[mask] + pred[mask])).  Integer output, bit-exact given
// the predictions; argmax takes the FIRST maximum (torch.max's rule; softmax is
// monotonic, so argmax of the logits is argmax of the probabilities except where two
// probabilities round to the same float).  One LDS histogram per workgroup, then int64
// atomics: the 159 MB [B,19,H,W] fp32 copy to the host disappears.
namespace {
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ logits, int ld,
                                                        const long* __restrict__ labels, long P, int C,
                                                        unsigned char* __restrict__ pred_out,
                                                        unsigned long long* __restrict__ hist) {
  SSA_DYN_LDS(unsigned int, lh);          // [C*C]
  for (int i = threadIdx.x; i < C * C; i += 256) lh[i] = 0u;
  __syncthreads();
  for (long p = blockIdx.x * 256L + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    const float* row = logits + p * ld;
    float best = row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = row[c];
      if (v > best || (v != v && best == best)) { best = v; arg = c; }   // first maximum; NaN wins once
    }
    if (pred_out) pred_out[p] = (unsigned char)arg;
    const long g = labels[p];
    if (g >= 0 && g < C) atomicAdd(&lh[(int)g * C + arg], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}
}  // namespace

extern "C" int ssa_confusion_matrix(const float* logits, int ld, const int64_t* labels, long P, int C,
                                    unsigned char* pred_out, int64_t* hist, void* stream) {
  if (!logits || !labels || !hist || P < 1 || C < 1 || C > 255 || ld < C) return SSA_EINVAL;
  const int blocks = (int)((P + 255) / 256 > 2048 ? 2048 : (P + 255) / 256);
  hipLaunchKernelGGL(confusion_kernel, dim3(blocks), dim3(256), (size_t)C * C * sizeof(unsigned int),
                     (hipStream_t)stream, logits, ld, (const long*)labels, P, C, pred_out,
                     (unsigned long long*)hist);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
