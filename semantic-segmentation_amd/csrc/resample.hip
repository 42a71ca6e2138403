// Bilinear resampling, align_corners=False, on NHWC tensors.
// Same index arithmetic as ATen's upsample_bilinear2d (fp32 source index,
// negative source clamped to 0, i1 = i0 + (i0 < in-1)), which is what
// F.interpolate does behind network/mynn.py:42-114 and
// network/hrnetv2.py:246-249,440-445.
// The backward is a gather over the output pixels that reference an input
// pixel -- deterministic, no atomics.
#include <type_traits>
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"

namespace {

struct Src { int i0, i1; float l0, l1; };

__device__ __forceinline__ Src src_index(int o, float scale, int in_size) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  Src r;
  r.i0 = (int)s;
  if (r.i0 > in_size - 1) r.i0 = in_size - 1;
  r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
  r.l1 = s - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// ---- forward, 8 bf16 channels per thread
__device__ __forceinline__ void bilinear_fwd_v8_body(const bf16_t* __restrict__ x, int B, int Hi, int Wi, int C, int ldx,
                                bf16_t* __restrict__ y, int Ho, int Wo, int ldy, float sh, float sw, const int bx, const int gx) {
  const int VC = C >> 3;
  const long n = (long)B * Ho * Wo * VC;
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < n; i += (long)gx * blockDim.x) {
    const int cg = (int)(i % VC);
    long t = i / VC;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const Src ys = src_index(oy, sh, Hi), xs = src_index(ox, sw, Wi);
    const bf16_t* base = x + (long)b * Hi * Wi * ldx + cg * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ys.i0 * Wi + xs.i0) * ldx), v00);
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ys.i0 * Wi + xs.i1) * ldx), v01);
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ys.i1 * Wi + xs.i0) * ldx), v10);
    unpack8(*reinterpret_cast<const uint4*>(base + ((long)ys.i1 * Wi + xs.i1) * ldx), v11);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = ys.l0 * (xs.l0 * v00[j] + xs.l1 * v01[j]) + ys.l1 * (xs.l0 * v10[j] + xs.l1 * v11[j]);
    *reinterpret_cast<uint4*>(y + ((long)(b * Ho + oy) * Wo + ox) * ldy + cg * 8) = pack8(o);
  }
}

// ---- forward, one element per thread (any C, any dtype pair)
template <typename InT, typename OutT>
__device__ __forceinline__ void bilinear_fwd_s_body(const InT* __restrict__ x, int B, int Hi, int Wi, int C, int ldx,
                               OutT* __restrict__ y, int Ho, int Wo, int ldy, float sh, float sw, const int bx, const int gx) {
  const long n = (long)B * Ho * Wo * C;
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < n; i += (long)gx * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const Src ys = src_index(oy, sh, Hi), xs = src_index(ox, sw, Wi);
    const InT* base = x + (long)b * Hi * Wi * ldx + c;
    const float v00 = ld_as_f32(base + ((long)ys.i0 * Wi + xs.i0) * ldx);
    const float v01 = ld_as_f32(base + ((long)ys.i0 * Wi + xs.i1) * ldx);
    const float v10 = ld_as_f32(base + ((long)ys.i1 * Wi + xs.i0) * ldx);
    const float v11 = ld_as_f32(base + ((long)ys.i1 * Wi + xs.i1) * ldx);
    const float o = ys.l0 * (xs.l0 * v00 + xs.l1 * v01) + ys.l1 * (xs.l0 * v10 + xs.l1 * v11);
    st_from_f32(y + ((long)(b * Ho + oy) * Wo + ox) * ldy + c, o);
  }
}

constexpr int kMaxCand = 12;      // candidate columns whose weights the gather kernels keep in registers

// candidate output range that can reference input index i
__device__ __forceinline__ void cand_range(int i, float scale, int out_size, int* lo, int* hi) {
  const float inv = 1.f / scale;
  int a = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
  int b = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
  *lo = a < 0 ? 0 : a;
  *hi = b > out_size - 1 ? out_size - 1 : b;
}
__device__ __forceinline__ float weight_for(int o, float scale, int in_size, int i) {
  const Src s = src_index(o, scale, in_size);
  float w = 0.f;
  if (s.i0 == i) w += s.l0;
  if (s.i1 == i) w += s.l1;
  return w;
}

// the candidates of an input index that really carry weight
__device__ __forceinline__ void tight_range(int i, float scale, int out_size, int in_size, int* lo, int* hi) {
  int a, b;
  cand_range(i, scale, out_size, &a, &b);
  while (a < b && weight_for(a, scale, in_size, i) == 0.f) ++a;
  while (b > a && weight_for(b, scale, in_size, i) == 0.f) --b;
  *lo = a; *hi = b;
}

// ---- backward gather, 8 bf16 channels per thread
__device__ __forceinline__ void bilinear_bwd_v8_body(const bf16_t* __restrict__ dy, int B, int Ho, int Wo, int C,
                                int lddy, bf16_t* __restrict__ dx, int Hi, int Wi, int lddx, float sh,
                                float sw, const int bx, const int gx) {
  const int VC = C >> 3;
  const long n = (long)B * Hi * Wi * VC;
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < n; i += (long)gx * blockDim.x) {
    const int cg = (int)(i % VC);
    long t = i / VC;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    int ylo, yhi, xlo, xhi;
    cand_range(iy, sh, Ho, &ylo, &yhi);
    cand_range(ix, sw, Wo, &xlo, &xhi);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const bf16_t* base = dy + (long)b * Ho * Wo * lddy + cg * 8;
    // column weights once per element, not once per (row, column): up to kMaxCand candidates stay in
    // registers (x4 upsampling has 11), wider ranges recompute
    float wxs[kMaxCand];
    const int nx = xhi - xlo + 1;
    const bool hoisted = nx <= kMaxCand;
#pragma unroll
    for (int k = 0; k < kMaxCand; ++k) wxs[k] = (hoisted && k < nx) ? weight_for(xlo + k, sw, Wi, ix) : 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = weight_for(oy, sh, Hi, iy);
      if (wy == 0.f) continue;
      const bf16_t* row = base + (long)oy * Wo * lddy;
      if (hoisted) {
#pragma unroll
        for (int k = 0; k < kMaxCand; ++k) {
          if (wxs[k] == 0.f) continue;
          float g[8];
          unpack8(*reinterpret_cast<const uint4*>(row + (long)(xlo + k) * lddy), g);
          const float w = wy * wxs[k];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
        }
      } else {
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = weight_for(ox, sw, Wi, ix);
          if (wx == 0.f) continue;
          float g[8];
          unpack8(*reinterpret_cast<const uint4*>(row + (long)ox * lddy), g);
          const float w = wy * wx;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += w * g[j];
        }
      }
    }
    *reinterpret_cast<uint4*>(dx + ((long)(b * Hi + iy) * Wi + ix) * lddx + cg * 8) = pack8(acc);
  }
}

template <typename InT, typename OutT>
__device__ __forceinline__ void bilinear_bwd_s_body(const InT* __restrict__ dy, int B, int Ho, int Wo, int C, int lddy,
                               OutT* __restrict__ dx, int Hi, int Wi, int lddx, float sh, float sw, const int bx, const int gx) {
  const long n = (long)B * Hi * Wi * C;
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < n; i += (long)gx * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    int ylo, yhi, xlo, xhi;
    cand_range(iy, sh, Ho, &ylo, &yhi);
    cand_range(ix, sw, Wo, &xlo, &xhi);
    float acc = 0.f;
    const InT* base = dy + (long)b * Ho * Wo * lddy + c;
    float wxs[kMaxCand];
    const int nx = xhi - xlo + 1;
    const bool hoisted = nx <= kMaxCand;
#pragma unroll
    for (int k = 0; k < kMaxCand; ++k) wxs[k] = (hoisted && k < nx) ? weight_for(xlo + k, sw, Wi, ix) : 0.f;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = weight_for(oy, sh, Hi, iy);
      if (wy == 0.f) continue;
      const InT* row = base + (long)oy * Wo * lddy;
      if (hoisted) {
#pragma unroll
        for (int k = 0; k < kMaxCand; ++k)
          if (wxs[k] != 0.f) acc += wy * wxs[k] * ld_as_f32(row + (long)(xlo + k) * lddy);
      } else {
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = weight_for(ox, sw, Wi, ix);
          if (wx == 0.f) continue;
          acc += wy * wx * ld_as_f32(row + (long)ox * lddy);
        }
      }
    }
    st_from_f32(dx + ((long)(b * Hi + iy) * Wi + ix) * lddx + c, acc);
  }
}

// ---- backward of an UPSAMPLING resize, separable: the gather above reads (window_y x window_x) output pixels per input
// element -- 8 x 8 at scale 4, 16 x 16 at scale 8, every gradient element ~4 times over and poorly coalesced (0.85-0.9
// TB/s measured on the 19-channel fp32 logits and on the trunk's 8x branch, profiles/r04_notes.md).  Bilinear weights
// factor, w(oy, ox -> iy, ix) = wy(oy -> iy) * wx(ox -> ix), so
//   pass X:  tmp[b, oy, ix, c] = sum_ox wx * dy[b, oy, ox, c]      (fp32, [B, Ho, Wi, C]; dy read ONCE, coalesced)
//   pass Y:  dx[b, iy, ix, c]  = sum_oy wy * tmp[b, oy, ix, c]     (unit-stride reads along (ix, c))
// -- window_x + window_y taps instead of their product.  Two dependent launches: the host issues the X passes of a
// level in one bracket and the Y passes in the next.
constexpr int kXRows = 1;         // output rows per thread of pass X (8, with the column weights computed once for all of them,
                                  // measured slower: 34 -> 46 us per launch, too few threads in flight; profiles/r04_notes.md call I)

template <typename InT, int V>
__device__ __forceinline__ void bilinear_bwd_x_body(const InT* __restrict__ dy, int B, int Ho, int Wo, int C, int lddy,
                                                    float* __restrict__ tmp, int Wi, float sw, const int bx, const int gx) {
  const int VC = C / V;
  const long rows = (long)B * Ho;                       // (b, oy) flattened: dy rows are Wo * lddy apart
  const long rblocks = (rows + kXRows - 1) / kXRows;
  const long n = rblocks * Wi * VC;
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < n; i += (long)gx * blockDim.x) {
    const int cg = (int)(i % VC);
    long t = i / VC;
    const int ix = (int)(t % Wi);
    const long r0 = (t / Wi) * kXRows;
    // the columns with a non-zero weight: 2 f for an integer factor f.  Up to 8 of them go as ONE batch of loads with
    // no branch in between (taps past the window re-read its last column with weight 0) -- with a zero-weight test
    // between the candidates' loads they were issued one at a time (profiles/r06_notes.md, call U: the same change made
    // the few-channel fp32 form 2.3x faster)
    int xlo, xhi;
    tight_range(ix, sw, Wo, Wi, &xlo, &xhi);
    const int nx = xhi - xlo + 1;
    for (int rr = 0; rr < kXRows; ++rr) {
      const long r = r0 + rr;
      if (r >= rows) break;
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = 0.f;
      const InT* row = dy + r * (long)Wo * lddy + cg * V;
      auto batch = [&](auto taps) {
        constexpr int T = decltype(taps)::value;
        float w[T];
        long off[T];
#pragma unroll
        for (int k = 0; k < T; ++k) {
          w[k] = k < nx ? weight_for(xlo + k, sw, Wi, ix) : 0.f;
          off[k] = (long)(xlo + (k < nx ? k : nx - 1)) * lddy;
        }
        if constexpr (V == 8) {
          uint4 raw[T];
#pragma unroll
          for (int k = 0; k < T; ++k) raw[k] = *reinterpret_cast<const uint4*>(row + off[k]);
#pragma unroll
          for (int k = 0; k < T; ++k) {
            float g[8];
            unpack8(raw[k], g);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w[k] * g[j];
          }
        } else {
          float g[T];
#pragma unroll
          for (int k = 0; k < T; ++k) g[k] = ld_as_f32(row + off[k]);
#pragma unroll
          for (int k = 0; k < T; ++k) acc[0] += w[k] * g[k];
        }
      };
      if (nx <= 4) {
        batch(std::integral_constant<int, 4>());
      } else if (nx <= 8) {
        batch(std::integral_constant<int, 8>());
      } else {
        for (int ox = xlo; ox <= xhi; ++ox) {
          const float wx = weight_for(ox, sw, Wi, ix);
          if (wx == 0.f) continue;
          if constexpr (V == 8) {
            float g[8];
            unpack8(*reinterpret_cast<const uint4*>(row + (long)ox * lddy), g);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wx * g[j];
          } else {
            acc[0] += wx * ld_as_f32(row + (long)ox * lddy);
          }
        }
      }
      float* o = tmp + (r * Wi + ix) * (long)C + cg * V;
      if constexpr (V == 8) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
      } else {
        o[0] = acc[0];
      }
    }
  }
}

template <typename OutT, int V>
__device__ __forceinline__ void bilinear_bwd_y_body(const float* __restrict__ tmp, int B, int Ho, int Wi, int C,
                                                    OutT* __restrict__ dx, int Hi, int lddx, float sh, const int bx,
                                                    const int gx) {
  const int VC = C / V;
  const long n = (long)B * Hi * Wi * VC;
  for (long i = bx * (long)blockDim.x + threadIdx.x; i < n; i += (long)gx * blockDim.x) {
    const int cg = (int)(i % VC);
    long t = i / VC;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    int ylo, yhi;
    tight_range(iy, sh, Ho, Hi, &ylo, &yhi);
    const int ny = yhi - ylo + 1;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const float* col = tmp + ((long)b * Ho * Wi + ix) * C + cg * V;
    auto batch = [&](auto taps) {          // as in pass X: the window's rows as one batch of loads
      constexpr int T = decltype(taps)::value;
      float w[T];
      const float* src[T];
#pragma unroll
      for (int k = 0; k < T; ++k) {
        w[k] = k < ny ? weight_for(ylo + k, sh, Hi, iy) : 0.f;
        src[k] = col + (long)(ylo + (k < ny ? k : ny - 1)) * Wi * C;
      }
      if constexpr (V == 8) {
        float4 a[T], c4[T];
#pragma unroll
        for (int k = 0; k < T; ++k) { a[k] = *reinterpret_cast<const float4*>(src[k]); c4[k] = *reinterpret_cast<const float4*>(src[k] + 4); }
#pragma unroll
        for (int k = 0; k < T; ++k) {
          acc[0] += w[k] * a[k].x; acc[1] += w[k] * a[k].y; acc[2] += w[k] * a[k].z; acc[3] += w[k] * a[k].w;
          acc[4] += w[k] * c4[k].x; acc[5] += w[k] * c4[k].y; acc[6] += w[k] * c4[k].z; acc[7] += w[k] * c4[k].w;
        }
      } else {
        float g[T];
#pragma unroll
        for (int k = 0; k < T; ++k) g[k] = src[k][0];
#pragma unroll
        for (int k = 0; k < T; ++k) acc[0] += w[k] * g[k];
      }
    };
    if (ny <= 4) {
      batch(std::integral_constant<int, 4>());
    } else if (ny <= 8) {
      batch(std::integral_constant<int, 8>());
    } else {
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = weight_for(oy, sh, Hi, iy);
        if (wy == 0.f) continue;
        const float* src = col + (long)oy * Wi * C;
        if constexpr (V == 8) {
          const float4 a = *reinterpret_cast<const float4*>(src), c4 = *reinterpret_cast<const float4*>(src + 4);
          acc[0] += wy * a.x; acc[1] += wy * a.y; acc[2] += wy * a.z; acc[3] += wy * a.w;
          acc[4] += wy * c4.x; acc[5] += wy * c4.y; acc[6] += wy * c4.z; acc[7] += wy * c4.w;
        } else {
          acc[0] += wy * src[0];
        }
      }
    }
    OutT* o = dx + ((long)(b * Hi + iy) * Wi + ix) * lddx + cg * V;
    if constexpr (V == 8) {
      *reinterpret_cast<uint4*>(o) = pack8(acc);
    } else {
      st_from_f32(o, acc[0]);
    }
  }
}

// NCHW fp32 image -> (optionally resized) NHWC bf16, channels zero padded.
// ResizeX(x, s) of network/mynn.py:101-114 fused with the layout change.
__global__ void image_resize_kernel(const float* __restrict__ x, int B, int C, int Hi, int Wi,
                                    bf16_t* __restrict__ y, int Ho, int Wo, int cpad, float sh,
                                    float sw) {
  const long n = (long)B * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long t = i;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const Src ys = src_index(oy, sh, Hi), xs = src_index(ox, sw, Wi);
    bf16_t* dst = y + i * cpad;
    for (int c0 = 0; c0 < cpad; c0 += 8) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        float v = 0.f;
        if (c < C) {
          const float* pl = x + ((long)b * C + c) * Hi * Wi;
          const float v00 = pl[(long)ys.i0 * Wi + xs.i0], v01 = pl[(long)ys.i0 * Wi + xs.i1];
          const float v10 = pl[(long)ys.i1 * Wi + xs.i0], v11 = pl[(long)ys.i1 * Wi + xs.i1];
          v = ys.l0 * (xs.l0 * v00 + xs.l1 * v01) + ys.l1 * (xs.l0 * v10 + xs.l1 * v11);
        }
        f[j] = v;
      }
      *reinterpret_cast<uint4*>(dst + c0) = pack8(f);
    }
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}


// Nearest-neighbour resize of uint8 label maps with PIL's index rule.  The
// source row/column of every destination row/column comes from host-computed
// tables (semseg_amd/datasets/transforms.py reproduces Pillow's running double
// sum bit for bit), so the gather itself is exact by construction.
__global__ void resize_nearest_u8_kernel(const unsigned char* __restrict__ src, int Hs, int Ws,
                                         unsigned char* __restrict__ dst, int Hd, int Wd,
                                         const int* __restrict__ iy, const int* __restrict__ ix,
                                         long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wd);
    long t = i / Wd;
    const int y = (int)(t % Hd);
    const long b = t / Hd;
    dst[i] = src[(b * Hs + iy[y]) * Ws + ix[x]];
  }
}


// ---- group-aware wrappers (group.h)
template <typename InT, typename OutT, bool V8, bool BWD>
struct BilinearK {
  // forward: (x [B,Hi,Wi,C] ldx) -> (y [B,Ho,Wo,C] ldy); backward: (dy [B,Ho,Wo,C]) -> (dx [B,Hi,Wi,C])
  struct Args { const InT* src; OutT* dst; int B, Hs, Ws, C, lds, Hd, Wd, ldd; float sh, sw; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    if constexpr (V8 && !BWD) bilinear_fwd_v8_body(a.src, a.B, a.Hs, a.Ws, a.C, a.lds, a.dst, a.Hd, a.Wd, a.ldd, a.sh, a.sw, bx, gx);
    else if constexpr (V8 && BWD) bilinear_bwd_v8_body(a.src, a.B, a.Hs, a.Ws, a.C, a.lds, a.dst, a.Hd, a.Wd, a.ldd, a.sh, a.sw, bx, gx);
    else if constexpr (!BWD) bilinear_fwd_s_body<InT, OutT>(a.src, a.B, a.Hs, a.Ws, a.C, a.lds, a.dst, a.Hd, a.Wd, a.ldd, a.sh, a.sw, bx, gx);
    else bilinear_bwd_s_body<InT, OutT>(a.src, a.B, a.Hs, a.Ws, a.C, a.lds, a.dst, a.Hd, a.Wd, a.ldd, a.sh, a.sw, bx, gx);
  }
};
template <typename InT, typename OutT, bool V8, bool BWD>
int launch_bilinear(const void* src, int B, int Hs, int Ws, int C, int lds, void* dst, int Hd, int Wd, int ldd,
                    float sh, float sw, long nthreads, hipStream_t s) {
  typedef BilinearK<InT, OutT, V8, BWD> K;
  typename K::Args a{(const InT*)src, (OutT*)dst, B, Hs, Ws, C, lds, Hd, Wd, ldd, sh, sw};
  return ssa::submit<K>(a, grid_for(nthreads), 1, 0, s);
}

template <typename InT, int V>
struct BilinearBwdXK {
  struct Args { const InT* dy; float* tmp; int B, Ho, Wo, C, lddy, Wi; float sw; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    bilinear_bwd_x_body<InT, V>(a.dy, a.B, a.Ho, a.Wo, a.C, a.lddy, a.tmp, a.Wi, a.sw, bx, gx);
  }
};
template <typename OutT, int V>
struct BilinearBwdYK {
  struct Args { const float* tmp; OutT* dx; int B, Ho, Wi, C, Hi, lddx; float sh; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    bilinear_bwd_y_body<OutT, V>(a.tmp, a.B, a.Ho, a.Wi, a.C, a.dx, a.Hi, a.lddx, a.sh, bx, gx);
  }
};

// ---- few-channel tensors (class logits [.., 19], attention maps [.., 1]; fp32 on this path): one
// thread per PIXEL.  The source indices / weights are computed once per pixel instead of once per
// element (and with 32-bit index arithmetic); the forward stages the 256-pixel block of outputs in LDS
// ([pixel][C], odd C = conflict free) and writes it with unit stride.
constexpr int kPxMaxC = 32;      // LDS tile [256][C] stays below 32 KB

template <typename InT, typename OutT>
__device__ __forceinline__ void bilinear_fwd_px_body(const InT* __restrict__ x, int B, int Hi, int Wi, int C, int ldx,
                                                     OutT* __restrict__ y, int Ho, int Wo, float sh, float sw,
                                                     const int bx, const int gx) {
  SSA_DYN_LDS(float, tile);               // [256][C]
  const int npix = B * Ho * Wo;
  const int tid = threadIdx.x;
  for (int p0 = bx * 256; p0 < npix; p0 += gx * 256) {
    const int p = p0 + tid;
    if (p < npix) {
      const int ox = p % Wo;
      const int t = p / Wo;
      const int oy = t % Ho, b = t / Ho;
      const Src ys = src_index(oy, sh, Hi), xs = src_index(ox, sw, Wi);
      const InT* r0 = x + ((long)(b * Hi + ys.i0) * Wi) * ldx;
      const InT* r1 = x + ((long)(b * Hi + ys.i1) * Wi) * ldx;
      const InT* p00 = r0 + (long)xs.i0 * ldx;
      const InT* p01 = r0 + (long)xs.i1 * ldx;
      const InT* p10 = r1 + (long)xs.i0 * ldx;
      const InT* p11 = r1 + (long)xs.i1 * ldx;
      const float w00 = ys.l0 * xs.l0, w01 = ys.l0 * xs.l1, w10 = ys.l1 * xs.l0, w11 = ys.l1 * xs.l1;
      for (int c = 0; c < C; ++c) {
        // same association as the element-wise kernel: l0y*(l0x*a + l1x*b) + l1y*(l0x*c + l1x*d)
        const float o = ys.l0 * (xs.l0 * ld_as_f32(p00 + c) + xs.l1 * ld_as_f32(p01 + c)) +
                        ys.l1 * (xs.l0 * ld_as_f32(p10 + c) + xs.l1 * ld_as_f32(p11 + c));
        tile[tid * C + c] = o;
      }
      (void)w00; (void)w01; (void)w10; (void)w11;
    }
    __syncthreads();
    const int nval = min(256, npix - p0) * C;
    OutT* dst = y + (long)p0 * C;
    for (int i = tid; i < nval; i += 256) st_from_f32(dst + i, tile[i]);
    __syncthreads();
  }
}

// ---- backward of an UPSAMPLING resize of a few-channel fp32 tensor (class logits [.., 19], attention maps [.., 1]):
// the per-element gather above reads 8 x 8 (scale 4) scattered gradient values per input element, 76 bytes apart --
// 0.84 TB/s on the 80 MB logit gradients of a 1024 x 1024 step, six launches of ~95 us.  Here a workgroup owns a tile
// of kTileY x kTileX INPUT pixels: pass X reduces the gradient rows the tile can see along x into LDS
// (tmp[row][tx][c], lanes = (tx, c): runs of C floats, every gradient line fetched once per workgroup and re-used from
// L1 by the neighbouring taps), pass Y reduces the LDS rows along y -- window_x + window_y taps per element instead of
// their product, no intermediate in HBM, one launch.  Weights and window origins per tile row / column are computed once
// into LDS.  Summation order differs from the gather (x first): fp32 rounding only.
constexpr int kTileX = 16;

template <typename OutT, int TY, int TAPS>
__device__ __forceinline__ void bilinear_bwd_tile_body(const float* __restrict__ dy, int B, int Ho, int Wo, int C, int lddy,
                                                       OutT* __restrict__ dx, int Hi, int Wi, int lddx, float sh, float sw,
                                                       const int maxrows, const int bx, const int gx) {
  SSA_DYN_LDS(float, lds);
  // layout: wx[kTileX][kMaxCand] | wy[TY][kMaxCand] | xlo[kTileX] nx[kTileX] ylo[TY] ny[TY] (ints) | tmp[rows][kTileX][C]
  float* wxs = lds;
  float* wys = wxs + kTileX * kMaxCand;
  int* xlo = reinterpret_cast<int*>(wys + TY * kMaxCand);
  int* nxs = xlo + kTileX;
  int* ylo = nxs + kTileX;
  int* nys = ylo + TY;
  float* tmp = reinterpret_cast<float*>(nys + TY);
  const int tid = threadIdx.x;
  const int tiles_x = (Wi + kTileX - 1) / kTileX, tiles_y = (Hi + TY - 1) / TY;
  const int ntiles = B * tiles_y * tiles_x;
  const int rowlen = kTileX * C;
  for (int tile = bx; tile < ntiles; tile += gx) {
    const int tx0 = (tile % tiles_x) * kTileX;
    const int t1 = tile / tiles_x;
    const int ty0 = (t1 % tiles_y) * TY, b = t1 / tiles_y;
    __syncthreads();                       // the previous tile's readers are done with the tables and tmp
    if (tid < kTileX) {
      const int ix = tx0 + tid;
      int lo = 0, hi = -1;
      if (ix < Wi) tight_range(ix, sw, Wo, Wi, &lo, &hi);
      int n = hi - lo + 1;
      if (n > TAPS) n = TAPS;              // (never: the host picks TAPS >= the widest window of the scale)
      xlo[tid] = lo; nxs[tid] = n < 0 ? 0 : n;
      for (int k = 0; k < kMaxCand; ++k) wxs[tid * kMaxCand + k] = k < n ? weight_for(lo + k, sw, Wi, ix) : 0.f;
    } else if (tid >= 64 && tid < 64 + TY) {
      const int j = tid - 64, iy = ty0 + j;
      int lo = 0, hi = -1;
      if (iy < Hi) tight_range(iy, sh, Ho, Hi, &lo, &hi);
      int n = hi - lo + 1;
      if (n > TAPS) n = TAPS;
      ylo[j] = lo; nys[j] = n < 0 ? 0 : n;
      for (int k = 0; k < kMaxCand; ++k) wys[j * kMaxCand + k] = k < n ? weight_for(lo + k, sh, Hi, iy) : 0.f;
    }
    __syncthreads();
    const int r0 = ylo[0];
    int r1 = r0 - 1;
#pragma unroll
    for (int j = 0; j < TY; ++j) { const int e = ylo[j] + nys[j] - 1; if (nys[j] > 0 && e > r1) r1 = e; }
    int nrows = r1 - r0 + 1;
    if (nrows > maxrows) nrows = maxrows;          // (never: the host sized tmp for the tile's span)
    // ---- pass X: tmp[r][tx][c] = sum_k wx[tx][k] * dy[b, r0 + r, xlo[tx] + k, c]
    // An item = four gradient rows of one (tx, c): 4 x kMaxCand loads with no branch between them (taps beyond the
    // window re-read its last column with weight 0), so that they are all in flight before the first multiply --
    // with a data-dependent trip count the loop ran one load at a time (69 us per 80 MB launch, call U).
    const float* img = dy + (long)b * Ho * Wo * lddy;
    const int rgroups = (nrows + 3) >> 2;
    for (int item = tid; item < rgroups * rowlen; item += 256) {
      const int rg = item / rowlen, q = item - rg * rowlen;
      const int tx = q / C, c = q - tx * C;
      const int n = nxs[tx];
      float w[TAPS];
      int off[TAPS];
#pragma unroll
      for (int k = 0; k < TAPS; ++k) {
        w[k] = wxs[tx * kMaxCand + k];
        off[k] = (k < n ? k : (n > 0 ? n - 1 : 0)) * lddy;
      }
      const float* src = img + ((long)(r0 + rg * 4) * Wo + xlo[tx]) * lddy + c;
      float acc[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = rg * 4 + rr;
        const float* sr = src + (long)(r < nrows ? rr : 0) * Wo * lddy;      // rows past the tile: re-read, not stored
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) a += w[k] * sr[off[k]];
        acc[rr] = a;
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (rg * 4 + rr < nrows) tmp[(rg * 4 + rr) * rowlen + q] = n > 0 ? acc[rr] : 0.f;
    }
    __syncthreads();
    // ---- pass Y: dx[b, ty0 + j, tx0 + tx, c] = sum_k wy[j][k] * tmp[ylo[j] - r0 + k][tx][c]
    for (int item = tid; item < TY * rowlen; item += 256) {
      const int j = item / rowlen, q = item - j * rowlen;
      const int tx = q / C, c = q - tx * C;
      const int iy = ty0 + j, ix = tx0 + tx;
      if (iy >= Hi || ix >= Wi) continue;
      const int n = nys[j];
      const int base = ylo[j] - r0;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < TAPS; ++k) {
        int r = base + (k < n ? k : n - 1);
        r = r < nrows ? r : nrows - 1;
        acc += wys[j * kMaxCand + k] * tmp[r * rowlen + q];
      }
      st_from_f32(dx + ((long)(b * Hi + iy) * Wi + ix) * lddx + c, acc);
    }
  }
}

template <typename OutT, int TY, int TAPS>
struct BilinearBwdTileK {
  struct Args { const float* dy; OutT* dx; int B, Ho, Wo, C, lddy, Hi, Wi, lddx; float sh, sw; int maxrows; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    bilinear_bwd_tile_body<OutT, TY, TAPS>(a.dy, a.B, a.Ho, a.Wo, a.C, a.lddy, a.dx, a.Hi, a.Wi, a.lddx, a.sh, a.sw, a.maxrows, bx, gx);
  }
};
// gradient rows / columns an input pixel can receive from: ceil(2 / scale) + 1 covers the two-tap footprint's inverse
static int bwd_window(float scale) { return (int)ceilf(2.f / scale) + 1; }
static int bwd_tile_rows(float sh, int TY) { return (int)ceilf((float)(TY + 1) / sh) + 2; }
static size_t bwd_tile_lds(float sh, int TY, int C) {
  return sizeof(float) * ((kTileX + TY) * kMaxCand + 2 * (kTileX + TY) + (size_t)bwd_tile_rows(sh, TY) * kTileX * C);
}
// taps the widest window of a scale has: 2 f for an integer upsampling factor f (what the tight ranges come to), else
// the inverse footprint's bound
static int bwd_taps(float scale) {
  const float f = 1.f / scale;
  return f == floorf(f) ? 2 * (int)f : bwd_window(scale);
}
template <typename OutT, int TY, int TAPS>
int launch_bilinear_bwd_tile(const void* dy, int B, int Ho, int Wo, int C, int lddy, void* dx, int Hi, int Wi, int lddx,
                             float sh, float sw, hipStream_t s) {
  typedef BilinearBwdTileK<OutT, TY, TAPS> K;
  typename K::Args a{(const float*)dy, (OutT*)dx, B, Ho, Wo, C, lddy, Hi, Wi, lddx, sh, sw, bwd_tile_rows(sh, TY)};
  const long ntiles = (long)B * ((Hi + TY - 1) / TY) * ((Wi + kTileX - 1) / kTileX);
  return ssa::submit<K>(a, (int)(ntiles > 16384 ? 16384 : ntiles), 1, bwd_tile_lds(sh, TY, C), s);
}
// routed here: fp32 gradient of an upsampling resize, few channels, windows that fit the weight tables and the LDS rows
static bool bwd_tile_ok(int B, int Ho, int Wo, int Hi, int Wi, int C, float sh, float sw, int TY) {
  if (C < 8 || C > kPxMaxC || sh > 1.f || sw > 1.f) return false;       // (one channel: 16 of 256 lanes busy -- the gather)
  if (bwd_taps(sw) > kMaxCand || bwd_taps(sh) > kMaxCand) return false;
  if (bwd_tile_lds(sh, TY, C) > 60 * 1024) return false;
  return (long)B * Ho * Wo < (1L << 30);
}

template <typename InT, typename OutT>
struct BilinearPxK {
  struct Args { const InT* src; OutT* dst; int B, Hs, Ws, C, lds, Hd, Wd, ldd; float sh, sw; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, int bx, int, int gx) {
    bilinear_fwd_px_body<InT, OutT>(a.src, a.B, a.Hs, a.Ws, a.C, a.lds, a.dst, a.Hd, a.Wd, a.sh, a.sw, bx, gx);
  }
};
template <typename InT, typename OutT>
int launch_bilinear_px(const void* src, int B, int Hs, int Ws, int C, int lds, void* dst, int Hd, int Wd, int ldd,
                       float sh, float sw, hipStream_t s) {
  typedef BilinearPxK<InT, OutT> K;
  typename K::Args a{(const InT*)src, (OutT*)dst, B, Hs, Ws, C, lds, Hd, Wd, ldd, sh, sw};
  const long npix = (long)B * Hd * Wd;
  return ssa::submit<K>(a, grid_for(npix), 1, (size_t)256 * C * sizeof(float), s);
}
// few channels, dense destination, 32-bit pixel counts
static bool px_ok(int B, int Hs, int Ws, int Hd, int Wd, int C, int ldd) {
  return C <= kPxMaxC && ldd == C && (long)B * Hs * Ws < (1L << 30) && (long)B * Hd * Wd < (1L << 30);
}

}  // namespace

extern "C" {

int ssa_bilinear_fwd(const void* x, int in_dtype, int B, int Hi, int Wi, int C, int ldx, void* y,
                     int out_dtype, int Ho, int Wo, int ldy, void* stream) {
  if (!x || !y || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const long n = (long)B * Ho * Wo * C;
  if (in_dtype == 0 && out_dtype == 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0)
    return launch_bilinear<bf16_t, bf16_t, true, false>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, n / 8, s);
  if (px_ok(B, Hi, Wi, Ho, Wo, C, ldy)) {
    if (in_dtype == 1 && out_dtype == 1) return launch_bilinear_px<float, float>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, s);
    if (in_dtype == 0 && out_dtype == 1) return launch_bilinear_px<bf16_t, float>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, s);
  }
  if (in_dtype == 0 && out_dtype == 0)
    return launch_bilinear<bf16_t, bf16_t, false, false>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, n, s);
  if (in_dtype == 0 && out_dtype == 1)
    return launch_bilinear<bf16_t, float, false, false>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, n, s);
  if (in_dtype == 1 && out_dtype == 1)
    return launch_bilinear<float, float, false, false>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, n, s);
  if (in_dtype == 1 && out_dtype == 0)
    return launch_bilinear<float, bf16_t, false, false>(x, B, Hi, Wi, C, ldx, y, Ho, Wo, ldy, sh, sw, n, s);
  return SSA_EINVAL;
}

int ssa_bilinear_bwd(const void* dy, int dy_dtype, int B, int Ho, int Wo, int C, int lddy, void* dx,
                     int dx_dtype, int Hi, int Wi, int lddx, void* stream) {
  if (!dy || !dx || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  const long n = (long)B * Hi * Wi * C;
  if (dy_dtype == 0 && dx_dtype == 0 && C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0)
    return launch_bilinear<bf16_t, bf16_t, true, true>(dy, B, Ho, Wo, C, lddy, dx, Hi, Wi, lddx, sh, sw, n / 8, s);
  // (a per-pixel gather for the few-channel fp32 tensors was measured 4x SLOWER than the per-element
  // kernel -- lanes = pixels read with a 76-byte stride -- and is not used; profiles/r02_notes.md)
  // few-channel fp32 gradient of an upsampling resize: the LDS-tiled separable kernel (one launch)
  static const bool tiled = [] { const char* e = getenv("SSA_BILINEAR_BWD_TILE"); return !(e && e[0] == '0'); }();
  if (tiled && dy_dtype == 1 && bwd_tile_ok(B, Ho, Wo, Hi, Wi, C, sh, sw, 4)) {
    const int taps = bwd_taps(sw) > bwd_taps(sh) ? bwd_taps(sw) : bwd_taps(sh);
#define SSA_BWD_TILE(T, N) launch_bilinear_bwd_tile<T, 4, N>(dy, B, Ho, Wo, C, lddy, dx, Hi, Wi, lddx, sh, sw, s)
    if (dx_dtype == 1) return taps <= 4 ? SSA_BWD_TILE(float, 4) : taps <= 8 ? SSA_BWD_TILE(float, 8) : SSA_BWD_TILE(float, 12);
    return taps <= 4 ? SSA_BWD_TILE(bf16_t, 4) : taps <= 8 ? SSA_BWD_TILE(bf16_t, 8) : SSA_BWD_TILE(bf16_t, 12);
#undef SSA_BWD_TILE
  }
  if (dy_dtype == 1 && dx_dtype == 1)
    return launch_bilinear<float, float, false, true>(dy, B, Ho, Wo, C, lddy, dx, Hi, Wi, lddx, sh, sw, n, s);
  if (dy_dtype == 1 && dx_dtype == 0)
    return launch_bilinear<float, bf16_t, false, true>(dy, B, Ho, Wo, C, lddy, dx, Hi, Wi, lddx, sh, sw, n, s);
  if (dy_dtype == 0 && dx_dtype == 0)
    return launch_bilinear<bf16_t, bf16_t, false, true>(dy, B, Ho, Wo, C, lddy, dx, Hi, Wi, lddx, sh, sw, n, s);
  return SSA_EINVAL;
}

int ssa_bilinear_bwd_x(const void* dy, int dy_dtype, int B, int Ho, int Wo, int C, int lddy, float* tmp, int Wi,
                       void* stream) {
  if (!dy || !tmp || B <= 0 || Ho <= 0 || Wo <= 0 || Wi <= 0 || C <= 0) return SSA_EINVAL;
  if (reinterpret_cast<uintptr_t>(tmp) & 15u) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float sw = (float)Wi / (float)Wo;
  const long n = (long)B * Ho * Wi * C;
  if (dy_dtype == 0 && C % 8 == 0 && lddy % 8 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15u) == 0) {
    typedef BilinearBwdXK<bf16_t, 8> K;
    K::Args a{(const bf16_t*)dy, tmp, B, Ho, Wo, C, lddy, Wi, sw};
    return ssa::submit<K>(a, grid_for(n / 8 / kXRows + 1), 1, 0, s);
  }
  if (dy_dtype == 0) {
    typedef BilinearBwdXK<bf16_t, 1> K;
    K::Args a{(const bf16_t*)dy, tmp, B, Ho, Wo, C, lddy, Wi, sw};
    return ssa::submit<K>(a, grid_for(n / kXRows + 1), 1, 0, s);
  }
  if (dy_dtype == 1) {
    typedef BilinearBwdXK<float, 1> K;
    K::Args a{(const float*)dy, tmp, B, Ho, Wo, C, lddy, Wi, sw};
    return ssa::submit<K>(a, grid_for(n / kXRows + 1), 1, 0, s);
  }
  return SSA_EINVAL;
}

int ssa_bilinear_bwd_y(const float* tmp, int B, int Ho, int Wi, int C, void* dx, int dx_dtype, int Hi, int lddx,
                       void* stream) {
  if (!tmp || !dx || B <= 0 || Ho <= 0 || Hi <= 0 || Wi <= 0 || C <= 0) return SSA_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float sh = (float)Hi / (float)Ho;
  const long n = (long)B * Hi * Wi * C;
  if (dx_dtype == 0 && C % 8 == 0 && lddx % 8 == 0 && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(tmp)) & 15u) == 0) {
    typedef BilinearBwdYK<bf16_t, 8> K;
    K::Args a{tmp, (bf16_t*)dx, B, Ho, Wi, C, Hi, lddx, sh};
    return ssa::submit<K>(a, grid_for(n / 8), 1, 0, s);
  }
  if (dx_dtype == 0) {
    typedef BilinearBwdYK<bf16_t, 1> K;
    K::Args a{tmp, (bf16_t*)dx, B, Ho, Wi, C, Hi, lddx, sh};
    return ssa::submit<K>(a, grid_for(n), 1, 0, s);
  }
  if (dx_dtype == 1) {
    typedef BilinearBwdYK<float, 1> K;
    K::Args a{tmp, (float*)dx, B, Ho, Wi, C, Hi, lddx, sh};
    return ssa::submit<K>(a, grid_for(n), 1, 0, s);
  }
  return SSA_EINVAL;
}

int ssa_image_resize_to_nhwc_bf16(const float* x, int B, int C, int Hi, int Wi, void* y, int Ho,
                                  int Wo, int cpad, void* stream) {
  if (!x || !y || cpad % 8 || cpad < C || Ho <= 0 || Wo <= 0) return SSA_EINVAL;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  hipLaunchKernelGGL(image_resize_kernel, dim3(grid_for((long)B * Ho * Wo)), dim3(256), 0,
                     (hipStream_t)stream, x, B, C, Hi, Wi, (bf16_t*)y, Ho, Wo, cpad, sh, sw);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_resize_nearest_u8(const unsigned char* src, int B, int Hs, int Ws, unsigned char* dst, int Hd,
                          int Wd, const int* iy_table, const int* ix_table, void* stream) {
  if (!src || !dst || !iy_table || !ix_table || B < 1 || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1)
    return SSA_EINVAL;
  const long n = (long)B * Hd * Wd;
  const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(resize_nearest_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, Hs,
                     Ws, dst, Hd, Wd, iy_table, ix_table, n);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
