// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations, packed
// bf16 filters [N][Kpad], fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
//   forward / data-gradient:  Y[m, n] = sum_k A[m, k] * Wp[n, k]
//        m = (b, oy, ox) output pixel, n = output channel, k = (kh, kw, ci)
//        A is never materialised: each 16-byte piece (8 channels of one tap of
//        one pixel) is gathered straight from the NHWC tensor into LDS.
//   weight-gradient:          dW[co, k] = sum_p dY[p, co] * A[p, k]
//        the reduction runs over pixels: both operands are staged pixel-major
//        (as loaded) and read with transposing ds_read_b64_tr_b16 fragments; the
//        pixel axis is split over blockIdx.y and the fp32 partials are summed by
//        wgrad_reduce_kernel (deterministic).  The head convs use
//        conv_wgrad_head.hip instead.
//
// Replaces cuDNN behind nn.Conv2d on the reference's hot path (SURVEY.md K1-K6;
// network/hrnetv2.py:31-34, network/ocrnet.py:54-58, network/utils.py:192-198).
//
// Tile geometry: a workgroup of WGM x WGN wavefronts; each wavefront owns
// MI x NI MFMA tiles of 32x32.  K advances 32 per stage (two MFMA k-steps),
// LDS is double buffered with the next stage's global loads issued before the
// current stage's MFMAs.  LDS rows are padded to 80 bytes so the 16 lanes that
// ds_read_b128 services together hit 16 distinct 16-byte slots.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"
#include <limits.h>
#include <string.h>
#include <stdlib.h>

namespace {

constexpr int BK = 32;
constexpr int LDT = BK + 8;

template <int MI, int NI, int BK = ::BK, int LDT = BK + 8>
__device__ __forceinline__ void mma_stage(const bf16_t* __restrict__ As,
                                          const bf16_t* __restrict__ Bs, int a_row0,
                                          int b_row0, int lane,
                                          f32x16_t (&acc)[MI][NI]) {
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    bf16x8_t af[MI], bfr[NI];
    const int koff = ks * 16 + (lane >> 5) * 8;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      af[mi] = *reinterpret_cast<const bf16x8_t*>(As + (a_row0 + mi * 32 + (lane & 31)) * LDT + koff);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      bfr[ni] = *reinterpret_cast<const bf16x8_t*>(Bs + (b_row0 + ni * 32 + (lane & 31)) * LDT + koff);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = ssa_mfma32(af[mi], bfr[ni], acc[mi][ni]);
  }
}

// ----------------------------------------------------------------------------
// forward / dgrad kernel
// ----------------------------------------------------------------------------
struct IgemmArgs {
  ssa_conv_desc d;
  const bf16_t* x; const bf16_t* w; const float* bias; void* y; double* stats;
  int tiles_n, tr_shift;
  // strided output (parity classes of a stride-2 data gradient, ssa_conv2d_dgrad_s2): GEMM row
  // m = (b, oy, ox) is stored at pixel b*o_HW + (oy*o_mul + o_py)*o_W + ox*o_mul + o_px; o_mul = 0: pixel m
  int o_mul, o_py, o_px, o_W, o_HW;
  // inference epilogue (ssa_conv2d_igemm_affine): z = act(scale * y + shift [+ residual]) on the 16-bit-rounded y
  const float* aff; const bf16_t* res; int ldres, relu;
};

// K-stage depth of the forward / data-gradient kernel: 64 for the small tiles.  Their launches are chains of
// dependent K-stages (store -> barrier -> fragment reads -> two MFMAs) on a mostly idle chip -- a 3x3 stride-2 conv
// with 192 input channels is 54 stages of 32 --, so halving the stage count is what shortens them; the 128-wide
// tiles keep 32 (two workgroups per CU).  The packed filter rows stay padded to 32 (the last stage is masked).
template <int MI, int NI> struct IgemmBK { static constexpr int value = MI * NI <= 2 ? 64 : 32; };

template <int WGM, int WGN, int MI, int NI>
struct ConvIgemm {
  typedef IgemmArgs Args;
  static constexpr int BK = IgemmBK<MI, NI>::value;
  static constexpr int LDT = BK + 8;
  static constexpr int NT = 64 * WGM * WGN;
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int /*gx*/) {
  const ssa_conv_desc& d = a.d;
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ w = a.w;
  const float* __restrict__ bias = a.bias;
  void* __restrict__ yv = a.y;
  double* __restrict__ stats = a.stats;
  const int tiles_n = a.tiles_n, tr_shift = a.tr_shift;
  constexpr int BM = WGM * MI * 32, BN = WGN * NI * 32;
  constexpr int PPR = BK / 8;
  constexpr int RPP = NT / PPR;
  constexpr int A_IT = (BM + RPP - 1) / RPP, B_IT = (BN + RPP - 1) / RPP;
  constexpr int STAGE = (BM + BN) * LDT;
  SSA_DYN_LDS(unsigned char, smem);
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tn = bx % tiles_n, tm = bx / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = d.B * d.Ho * d.Wo;
  const int HoWo = d.Ho * d.Wo;
  const int pc = tid % PPR, r0 = tid / PPR;

  int iy0[A_IT], ix0[A_IT], pb[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int r = r0 + i * RPP, m = m0 + r;
    if (r < BM && m < M) {
      const int b = m / HoWo, rem = m - b * HoWo;
      const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
      if (d.transposed) { iy0[i] = oy - d.pad; ix0[i] = ox - d.pad; }
      else { iy0[i] = oy * d.stride - d.pad; ix0[i] = ox * d.stride - d.pad; }
      pb[i] = b * d.H * d.W;
    } else {
      iy0[i] = INT_MIN / 2; ix0[i] = INT_MIN / 2; pb[i] = 0;
    }
  }
  // this thread's K cursor: channel offset inside the tap, and the tap itself
  int kc, kh, kw;
  {
    const int kpos = pc * 8;
    const int tap = kpos / d.Cin;
    kc = kpos - tap * d.Cin;
    kh = tap / d.KW;
    kw = tap - kh * d.KW;
  }
  const int tr_mask = (1 << tr_shift) - 1;
  const int nk = (d.Kpad + BK - 1) / BK;
  const int o_mul = a.o_mul, o_py = a.o_py, o_px = a.o_px, o_W = a.o_W, o_HW = a.o_HW;
  auto opix = [&](int m) -> long {
    if (o_mul == 0) return m;
    const int b = m / HoWo, rem = m - b * HoWo;
    const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
    return (long)b * o_HW + (long)(oy * o_mul + o_py) * o_W + ox * o_mul + o_px;
  };

  // Global loads run PD - 1 K-stages ahead of the MFMAs, through a ring of PD register sets.  With one stage in
  // flight (the first version) a 32-deep K-step cost a full memory latency: these launches are small (a 64x64 tile
  // is two MFMAs per wave and K-step), so the loop ran at ~1 us per step whatever it computed (profiles/r03_notes.md).
  // Every lane loads (pieces outside the image / past Cout read a valid address and are zeroed when they are staged):
  // a select on the loaded value would put the wait inside the load phase.
  constexpr int PD = 4;
  uint4 ra[PD][A_IT], rb[PD][B_IT];
  unsigned amask[PD], bmask[PD];
  auto gload = [&](int kt, int slot) {
    const bool kvalid = kh < d.KH;
    unsigned am = 0, bm = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int iy = iy0[i] + kh * d.dil, ix = ix0[i] + kw * d.dil;
      bool ok = kvalid;
      if (d.transposed) {
        ok = ok && iy >= 0 && ix >= 0 && ((iy & tr_mask) == 0) && ((ix & tr_mask) == 0);
        iy >>= tr_shift; ix >>= tr_shift;
        ok = ok && iy < d.H && ix < d.W;
      } else {
        ok = ok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
      }
      const long pix = ok ? (long)(pb[i] + iy * d.W + ix) : 0;
      const bf16_t* p = x + pix * d.ldx + (ok ? kc : 0);
      ra[slot][i] = *reinterpret_cast<const uint4*>(p);
      am |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      const int r = r0 + j * RPP, n = n0 + r;
      const bool ok = (r < BN) && (n < d.Cout) && (kt * BK + pc * 8 < d.Kpad);
      const bf16_t* p = w + (ok ? (long)n * d.Kpad + kt * BK + pc * 8 : 0);
      rb[slot][j] = *reinterpret_cast<const uint4*>(p);
      bm |= (ok ? 1u : 0u) << j;
    }
    amask[slot] = am;
    bmask[slot] = bm;
    // advance the cursor by one stage
    kc += BK;
    while (kc >= d.Cin) {
      kc -= d.Cin;
      if (++kw == d.KW) { kw = 0; ++kh; }
    }
  };
  auto lstore = [&](int buf, int slot) {
    bf16_t* As = lds + buf * STAGE;
    bf16_t* Bs = As + BM * LDT;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int r = r0 + i * RPP;
      if (r < BM)
        *reinterpret_cast<uint4*>(As + r * LDT + pc * 8) = ((amask[slot] >> i) & 1u) ? ra[slot][i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      const int r = r0 + j * RPP;
      if (r < BN)
        *reinterpret_cast<uint4*>(Bs + r * LDT + pc * 8) = ((bmask[slot] >> j) & 1u) ? rb[slot][j] : make_uint4(0, 0, 0, 0);
    }
  };

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#pragma unroll
  for (int j = 0; j < PD - 1; ++j)
    if (j < nk) gload(j, j);
  lstore(0, 0);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
    for (int j = 0; j < PD; ++j) {
      const int kt = kt0 + j;                    // stage kt lives in ring slot kt % PD = j (kt0 is a multiple of PD)
      if (kt < nk) {
        const int buf = kt & 1;
        if (kt + PD - 1 < nk) gload(kt + PD - 1, (j + PD - 1) % PD);
        const bf16_t* As = lds + buf * STAGE;
        mma_stage<MI, NI, BK, LDT>(As, As + BM * LDT, wm * MI * 32, wn * NI * 32, lane, acc);
        if (kt + 1 < nk) lstore(buf ^ 1, (j + 1) % PD);
        __syncthreads();
      }
    }
  }

  // ------------------------------------------------------------- epilogue
  if (d.out_f32) {
    float* y = reinterpret_cast<float*>(yv);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * NI * 32 + ni * 32 + (lane & 31);
        const float bv = (bias != nullptr && n < d.Cout) ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * MI * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int m = m0 + row;
          if (m < M && n < d.Cout) y[opix(m) * d.ldy + n] = acc[mi][ni][r] + bv;
        }
      }
    return;
  }
  constexpr int LDC = BN + 8;
  bf16_t* Cs = lds;  // safe: the K loop ended on a barrier
  float* red = reinterpret_cast<float*>(smem + (size_t)BM * LDC * 2);   // [WGM][2][BN] BN-statistics partials
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = wn * NI * 32 + ni * 32 + (lane & 31);
    const int n = n0 + col;
    const float bv = (bias != nullptr && n < d.Cout) ? bias[n] : 0.f;
    float ssum = 0.f, sq = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * MI * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bf16_t o = f2bf(acc[mi][ni][r] + bv);
        Cs[row * LDC + col] = o;
        if (stats != nullptr) {          // batch statistics of the bf16-rounded outputs (fused bn_stats)
          const float f = (m0 + row < M) ? bf2f(o) : 0.f;
          ssum += f;
          sq += f * f;
        }
      }
    }
    if (stats != nullptr) {
      ssum += __shfl_xor(ssum, 32, 64);
      sq += __shfl_xor(sq, 32, 64);
      if (lane < 32) {
        red[(wm * 2 + 0) * BN + col] = ssum;
        red[(wm * 2 + 1) * BN + col] = sq;
      }
    }
  }
  __syncthreads();
  if (stats != nullptr) {
    double* st = stats + (long)(bx % 8) * 2 * d.Cout;     // replica, as conv_tile.hip
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, col = i - which * BN;
      const int n = n0 + col;
      if (n < d.Cout) {
        float v = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WGM; ++w_) v += red[(w_ * 2 + which) * BN + col];
        atomicAdd(&st[which * d.Cout + n], (double)v);
      }
    }
  }
  bf16_t* y = reinterpret_cast<bf16_t*>(yv);
  constexpr int CPR = BN / 8;
  for (int idx = tid; idx < BM * CPR; idx += NT) {
    const int row = idx / CPR, cp = idx - row * CPR;
    const int m = m0 + row, n = n0 + cp * 8;
    if (m >= M || n >= d.Cout) continue;
    bf16_t* dst = y + opix(m) * d.ldy + n;
    const bf16_t* src = Cs + row * LDC + cp * 8;
    if (a.aff != nullptr) {
      // the conv's inference BatchNorm (+ residual, ReLU) with the arithmetic of bn_apply_rows on the rounded output
      const float* sc = a.aff + n;
      const float* sh = a.aff + d.Cout + n;
      const bf16_t* rp = a.res ? a.res + opix(m) * a.ldres + n : nullptr;
      if (n + 8 <= d.Cout) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(src), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = f[j] * sc[j] + sh[j];
        if (rp) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(rp), r);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += r[j];
        }
        if (a.relu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
        }
        *reinterpret_cast<uint4*>(dst) = pack8(f);
      } else {
        for (int j = 0; n + j < d.Cout; ++j) {
          float f = bf2f(src[j]) * sc[j] + sh[j];
          if (rp) f += bf2f(rp[j]);
          if (a.relu) f = fmaxf(f, 0.f);
          dst[j] = f2bf(f);
        }
      }
      continue;
    }
    if (n + 8 <= d.Cout) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      for (int j = 0; n + j < d.Cout; ++j) dst[j] = src[j];
    }
  }
  }
};

// ----------------------------------------------------------------------------
// weight-gradient kernel: [pixel][channel] LDS images (16-byte stores, exactly
// as loaded) + ds_read_b64_tr_b16 transposing fragment reads.
//
// ds_read_b64_tr_b16 lane map (probed on gfx950, tests/test_kernels_gpu.py::
// test_probe_tr16): inside a 16-lane group, lane i supplies the address of an
// 8-byte piece (4 bf16); result lane c, element j = element (c&3) of the piece
// supplied by lane 4j + (c>>2).  With lane i = 4j+q pointing at
// (pixel p0+j, channels n0+4q..+3), lane c receives channel n0+c at pixels
// p0..p0+3 -- the MFMA operand layout (8 consecutive k per lane = two reads).
// Row stride S of the LDS image satisfies S % 256 == 64 bytes, which spreads
// the 32 lanes the LDS services together over 32 distinct 8-byte slots.
// ----------------------------------------------------------------------------
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int tr_row_stride(int cols) {  // in elements
  int bytes = cols * 2;
  int s = (bytes + 255) / 256 * 256 + 64;
  if (s - 256 >= bytes) s -= 256;
  return s / 2;
}

__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* tile, int stride, int row0, int kbase,
                                            int lane) {
  const int i = lane & 15, j = i >> 2, q = i & 3;
  const int col = row0 + 16 * ((lane >> 4) & 1) + 4 * q;
  const int pix = kbase + 8 * (lane >> 5) + j;
  const bf16_t* p0 = tile + pix * stride + col;
  const s16x4_t lo = ssa_tr16_b64(p0);
  const s16x4_t hi = ssa_tr16_b64(p0 + 4 * stride);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// BKP: pixels (GEMM k) per LDS stage, 32 or 64.
struct WgradTrArgs {
  ssa_conv_desc d;
  const bf16_t* x; const bf16_t* dy; float* partial;
  int lddy, cout_pad, tiles_n, chunk;
};

template <int WGM, int WGN, int MI, int NI, int BKP>
struct ConvWgradTr {
  typedef WgradTrArgs Args;
  static constexpr int NT = 64 * WGM * WGN;
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const ssa_conv_desc& d = a.d;
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ dy = a.dy;
  float* __restrict__ partial = a.partial;
  const int lddy = a.lddy, cout_pad = a.cout_pad, tiles_n = a.tiles_n, chunk = a.chunk;
  constexpr int BM = WGM * MI * 32, BN = WGN * NI * 32;
  constexpr int SA = tr_row_stride(BM), SB = tr_row_stride(BN);
  constexpr int PA = BM / 8, PB = BN / 8;          // 16-byte pieces per pixel row
  constexpr int A_PIECES = BKP * PA, B_PIECES = BKP * PB;
  constexpr int A_IT = (A_PIECES + NT - 1) / NT, B_IT = (B_PIECES + NT - 1) / NT;
  constexpr int STAGE = BKP * (SA + SB);
  SSA_DYN_LDS(unsigned char, smem);
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int tn = bx % tiles_n, tm = bx / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int P = d.B * d.Ho * d.Wo, HoWo = d.Ho * d.Wo;
  const int Kflat = d.KH * d.KW * d.Cin;
  const int p_begin = by * chunk;
  const int p_end = min(P, p_begin + chunk);

  // piece -> (pixel row, channel piece): channel piece fastest => coalesced global loads
  int a_pix[A_IT], a_co[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int idx = tid + i * NT;
    a_pix[i] = idx / PA;
    a_co[i] = (idx - a_pix[i] * PA) * 8;
    a_ok[i] = (idx < A_PIECES) && (m0 + a_co[i] < cout_pad);
  }
  int b_pix[B_IT], b_col[B_IT], b_ci[B_IT], b_dy[B_IT], b_dx[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int j = 0; j < B_IT; ++j) {
    const int idx = tid + j * NT;
    b_pix[j] = idx / PB;
    b_col[j] = (idx - b_pix[j] * PB) * 8;
    const int kcol = n0 + b_col[j];
    b_ok[j] = (idx < B_PIECES) && (kcol < Kflat);
    const int tap = kcol / d.Cin;
    b_ci[j] = kcol - tap * d.Cin;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    b_dy[j] = kh * d.dil - d.pad;
    b_dx[j] = kw * d.dil - d.pad;
  }
  uint4 ra[A_IT], rb[B_IT];
  auto gload = [&](int kt) {
    const int pbase = p_begin + kt * BKP;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int p = pbase + a_pix[i];
      const bool ok = a_ok[i] && p < p_end;
      const bf16_t* ptr = dy + (long)(ok ? p : 0) * lddy + m0 + a_co[i];
      ra[i] = ok ? *reinterpret_cast<const uint4*>(ptr) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
      const int p = pbase + b_pix[j];
      bool ok = b_ok[j] && p < p_end;
      const int pp = ok ? p : 0;
      const int b = pp / HoWo, rem = pp - b * HoWo;
      const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
      const int iy = oy * d.stride + b_dy[j], ix = ox * d.stride + b_dx[j];
      ok = ok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
      const long pix = ok ? (long)((b * d.H + iy) * d.W + ix) : 0;
      rb[j] = ok ? *reinterpret_cast<const uint4*>(x + pix * d.ldx + b_ci[j]) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* As = lds + buf * STAGE;
    bf16_t* Bs = As + BKP * SA;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (tid + i * NT < A_PIECES) *reinterpret_cast<uint4*>(As + a_pix[i] * SA + a_co[i]) = ra[i];
#pragma unroll
    for (int j = 0; j < B_IT; ++j)
      if (tid + j * NT < B_PIECES) *reinterpret_cast<uint4*>(Bs + b_pix[j] * SB + b_col[j]) = rb[j];
  };

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int nk = (p_end - p_begin + BKP - 1) / BKP;
  if (nk > 0) {
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nk) gload(kt + 1);
      const bf16_t* As = lds + buf * STAGE;
      const bf16_t* Bs = As + BKP * SA;
#pragma unroll
      for (int ks = 0; ks < BKP / 16; ++ks) {
        bf16x8_t af[MI], bfr[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = tr_frag(As, SA, wm * MI * 32 + mi * 32, ks * 16, lane);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bfr[ni] = tr_frag(Bs, SB, wn * NI * 32 + ni * 32, ks * 16, lane);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = ssa_mfma32(af[mi], bfr[ni], acc[mi][ni]);
      }
      if (kt + 1 < nk) lstore(buf ^ 1);
      __syncthreads();
    }
  }
  float* out = partial + (long)by * cout_pad * Kflat;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int kcol = n0 + wn * NI * 32 + ni * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * MI * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < cout_pad && kcol < Kflat) out[(long)co * Kflat + kcol] = acc[mi][ni][r];
      }
    }
  }
};

// Workgroup (co, k-chunk of 64): 64 k-columns x 4 split lanes; each thread sums
// every 4th split (coalesced 256-byte rows), the 4 lanes meet in LDS, and the
// chunk is written to its OIHW positions.  k = (kh,kw,ci) -> (ci,kh,kw).
// One workgroup per output channel co: sums the splits of partial[.][co][k] for every k (coalesced:
// consecutive threads read consecutive k), transposes k = (kh,kw,ci) -> (ci,kh,kw) through LDS and
// writes -- or adds to -- the channel's contiguous run of the OIHW gradient with unit stride.
struct WgradReduceK {
  struct Args { const float* partial; float* dw; int nsplit, cout_pad, Cout, Cin_pad, Cin, KH, KW, accumulate, vec4; };
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int /*gx*/) {
    SSA_DYN_LDS(float, sh);                 // [Cin * taps] in OIHW order
    const float* __restrict__ partial = a.partial;
    const int nsplit = a.nsplit, Cin_pad = a.Cin_pad, Cin = a.Cin;
    const int co = bx;
    const int taps = a.KH * a.KW;
    const int Kflat = taps * Cin_pad;
    const long split_stride = (long)a.cout_pad * Kflat;
    const float* src0 = partial + (long)co * Kflat;
    // 16-byte loads (Kflat is a multiple of 8), eight splits in flight per thread: 128 bytes per lane on their way
    // (round 3 read one float per lane and split: 1.0 ms per step at 3.2 TB/s for a pure streaming sum)
    for (int k = threadIdx.x * 4; a.vec4 && k < Kflat; k += NT * 4) {
      const float* src = src0 + k;
      float4 acc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      int sp = 0;
      for (; sp + 7 < nsplit; sp += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long)(sp + u) * split_stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
      }
      for (; sp < nsplit; ++sp) {      // (a run-time index into acc[] would put the array in scratch memory: call H)
        const float4 v = *reinterpret_cast<const float4*>(src + (long)sp * split_stride);
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
      }
      float r[4];
      r[0] = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
      r[1] = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
      r[2] = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
      r[3] = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = k + j;
        const int tap = kk / Cin_pad, ci = kk - tap * Cin_pad;
        if (ci < Cin) sh[ci * taps + tap] = r[j];
      }
    }
    // rows that are not runs of 16-byte pieces (the OCR matrix products' odd shapes): one float per lane
    for (int k = threadIdx.x; !a.vec4 && k < Kflat; k += NT) {
      const float* src = src0 + k;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int sp = 0;
      for (; sp + 3 < nsplit; sp += 4) {
        s0 += src[(long)sp * split_stride];
        s1 += src[(long)(sp + 1) * split_stride];
        s2 += src[(long)(sp + 2) * split_stride];
        s3 += src[(long)(sp + 3) * split_stride];
      }
      for (; sp < nsplit; ++sp) s0 += src[(long)sp * split_stride];
      const int tap = k / Cin_pad, ci = k - tap * Cin_pad;
      if (ci < Cin) sh[ci * taps + tap] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    float* __restrict__ dst = a.dw + (long)co * Cin * taps;
    // accumulate: dw is the layer's slice of the step's gradient arena (cleared once per step);
    // one reduce job per layer and launch, so the read-modify-write has no concurrent writer
    if (a.accumulate) {
      for (int i = threadIdx.x; i < Cin * taps; i += NT) dst[i] += sh[i];
    } else {
      for (int i = threadIdx.x; i < Cin * taps; i += NT) dst[i] = sh[i];
    }
  }
};

struct PackJob {           // mirror of ssa_pack_job (include/semseg_hip.h)
  const float* w;
  bf16_t* out;
  long elem_begin;         // tiled repack: 1 + index of the next job packed from the SAME source tile (0: none)
  int Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows, layout;   // layout: reserved, 0 (one fragment order)
};

// Stride-2 data gradient by output parity (ssa_conv2d_dgrad_s2): dx[2m+p] along one axis of a 3-tap,
// pad-1, stride-2 conv is  p = 0: w[1]*dy[m];  p = 1: w[2]*dy[m] + w[0]*dy[m+1]  -- a correlation of dy
// with 1 + p taps; tap j of class p is forward tap s2_tap(p, j).
__host__ __device__ __forceinline__ int s2_tap(int p, int j) { return p == 0 ? 1 : 2 - 2 * j; }

// mode 0: forward operand, 1: data-gradient operand (transposed, taps flipped), 4 + 2*py + px: the
// data-gradient operand of parity class (py, px) of a 3x3 stride-2 conv: rows = Cin, k = (jy, jx, co).
__device__ __forceinline__ bf16_t pack_one(const float* __restrict__ w, int Cout, int Cin, int KH,
                                           int KW, int cin_pad, int cout_pad, int mode, int r, int k) {
  float v = 0.f;
  if (mode >= 4) {
    const int py = (mode - 4) >> 1, px = (mode - 4) & 1;
    const int kw_n = 1 + px, ntap = (1 + py) * kw_n;
    const int tap = k / cout_pad, co = k - tap * cout_pad;
    if (tap < ntap && co < Cout && r < Cin) {
      const int jy = tap / kw_n, jx = tap - jy * kw_n;
      v = w[(((long)co * Cin + r) * KH + s2_tap(py, jy)) * KW + s2_tap(px, jx)];
    }
  } else if (mode == 0) {
    const int tap = k / cin_pad, ci = k - tap * cin_pad;
    if (tap < KH * KW && ci < Cin && r < Cout) {
      const int kh = tap / KW, kw = tap - kh * KW;
      v = w[(((long)r * Cin + ci) * KH + kh) * KW + kw];
    }
  } else {
    const int tap = k / cout_pad, co = k - tap * cout_pad;
    if (tap < KH * KW && co < Cout && r < Cin) {
      const int khp = tap / KW, kwp = tap - khp * KW;
      const int kh = KH - 1 - khp, kw = KW - 1 - kwp;
      v = w[(((long)co * Cin + r) * KH + kh) * KW + kw];
    }
  }
  return f2bf(v);
}

// Element offset of (row r, k = tap * cpad + c) in the MFMA-fragment orders of mode 2 / 3.
// The one fragment order (conv_tile.hip, conv_tile_p.hip, conv_halo_gemm.hip): [n-block of 32][k-step of 16][lane][8] with
//   row = nb*32 + (lane&31), k = ks*16 + 8*(lane>>5) + j;  ksteps = Kpad / 16.
__device__ __forceinline__ long frag_offset(int /*layout*/, int r, int k, int ksteps, int /*cpad*/) {
  return ((((long)(r >> 5) * ksteps + (k >> 4)) * 64) + (r & 31) + 32 * ((k & 15) >> 3)) * 8 + (k & 7);
}

// All filters of the network in one launch: blockIdx.y = job, grid-stride in x.
// flat output index -> (row, k) of the GEMM operand.  mode 0/1: row-major
// [rows][Kpad].  mode 2/3: the inverse of frag_offset (k beyond the real taps: a zero element).
__device__ __forceinline__ void pack_index(long i, int Kpad, int mode, int layout, int cpad, int* r, int* k) {
  if (mode < 2 || mode >= 4) {
    *r = (int)(i / Kpad);
    *k = (int)(i - (long)*r * Kpad);
  } else {
    const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
    const long blk = i >> 9;
    const int ksteps = Kpad >> 4;
    const int ks = (int)(blk % ksteps), nb = (int)(blk / ksteps);
    *r = nb * 32 + (l & 31);
    *k = ks * 16 + 8 * (l >> 5) + j;
  }
}

// All filters of the network in one launch: blockIdx.y = job, blockIdx.x strides over the
// operand's rows.  A row's source elements are first copied into LDS in SOURCE order
// (mode 0/2: the Cin*KH*KW contiguous floats of output channel r; mode 1/3: the KH*KW
// runs of input channel r, one per output channel) so that the OIHW tensor is read with
// unit stride instead of a stride of KH*KW floats per consecutive k; the permuted,
// bf16-rounded row is then written from LDS.  Rows that do not fit (48 KiB) fall back to
// the element-wise gather.
__global__ __launch_bounds__(256) void pack_filters_batched_kernel(const PackJob* __restrict__ jobs) {
  SSA_DYN_LDS(float, rowbuf);
  const PackJob j = jobs[blockIdx.y];
  if (j.mode >= 4) {                                       // parity-class operands: element-wise
    for (int r = blockIdx.x; r < j.rows; r += gridDim.x)
      for (int k = threadIdx.x; k < j.Kpad; k += 256)
        j.out[(long)r * j.Kpad + k] = pack_one(j.w, j.Cout, j.Cin, j.KH, j.KW, j.cin_pad, j.cout_pad, j.mode, r, k);
    return;
  }
  const int taps = j.KH * j.KW;
  const int transposed = j.mode & 1;
  const int rows_real = transposed ? j.Cin : j.Cout;      // rows that carry data
  const int csrc = transposed ? j.Cout : j.Cin;            // channels along k in the source
  const int cpad = transposed ? j.cout_pad : j.cin_pad;
  const bool fits = (long)csrc * taps * sizeof(float) <= 48 * 1024;
  const int ksteps = j.Kpad >> 4;
  for (int r = blockIdx.x; r < j.rows; r += gridDim.x) {
    if (fits && r < rows_real) {
      __syncthreads();
      if (!transposed) {
        const float* src = j.w + (long)r * csrc * taps;
        for (int i = threadIdx.x; i < csrc * taps; i += 256) rowbuf[i] = src[i];
      } else {
        for (int i = threadIdx.x; i < csrc * taps; i += 256) {
          const int co = i / taps, t = i - co * taps;
          rowbuf[i] = j.w[((long)co * j.Cin + r) * taps + t];
        }
      }
      __syncthreads();
    }
    for (int k = threadIdx.x; k < j.Kpad; k += 256) {
      float v = 0.f;
      if (r < rows_real) {
        const int tap = k / cpad, c = k - tap * cpad;
        if (tap < taps && c < csrc) {
          const int tsrc = transposed ? (taps - 1 - tap) : tap;     // flipped taps for the data gradient
          v = fits ? rowbuf[c * taps + tsrc]
                   : (transposed ? j.w[((long)c * j.Cin + r) * taps + tsrc] : j.w[((long)r * j.Cin + c) * taps + tsrc]);
        }
      }
      const long o = j.mode < 2 ? (long)r * j.Kpad + k : frag_offset(j.layout, r, k, ksteps, cpad);
      j.out[o] = f2bf(v);
    }
  }
}

// Tiled form of the batched repack: one workgroup per (job, 32 output channels x CT input channels)
// tile of the OIHW tensor.  The tile is read with unit stride (32 runs of CT*taps floats) into LDS
// for BOTH operand forms -- the transposed (data-gradient) form used to gather 36-byte runs at a
// stride of Cin*taps floats --, permuted there, and written as 16-byte pieces of 8 consecutive k
// (fragment-major forms: lanes run over the operand row, i.e. whole 1-KiB fragment blocks; row-major
// forms: lanes run along k).  Work is balanced by tile, not by job (the old grid gave the 3.3 M-element
// head filters and the 20 K-element branch filters 32 workgroups each).  Zero padding (rows beyond the
// real ones, k beyond Kdim) is never written here: the destination is cleared when it is allocated and
// packed once in full by ssa_pack_filter.
__global__ __launch_bounds__(256) void pack_filters_tiled_kernel(const PackJob* __restrict__ jobs,
                                                                 const int4* __restrict__ tiles) {
  SSA_DYN_LDS(float, tbuf);
  const int4 t = tiles[blockIdx.x];
  const PackJob j0 = jobs[t.x];
  const PackJob& j = j0;
  const int taps = j.KH * j.KW;
  const int co0 = t.y, ci0 = t.z, CT = t.w;
  const int nco = min(32, j.Cout - co0), nci = min(CT, j.Cin - ci0);
  const int run = nci * taps, rowlen = (CT * taps) | 1;
  if ((run & 3) == 0 && ((j.Cin * taps) & 3) == 0 && ((ci0 * taps) & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(j.w) & 15u) == 0) {
    // 16-byte loads: every run (row of the tile) starts on a 16-byte boundary
    const int run4 = run >> 2;
    for (int idx = threadIdx.x; idx < nco * run4; idx += 256) {
      const int row = idx / run4, i = (idx - row * run4) * 4;
      const float4 v = *reinterpret_cast<const float4*>(j.w + ((long)(co0 + row) * j.Cin + ci0) * taps + i);
      float* d = tbuf + row * rowlen + i;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int idx = threadIdx.x; idx < nco * run; idx += 256) {
      const int row = idx / run, i = idx - row * run;
      tbuf[row * rowlen + i] = j.w[((long)(co0 + row) * j.Cin + ci0) * taps + i];
    }
  }
  __syncthreads();
  // every operand form of this parameter from the one staged tile: forward, data gradient, the four parity classes of
  // a stride-2 data gradient are jobs chained through elem_begin (a job per form used to fetch the tile again: 614 MB
  // read for 288 MB of parameters)
  for (PackJob j = jobs[t.x];; j = jobs[j.elem_begin - 1]) {
  const int cls = j.mode >= 4 ? j.mode - 4 : -1;          // parity class (py, px) of a stride-2 data gradient
  const int transposed = (j.mode & 1) || cls >= 0;
  const int cpy = cls >> 1, cpx = cls & 1, ckw = 1 + cpx;
  const int otaps = cls >= 0 ? (1 + cpy) * ckw : taps;     // taps of the OPERAND
  // operand row r and the channel c that runs along k inside a tap
  const int nr = transposed ? nci : nco, nc = transposed ? nco : nci;
  const int r0 = transposed ? ci0 : co0, c0 = transposed ? co0 : ci0;
  const int cpad = transposed ? j.cout_pad : j.cin_pad;
  const int ng = (nc + 7) >> 3;                          // 8-channel groups of this tile (c0 is a multiple of 8)
  const int ksteps = j.Kpad >> 4;
  const int items = otaps * ng * nr;
  const bool frag = j.mode == 2 || j.mode == 3;
  for (int idx = threadIdx.x; idx < items; idx += 256) {
    int rl, g, tap;
    if (frag) { rl = idx % nr; const int q = idx / nr; g = q % ng; tap = q / ng; }
    else { g = idx % ng; const int q = idx / ng; tap = q % otaps; rl = q / otaps; }
    int tsrc = transposed ? (taps - 1 - tap) : tap;                  // flipped taps for the data gradient
    if (cls >= 0) tsrc = s2_tap(cpy, tap / ckw) * j.KW + s2_tap(cpx, tap % ckw);
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cl = g * 8 + e;
      float f = 0.f;
      if (cl < nc) f = transposed ? tbuf[cl * rowlen + rl * taps + tsrc] : tbuf[rl * rowlen + cl * taps + tsrc];
      v[e] = f2bf(f);
    }
    const int r = r0 + rl, k = tap * cpad + c0 + g * 8;
    const long o = !frag ? (long)r * j.Kpad + k : frag_offset(j.layout, r, k, ksteps, cpad);
    uint4 pk;
    pk.x = v[0] | ((unsigned)v[1] << 16); pk.y = v[2] | ((unsigned)v[3] << 16);
    pk.z = v[4] | ((unsigned)v[5] << 16); pk.w = v[6] | ((unsigned)v[7] << 16);
    *reinterpret_cast<uint4*>(j.out + o) = pk;
  }
  if (j.elem_begin <= 0) break;
  }
}

__global__ void pack_filter_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout,
                                   int Cin, int KH, int KW, int cin_pad, int cout_pad, int Kpad,
                                   int mode, int rows, int layout) {
  const long n = (long)rows * Kpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r, k;
    pack_index(i, Kpad, mode, layout, (mode & 1) ? cout_pad : cin_pad, &r, &k);
    out[i] = pack_one(w, Cout, Cin, KH, KW, cin_pad, cout_pad, mode >= 4 ? mode : (mode & 1), r, k);
  }
}

__global__ void pad_cast_kernel(const float* __restrict__ x, long P, int C, int ldx,
                                bf16_t* __restrict__ y, int Cpad) {
  const long n = P * Cpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long p = i / Cpad;
    const int c = (int)(i - p * Cpad);
    y[i] = f2bf(c < C ? x[p * ldx + c] : 0.f);
  }
}

// ---------------------------------------------------------------- dispatch
struct TileChoice { int id, bm, bn; float eff; };
// id: 0 = 128x128 (2x2 waves, 2x2 tiles)   1 = 256x64 (4x1, 2x2)
//     2 = 128x96  (4x1, 1x3)               3 = 256x32 (4x1, 2x1)
//     4 = 64x64   (2x2, 1x1)               5 = 128x64 (2x2, 2x1)
constexpr TileChoice kFwdTiles[] = {
    {0, 128, 128, 1.00f}, {1, 256, 64, 0.90f}, {2, 128, 96, 0.85f},
    {3, 256, 32, 0.55f},  {4, 64, 64, 0.50f},  {5, 128, 64, 0.75f}};

int choose_fwd_tile(long M, int N) {
  // (Measured and rejected, profiles/r02_notes.md: inside group brackets, picking the largest tile that N
  // fills instead of this cost model -- the stride-2 / 1x1 fuse convs got slower, 3.0 -> 4.2 ms per step.)
  int best = 0;
  double best_cost = 1e300;
  for (const TileChoice& t : kFwdTiles) {
    const long tiles = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
    // work is issued in rounds of ~2 workgroups per CU on 256 CUs
    const double rounds = (double)((tiles + 511) / 512);
    const double full = (double)tiles / 512.0;
    const double cost = (0.5 * rounds + 0.5 * full) * t.bm * t.bn / t.eff;
    if (cost < best_cost) { best_cost = cost; best = t.id; }
  }
  return best;
}

struct OutMap { int mul, py, px, W, HW; };
struct IgemmAffine { const float* aff; const bf16_t* res; int ldres, relu; };
static thread_local IgemmAffine g_affine = {nullptr, nullptr, 0, 0};      // set around one launch by ssa_conv2d_igemm_affine

template <int WGM, int WGN, int MI, int NI>
int launch_fwd(const ssa_conv_desc& d, const void* x, const void* w, const float* bias, void* y,
               hipStream_t s, int tr_shift, double* stats = nullptr, const OutMap* om = nullptr) {
  constexpr int BM = WGM * MI * 32, BN = WGN * NI * 32;
  const long M = (long)d.B * d.Ho * d.Wo;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (d.Cout + BN - 1) / BN;
  size_t lds = (size_t)2 * (BM + BN) * (IgemmBK<MI, NI>::value + 8) * 2;
  const size_t cs = (size_t)BM * (BN + 8) * 2 + (size_t)WGM * 2 * BN * sizeof(float);
  if (cs > lds) lds = cs;
  IgemmArgs a;
  a.d = d; a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = y; a.stats = stats;
  a.tiles_n = tiles_n; a.tr_shift = tr_shift;
  a.o_mul = om ? om->mul : 0; a.o_py = om ? om->py : 0; a.o_px = om ? om->px : 0;
  a.o_W = om ? om->W : 0; a.o_HW = om ? om->HW : 0;
  a.aff = g_affine.aff; a.res = g_affine.res; a.ldres = g_affine.ldres; a.relu = g_affine.relu;
  return ssa::submit<ConvIgemm<WGM, WGN, MI, NI>>(a, tiles_m * tiles_n, 1, lds, s);
}

template <int WGM, int WGN, int MI, int NI>
int launch_wgrad(const ssa_conv_desc& d, const void* x, const void* dy, int lddy, int cout_pad,
                 int nsplit, float* partial, hipStream_t s) {
  constexpr int BM = WGM * MI * 32, BN = WGN * NI * 32;
  const int Kflat = d.KH * d.KW * d.Cin;
  const int tiles_m = (cout_pad + BM - 1) / BM, tiles_n = (Kflat + BN - 1) / BN;
  const long P = (long)d.B * d.Ho * d.Wo;
  long chunk = (P + nsplit - 1) / nsplit;
  // 32 pixels per LDS stage.  Measured and rejected on MI355X: 64-pixel
  // stages (80 KB of LDS -> one workgroup per CU: 720->512 wgrad 2.22 -> 3.20 ms) and a second
  // register set for two-stage-ahead prefetch (96@128^2: 36 -> 52 us).
  chunk = (chunk + 31) / 32 * 32;
  const size_t lds = (size_t)2 * 32 * (tr_row_stride(BM) + tr_row_stride(BN)) * 2;
  WgradTrArgs a;
  a.d = d; a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.partial = partial;
  a.lddy = lddy; a.cout_pad = cout_pad; a.tiles_n = tiles_n; a.chunk = (int)chunk;
  return ssa::submit<ConvWgradTr<WGM, WGN, MI, NI, 32>>(a, tiles_m * tiles_n, nsplit, lds, s);
}

// wgrad tile ids: 0 = 128x128, 1 = 64x64, 2 = 32x128, 3 = 96x128
int choose_wgrad_tile(int cout_pad) {
  if (cout_pad <= 32) return 2;
  if (cout_pad <= 64) return 1;
  if (cout_pad % 96 == 0 && cout_pad % 128 != 0) return 3;
  return 0;
}
void wgrad_tile_dims(int id, int* bm, int* bn) {
  switch (id) {
    case 1: *bm = 64; *bn = 64; break;
    case 2: *bm = 32; *bn = 128; break;
    case 3: *bm = 96; *bn = 128; break;
    default: *bm = 128; *bn = 128; break;
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

int ssa_version(void) { return 1; }

int ssa_conv2d_igemm_stats(const ssa_conv_desc* dp, const void* x, const void* w_packed,
                           const float* bias, void* y, double* stats, void* stream) {
  if (!dp || !x || !w_packed || !y) return SSA_EINVAL;
  if (stats && (dp->out_f32 || dp->transposed)) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  if (d.Cin % 8 || d.ldx % 8 || d.Kpad % BK || d.Kpad < d.KH * d.KW * d.Cin) return SSA_EINVAL;
  if (!aligned16(x) || !aligned16(w_packed)) return SSA_EINVAL;
  if (!d.out_f32 && (d.ldy % 8 || !aligned16(y))) return SSA_EINVAL;
  if (d.B <= 0 || d.Ho <= 0 || d.Wo <= 0 || d.Cout <= 0) return SSA_EINVAL;
  if ((long)d.B * d.H * d.W >= INT_MAX / 2 || (long)d.B * d.Ho * d.Wo >= INT_MAX / 2) return SSA_EUNSUPPORTED;
  int tr_shift = 0;
  if (d.transposed) {
    while ((1 << tr_shift) < d.stride) ++tr_shift;
    if ((1 << tr_shift) != d.stride) return SSA_EUNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  const long M = (long)d.B * d.Ho * d.Wo;
  const int cfg = d.cfg >= 0 ? d.cfg : choose_fwd_tile(M, d.Cout);
  switch (cfg) {
    case 0: return launch_fwd<2, 2, 2, 2>(d, x, w_packed, bias, y, s, tr_shift, stats);
    case 1: return launch_fwd<4, 1, 2, 2>(d, x, w_packed, bias, y, s, tr_shift, stats);
    case 2: return launch_fwd<4, 1, 1, 3>(d, x, w_packed, bias, y, s, tr_shift, stats);
    case 3: return launch_fwd<4, 1, 2, 1>(d, x, w_packed, bias, y, s, tr_shift, stats);
    case 4: return launch_fwd<2, 2, 1, 1>(d, x, w_packed, bias, y, s, tr_shift, stats);
    case 5: return launch_fwd<2, 2, 2, 1>(d, x, w_packed, bias, y, s, tr_shift, stats);
    default: return SSA_EINVAL;
  }
}

int ssa_conv2d_igemm_affine(const ssa_conv_desc* dp, const void* x, const void* w_packed, const float* bias, void* y,
                            const float* coef, const void* residual, int ldres, int relu, void* stream) {
  if (!dp || !coef || dp->out_f32 || dp->transposed) return SSA_EINVAL;
  if (residual && (ldres % 8 || !aligned16(residual))) return SSA_EINVAL;
  g_affine = IgemmAffine{coef, (const bf16_t*)residual, ldres, relu ? 1 : 0};
  const int rc = ssa_conv2d_igemm_stats(dp, x, w_packed, bias, y, nullptr, stream);
  g_affine = IgemmAffine{nullptr, nullptr, 0, 0};
  return rc;
}

int ssa_conv2d_dgrad_s2(int B, int H, int W, int Cin, int lddx, int Ho, int Wo, int cout_pad, int lddy,
                        const void* dy, const void* const* w_cls, const int* kpad_cls, void* dx,
                        void* stream) {
  if (!dy || !w_cls || !kpad_cls || !dx) return SSA_EINVAL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 8 || lddx % 8 || cout_pad % 8 || lddy % 8) return SSA_EINVAL;
  if (Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return SSA_EINVAL;      // 3x3, stride 2, pad 1
  if (!aligned16(dy) || !aligned16(dx)) return SSA_EINVAL;
  if ((long)B * H * W >= INT_MAX / 2) return SSA_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    const int Hc = (H - py + 1) / 2, Wc = (W - px + 1) / 2;      // pixels of dx with this parity
    if (Hc <= 0 || Wc <= 0) continue;
    if (!w_cls[cls] || !aligned16(w_cls[cls])) return SSA_EINVAL;
    ssa_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.H = Ho; d.W = Wo; d.Cin = cout_pad; d.ldx = lddy;
    d.Ho = Hc; d.Wo = Wc; d.Cout = Cin; d.ldy = lddx;
    d.KH = 1 + py; d.KW = 1 + px; d.stride = 1; d.pad = 0; d.dil = 1; d.transposed = 0;
    d.Kpad = kpad_cls[cls]; d.out_f32 = 0; d.cfg = -1;
    if (d.Kpad % BK || d.Kpad < d.KH * d.KW * d.Cin) return SSA_EINVAL;
    const OutMap om{2, py, px, W, H * W};
    int rc;
    switch (choose_fwd_tile((long)B * Hc * Wc, Cin)) {
      case 0: rc = launch_fwd<2, 2, 2, 2>(d, dy, w_cls[cls], nullptr, dx, s, 0, nullptr, &om); break;
      case 1: rc = launch_fwd<4, 1, 2, 2>(d, dy, w_cls[cls], nullptr, dx, s, 0, nullptr, &om); break;
      case 2: rc = launch_fwd<4, 1, 1, 3>(d, dy, w_cls[cls], nullptr, dx, s, 0, nullptr, &om); break;
      case 3: rc = launch_fwd<4, 1, 2, 1>(d, dy, w_cls[cls], nullptr, dx, s, 0, nullptr, &om); break;
      case 4: rc = launch_fwd<2, 2, 1, 1>(d, dy, w_cls[cls], nullptr, dx, s, 0, nullptr, &om); break;
      default: rc = launch_fwd<2, 2, 2, 1>(d, dy, w_cls[cls], nullptr, dx, s, 0, nullptr, &om); break;
    }
    if (rc) return rc;
  }
  return SSA_OK;
}

int ssa_conv2d_igemm(const ssa_conv_desc* dp, const void* x, const void* w_packed,
                     const float* bias, void* y, void* stream) {
  return ssa_conv2d_igemm_stats(dp, x, w_packed, bias, y, nullptr, stream);
}

int ssa_conv2d_igemm_tile(const ssa_conv_desc* dp) {
  if (!dp) return SSA_EINVAL;
  if (dp->cfg >= 0) return dp->cfg;
  return choose_fwd_tile((long)dp->B * dp->Ho * dp->Wo, dp->Cout);
}

int ssa_pack_filter(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW,
                    int cin_pad, int cout_pad, int Kpad, int mode, void* stream) {
  if (!w_oihw || !w_packed || mode < 0) return SSA_EINVAL;
  const int layout = 0;
  if (mode > 7) return SSA_EINVAL;
  if (mode >= 4 && (KH != 3 || KW != 3)) return SSA_EINVAL;
  int rows = (mode & 1) == 0 && mode < 4 ? Cout : Cin;
  long kneed = (long)KH * KW * ((mode & 1) == 0 ? cin_pad : cout_pad);
  if (mode >= 4) kneed = (long)(1 + ((mode - 4) >> 1)) * (1 + ((mode - 4) & 1)) * cout_pad;
  if (mode < 2 || mode >= 4) {
    if (Kpad % BK || Kpad < kneed) return SSA_EINVAL;
  } else {
    if (Kpad != kneed || Kpad % 16) return SSA_EINVAL;
    rows = (rows + 31) / 32 * 32;
  }
  const long n = (long)rows * Kpad;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(pack_filter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (bf16_t*)w_packed, Cout, Cin, KH, KW, cin_pad, cout_pad, Kpad, mode, rows, layout);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_conv2d_wgrad_plan(const ssa_conv_desc* d, int cout_pad, int* nsplit, size_t* ws_bytes) {
  if (!d || !nsplit || !ws_bytes) return SSA_EINVAL;
  int bm, bn;
  wgrad_tile_dims(choose_wgrad_tile(cout_pad), &bm, &bn);
  const int Kflat = d->KH * d->KW * d->Cin;
  const long tiles = (long)((cout_pad + bm - 1) / bm) * ((Kflat + bn - 1) / bn);
  const long P = (long)d->B * d->Ho * d->Wo;
  static const long target_wgs = getenv("SSA_WGRAD_WGS") ? atol(getenv("SSA_WGRAD_WGS")) : 640;
  long ns = (target_wgs + tiles - 1) / tiles;
  // d->cfg > 0: 128-pixel stages per split asked for by the caller (grouped launches, see
  // ssa_conv2d_wgrad_tile_plan)
  if (d->cfg > 0) ns = (P + 128L * d->cfg - 1) / (128L * d->cfg);
  const long max_by_pixels = (P + 127) / 128;  // at least 4 stages per split
  if (ns > max_by_pixels) ns = max_by_pixels;
  if (ns < 1) ns = 1;
  if (ns > 512) ns = 512;
  *nsplit = (int)ns;
  *ws_bytes = (size_t)ns * cout_pad * Kflat * sizeof(float);
  return SSA_OK;
}

int ssa_conv2d_wgrad(const ssa_conv_desc* dp, const void* x, const void* dy, int lddy,
                     int cout_pad, int nsplit, float* partial, void* stream) {
  if (!dp || !x || !dy || !partial || nsplit < 1) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  if (d.transposed) return SSA_EUNSUPPORTED;
  if (d.Cin % 8 || d.ldx % 8 || cout_pad % 8 || lddy % 8) return SSA_EINVAL;
  if (!aligned16(x) || !aligned16(dy)) return SSA_EINVAL;
  if ((long)d.B * d.H * d.W >= INT_MAX / 2 || (long)d.B * d.Ho * d.Wo >= INT_MAX / 2) return SSA_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  switch (choose_wgrad_tile(cout_pad)) {
    case 1: return launch_wgrad<2, 2, 1, 1>(d, x, dy, lddy, cout_pad, nsplit, partial, s);
    case 2: return launch_wgrad<1, 4, 1, 1>(d, x, dy, lddy, cout_pad, nsplit, partial, s);
    case 3: return launch_wgrad<1, 4, 3, 1>(d, x, dy, lddy, cout_pad, nsplit, partial, s);
    default: return launch_wgrad<2, 2, 2, 2>(d, x, dy, lddy, cout_pad, nsplit, partial, s);
  }
}

int ssa_conv2d_wgrad_reduce(const float* partial, int nsplit, int cout_pad, int Cout, int Cin_pad,
                            int Cin, int KH, int KW, float* dw_oihw, int accumulate, void* stream) {
  if (!partial || !dw_oihw || Cout > cout_pad || Cin > Cin_pad || nsplit < 1) return SSA_EINVAL;
  const size_t lds = (size_t)Cin * KH * KW * sizeof(float);
  if (lds > 160 * 1024) return SSA_EUNSUPPORTED;
  const long kflat = (long)KH * KW * Cin_pad;
  const int vec4 = kflat % 4 == 0 && (reinterpret_cast<uintptr_t>(partial) & 15u) == 0;
  WgradReduceK::Args a{partial, dw_oihw, nsplit, cout_pad, Cout, Cin_pad, Cin, KH, KW, accumulate, vec4};
  return ssa::submit<WgradReduceK>(a, Cout, 1, lds, (hipStream_t)stream);
}

int ssa_pack_filters_batched(const void* jobs_dev, int njobs, int blocks_per_job, void* stream) {
  if (!jobs_dev || njobs < 1 || blocks_per_job < 1) return SSA_EINVAL;
  static_assert(sizeof(PackJob) == 64, "ssa_pack_job layout");
  hipLaunchKernelGGL(pack_filters_batched_kernel, dim3(blocks_per_job, njobs), dim3(256), 48 * 1024,
                     (hipStream_t)stream, (const PackJob*)jobs_dev);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_pack_tile_channels(int KH, int KW) {
  const int taps = KH * KW;
  if (taps < 1) return 0;
  int ct = (288 / taps) & ~7;
  if (ct > 256) ct = 256;
  if (ct < 8) ct = 8;
  return ((size_t)32 * ((ct * taps) | 1) * sizeof(float) <= 64 * 1024) ? ct : 0;
}

int ssa_pack_filters_tiled(const void* jobs_dev, const void* tiles_dev, int ntiles, int max_ct_taps,
                           void* stream) {
  if (!jobs_dev || !tiles_dev || ntiles < 1 || max_ct_taps < 1) return SSA_EINVAL;
  const size_t lds = (size_t)32 * (max_ct_taps | 1) * sizeof(float);     // the largest ct * KH * KW of the tiles
  if (lds > 64 * 1024) return SSA_EINVAL;
  hipLaunchKernelGGL(pack_filters_tiled_kernel, dim3(ntiles), dim3(256), lds, (hipStream_t)stream,
                     (const PackJob*)jobs_dev, (const int4*)tiles_dev);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

int ssa_pad_cast_f32_bf16(const float* x, long P, int C, int ldx, void* y, int Cpad, void* stream) {
  if (!x || !y || Cpad < C) return SSA_EINVAL;
  const long n = P * Cpad;
  const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  hipLaunchKernelGGL(pad_cast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, P, C, ldx,
                     (bf16_t*)y, Cpad);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}

}  // extern "C"
