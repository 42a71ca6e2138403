// Halo-tile convolution for the small-channel HRNet branches (gfx950 / MI355X).
//
// The 48/64/96/192/384-channel 3x3 convs of the HRNetV2 trunk (network/hrnetv2.py:31-66,
// SURVEY.md K1; 24 % of the step's FLOPs, 1,046 launches per training step) are
// HBM/latency bound: K = 9*Cin is only 432..864, so a K-pipelined implicit GEMM
// spends its time in 14..27 dependent global->LDS stages.  Here one workgroup
//   * loads the (TH+2)x(TW+2) input halo tile ONCE (one burst of 16-byte loads,
//     every input byte fetched from L2/HBM once per tile instead of 9 times),
//   * runs all 9 taps x Cin straight out of LDS: the MFMA A fragment of tap
//     (kh,kw) is the same LDS image read at a shifted pixel offset, no im2col,
//   * takes the B operand (filter) from a copy packed in MFMA-fragment order
//     ([n-block][k-step][lane][8], ssa_pack_filter mode 2/3): it is DMA'd into
//     LDS with global_load_lds_dwordx4 (one 1-KiB fragment block per wave
//     instruction, no VGPRs, issued together with the halo loads), whole for
//     Cin <= 64, in 3-tap chunks double-buffered against the MFMAs for Cin = 96,
//   * has one barrier per filter chunk (1 or 3 in the whole kernel),
//   * optionally accumulates the BatchNorm batch statistics (sum, sum of squares
//     of the bf16-rounded outputs) in the epilogue -> no separate stats pass.
// The same kernel computes the data gradient (flipped, transposed filter).
//
// LDS image: [halo pixel][Cin] bf16 with pixel stride Cin*2+16 bytes: an odd
// number of 16-byte slots, so the 16 lanes ds_read_b128 services together
// (consecutive pixels, same channel offset) fall on 16 distinct slots.
// Data-gradient variants with a fused epilogue tile (ssa_conv2d_tile_aux): a second tile
// `aux`, congruent with the output tile, is staged through LDS in the epilogue:
//   aux_mode 1: added to the output -- the residual branch's gradient, dX = dgrad + dres,
//               rounded as the unfused bf16 add rounds (bf16(bf16(dgrad) + dres));
//   aux_mode 2: taken as the INPUT x of the BatchNorm+ReLU layer whose output this conv
//               consumed (network/hrnetv2.py:44-64: conv1 -> bn1 -> relu -> conv2): the epilogue
//               accumulates that layer's backward sums  sum(m*dz)  and  sum(m*dz*xhat),
//               m = [scale*x+shift > 0], xhat = (x-mean)*invstd, over the bf16-rounded dz it
//               stores (coef = [4][Cout]: scale, shift, mean, invstd) into `stats`
//               ([replica][2][Cout] fp64) -- what bn_bwd_reduce_kernel (bn.hip) computes in a
//               pass of its own over (x, dz).
//
// Group-aware (group.h): inside an ssa_group_begin/ssa_group_end bracket the launches of the
// independent problems of one depth level (branches x scale passes) that share a kernel
// instantiation become one launch.
#include "common.h"
#include "group.h"
#include <stdlib.h>
#include "../../include/semseg_hip.h"

namespace {

constexpr int kStatReplicas = 8;   // BN partial sums are spread over 8 replicas (atomic contention)

// Filter stage (channel chunk cc, taps [tap0, tap0+TPC)) of the NB n-blocks -> LDS
// buffer, as direct global->LDS DMA: one global_load_lds_dwordx4 per wave moves one
// 1-KiB (n-block, k-step) fragment block; the LDS image is lane-linear, which is
// exactly the order ds_read_b128 wants the B fragment in.
template <int NB, int TPC, int CST>
__device__ __forceinline__ void stage_filter_chunk(const uint4* __restrict__ wfrag, int nb0,
                                                   int nb_total, int ksteps_total, int csteps_total,
                                                   int cc, int tap0, unsigned char* dst, int wave,
                                                   int lane) {
  constexpr int PER_NB = TPC * CST;
  constexpr int NFRAG = NB * PER_NB;
#pragma unroll
  for (int f = 0; f < (NFRAG + 3) / 4; ++f) {
    const int fi = f * 4 + wave;               // wave-uniform
    if (fi < NFRAG) {
      const int nb = fi / PER_NB, rem = fi - nb * PER_NB;
      const int tl = rem / CST, j = rem - tl * CST;
      const int nbg = min(nb0 + nb, nb_total - 1);   // n-blocks past the end re-read the last one (never stored)
      const uint4* src = wfrag + ((long)nbg * ksteps_total + (tap0 + tl) * csteps_total + cc * CST + j) * 64 + lane;
      ssa_glds16(src, dst + (size_t)fi * 1024);
    }
  }
}

// CK: input channels staged per halo image (= Cin for 48/64/96; 192 for Cin = 192/384,
// which run Cin/CK passes over the taps), TPC: taps per filter stage.
struct TileArgs {
  const bf16_t* x; const uint4* wfrag; const float* bias; bf16_t* y; double* stats;
  const bf16_t* aux; const float* coef;
  int ldx, Cin, ldy, B, H, W, Cout, nb_total, tiles_x, tiles_y, ldaux, aux_mode;
};

template <int CK, int KS, int NB, int MI, int TW, int TPC, bool AUX>
struct ConvTile {
  typedef TileArgs Args;
  static constexpr int NT = 256;
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const bf16_t* __restrict__ x = a.x;
  const uint4* __restrict__ wfrag = a.wfrag;
  const float* __restrict__ bias = a.bias;
  bf16_t* __restrict__ y = a.y;
  double* __restrict__ stats = a.stats;
  const bf16_t* __restrict__ aux = a.aux;
  const float* __restrict__ coef = a.coef;
  const int ldx = a.ldx, Cin = a.Cin, ldy = a.ldy, H = a.H, W = a.W, Cout = a.Cout, nb_total = a.nb_total;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, ldaux = a.ldaux, aux_mode = a.aux_mode;
  constexpr int R = KS / 2;
  constexpr int BM = 4 * MI * 32;              // output pixels per workgroup
  constexpr int TH = BM / TW;                  // tile rows
  constexpr int HW_ = TW + 2 * R, HH_ = TH + 2 * R;
  constexpr int PSB = CK * 2 + 16;             // halo pixel stride in bytes
  constexpr int CP = CK / 8;                   // 16-byte pieces per pixel
  constexpr int NPIECE = HH_ * HW_ * CP;
  constexpr int CST = CK / 16;                 // c-steps per halo chunk
  constexpr int TAPS = KS * KS;
  constexpr int NSTAGE = TAPS / TPC;           // filter stages per channel chunk
  static_assert(NSTAGE * TPC == TAPS, "taps per stage must divide the tap count");
  constexpr int STAGE_KS = TPC * CST;          // k-steps per filter stage
  constexpr int STAGE_BYTES = NB * STAGE_KS * 1024;
  constexpr int NBUF = NSTAGE > 1 ? 2 : 1;
  constexpr int HALO_BYTES = (HH_ * HW_ * PSB + 1023) / 1024 * 1024;
  SSA_DYN_LDS(unsigned char, smem);
  unsigned char* Bs = smem + HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = bx;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int nb0 = by * NB;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int nchunk = Cin / CK;
  const int csteps_total = Cin / 16;
  const int ksteps_total = TAPS * csteps_total;

  // ---- per-lane A addressing: MFMA row block mi of this wave -> tile pixels
  int a_off[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = (wave * MI + mi) * 32 + (lane & 31);
    const int ty = m / TW, tx = m - ty * TW;
    a_off[mi] = (ty * HW_ + tx) * PSB + (lane >> 5) * 16;
  }

  // fused-epilogue tile (data-gradient variants): its global loads are issued HERE, before the halo /
  // MFMA phase, and land in registers while that runs (after the MFMA loop they cost a full, exposed
  // memory latency: measured +4-5 us per launch)
  constexpr int AUX_IT = AUX ? (4 * MI * 32 * NB * 4 + 255) / 256 : 1;
  uint4 auxv[AUX_IT];
  if constexpr (AUX) {
    constexpr int CPR_ = NB * 4;
    const bf16_t* ab = aux + (long)b * H * W * ldaux;
#pragma unroll
    for (int i = 0; i < AUX_IT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR_, cp = idx - row * CPR_;
      const int ty = row / TW, tx = row - ty * TW;
      const int oy = y0 + ty, ox = x0 + tx, n = nb0 * 32 + cp * 8;
      auxv[i] = make_uint4(0, 0, 0, 0);
      if (idx < 4 * MI * 32 * CPR_ && oy < H && ox < W && n + 8 <= Cout)
        auxv[i] = *reinterpret_cast<const uint4*>(ab + ((long)oy * W + ox) * ldaux + n);
    }
  }

  f32x16_t acc[MI][NB];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nb][r] = 0.f;

  const bf16_t* xb = x + (long)b * H * W * ldx;
  // this thread's halo pieces: global element offset of channel chunk 0 (-1: outside the image / no piece)
  constexpr int IT = (NPIECE + 255) / 256;
  int g_off[IT];
  uint4 v[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int piece = tid + i * 256;
    const int pix = piece / CP, cp = piece - pix * CP;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    const int iy = y0 - R + hy, ix = x0 - R + hx;
    const bool ok = piece < NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    g_off[i] = ok ? (iy * W + ix) * ldx + cp * 8 : -1;
  }
  auto halo_gload = [&](int cc) {
#pragma unroll
    for (int i = 0; i < IT; ++i)
      v[i] = g_off[i] >= 0 ? *reinterpret_cast<const uint4*>(xb + g_off[i] + cc * CK) : make_uint4(0, 0, 0, 0);
  };
  halo_gload(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    if (cc > 0) __syncthreads();                // previous chunk's halo image and filter buffers are free
    // ---- filter stage 0 -> LDS (DMA), halo registers -> LDS
    stage_filter_chunk<NB, TPC, CST>(wfrag, nb0, nb_total, ksteps_total, csteps_total, cc, 0, Bs, wave, lane);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int piece = tid + i * 256;
      const int pix = piece / CP, cp = piece - pix * CP;
      if (piece < NPIECE) *reinterpret_cast<uint4*>(smem + pix * PSB + cp * 16) = v[i];
    }
    __syncthreads();                            // halo + filter stage 0 landed (vmcnt(0) + barrier)
    // the next channel chunk's halo: issued now, in flight during this chunk's MFMAs (192/384 channels =
    // 2/4 chunks; their load -> barrier -> MFMA chains are the longest workgroups of a grouped launch)
    if (cc + 1 < nchunk) halo_gload(cc + 1);

#pragma unroll
    for (int st = 0; st < NSTAGE; ++st) {
      if (st + 1 < NSTAGE)
        stage_filter_chunk<NB, TPC, CST>(wfrag, nb0, nb_total, ksteps_total, csteps_total, cc, (st + 1) * TPC,
                                         Bs + ((st + 1) % NBUF) * STAGE_BYTES, wave, lane);
      const unsigned char* Bc = Bs + (st % NBUF) * STAGE_BYTES + lane * 16;
#pragma unroll
      for (int ksl = 0; ksl < STAGE_KS; ++ksl) {
        const int tap = st * TPC + ksl / CST, cs = ksl % CST;
        const int kh = tap / KS, kw = tap - kh * KS;
        bf16x8_t af[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          af[mi] = *reinterpret_cast<const bf16x8_t*>(smem + a_off[mi] + (kh * HW_ + kw) * PSB + cs * 32);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const bf16x8_t bfr = *reinterpret_cast<const bf16x8_t*>(Bc + (nb * STAGE_KS + ksl) * 1024);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][nb] = ssa_mfma32(af[mi], bfr, acc[mi][nb]);
        }
      }
      if (st + 1 < NSTAGE) __syncthreads();     // next stage landed; this buffer is free for stage st+2
    }
  }

  // ---- epilogue: (+bias) -> bf16 -> LDS -> coalesced 16-byte stores; BN statistics
  __syncthreads();                              // everyone is done reading the halo image
  constexpr int LDC = NB * 32 + 8;              // staging row stride (elements)
  constexpr int CPR = NB * 4;                   // 16-byte pieces per staged row
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)BM * LDC * 2);   // [4 waves][2][NB*32]
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem + (size_t)BM * LDC * 2 + (size_t)4 * 2 * NB * 32 * sizeof(float));
  if constexpr (AUX) {
    // the aux tile (same pixels, same channels as the output tile; zero outside the image): registers -> LDS
#pragma unroll
    for (int i = 0; i < AUX_IT; ++i) {
      const int idx = tid + i * 256;
      if (idx < BM * CPR) *reinterpret_cast<uint4*>(Xs + (idx / CPR) * LDC + (idx % CPR) * 8) = auxv[i];
    }
    __syncthreads();
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = nb * 32 + (lane & 31);
    const int n = nb0 * 32 + col;
    const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
    float s = 0.f, q = 0.f;
    float ma = 0.f, mb = 0.f, mu = 0.f, is = 0.f;
    if constexpr (AUX) {
      if (aux_mode == 2 && n < Cout) { ma = coef[n]; mb = coef[Cout + n]; mu = coef[2 * Cout + n]; is = coef[3 * Cout + n]; }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wave * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        bf16_t o = f2bf(acc[mi][nb][r] + bv);
        if constexpr (AUX) {
          const float xv = bf2f(Xs[row * LDC + col]);
          if (aux_mode == 1) {
            o = f2bf(bf2f(o) + xv);
          } else {
            const int ty = row / TW, tx = row - ty * TW;
            const float gz = (y0 + ty < H && x0 + tx < W) ? bf2f(o) : 0.f;
            const float gm = (xv * ma + mb) > 0.f ? gz : 0.f;
            s += gm;
            q += gm * (xv - mu) * is;
          }
        }
        Cs[row * LDC + col] = o;
        if (!(AUX) && stats != nullptr) {
          const int ty = row / TW, tx = row - ty * TW;
          const float f = (y0 + ty < H && x0 + tx < W) ? bf2f(o) : 0.f;
          s += f;
          q += f * f;
        }
      }
    }
    if (stats != nullptr) {
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 32) {
        red[(wave * 2 + 0) * NB * 32 + col] = s;
        red[(wave * 2 + 1) * NB * 32 + col] = q;
      }
    }
  }
  __syncthreads();
  if (stats != nullptr) {
    // stats layout: [replica][2][C]; replica = workgroup index mod kStatReplicas
    double* st = stats + (long)(bx % kStatReplicas) * 2 * Cout;
    for (int i = tid; i < 2 * NB * 32; i += 256) {
      const int which = i / (NB * 32), col = i - which * NB * 32;
      const int n = nb0 * 32 + col;
      if (n < Cout) {
        const float v = (red[(0 * 2 + which) * NB * 32 + col] + red[(1 * 2 + which) * NB * 32 + col]) +
                        (red[(2 * 2 + which) * NB * 32 + col] + red[(3 * 2 + which) * NB * 32 + col]);
        atomicAdd(&st[which * Cout + n], (double)v);
      }
    }
  }
  bf16_t* yb = y + (long)b * H * W * ldy;
  for (int idx = tid; idx < BM * CPR; idx += 256) {
    const int row = idx / CPR, cp = idx - row * CPR;
    const int ty = row / TW, tx = row - ty * TW;
    const int oy = y0 + ty, ox = x0 + tx, n = nb0 * 32 + cp * 8;
    if (oy >= H || ox >= W || n >= Cout) continue;
    bf16_t* dst = yb + ((long)oy * W + ox) * ldy + n;
    const bf16_t* src = Cs + row * LDC + cp * 8;
    if (n + 8 <= Cout) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      for (int j = 0; n + j < Cout; ++j) dst[j] = src[j];
    }
  }
  }
};

struct AuxArgs {
  const void* aux;
  int ld;
  const float* coef;
  int mode;               // 0: none, 1: add, 2: BatchNorm backward sums
};

template <int CK, int KS, int NB, int MI, int TW, int TPC, bool AUX>
int launch_tile_v(const ssa_conv_desc& d, const void* x, const void* wfrag, const float* bias, void* y,
                  double* stats, hipStream_t s, const AuxArgs& ax) {
  constexpr int R = KS / 2, BM = 4 * MI * 32, TH = BM / TW;
  constexpr int NSTAGE = KS * KS / TPC;
  constexpr size_t halo = ((size_t)(TH + 2 * R) * (TW + 2 * R) * (CK * 2 + 16) + 1023) / 1024 * 1024;
  constexpr size_t filt = (size_t)(NSTAGE > 1 ? 2 : 1) * NB * TPC * (CK / 16) * 1024;
  // epilogue staging: output tile (+ aux tile) + reduction scratch
  constexpr size_t stage = (size_t)(AUX ? 2 : 1) * BM * (NB * 32 + 8) * 2 + 4 * 2 * NB * 32 * sizeof(float);
  constexpr size_t lds = halo + filt > stage ? halo + filt : stage;
  static_assert(lds <= 160 * 1024, "tile does not fit in LDS");
  TileArgs a;
  a.x = (const bf16_t*)x; a.wfrag = (const uint4*)wfrag; a.bias = bias; a.y = (bf16_t*)y; a.stats = stats;
  a.aux = (const bf16_t*)ax.aux; a.coef = ax.coef;
  a.ldx = d.ldx; a.Cin = d.Cin; a.ldy = d.ldy; a.B = d.B; a.H = d.H; a.W = d.W; a.Cout = d.Cout;
  a.nb_total = (d.Cout + 31) / 32;
  a.tiles_x = (d.W + TW - 1) / TW; a.tiles_y = (d.H + TH - 1) / TH;
  a.ldaux = ax.ld; a.aux_mode = ax.mode;
  return ssa::submit<ConvTile<CK, KS, NB, MI, TW, TPC, AUX>>(a, a.tiles_x * a.tiles_y * d.B,
                                                             (a.nb_total + NB - 1) / NB, lds, s);
}

template <int CK, int KS, int NB, int MI, int TW, int TPC>
int launch_tile(const ssa_conv_desc& d, const void* x, const void* wfrag, const float* bias, void* y,
                double* stats, hipStream_t s, const AuxArgs& ax) {
  if (ax.mode) return launch_tile_v<CK, KS, NB, MI, TW, TPC, true>(d, x, wfrag, bias, y, stats, s, ax);
  return launch_tile_v<CK, KS, NB, MI, TW, TPC, false>(d, x, wfrag, bias, y, stats, s, ax);
}

// tile shape by image width: 32-, 16- or 8-pixel-wide tiles of 128 pixels
template <int CK, int KS, int NB, int TPC>
int dispatch_geom(const ssa_conv_desc& d, const void* x, const void* w, const float* bias, void* y,
                  double* stats, hipStream_t s, const AuxArgs& ax) {
  if (d.W >= 32) return launch_tile<CK, KS, NB, 1, 32, TPC>(d, x, w, bias, y, stats, s, ax);
  if (d.W >= 16) return launch_tile<CK, KS, NB, 1, 16, TPC>(d, x, w, bias, y, stats, s, ax);
  return launch_tile<CK, KS, NB, 1, 8, TPC>(d, x, w, bias, y, stats, s, ax);
}

// What is left for this kernel since the persistent one (conv_tile_p.hip) took the trunk's 48/96/192/384-channel
// layers: the 64-channel stem / layer1 convs, images narrower than 16 pixels, convs with a bias.  One instantiation
// per channel count -- 128-pixel tiles, one n-block per workgroup (measured best for these latency-bound layers,
// profiles/r01_convbench.txt); round 2's variants (two / three n-blocks, 256-pixel tiles, the 96-channel-chunk
// instantiation and the ConvTileAny kernel of the grouped levels) went with their callers in round 4.
int tile_impl(const ssa_conv_desc* dp, const void* x, const void* w_frag, const float* bias, void* y,
              double* stats, const AuxArgs& ax, void* stream) {
  if (!dp || !x || !w_frag || !y) return SSA_EINVAL;
  if (!ssa_conv2d_tile_supported(dp)) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_frag)) & 15u)
    return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  switch (d.Cin) {
    case 48: return dispatch_geom<48, 3, 1, 9>(d, x, w_frag, bias, y, stats, s, ax);
    case 64: return dispatch_geom<64, 3, 1, 9>(d, x, w_frag, bias, y, stats, s, ax);
    case 96: return dispatch_geom<96, 3, 1, 9>(d, x, w_frag, bias, y, stats, s, ax);
    case 192:
    case 384: return dispatch_geom<192, 3, 1, 1>(d, x, w_frag, bias, y, stats, s, ax);   // Cin/192 passes, one tap per filter stage
    default: return SSA_EUNSUPPORTED;
  }
}

}  // namespace

extern "C" {

int ssa_conv2d_tile_supported(const ssa_conv_desc* d) {
  if (!d) return 0;
  if (d->KH != d->KW || d->KH != 3) return 0;
  if (d->stride != 1 || d->dil != 1 || d->transposed || d->pad != d->KH / 2) return 0;
  if (d->Ho != d->H || d->Wo != d->W || d->out_f32) return 0;
  if (d->Cout % 8 || d->ldy % 8 || d->ldx % 8) return 0;
  if ((long)d->H * d->W * d->ldx >= (1L << 31)) return 0;      // 32-bit element offsets inside one image
  return d->Cin == 48 || d->Cin == 64 || d->Cin == 96 || d->Cin == 192 || d->Cin == 384;
}

int ssa_conv2d_tile(const ssa_conv_desc* dp, const void* x, const void* w_frag, const float* bias,
                    void* y, double* stats, void* stream) {
  return tile_impl(dp, x, w_frag, bias, y, stats, AuxArgs{nullptr, 0, nullptr, 0}, stream);
}

int ssa_conv2d_tile_aux(const ssa_conv_desc* dp, const void* x, const void* w_frag, const float* bias,
                        void* y, double* stats, const void* aux, int ldaux, const float* coef,
                        int aux_mode, void* stream) {
  if (aux_mode != 1 && aux_mode != 2) return SSA_EINVAL;
  if (!aux || ldaux % 8 || (reinterpret_cast<uintptr_t>(aux) & 15u)) return SSA_EINVAL;
  if (aux_mode == 2 && (!coef || !stats)) return SSA_EINVAL;
  if (aux_mode == 1 && stats) return SSA_EINVAL;          // forward statistics are not part of this epilogue
  return tile_impl(dp, x, w_frag, bias, y, stats, AuxArgs{aux, ldaux, coef, aux_mode}, stream);
}

int ssa_bn_stat_replicas(void) { return kStatReplicas; }

}  // extern "C"
