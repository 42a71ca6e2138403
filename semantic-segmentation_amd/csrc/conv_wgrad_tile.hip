// Halo-staged weight gradient for the 3x3 stride-1 convs of the HRNet trunk
// (Cin = Cout = 48 / 64 / 96 / 192 / 384; network/hrnetv2.py:31-66, SURVEY.md K1:
// 522 of the 641 wgrad launches of a training step).
//
//   dW[co][tap][ci] = sum over pixels p of  dy[p][co] * x[p + tap][ci]
//
// as a GEMM: M = co, N = (tap, ci), K = pixels.  The K-pipelined wgrad kernel
// (conv_igemm.hip) gathers x im2col-style (every input pixel read 9 times, 9
// bounds checks per element) and splits the pixel axis over ~90 workgroups per
// output tile, each writing an fp32 partial: 30-40 MB of partial traffic for a
// 6 MB layer (profiles/r01_pmc_traffic.txt).  Here
//   * a workgroup is PERSISTENT over a strip of 128-pixel tiles (4 rows x 32) and
//     keeps its whole block of dW in MFMA accumulators, so only G <= 96 partials
//     exist per layer;
//   * per tile the x HALO image (6x34 pixels) and the dy tile are staged in LDS
//     once, pixel-major exactly as they lie in HBM (16-byte copies, no transposing
//     stores); all 9 taps are the same image at shifted pixel offsets;
//   * both MFMA operands need 8 consecutive PIXELS per lane for one channel, i.e.
//     a transposed read of the pixel-major image: ds_read_b64_tr_b16 (lane map
//     probed on the device, tests/test_kernels_gpu.py::test_probe_tr16).  The
//     pixel stride is 64*odd bytes, which puts the 4 pixels a 16-lane group reads
//     in 4 distinct 64-byte windows of the 256-byte bank row: conflict free.
// grid.y partitions the output (co part, n part) so that a workgroup owns at
// most 12 accumulator tiles per wave:
//     C=48: 2 m-blocks x 14 n-blocks (all taps)      C=64: 2 x 18
//     C=96: 3 x 9 (one kernel row kh per part)       C=192/384: the C=96 instantiation over
//     (96 output channels) x (kernel row, 96 input channels) sub-problems -- 12 / 48 parts
//     (SSA_WGRAD_C96=0: 6 x 6, one tap of 192 input x 192 output channels per part)
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef short s16x8_t __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int tr_stride_bytes(int row_bytes) {   // smallest 64*odd >= row_bytes
  int s = (row_bytes + 63) / 64;
  if ((s & 1) == 0) ++s;
  return s * 64;
}

__device__ __forceinline__ bf16x8_t tr_read8(const unsigned char* p, int stride_bytes) {
  const s16x4_t lo = ssa_tr16_b64(p);
  const s16x4_t hi = ssa_tr16_b64(p + 4 * stride_bytes);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ uint4 g_zero_piece_t;    // 16 zero bytes: what a piece outside the image loads (ConvWgradTileA's DMAs)

struct WgradTileArgs {
  const bf16_t* x; const bf16_t* dy; float* partial;
  int ldx, Cin, lddy, cout_pad, B, H, W, tiles_x, tiles_y, tiles_per_wg, n_parts;
};

template <int CX, int MB, int NBW>
struct ConvWgradTile {
  typedef WgradTileArgs Args;
  static constexpr int NT = 256;
  static constexpr int MAXJOBS = 32;     // (group.h: twice the default -- the strips get longer, the partials fewer)
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ dy = a.dy;
  float* __restrict__ partial = a.partial;
  const int ldx = a.ldx, Cin = a.Cin, lddy = a.lddy, cout_pad = a.cout_pad, B = a.B, H = a.H, W = a.W;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, tiles_per_wg = a.tiles_per_wg, n_parts = a.n_parts;
  constexpr int TW = 32, TH = 4, HW_ = TW + 2, HH_ = TH + 2;
  constexpr int SX = tr_stride_bytes(CX * 2), SD = tr_stride_bytes(MB * 64);
  constexpr int HALO_BYTES = HH_ * HW_ * SX;
  constexpr int NBL = (NBW + 3) / 4;           // n-blocks per wave
  constexpr int XP = CX / 8, DP = MB * 4;      // 16-byte pieces per pixel
  SSA_DYN_LDS(unsigned char, smem);
  unsigned char* Xs = smem;
  unsigned char* Ds = smem + HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_part = by % n_parts, co_part = by / n_parts;
  const int co0 = co_part * MB * 32;
  const int Kflat = 9 * Cin;
  // Column parts of dW[co][(tap, ci)].  CX = 96 ("row mode", Cin = 96 / 192 / 384): a part is one kernel
  // row kh x one block of 96 input channels -- n-block nb = (kw = nb / 3, 32 channels (nb % 3) of the
  // block) --, so the 192- and 384-channel layers run as 2x2 / 4x4 (co block, ci block) sub-problems of
  // the 96-channel instantiation.  Otherwise a part is NBW*32 consecutive columns.
  constexpr bool ROWS = CX == 96;
  const int n0 = ROWS ? 0 : n_part * NBW * 32;
  const int ci_base = ROWS ? (n_part / 3) * 96 : n0 % Cin;

  // per-lane constants of the B (x) fragments of this wave's n-blocks
  const int li = lane & 15, lj = li >> 2, lq = li & 3, lg = (lane >> 4) & 1, lh = lane >> 5;
  int b_off[NBL];
  int kcol0[NBL];             // first dW column of n-block l (-1: none)
#pragma unroll
  for (int l = 0; l < NBL; ++l) {
    const int nb = wave + 4 * l;
    if constexpr (ROWS) {
      const int kh = n_part % 3, kw = nb / 3, c32 = (nb - kw * 3) * 32;
      const bool ok = nb < NBW;
      b_off[l] = ok ? (kh * HW_ + kw) * SX + (c32 + 16 * lg + 4 * lq) * 2 : 0;
      kcol0[l] = ok ? (kh * 3 + kw) * Cin + ci_base + c32 : -1;
    } else {
      const int n16 = n0 + nb * 32 + 16 * lg;
      const bool ok = nb < NBW && n16 < Kflat;
      const int tap = ok ? n16 / Cin : 0;
      const int ci = ok ? n16 - tap * Cin - ci_base : -4 * lq;   // invalid: offset 0 (in bounds, discarded)
      const int kh = tap / 3, kw = tap - kh * 3;
      b_off[l] = (kh * HW_ + kw) * SX + (ci + 4 * lq) * 2;
      kcol0[l] = nb < NBW ? n0 + nb * 32 : -1;
    }
  }
  const int a_col = (16 * lg + 4 * lq) * 2;     // byte offset of this lane's 4 channels inside an m-block

  f32x16_t acc[MB][NBL];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int l = 0; l < NBL; ++l)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][l][r] = 0.f;

  const int total_tiles = B * tiles_x * tiles_y;
  const int t_begin = bx * tiles_per_wg;
  const int t_end = min(total_tiles, t_begin + tiles_per_wg);
  constexpr int XN = HH_ * HW_ * XP, DN = TH * TW * DP;
  constexpr int XI = (XN + 255) / 256, DI = (DN + 255) / 256;
  uint4 xv[XI], dv[DI];
  unsigned xmask = 0, dmask = 0;               // bit i = piece i lies inside the image
  // tile-invariant part of every piece this thread moves: packed (row, column) inside the
  // tile, element offset relative to the tile origin, LDS byte offset (-1: no such piece).
  // Kept in registers when there are few pieces per thread (C <= 96); recomputed per tile
  // for the 192-channel variant, whose accumulators leave no registers for them.
  constexpr bool PRE = (XI + DI) <= 16;
  int x_rc_[PRE ? XI : 1], x_go_[PRE ? XI : 1], x_lo_[PRE ? XI : 1];
  int d_rc_[PRE ? DI : 1], d_go_[PRE ? DI : 1], d_lo_[PRE ? DI : 1];
  auto x_piece = [&](int i, int* rc, int* go, int* lo) {
    const int piece = tid + i * 256;
    const int pix = piece / XP, cp = piece - pix * XP;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    *rc = (hy << 8) | hx;
    *go = (hy * W + hx) * ldx + cp * 8;
    *lo = piece < XN ? pix * SX + cp * 16 : -1;
  };
  auto d_piece = [&](int i, int* rc, int* go, int* lo) {
    const int piece = tid + i * 256;
    const int pix = piece / DP, cp = piece - pix * DP;
    const int ty = pix / TW, tx = pix - ty * TW;
    *rc = (ty << 8) | tx;
    *go = (ty * W + tx) * lddy + cp * 8;
    *lo = (piece < DN && co0 + cp * 8 < cout_pad) ? pix * SD + cp * 16 : -1;
  };
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < XI; ++i) x_piece(i, &x_rc_[i], &x_go_[i], &x_lo_[i]);
#pragma unroll
    for (int i = 0; i < DI; ++i) d_piece(i, &d_rc_[i], &d_go_[i], &d_lo_[i]);
  }
  // global -> registers for tile t (x halo image, zero outside the image; dy tile, zero
  // outside the image / past cout_pad); issued one tile ahead of the MFMAs
  auto fetch = [&](int t) {
    int r_ = t;
    const int tx_i = r_ % tiles_x; r_ /= tiles_x;
    const int ty_i = r_ % tiles_y;
    const int b = r_ / tiles_y;
    const int x0 = tx_i * TW, y0 = ty_i * TH;
    const bf16_t* xb = x + ((long)b * H * W + (long)(y0 - 1) * W + (x0 - 1)) * ldx + ci_base;
    const bf16_t* db = dy + ((long)b * H * W + (long)y0 * W + x0) * lddy + co0;
    // every lane loads (pieces outside the image read the tile's first output pixel and are zeroed when they
    // are staged): a select on the loaded value made the compiler wait for each batch of loads inside this
    // function -- the "prefetch" of the next tile then stalled for a full memory latency, twice per tile
    xmask = 0;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int rc, go, lo;
      if constexpr (PRE) { rc = x_rc_[i]; go = x_go_[i]; lo = x_lo_[i]; } else { x_piece(i, &rc, &go, &lo); }
      const int iy = y0 - 1 + (rc >> 8), ix = x0 - 1 + (rc & 255);
      const bool ok = lo >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      xv[i] = *reinterpret_cast<const uint4*>(xb + (ok ? go : (W + 1) * ldx));
      xmask |= (ok ? 1u : 0u) << i;
    }
    dmask = 0;
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      int rc, go, lo;
      if constexpr (PRE) { rc = d_rc_[i]; go = d_go_[i]; lo = d_lo_[i]; } else { d_piece(i, &rc, &go, &lo); }
      const bool ok = lo >= 0 && y0 + (rc >> 8) < H && x0 + (rc & 255) < W;
      dv[i] = *reinterpret_cast<const uint4*>(db + (ok ? go : 0));
      dmask |= (ok ? 1u : 0u) << i;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int lo;
      if constexpr (PRE) { lo = x_lo_[i]; } else { int rc, go; x_piece(i, &rc, &go, &lo); }
      if (lo >= 0) *reinterpret_cast<uint4*>(Xs + lo) = ((xmask >> i) & 1u) ? xv[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int piece = tid + i * 256;
      if (piece < DN)
        *reinterpret_cast<uint4*>(Ds + (piece / DP) * SD + (piece % DP) * 16) = ((dmask >> i) & 1u) ? dv[i] : make_uint4(0, 0, 0, 0);
    }
  };
  if (t_begin < t_end) fetch(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    if (t > t_begin) __syncthreads();          // previous tile's fragments are read
    stage();
    __syncthreads();
    if (t + 1 < t_end) fetch(t + 1);           // next tile's loads fly during this tile's MFMAs
    // ---- 8 k-steps of 16 pixels (half a tile row each).  The wave's n-block count (NBL or NBL - 1) is hoisted out
    // of the loop, so each variant is straight-line code, and the fragments of k-step ks + 1 are read (into the other
    // half of a register ring) while the MFMAs of k-step ks run -- left to itself the compiler read, waited,
    // multiplied, and only then read again.
    // columns n >= 9*Cin of the last n-block read whatever lies at offset 0 of the image: a B column only feeds its
    // own output column, and those columns are never stored
    auto k_loop = [&](auto nl_c) {
      constexpr int NL = decltype(nl_c)::value;
      bf16x8_t af[2][MB], bfr[2][NL > 0 ? NL : 1];
      auto rd = [&](int ks, int slot) {
        const int ty = ks >> 1, tx0 = (ks & 1) * 16;
        const int kp = tx0 + 8 * lh + lj;         // this lane's first pixel column inside the row
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) af[slot][mb] = tr_read8(Ds + (ty * TW + kp) * SD + mb * 64 + a_col, SD);
#pragma unroll
        for (int l = 0; l < NL; ++l) bfr[slot][l] = tr_read8(Xs + (ty * HW_ + kp) * SX + b_off[l], SX);
      };
      constexpr bool PIPE = CX != 192;          // the 192-channel instantiation has no registers for a second set
      if constexpr (PIPE) {
        rd(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MB + NL), 0);    // the prologue reads are a group of their own
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if constexpr (PIPE) {
          if (ks + 1 < 8) rd(ks + 1, (ks + 1) & 1);
        } else {
          rd(ks, 0);
        }
        const int slot = PIPE ? (ks & 1) : 0;
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
            acc[mb][l] = ssa_mfma32(af[slot][mb], bfr[slot][l], acc[mb][l]);
        if constexpr (PIPE) {
          if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MB + NL), 0);
          __builtin_amdgcn_sched_group_barrier(0x008, MB * NL, 0);
        }
      }
    };
    constexpr int NL_LO = NBW / 4, NL_HI = (NBW + 3) / 4;      // n-blocks of waves >= NBW % 4 / of the others
    if (NL_HI == NL_LO || wave < NBW % 4) k_loop(std::integral_constant<int, NL_HI>());
    else k_loop(std::integral_constant<int, NL_LO>());
  }

  // ---- this workgroup's block of partial[g][co][k] (zeros if it had no tile)
  float* out = partial + (long)bx * cout_pad * Kflat;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int l = 0; l < NBL; ++l) {
      const int kcol = kcol0[l] + (lane & 31);
      if (kcol0[l] < 0 || kcol >= Kflat) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < cout_pad) out[(long)co * Kflat + kcol] = acc[mb][l][r];
      }
    }
  }
};

// ---- All nine taps per staged tile, LDS-DMA double buffer (round 6) ----------------------------------------------------
// ConvWgradTile above holds 9 accumulator tiles per wave AND the next tile's x / dy pieces in registers (it stages
// through them) AND their addresses: 467 registers per lane, i.e. ONE 4-wave workgroup per CU (the host sized its
// launches for two), whose single wave per SIMD alternates between staging, two barriers and 48-72 MFMAs per tile --
// MFMA pipe busy 0.12 (profiles/r05_pmc.txt); and a 96-channel block's tile is staged three times, once per kernel row.
// Here, for the 96-channel blocks (Cin = Cout = 96 / 192 / 384: three quarters of the trunk's weight-gradient FLOPs),
//   * a workgroup (4 waves, one per SIMD) owns HALF of the 27 (tap, 32-channel) n-blocks -- 14 or 13 -- x 3 m-blocks of
//     its (96 co x 96 ci) block: wave w holds n-blocks w, w + 4, w + 8, w + 12 of its half x 3 m-blocks = 12
//     accumulator tiles, 192 registers -- the matrix instructions of one function accumulate EITHER in the accumulator
//     half of the register file OR in the vector half (256 registers = 16 tiles each), so 21 tiles per wave, a whole
//     block on four waves, spill (tried: 817 spilled registers).  The staged tile feeds 336 MFMAs instead of 216 -- 96 per
//     wave between barriers instead of 48-72 -- and is fetched twice per block instead of three times;
//   * x halo image and dy tile arrive by LDS DMA (global_load_lds_dwordx4; pieces outside the image fetch a zero block)
//     into the other half of a double buffer while the current tile's MFMAs run -- no staging registers, no ds_write, no
//     address tables -- and a tile costs ONE barrier (s_waitcnt vmcnt(0) ; s_barrier: the next tile has landed,
//     everybody is through with the buffer the DMAs after it overwrite);
//   * the fragments of k-step ks + 1 are read (second register set) while the MFMAs of k-step ks run, as above;
//   * every wave runs the same straight-line body over 4 n-block slots; a slot past the half's count multiplies the
//     image's first pixels into an accumulator that is never stored (2 of 16 slots in one half, 3 in the other).
// Same pixel-major LDS image (192-byte pixel stride = 64 x odd: ds_read_b64_tr_b16 conflict free), same partials.
template <int DUMMY>
struct ConvWgradTileA {
  typedef WgradTileArgs Args;
  static constexpr int NT = 256, NW = 4;
  static constexpr int MAXJOBS = 32;
  static constexpr int CX = 96, MB = 3, NBW = 27, NBL = 4, TAP_PARTS = 2, NB_FIRST = 14;
  static constexpr int TW = 32, TH = 4, HW_ = TW + 2, HH_ = TH + 2;
  static constexpr int SX = 192, SD = 192, XP = 12, DP = 12;       // pixel strides (bytes), 16-byte pieces per pixel
  static constexpr int XN = HH_ * HW_ * XP, DN = TH * TW * DP;     // pieces: 2448, 1536
  static constexpr int XW = (XN + 63) / 64, DW = DN / 64;          // wave instructions: 39, 24
  static constexpr int XI = (XW + NW - 1) / NW, DI = DW / NW;      // per wave: 10 (the last wave 9), 6
  static constexpr int X_BYTES = XW * 1024, D_BYTES = DW * 1024, STAGE_BYTES = X_BYTES + D_BYTES;
  static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE_BYTES;
  static_assert(DN % 64 == 0 && DW % NW == 0 && LDS_BYTES <= 160 * 1024, "geometry");

  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ dy = a.dy;
  float* __restrict__ partial = a.partial;
  const int ldx = a.ldx, Cin = a.Cin, lddy = a.lddy, cout_pad = a.cout_pad, B = a.B, H = a.H, W = a.W;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, tiles_per_wg = a.tiles_per_wg, n_parts = a.n_parts;
  SSA_DYN_LDS(unsigned char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // by = (co block, ci block, half of the n-blocks): n_parts = 2 x the ci blocks
  const int n_part = by % n_parts, co_part = by / n_parts;
  const int half = n_part % TAP_PARTS, ci_part = n_part / TAP_PARTS;
  const int co0 = co_part * 96, ci_base = ci_part * 96;
  const int nb_begin = half * NB_FIRST, nb_count = half == 0 ? NB_FIRST : NBW - NB_FIRST;
  const int Kflat = 9 * Cin;

  // per-lane constants of the B (x) fragments of this wave's n-blocks: n-block nb = (tap nb / 3, channels 32 (nb % 3))
  const int li = lane & 15, lj = li >> 2, lq = li & 3, lg = (lane >> 4) & 1, lh = lane >> 5;
  int b_off[NBL], kcol0[NBL];
#pragma unroll
  for (int l = 0; l < NBL; ++l) {
    const bool ok = wave + NW * l < nb_count;
    const int nb = nb_begin + wave + NW * l;
    const int tap = ok ? nb / 3 : 0, c32 = ok ? (nb - tap * 3) * 32 : 0;
    const int kh = tap / 3, kw = tap - kh * 3;
    b_off[l] = (kh * HW_ + kw) * SX + (c32 + 16 * lg + 4 * lq) * 2;
    kcol0[l] = ok ? tap * Cin + ci_base + c32 : -1;
  }
  const int a_col = (16 * lg + 4 * lq) * 2;     // byte offset of this lane's 4 channels inside an m-block

  f32x16_t acc[MB][NBL];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int l = 0; l < NBL; ++l)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][l][r] = 0.f;

  // this lane's pieces of a tile, tile-invariant part: x piece i = slot 64 (wave + 8 i) + lane -> (halo pixel, piece);
  // packed (hy << 16) | (hx << 8) | piece, -1: no such piece (its DMA fetches the zero block)
  int xd[XI], dd[DI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int slot = (wave + NW * i) * 64 + lane;
    const int pix = slot / XP, pc = slot - pix * XP;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    xd[i] = (wave + NW * i < XW && slot < XN) ? ((hy << 16) | (hx << 8) | pc) : -1;
  }
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    const int slot = (wave + NW * i) * 64 + lane;
    const int pix = slot / DP, pc = slot - pix * DP;
    const int ty = pix / TW, tx = pix - ty * TW;
    dd[i] = (co0 + pc * 8 < cout_pad) ? ((ty << 16) | (tx << 8) | pc) : -1;
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(&g_zero_piece_t);
  const int total_tiles = B * tiles_x * tiles_y;
  const int t_begin = bx * tiles_per_wg;
  const int t_end = min(total_tiles, t_begin + tiles_per_wg);

  auto issue = [&](int t, int buf) {
    int r_ = t;
    const int tx_i = r_ % tiles_x; r_ /= tiles_x;
    const int ty_i = r_ % tiles_y;
    const int b = r_ / tiles_y;
    const int x0 = tx_i * TW, y0 = ty_i * TH;
    unsigned char* Xs = smem + buf * STAGE_BYTES;
    unsigned char* Ds = Xs + X_BYTES;
    const bf16_t* xb = x + (long)b * H * W * ldx + ci_base;
    const bf16_t* db = dy + (long)b * H * W * lddy + co0;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int wi = wave + NW * i;
      if (wi < XW) {                            // (wave-uniform: the last wave has one instruction fewer)
        const int iy = y0 - 1 + (xd[i] >> 16), ix = x0 - 1 + ((xd[i] >> 8) & 255);
        const bool ok = xd[i] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const bf16_t* src = ok ? xb + ((long)iy * W + ix) * ldx + (xd[i] & 255) * 8 : zero;
        ssa_glds16_untracked(src, Xs + wi * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int wi = wave + NW * i;
      const int oy = y0 + (dd[i] >> 16), ox = x0 + ((dd[i] >> 8) & 255);
      const bool ok = dd[i] >= 0 && oy < H && ox < W;
      const bf16_t* src = ok ? db + ((long)oy * W + ox) * lddy + (dd[i] & 255) * 8 : zero;
      ssa_glds16_untracked(src, Ds + wi * 1024);
    }
  };

  // The fragments of k-step ks + 1 are read into the other half of a register ring while the MFMAs of k-step ks run.
  auto compute = [&](const unsigned char* Xs, const unsigned char* Ds) {
    constexpr int NL = NBL;
    bf16x8_t af[2][MB], bfr[2][NL];
    auto rd = [&](int ks, int slot) {
      const int ty = ks >> 1, tx0 = (ks & 1) * 16;
      const int kp = tx0 + 8 * lh + lj;         // this lane's first pixel column inside the row
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[slot][mb] = tr_read8(Ds + (ty * TW + kp) * SD + mb * 64 + a_col, SD);
#pragma unroll
      for (int l = 0; l < NL; ++l) bfr[slot][l] = tr_read8(Xs + (ty * HW_ + kp) * SX + b_off[l], SX);
    };
    rd(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MB + NL), 0);      // the prologue reads are a group of their own
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) rd(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb][l] = ssa_mfma32(af[ks & 1][mb], bfr[ks & 1][l], acc[mb][l]);
      if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MB + NL), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MB * NL, 0);
    }
  };

  int buf = 0;
  if (t_begin < t_end) issue(t_begin, 0);
  for (int t = t_begin; t < t_end; ++t) {
    ssa_wait_vm_barrier<0, 0>();                // tile t has landed; every wave is through with the other buffer
    if (t + 1 < t_end) issue(t + 1, buf ^ 1);
    const unsigned char* Xs = smem + buf * STAGE_BYTES;
    const unsigned char* Ds = Xs + X_BYTES;
    compute(Xs, Ds);
    buf ^= 1;
  }

  // ---- this workgroup's block of partial[g][co][k] (zeros if it had no tile)
  float* out = partial + (long)bx * cout_pad * Kflat;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int l = 0; l < NBL; ++l) {
      if (kcol0[l] < 0) continue;
      const int kcol = kcol0[l] + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < cout_pad) out[(long)co * Kflat + kcol] = acc[mb][l][r];
      }
    }
  }
};

struct Plan { int cx, mb, nbw, n_parts, co_parts; };

bool c96_on() {
  static const bool on = !(getenv("SSA_WGRAD_C96") && atoi(getenv("SSA_WGRAD_C96")) == 0);
  return on;
}

// SSA_WGRAD_ALL=0: the 96-channel blocks stay on the 4-wave, one-kernel-row-per-workgroup form
bool all_taps_on() {
  static const bool on = !(getenv("SSA_WGRAD_ALL") && atoi(getenv("SSA_WGRAD_ALL")) == 0);
  return on;
}

bool make_plan(int Cin, int cout_pad, Plan* p) {
  if (Cin != cout_pad) return false;
  if (all_taps_on() && (Cin == 96 || Cin == 192 || Cin == 384)) {      // ConvWgradTileA: cx = 0 marks it
    *p = Plan{0, 3, 27, 2 * (Cin / 96), Cin / 96};       // n_parts = 2 halves of the n-blocks x ci blocks
    return true;
  }
  switch (Cin) {
    case 48: *p = {48, 2, 14, 1, 1}; return true;
    case 64: *p = {64, 2, 18, 1, 1}; return true;
    case 96: *p = {96, 3, 9, 3, 1}; return true;
    case 192: *p = c96_on() ? Plan{96, 3, 9, 6, 2} : Plan{192, 6, 6, 9, 1}; return true;
    case 384: *p = c96_on() ? Plan{96, 3, 9, 12, 4} : Plan{192, 6, 6, 18, 2}; return true;
    default: return false;
  }
}

constexpr size_t wgrad_tile_lds(int cx, int mb) {
  return (size_t)6 * 34 * tr_stride_bytes(cx * 2) + (size_t)128 * tr_stride_bytes(mb * 64);
}

template <int CX, int MB, int NBW>
int launch(const ssa_conv_desc& d, const Plan& p, const void* x, const void* dy, int lddy, int cout_pad,
           int G, int tiles_per_wg, float* partial, hipStream_t s) {
  constexpr size_t lds = wgrad_tile_lds(CX, MB);
  static_assert(lds <= 160 * 1024, "does not fit in LDS");
  WgradTileArgs a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.partial = partial;
  a.ldx = d.ldx; a.Cin = d.Cin; a.lddy = lddy; a.cout_pad = cout_pad; a.B = d.B; a.H = d.H; a.W = d.W;
  a.tiles_x = (d.W + 31) / 32; a.tiles_y = (d.H + 3) / 4; a.tiles_per_wg = tiles_per_wg; a.n_parts = p.n_parts;
  // (Measured and rejected, profiles/r02_notes.md call P: the 48- and 96-channel instantiations behind one
  // kernel, as conv_tile.hip's ConvTileAny -- no gain here, a flush's launches are 60-160 us each.)
  return ssa::submit<ConvWgradTile<CX, MB, NBW>>(a, G, p.n_parts * p.co_parts, lds, s);
}

bool shape_ok(const ssa_conv_desc* d) {
  return d && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && !d->transposed &&
         d->Ho == d->H && d->Wo == d->W && d->ldx % 8 == 0;
}

}  // namespace

extern "C" {

int ssa_conv2d_wgrad_tile_plan(const ssa_conv_desc* d, int cout_pad, int* nsplit, size_t* ws_bytes) {
  Plan p;
  if (!nsplit || !ws_bytes || !shape_ok(d) || !make_plan(d->Cin, cout_pad, &p)) return SSA_EUNSUPPORTED;
  const long tiles = (long)d->B * ((d->W + 31) / 32) * ((d->H + 3) / 4);
  const long per_split = (long)cout_pad * 9 * d->Cin * sizeof(float);
  static const long budget_mb = getenv("SSA_WGRAD_TILE_MB") ? atol(getenv("SSA_WGRAD_TILE_MB")) : 12;
  long g = (budget_mb << 20) / per_split;       // bounded partial traffic per layer
  if (g > 192) g = 192;
  if (g < 2) g = 2;
  // d->cfg > 0: tiles per workgroup asked for by the caller -- grouped launches (group.h) get
  // their parallelism from the number of layers in the launch, so a layer is split only far
  // enough to bound a workgroup's serial strip; the partials then cost less than the operands
  if (d->cfg > 0) g = (tiles + d->cfg - 1) / d->cfg;
  if (g < 1) g = 1;
  if (g > tiles) g = tiles;
  const long tpw = (tiles + g - 1) / g;
  g = (tiles + tpw - 1) / tpw;
  *nsplit = (int)g;
  *ws_bytes = (size_t)g * per_split;
  return SSA_OK;
}

// What the host's launch planning needs to know about a layer's launch: workgroups per strip (`parts`) and how many
// workgroups of this instantiation the chip holds at once (`slots`: 64 KB of LDS and <= 256 registers x 4 waves = two per
// CU on paper -- the 4-wave form in fact holds 467 registers and gets ONE; the 8-wave all-taps form: one per CU).
int ssa_conv2d_wgrad_tile_geometry(int Cin, int cout_pad, int* parts, int* slots, int* kind) {
  Plan p;
  if (!parts || !slots || !kind || !make_plan(Cin, cout_pad, &p)) return SSA_EUNSUPPORTED;
  *parts = p.n_parts * p.co_parts;
  *slots = 256;
  *kind = p.cx == 0 ? 0 : (p.cx == 96 ? 96 : p.cx);      // launches group by instantiation
  return SSA_OK;
}

int ssa_conv2d_wgrad_tile(const ssa_conv_desc* dp, const void* x, const void* dy, int lddy, int cout_pad,
                          int nsplit, float* partial, void* stream) {
  Plan p;
  if (!dp || !x || !dy || !partial || nsplit < 1) return SSA_EINVAL;
  if (!shape_ok(dp) || !make_plan(dp->Cin, cout_pad, &p) || lddy % 8) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15u) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  const long tiles = (long)d.B * ((d.W + 31) / 32) * ((d.H + 3) / 4);
  const int tpw = (int)((tiles + nsplit - 1) / nsplit);
  hipStream_t s = (hipStream_t)stream;
  if (p.cx == 0) {
    typedef ConvWgradTileA<0> K;
    WgradTileArgs a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.partial = partial;
    a.ldx = d.ldx; a.Cin = d.Cin; a.lddy = lddy; a.cout_pad = cout_pad; a.B = d.B; a.H = d.H; a.W = d.W;
    a.tiles_x = (d.W + 31) / 32; a.tiles_y = (d.H + 3) / 4; a.tiles_per_wg = tpw; a.n_parts = p.n_parts;
    return ssa::submit<K>(a, nsplit, p.n_parts * p.co_parts, K::LDS_BYTES, s);
  }
  switch (d.Cin) {
    case 48: return launch<48, 2, 14>(d, p, x, dy, lddy, cout_pad, nsplit, tpw, partial, s);
    case 64: return launch<64, 2, 18>(d, p, x, dy, lddy, cout_pad, nsplit, tpw, partial, s);
    case 96: return launch<96, 3, 9>(d, p, x, dy, lddy, cout_pad, nsplit, tpw, partial, s);
    default:
      if (p.cx == 96) return launch<96, 3, 9>(d, p, x, dy, lddy, cout_pad, nsplit, tpw, partial, s);
      return launch<192, 6, 6>(d, p, x, dy, lddy, cout_pad, nsplit, tpw, partial, s);
  }
}

}  // extern "C"
