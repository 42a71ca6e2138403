// Weight gradient of the large-channel head convolutions (conv3x3_ocr 720->512,
// attn 512->256 / 256->256, the 1x1 convs 1024->512, 720->720, 512->256, ...;
// network/ocrnet.py:54-58, network/utils.py:348-357, network/ocr_utils.py:68-93):
//
//   dW[co][tap][ci] = sum over pixels p of  dy[p][co] * x[p + tap][ci]
//
// GEMM with M = co, N = (tap, ci), K = pixels (65,536 per image at 1024x1024).
// One workgroup (8 wave64s) owns a 128(co) x 128(ci) x 3(kw)  [3x3, one kernel row
// kh]  or a 128(co) x 256(ci)  [1x1]  block of dW in MFMA accumulators and is
// PERSISTENT over a strip of 128-pixel tiles (4 rows x 32 columns):
//   * per tile the dy tile and the x rows it needs (4 x 34 pixels for the three kw
//     shifts of one kh) are staged in LDS pixel-major, exactly as they lie in HBM;
//     the three taps reuse the same x image at shifted pixel offsets;
//   * both MFMA operands want 8 consecutive PIXELS per lane for one channel:
//     ds_read_b64_tr_b16 transposing reads, pixel stride 64*odd bytes (conflict free);
//   * the next tile's global loads are issued into registers before the current
//     tile's MFMAs and stored to LDS after them (one barrier pair per tile);
//   * grid.x = pixel split: only G = 256 / (#output blocks) partials per layer
//     ([G][cout_pad][taps*Cin] fp32, summed by wgrad_reduce_kernel).
// Replaces the K-pipelined conv_wgrad_tr_kernel (195 TFLOP/s on 720->512) for these shapes.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"

namespace {

typedef short s16x8_t __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int trs(int row_bytes) {   // smallest 64*odd >= row_bytes
  int s = (row_bytes + 63) / 64;
  if ((s & 1) == 0) ++s;
  return s * 64;
}

__device__ __forceinline__ bf16x8_t tr8(const unsigned char* p, int stride_bytes) {
  const s16x4_t lo = ssa_tr16_b64(p);
  const s16x4_t hi = ssa_tr16_b64(p + 4 * stride_bytes);
  s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// KS = 3: NT = 3 (the kw taps of kernel row kh), 128 input channels per workgroup.
// KS = 1: NT = 2 (two 32-channel blocks per wave), 256 input channels per workgroup.
struct WgradHeadArgs {
  const bf16_t* x; const bf16_t* dy; float* partial;
  int ldx, Cin, lddy, cout_pad, B, H, W, tiles_x, tiles_y, tiles_per_wg, ci_tiles, G, wgs_y;
};

// Bijective XCD-aware order (block b runs on XCD b % 8): XCD x gets one contiguous range of work items.
__device__ __forceinline__ int xcd_order(int v, int n) {
  const int q = n >> 3, r = n & 7, x = v & 7, k = v >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

template <int KS>
struct ConvWgradHead {
  typedef WgradHeadArgs Args;
  static constexpr int NT = 512;
  static __device__ __forceinline__ void run(const Args& a, const int bx_, const int by0_, const int /*gx*/) {
  // Work items in the order (pixel strip, co tile, ci tile, kernel row), dealt to the XCDs in contiguous ranges: the
  // 18 workgroups (6 ci tiles x 3 kernel rows) that stream the same dy block, and the 12 that stream the same x
  // block, then run on ONE XCD at about the same pixel and share its L2 -- in launch order (strip fastest) every
  // workgroup fetched its operands from HBM by itself: 3.9 GB of fabric traffic for 0.72 GB of operands
  // (profiles/r03_pmc.txt), the kernel ran at the fabric's bandwidth, not the MFMA's.
  const int w_ = xcd_order(by0_ * a.G + bx_, a.G * a.wgs_y);
  const int bx = w_ / a.wgs_y, by_ = w_ - bx * a.wgs_y;
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ dy = a.dy;
  float* __restrict__ partial = a.partial;
  const int ldx = a.ldx, Cin = a.Cin, lddy = a.lddy, cout_pad = a.cout_pad, B = a.B, H = a.H, W = a.W;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, tiles_per_wg = a.tiles_per_wg, ci_tiles = a.ci_tiles;
  constexpr int TW = 32, TH = 4, NT = KS == 3 ? 3 : 2;
  constexpr int CX = KS == 3 ? 128 : 256;                // input channels per workgroup
  constexpr int XW = KS == 3 ? TW + 2 : TW;              // x image width in pixels
  constexpr int SX = trs(CX * 2), SD = trs(256);
  constexpr int X_BYTES = TH * XW * SX;
  constexpr int XP = CX / 8, DP = 16;                    // 16-byte pieces per pixel
  constexpr int XN = TH * XW * XP, DN = TH * TW * DP;
  constexpr int XI = (XN + 511) / 512, DI = (DN + 511) / 512;
  SSA_DYN_LDS(unsigned char, smem);
  unsigned char* Xs = smem;
  unsigned char* Ds = smem + X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // grid.y -> (co tile, ci tile, kh)
  int by = by_;
  const int kh = KS == 3 ? by % 3 : 0;
  if (KS == 3) by /= 3;
  const int ci_t = by % ci_tiles, co_t = by / ci_tiles;
  const int co0 = co_t * 128, ci0 = ci_t * CX;
  const int taps = KS * KS, Kflat = taps * Cin;

  const int li = lane & 15, lj = li >> 2, lq = li & 3, lg = (lane >> 4) & 1, lh = lane >> 5;
  // B fragment j of this wave: KS=3 -> tap (kh, kw=j), channel block wn; KS=1 -> channel block 2*wn+j
  int b_off[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int cblk = KS == 3 ? wn : 2 * wn + j;
    const int shift = KS == 3 ? j : 0;
    b_off[j] = shift * SX + (cblk * 32 + 16 * lg + 4 * lq) * 2;
  }
  const int a_col = (16 * lg + 4 * lq) * 2;

  f32x16_t acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  const int total_tiles = B * tiles_x * tiles_y;
  const int t_begin = bx * tiles_per_wg;
  const int t_end = min(total_tiles, t_begin + tiles_per_wg);
  uint4 xv[XI], dv[DI];
  unsigned xmask = 0, dmask = 0;               // bit i: piece i lies inside the image (others are zeroed when staged)
  auto fetch = [&](int t) {
    int r_ = t;
    const int tx_i = r_ % tiles_x; r_ /= tiles_x;
    const int ty_i = r_ % tiles_y;
    const int b = r_ / tiles_y;
    const int x0 = tx_i * TW, y0 = ty_i * TH;
    const int yoff = KS == 3 ? kh - 1 : 0, xoff = KS == 3 ? -1 : 0;
    const bf16_t* xb = x + (long)b * H * W * ldx + ci0;
    const bf16_t* db = dy + (long)b * H * W * lddy + co0;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int piece = tid + i * 512;
      const int pix = piece / XP, cp = piece - pix * XP;
      const int hy = pix / XW, hx = pix - hy * XW;
      const int iy = y0 + hy + yoff, ix = x0 + hx + xoff;
      const bool ok = piece < XN && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && ci0 + cp * 8 < Cin;
      // every lane loads (a select on the loaded value makes the compiler wait for the loads right here)
      xv[i] = *reinterpret_cast<const uint4*>(ok ? xb + ((long)iy * W + ix) * ldx + cp * 8 : x);
      xmask = i == 0 ? (ok ? 1u : 0u) : (xmask | ((ok ? 1u : 0u) << i));
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int piece = tid + i * 512;
      const int pix = piece / DP, cp = piece - pix * DP;
      const int oy = y0 + pix / TW, ox = x0 + pix % TW;
      const bool ok = piece < DN && oy < H && ox < W && co0 + cp * 8 < cout_pad;
      dv[i] = *reinterpret_cast<const uint4*>(ok ? db + ((long)oy * W + ox) * lddy + cp * 8 : dy);
      dmask = i == 0 ? (ok ? 1u : 0u) : (dmask | ((ok ? 1u : 0u) << i));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int piece = tid + i * 512;
      if (piece < XN)
        *reinterpret_cast<uint4*>(Xs + (piece / XP) * SX + (piece % XP) * 16) = ((xmask >> i) & 1u) ? xv[i] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int piece = tid + i * 512;
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
    // 16 pixels (half a tile row) per k-step; the fragments of k-step ks + 1 are read while the MFMAs of ks run
    bf16x8_t af[2][2], bfr[2][NT];
    auto rd = [&](int ks, int slot) {
      const int ty = ks >> 1, kp = (ks & 1) * 16 + 8 * lh + lj;
#pragma unroll
      for (int m = 0; m < 2; ++m) af[slot][m] = tr8(Ds + (ty * TW + kp) * SD + (wm * 2 + m) * 64 + a_col, SD);
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[slot][j] = tr8(Xs + (ty * XW + kp) * SX + b_off[j], SX);
    };
    rd(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (2 + NT), 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) rd(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[m][j] = ssa_mfma32(af[ks & 1][m], bfr[ks & 1][j], acc[m][j]);
      if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (2 + NT), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
    }
  }

  float* out = partial + (long)bx * cout_pad * Kflat;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int cblk = KS == 3 ? wn : 2 * wn + j;
      const int ci = ci0 + cblk * 32 + (lane & 31);
      const int tap = KS == 3 ? kh * 3 + j : 0;
      if (ci >= Cin) continue;
      const long kcol = (long)tap * Cin + ci;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * 2 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < cout_pad) out[(long)co * Kflat + kcol] = acc[m][j][r];
      }
    }
  }
};

__device__ uint4 g_zero_piece;      // 16 zero bytes: what a piece outside the image / the channel range fetches

// 3x3 (one kernel row kh per workgroup): the same tiling with both operand images brought in by LDS DMA
// (global_load_lds_dwordx4), double buffered by tile.  Round 2 moved every tile through registers -- 9 x 16-byte loads
// per thread whose destination registers the compiler reused for address arithmetic (a wait for the first load
// right in the issue sequence), then 9 ds_write_b128 between two barriers; measured on 720->512 @ 256x256
// (tools/headbench.py, experiment builds): 569 us = 346 us of MFMA loop + 96 us staging + 127 us of exposed fetch,
// nothing overlapping because all eight waves of the one resident workgroup are in the same phase.  A DMA needs no
// registers and no store phase, and the next tile's images land while this tile's 48 MFMAs per wave run.
//   * DMA fills 64 CONSECUTIVE 16-byte slots per wave instruction, so the images are unpadded (pixel-major, 256
//     bytes per pixel) and the conflict-free layout the transposing reads need comes from a swizzle instead: the
//     16-byte piece q of pixel p lies in slot  q ^ ((p & 3) << 2)  -- which piece a lane fetches is free.  The 32
//     lanes ds_read_b64_tr_b16 services together read 4 consecutive pixels x one 64-byte channel block: with the
//     swizzle the four pixels' blocks are the four different 64-byte windows of the 256-byte bank row.
//   * pieces outside the image or the channel range fetch a zero block; every wave issues the same 9 DMAs per tile.
struct ConvWgradHead3 {
  typedef WgradHeadArgs Args;
  static constexpr int NT = 512;
  static constexpr int TW = 32, TH = 4, XW = TW + 2, CX = 128;
  static constexpr int X_BYTES = TH * XW * 256, D_BYTES = TH * TW * 256, BUF_BYTES = X_BYTES + D_BYTES;
  static constexpr int XWI = X_BYTES / 1024;           // 34 wave instructions fill the x image, 32 the dy image
  static constexpr int XI = (XWI + 7) / 8, DI = D_BYTES / 1024 / 8, NI = XI + DI;   // DMAs per wave and tile: 5 + 4
  static_assert(NI == 9, "one DMA in front of the k loop and one behind each of its 8 steps");
  static constexpr size_t LDS = 2 * (size_t)BUF_BYTES + 1024;                        // + the dump block
  static __device__ __forceinline__ void run(const Args& a, const int bx_, const int by0_, const int /*gx*/) {
  const int w_ = xcd_order(by0_ * a.G + bx_, a.G * a.wgs_y);
  const int bx = w_ / a.wgs_y;
  int by = w_ - bx * a.wgs_y;
  const bf16_t* __restrict__ x = a.x;
  const bf16_t* __restrict__ dy = a.dy;
  float* __restrict__ partial = a.partial;
  const int ldx = a.ldx, Cin = a.Cin, lddy = a.lddy, cout_pad = a.cout_pad, B = a.B, H = a.H, W = a.W;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, tiles_per_wg = a.tiles_per_wg, ci_tiles = a.ci_tiles;
  SSA_DYN_LDS(unsigned char, smem);
  unsigned char* dump = smem + 2 * BUF_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int kh = by % 3; by /= 3;
  const int ci_t = by % ci_tiles, co_t = by / ci_tiles;
  const int co0 = co_t * 128, ci0 = ci_t * CX;
  const int Kflat = 9 * Cin;

  // ---- DMA role of this lane: piece (pixel p, slot) of wave instruction i * 8 + wave; the slot holds piece
  // q = slot ^ ((p & 3) << 2) of the pixel.  rel: element offset from the tile's first pixel; yx: (row << 8 | col)
  // inside the image window, 0xffff if the piece is never real (channel range / past the image)
  int rel[NI], yx[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const bool isx = i < XI;
    const int wi = (isx ? i : i - XI) * 8 + wave;
    const int piece = wi * 64 + lane;
    const int p = piece >> 4, q = (piece & 15) ^ ((p & 3) << 2);
    const int r = isx ? p / XW : p >> 5, c = isx ? p - r * XW : p & 31;
    const bool real = isx ? (wi < XWI && ci0 + q * 8 < Cin) : (co0 + q * 8 < cout_pad);
    rel[i] = (r * W + c) * (isx ? ldx : lddy) + q * 8;
    yx[i] = real ? ((r << 8) | c) : 0xffff;
  }
  // tile walk without divisions
  int f_tx, f_ty, f_b;
  const int total_tiles = B * tiles_x * tiles_y;
  const int t_begin = bx * tiles_per_wg;
  const int t_end = min(total_tiles, t_begin + tiles_per_wg);
  {
    f_tx = t_begin % tiles_x;
    const int r_ = t_begin / tiles_x;
    f_ty = r_ % tiles_y;
    f_b = r_ / tiles_y;
  }
  // the zero block's address, fetched once: left to itself the compiler reloads it from the GOT (s_load + a wait
  // for lgkmcnt(0), i.e. for all fragment reads in flight) in front of every DMA inside the k loop
  const bf16_t* zero_src = reinterpret_cast<const bf16_t*>(&g_zero_piece);
#ifndef SSA_EMU
  asm volatile("" : "+v"(zero_src));
#endif
  // DMA i (of NI) of the tile (f_b, f_ty, f_tx) -> buffer buf
  auto dma = [&](int i, int buf) {
    const int x0 = f_tx * TW, y0 = f_ty * TH;
    const bool isx = i < XI;
    const int wi = (isx ? i : i - XI) * 8 + wave;
    const int oy = isx ? y0 + kh - 1 : y0, ox = isx ? x0 - 1 : x0;
    const bf16_t* base = (isx ? x + ci0 : dy + co0) + ((long)f_b * H * W + (long)oy * W + ox) * (isx ? ldx : lddy);
    const int iy = oy + (yx[i] >> 8), ix = ox + (yx[i] & 255);
    const bool ok = yx[i] != 0xffff && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const bf16_t* src = ok ? base + rel[i] : zero_src;
    unsigned char* Xb = smem + buf * BUF_BYTES;
    unsigned char* dst = isx ? (wi < XWI ? Xb + wi * 1024 : dump) : Xb + X_BYTES + wi * 1024;
    ssa_glds16_untracked(src, dst);
  };
  auto next_tile = [&]() {
    if (++f_tx == tiles_x) {
      f_tx = 0;
      if (++f_ty == tiles_y) { f_ty = 0; ++f_b; }
    }
  };

  // ---- fragment addresses (bytes inside a buffer): pixel row 8 * lh + lj (+ 4 for the upper half of the 8 pixels),
  // 64-byte channel block  blk ^ (pixel & 3),  then 32 * lg + 8 * lq inside the block
  const int li = lane & 15, lj = li >> 2, lq = li & 3, lg = (lane >> 4) & 1, lh = lane >> 5;
  const int lrow = (8 * lh + lj) * 256 + lg * 32 + lq * 8;
  int a_base[2], b_base[4];
#pragma unroll
  for (int m = 0; m < 2; ++m) a_base[m] = X_BYTES + lrow + (((wm * 2 + m) ^ lj) << 6);      // dy: 32 pixels per row
#pragma unroll
  for (int r = 0; r < 4; ++r) b_base[r] = lrow + ((wn ^ ((lj + r) & 3)) << 6);              // x: pixel & 3 = (lj + r) & 3

  f32x16_t acc[2][3];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  if (t_begin < t_end) {
#pragma unroll
    for (int i = 0; i < NI; ++i) dma(i, 0);
    next_tile();
    ssa_wait_vm_barrier<0, 0>();
  }
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    // the next tile's DMAs (into the buffer last read in tile t - 1) are issued one per k-step BEHIND that k-step's
    // MFMAs: an LDS DMA costs its wave 60..185 clocks of issue, which the six queued MFMAs (192 clocks) cover --
    // nine of them in front of the loop were 0.5 us per tile during which the SIMD had no MFMA to run
    const bool more = t + 1 < t_end;
    if (more) dma(0, buf ^ 1);
    const unsigned char* Tb = smem + buf * BUF_BYTES;
    // 16 pixels (half a tile row) per k-step; the fragments of k-step ks + 1 are read while the MFMAs of ks run
    bf16x8_t af[2][2], bfr[2][3];
    auto rd = [&](int ks, int slot) {
      const int ty = ks >> 1, kp = (ks & 1) * 16;
#pragma unroll
      for (int m = 0; m < 2; ++m) af[slot][m] = tr8(Tb + a_base[m] + (ty * TW + kp) * 256, 256);
#pragma unroll
      for (int j = 0; j < 3; ++j) bfr[slot][j] = tr8(Tb + b_base[(2 * ty + j) & 3] + (ty * XW + kp + j) * 256, 256);
    };
    rd(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) rd(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[m][j] = ssa_mfma32(af[ks & 1][m], bfr[ks & 1][j], acc[m][j]);
      if (more) dma(ks + 1, buf ^ 1);
      if (ks + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    if (more) next_tile();
    ssa_wait_vm_barrier<0, 0>();               // the next tile has landed; everyone is done reading this one
  }

  float* out = partial + (long)bx * cout_pad * Kflat;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ci = ci0 + wn * 32 + (lane & 31);
      if (ci >= Cin) continue;
      const long kcol = (long)(kh * 3 + j) * Cin + ci;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * 2 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < cout_pad) out[(long)co * Kflat + kcol] = acc[m][j][r];
      }
    }
  }
};

bool head_shape_ok(const ssa_conv_desc* d, int cout_pad) {
  if (!d || d->KH != d->KW || (d->KH != 3 && d->KH != 1)) return false;
  if (d->stride != 1 || d->dil != 1 || d->pad != d->KH / 2 || d->transposed) return false;
  if (d->Ho != d->H || d->Wo != d->W || d->ldx % 8 || d->Cin % 8 || cout_pad % 8) return false;
  // large-channel layers on large images only: small ones are latency bound either way
  return d->Cin >= 128 && cout_pad >= 64 && d->W >= 32 && (long)d->B * d->H * d->W >= 16384;
}

struct HeadPlan { int co_tiles, ci_tiles, wgs_y, G, tpw; long per_split; };

HeadPlan head_plan(const ssa_conv_desc& d, int cout_pad) {
  HeadPlan p;
  const int cx = d.KH == 3 ? 128 : 256;
  p.co_tiles = (cout_pad + 127) / 128;
  p.ci_tiles = (d.Cin + cx - 1) / cx;
  p.wgs_y = p.co_tiles * p.ci_tiles * (d.KH == 3 ? 3 : 1);
  const long tiles = (long)d.B * ((d.W + 31) / 32) * ((d.H + 3) / 4);
  long g = 256 / p.wgs_y;                         // one workgroup per CU
  if (g < 1) g = 1;
  if (g > tiles) g = tiles;
  p.tpw = (int)((tiles + g - 1) / g);
  p.G = (int)((tiles + p.tpw - 1) / p.tpw);
  p.per_split = (long)cout_pad * d.KH * d.KW * d.Cin * sizeof(float);
  return p;
}

void fill_args(WgradHeadArgs& a, const ssa_conv_desc& d, const HeadPlan& p, const void* x, const void* dy, int lddy,
               int cout_pad, float* partial) {
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.partial = partial;
  a.ldx = d.ldx; a.Cin = d.Cin; a.lddy = lddy; a.cout_pad = cout_pad; a.B = d.B; a.H = d.H; a.W = d.W;
  a.tiles_x = (d.W + 31) / 32; a.tiles_y = (d.H + 3) / 4; a.tiles_per_wg = p.tpw; a.ci_tiles = p.ci_tiles;
  a.G = p.G; a.wgs_y = p.wgs_y;
}

template <int KS>
int launch_head(const ssa_conv_desc& d, const HeadPlan& p, const void* x, const void* dy, int lddy,
                int cout_pad, float* partial, hipStream_t s) {
  constexpr int XW = KS == 3 ? 34 : 32, CX = KS == 3 ? 128 : 256;
  constexpr size_t lds = (size_t)4 * XW * trs(CX * 2) + (size_t)128 * trs(256);
  static_assert(lds <= 160 * 1024, "does not fit in LDS");
  WgradHeadArgs a;
  fill_args(a, d, p, x, dy, lddy, cout_pad, partial);
  return ssa::submit<ConvWgradHead<KS>>(a, p.G, p.wgs_y, lds, s);
}

}  // namespace

extern "C" {

int ssa_conv2d_wgrad_head_plan(const ssa_conv_desc* d, int cout_pad, int* nsplit, size_t* ws_bytes) {
  if (!nsplit || !ws_bytes || !head_shape_ok(d, cout_pad)) return SSA_EUNSUPPORTED;
  const HeadPlan p = head_plan(*d, cout_pad);
  *nsplit = p.G;
  *ws_bytes = (size_t)p.G * p.per_split;
  return SSA_OK;
}

int ssa_conv2d_wgrad_head(const ssa_conv_desc* dp, const void* x, const void* dy, int lddy, int cout_pad,
                          int nsplit, float* partial, void* stream) {
  if (!dp || !x || !dy || !partial) return SSA_EINVAL;
  if (!head_shape_ok(dp, cout_pad) || lddy % 8) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15u) return SSA_EINVAL;
  const HeadPlan p = head_plan(*dp, cout_pad);
  if (nsplit != p.G) return SSA_EINVAL;
  if (dp->KH == 3) {
    WgradHeadArgs a;
    fill_args(a, *dp, p, x, dy, lddy, cout_pad, partial);
    return ssa::submit<ConvWgradHead3>(a, p.G, p.wgs_y, ConvWgradHead3::LDS, (hipStream_t)stream);
  }
  return launch_head<1>(*dp, p, x, dy, lddy, cout_pad, partial, (hipStream_t)stream);
}

}  // extern "C"
