// Halo-chunk implicit GEMM for the large-channel convolutions of the OCR /
// attention heads (gfx950 / MI355X): conv3x3_ocr 720->512 (28 % of the step's
// FLOPs), attn 512->256 and 256->256, the 1x1 convs 1024->512 / 720->720 /
// 512->256 and their data gradients (network/ocrnet.py:54-58,
// network/utils.py:348-357, network/ocr_utils.py:68-93,142-147).
//
// Workgroup = 8 wave64s, output tile 256 pixels (8 rows x 32 columns) x 128
// output channels; each wave owns 64 pixels x 64 channels (2x2 MFMA 32x32x16
// tiles).  K runs over channel chunks of CK (48 or 64) input channels:
//   * A operand: the (8+2)x(32+2) input HALO tile of the chunk is staged in LDS
//     once and serves all 9 taps (each tap is the same image at a shifted pixel
//     offset) -- 9x less L2->LDS traffic for A than an im2col gather, which is
//     what makes a 128-wide N tile affordable.  It arrives by DMA as well
//     (global_load_lds_dwordx4, no registers, no ds_write), pixel-major with the
//     16-byte pieces of a pixel XOR-swizzled so that ds_read_b128 is conflict free
//     without padding (a DMA fills 64 CONSECUTIVE slots).  3x3: double buffered per
//     chunk; 1x1: in the same ring as the filter.
//   * B operand: the filter in MFMA-fragment order ([n-block][k-step][lane][8],
//     ssa_pack_filter mode 2/3), streamed per (chunk, tap) stage straight into a
//     RING of LDS buffers with global_load_lds_dwordx4 (1 KiB per wave
//     instruction), RING - 1 stages ahead of the MFMAs.  A stage is only 12..16
//     MFMAs per wave (~1000 clocks per SIMD with two waves); an L2 hit takes about
//     twice that under load, so with one stage of lookahead (round 2: double
//     buffer + __syncthreads, whose fence waits vmcnt(0)) every stage ended in a
//     wait for its successor's filter: 0.85 us per stage, 38 % of the MFMA peak.
//     The stage now ends in  s_waitcnt vmcnt(K) ; s_barrier  with K = the number
//     of vector-memory operations issued after the NEXT stage's filter (memory
//     operations retire in order): only that one has to have landed.  Every wave
//     issues the same operations in every stage (out-of-range prefetches are
//     clamped, not skipped), so K is a compile-time constant per tap.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kStatReplicasG = 8;   // must equal conv_tile.hip's kStatReplicas

__device__ uint4 g_zero_piece;      // 16 zero bytes: what a halo piece outside the image loads

struct HaloArgs {
  const bf16_t* x; const uint4* wfrag; const float* bias; void* y; double* stats;
  int ldx, Cin, ldy, out_f32, B, H, W, Cout, nb_total, tiles_x, tiles_y;
};

// Epilogue shared by the two kernels: bias, bf16 rounding, BatchNorm partial sums of the ROUNDED values, the tile
// staged through LDS for 16-byte row stores (or plain fp32 stores for the logit convs).
struct HaloTile { int bx, nb0, b, y0, x0, wm, wn; };

__device__ __forceinline__ void halo_epilogue(const HaloArgs& a, const HaloTile& k, f32x16_t (&acc)[2][2], unsigned char* smem) {
  constexpr int NB = 4, TW = 32, BM = 256;
  const float* __restrict__ bias = a.bias;
  void* __restrict__ yv = a.y;
  double* __restrict__ stats = a.stats;
  const int ldy = a.ldy, out_f32 = a.out_f32, H = a.H, W = a.W, Cout = a.Cout;
  const int bx = k.bx, nb0 = k.nb0, b = k.b, y0 = k.y0, x0 = k.x0, wm = k.wm, wn = k.wn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int n_base = nb0 * 32 + wn * 64;
  if (out_f32) {
    float* y = reinterpret_cast<float*>(yv) + (long)b * H * W * ldy;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n_base + ni * 32 + (lane & 31);
      const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * 2 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int oy = y0 + row / TW, ox = x0 + row % TW;
          if (oy < H && ox < W && n < Cout) y[((long)oy * W + ox) * ldy + n] = acc[mi][ni][r] + bv;
        }
    }
    return;
  }
  constexpr int LDC = NB * 32 + 8;
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
  float* red = reinterpret_cast<float*>(smem + (size_t)BM * LDC * 2);   // [4 wm][2][128]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = wn * 64 + ni * 32 + (lane & 31);
    const int n = nb0 * 32 + col;
    const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
    float sacc = 0.f, qacc = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * 2 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const bf16_t o = f2bf(acc[mi][ni][r] + bv);
        Cs[row * LDC + col] = o;
        if (stats != nullptr) {
          const float f = (y0 + row / TW < H && x0 + row % TW < W) ? bf2f(o) : 0.f;
          sacc += f;
          qacc += f * f;
        }
      }
    if (stats != nullptr) {
      sacc += __shfl_xor(sacc, 32, 64);
      qacc += __shfl_xor(qacc, 32, 64);
      if (lane < 32) {
        red[(wm * 2 + 0) * 128 + col] = sacc;
        red[(wm * 2 + 1) * 128 + col] = qacc;
      }
    }
  }
  __syncthreads();
  if (stats != nullptr) {
    double* st = stats + (long)(bx % kStatReplicasG) * 2 * Cout;
    if (tid < 256) {
      const int which = tid >> 7, col = tid & 127;
      const int n = nb0 * 32 + col;
      if (n < Cout) {
        const float v = (red[(0 * 2 + which) * 128 + col] + red[(1 * 2 + which) * 128 + col]) +
                        (red[(2 * 2 + which) * 128 + col] + red[(3 * 2 + which) * 128 + col]);
        atomicAdd(&st[which * Cout + n], (double)v);
      }
    }
  }
  bf16_t* yb = reinterpret_cast<bf16_t*>(yv) + (long)b * H * W * ldy;
  constexpr int CPR = NB * 4;
  for (int idx = tid; idx < BM * CPR; idx += 512) {
    const int row = idx / CPR, cp = idx - row * CPR;
    const int oy = y0 + row / TW, ox = x0 + row % TW, n = nb0 * 32 + cp * 8;
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

// 1x1 convs: a stage is a whole chunk -- 32 KiB of input tile + 16 KiB of filter per 12..16 MFMAs per wave, the
// L1 -> LDS path (64 B/clk/CU) is what bounds them, not the stage latency; they keep the register-staged, padded
// halo image and the double-buffered filter of round 2 (the DMA/swizzle pipeline of the 3x3 kernel measured
// 109 -> 132 us on 720->720 @ 256x256: its zero-source padding pieces and duplicate filter blocks cost L1 cycles).
template <int CK>
struct ConvHaloGemm1 {
  static constexpr int KS = 1;
  // CK = 48: the pipeline is 80 KiB of LDS -- two workgroups per CU when the kernel stays within 128 registers
  // (it did by itself in round 2; with 142 the 720->720 conv ran 105 -> 136 us)
  static constexpr int WPE = CK == 48 ? 4 : 2;
  typedef HaloArgs Args;
  static constexpr int NT = 512;
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const bf16_t* __restrict__ x = a.x;
  const uint4* __restrict__ wfrag = a.wfrag;
  const float* __restrict__ bias = a.bias;
  void* __restrict__ yv = a.y;
  double* __restrict__ stats = a.stats;
  const int ldx = a.ldx, Cin = a.Cin, ldy = a.ldy, out_f32 = a.out_f32, H = a.H, W = a.W, Cout = a.Cout;
  const int nb_total = a.nb_total, tiles_x = a.tiles_x, tiles_y = a.tiles_y;
  constexpr int NB = 4, TW = 32, TH = 8, R = KS / 2;
  constexpr int HW_ = TW + 2 * R, HH_ = TH + 2 * R;
  constexpr int PSB = CK * 2 + 16;
  constexpr int CP = CK / 8;
  constexpr int NPIECE = HH_ * HW_ * CP;
  constexpr int IT = (NPIECE + 511) / 512;
  constexpr int CST = CK / 16;                  // c-steps per chunk
  constexpr int TAPS = KS * KS;
  constexpr int HALO_BYTES = (HH_ * HW_ * PSB + 1023) / 1024 * 1024;
  constexpr int BST_BYTES = NB * CST * 1024;    // one (chunk, tap) stage of the filter
  SSA_DYN_LDS(unsigned char, smem);
  unsigned char* Hs = smem;                     // [2][HALO_BYTES]
  unsigned char* Bs = smem + 2 * HALO_BYTES;    // [2][BST_BYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 x 2 waves
  int bid = bx;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int nb0 = by * NB;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int nchunk = Cin / CK;
  const int csteps = Cin / 16;                  // c-steps per tap in the packed filter
  const int ksteps = TAPS * csteps;
  const int nstage = nchunk * TAPS;

  // this thread's halo pieces: global element offset (or -1) and LDS byte offset
  long g_off[IT];
  int l_off[IT];
  const bf16_t* xb = x + (long)b * H * W * ldx;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int piece = tid + i * 512;
    const int pix = piece / CP, cp = piece - pix * CP;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    const int iy = y0 - R + hy, ix = x0 - R + hx;
    const bool ok = piece < NPIECE && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    g_off[i] = ok ? ((long)iy * W + ix) * ldx + cp * 8 : -1;
    l_off[i] = piece < NPIECE ? pix * PSB + cp * 16 : -1;
  }
  uint4 hv[IT];
  auto halo_load = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < IT; ++i)
      hv[i] = g_off[i] >= 0 ? *reinterpret_cast<const uint4*>(xb + g_off[i] + chunk * CK)
                            : make_uint4(0, 0, 0, 0);
  };
  auto halo_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < IT; ++i)
      if (l_off[i] >= 0) *reinterpret_cast<uint4*>(Hs + buf * HALO_BYTES + l_off[i]) = hv[i];
  };
  // filter stage (chunk c, tap t): NB x CST fragment blocks of 1 KiB, 8 waves share them
  auto filt_stage = [&](int c, int t, int buf) {
    constexpr int NFRAG = NB * CST;
#pragma unroll
    for (int f = 0; f < (NFRAG + 7) / 8; ++f) {
      const int fi = f * 8 + wave;
      if (fi < NFRAG) {
        const int nb = fi / CST, j = fi - nb * CST;
        const int nbg = min(nb0 + nb, nb_total - 1);
        const uint4* src = wfrag + ((long)nbg * ksteps + t * csteps + c * CST + j) * 64 + lane;
        ssa_glds16(src, Bs + buf * BST_BYTES + fi * 1024);
      }
    }
  };

  int a_off[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = (wm * 2 + mi) * 32 + (lane & 31);
    const int ty = m / TW, tx = m - ty * TW;
    a_off[mi] = (ty * HW_ + tx) * PSB + (lane >> 5) * 16;
  }
  f32x16_t acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  halo_load(0);
  filt_stage(0, 0, 0);
  halo_store(0);
  __syncthreads();

  int c = 0, t = 0;
  for (int s = 0; s < nstage; ++s) {
    int cn = c, tn = t + 1;
    if (tn == TAPS) { tn = 0; cn = c + 1; }
    const bool more = s + 1 < nstage;
    if (more) filt_stage(cn, tn, (s + 1) & 1);
    const bool fetch_halo = (t == 0) && (c + 1 < nchunk);
    if (fetch_halo) halo_load(c + 1);
    const int kh = t / KS, kw = t - kh * KS;
    const unsigned char* Ha = Hs + (c & 1) * HALO_BYTES + (kh * HW_ + kw) * PSB;
    const unsigned char* Bc = Bs + (s & 1) * BST_BYTES + lane * 16;
#pragma unroll
    for (int j = 0; j < CST; ++j) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        af[mi] = *reinterpret_cast<const bf16x8_t*>(Ha + a_off[mi] + j * 32);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        bfr[ni] = *reinterpret_cast<const bf16x8_t*>(Bc + ((wn * 2 + ni) * CST + j) * 1024);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = ssa_mfma32(af[mi], bfr[ni], acc[mi][ni]);
    }
    if (fetch_halo) halo_store((c + 1) & 1);
    __syncthreads();
    c = cn; t = tn;
  }
  HaloTile k = {bx, nb0, b, y0, x0, wm, wn};
  halo_epilogue(a, k, acc, smem);
  }
};

// 3x3 convs.
template <int CK, int RING>
struct ConvHaloGemm3 {
  typedef HaloArgs Args;
  static constexpr int NT = 512;
  static constexpr int KS = 3, NB = 4, TW = 32, TH = 8, R = 1;
  static constexpr int HW_ = TW + 2 * R, HH_ = TH + 2 * R;
  static constexpr int CP = CK / 8;                    // 16-byte pieces of a pixel's chunk that carry data
  static constexpr int NPIECE = HH_ * HW_ * 8;         // LDS slots: 8 per pixel (128 bytes; CK = 48 leaves 2 unused)
  static constexpr int IT = (NPIECE + 511) / 512;      // halo DMAs per wave and chunk
  static constexpr int HWI = (NPIECE + 63) / 64;       // wave instructions that land inside the halo image
  static constexpr int HALO_BYTES = HWI * 1024;
  static constexpr int CST = CK / 16;                  // c-steps per chunk
  static constexpr int TAPS = 9;
  static constexpr int BST_BYTES = NB * CST * 1024;    // one (chunk, tap) stage of the filter
  static constexpr int NFRAG = NB * CST;               // 1 KiB fragment blocks per stage, dealt to the 8 waves
  static constexpr int NF = (NFRAG + 7) / 8;           // filter DMAs per wave and stage
  static constexpr int D = RING - 1;                   // filter stages in flight
  static constexpr int RS = CST % 2 == 0 ? 2 : 3;      // fragment register ring (CST % RS == 0: slot 0 starts every stage)
  static constexpr int DUMP_BYTES = IT * 8 > HWI ? 1024 : 0;   // where the wave instructions past the image land
  static constexpr size_t PIPE_BYTES = 2 * (size_t)HALO_BYTES + (size_t)RING * BST_BYTES + DUMP_BYTES;
  static_assert(RING >= 4, "stage s reads ahead into stage s + 1: its filter must have landed one barrier earlier");

  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const bf16_t* __restrict__ x = a.x;
  const uint4* __restrict__ wfrag = a.wfrag;
  const int ldx = a.ldx, Cin = a.Cin, H = a.H, W = a.W;
  const int nb_total = a.nb_total, tiles_x = a.tiles_x, tiles_y = a.tiles_y;
  SSA_DYN_LDS(unsigned char, smem);
  unsigned char* Hs = smem;                                 // [2][HALO_BYTES]
  unsigned char* Bs = smem + 2 * HALO_BYTES;                // [RING][BST_BYTES]
  unsigned char* dump = Bs + RING * BST_BYTES;              // [DUMP_BYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;      // 4 x 2 waves
  int bid = bx;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int nb0 = by * NB;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int nchunk = Cin / CK;
  const int csteps = Cin / 16;                  // c-steps per tap in the packed filter
  const int ksteps = TAPS * csteps;

  // Halo image in LDS: pixel-major, 128 bytes per pixel, the 16-byte piece q of pixel p in slot  q ^ ((p >> 1) & 7)
  // -- written by DMA (a wave instruction fills 64 consecutive slots; WHICH piece of the chunk a lane fetches is
  // free), read with ds_read_b128: the 16 lanes the LDS services together read the same q of 16 pixels with
  // distinct p mod 16, i.e. 16 distinct slots of the 256-byte bank row.  A thread's IT pieces: source pointer and
  // channel step per chunk; pieces outside the image or the chunk fetch a zero block (step 0).
  const bf16_t* g_ptr[IT];
  int g_step[IT];
  const bf16_t* xb = x + (long)b * H * W * ldx;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int piece = tid + i * 512;
    const int pix = piece >> 3, q = (piece & 7) ^ ((pix >> 1) & 7);
    const int hy = pix / HW_, hx = pix - hy * HW_;
    const int iy = y0 - R + hy, ix = x0 - R + hx;
    const bool ok = piece < NPIECE && q < CP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    g_ptr[i] = ok ? xb + ((long)iy * W + ix) * ldx + q * 8 : reinterpret_cast<const bf16_t*>(&g_zero_piece);
    g_step[i] = ok ? CK : 0;
  }
  auto halo_dma = [&](int chunk, int buf) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int wi = i * 8 + wave;              // this wave instruction's 1 KiB of the image (or the dump block)
      ssa_glds16(g_ptr[i] + chunk * g_step[i], wi < HWI ? Hs + buf * HALO_BYTES + wi * 1024 : dump);
    }
  };
  // filter stage (chunk c, tap t): NB x CST fragment blocks of 1 KiB dealt to the 8 waves.  Every wave issues NF
  // DMAs: when NFRAG is not a multiple of 8 (CK = 48: 12 blocks) the waves without a block of their own fetch one
  // another wave also fetches (same bytes to the same place) -- the stage-end wait needs the same count in every wave.
  auto filt_stage = [&](int c, int t, int slot) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int fi = f * 8 + wave;
      if (NFRAG % 8 != 0 && f == NF - 1 && fi >= NFRAG) fi -= 8;
      const int nb = fi / CST, j = fi - nb * CST;
      const int nbg = min(nb0 + nb, nb_total - 1);
      const uint4* src = wfrag + ((long)nbg * ksteps + t * csteps + c * CST + j) * 64 + lane;
      ssa_glds16(src, Bs + slot * BST_BYTES + fi * 1024);
    }
  };

  // A fragment of this lane: output pixel (row wm * 2 + mi, column lane & 31) of the tile, k half lane >> 5
  int a_pix[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) a_pix[mi] = (wm * 2 + mi) * HW_ + (lane & 31);
  const int a_half = (lane >> 5) << 4;
  f32x16_t acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // Fragment registers: a ring of RS c-steps that runs ACROSS the stage barriers -- the last c-step of a stage
  // reads the first fragments of the next stage (its halo image and filter are visible since the previous barrier),
  // so the MFMAs resume right behind the barrier instead of behind an LDS round trip of all eight waves at once.
  bf16x8_t af[RS][2], bfr[RS][2];
  auto rd = [&](const unsigned char* Hbuf, const int tap_off, const unsigned char* Bc, const int j, const int sl) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int p = a_pix[mi] + tap_off;
      const int adr = (p << 7) | (((p << 3) & 0x70) ^ a_half);     // piece 2j + half of pixel p: slot ^ ((p >> 1) & 7)
      af[sl][mi] = *reinterpret_cast<const bf16x8_t*>(Hbuf + (adr ^ (j << 5)));
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bfr[sl][ni] = *reinterpret_cast<const bf16x8_t*>(Bc + ((wn * 2 + ni) * CST + j) * 1024);
  };

  // Stage s = (chunk c, tap t) issues the filter DMAs of stage s + D (and at tap 0 the next chunk's halo) and ends in
  //     s_waitcnt vmcnt(K) lgkmcnt(4) ; s_barrier
  // K = number of DMAs this wave issued after those of stage s + 2: memory operations retire in order, so the
  // operands of stage s + 2 have landed -- for every wave once all are through the barrier, i.e. stage s + 1 may
  // read ahead into them.  Every wave has then also finished reading stage s (lgkmcnt(4): all LDS reads but the
  // four read-ahead ones have returned), whose filter slot stage s + 1 hands to the DMAs of stage s + 1 + D.
  const int last = nchunk - 1;
  int slot = 0;                                 // filter ring slot of the current stage
  halo_dma(0, 0);
#pragma unroll
  for (int q = 0; q < D; ++q) filt_stage(min(q / TAPS, last), q % TAPS, q);
  ssa_wait_vm_barrier<0, 0>();
  rd(Hs, 0, Bs + lane * 16, 0, 0);
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
  for (int c = 0; c < nchunk; ++c) {
    auto tap = [&](auto t_) {
      constexpr int t = decltype(t_)::value;
      constexpr int td = (t + D) % TAPS, cd_inc = (t + D) / TAPS;
      int sd = slot + D; if (sd >= RING) sd -= RING;
      int sn = slot + 1; if (sn >= RING) sn -= RING;
      filt_stage(min(c + cd_inc, last), td, sd);
      if (t == 0) halo_dma(min(c + 1, last), (c + 1) & 1);        // read from stage (c, 8)'s last c-step on
      constexpr int tap_off = (t / 3) * HW_ + t % 3;
      constexpr int tn = (t + 1) % TAPS, tap_off_n = (tn / 3) * HW_ + tn % 3;
      const unsigned char* Hc = Hs + (c & 1) * HALO_BYTES;
      const unsigned char* Hn = Hs + ((t + 1 == TAPS ? c + 1 : c) & 1) * HALO_BYTES;
      const unsigned char* Bc = Bs + slot * BST_BYTES + lane * 16;
      const unsigned char* Bn = Bs + sn * BST_BYTES + lane * 16;
#pragma unroll
      for (int j = 0; j < CST; ++j) {
        if (j + 1 < CST) rd(Hc, tap_off, Bc, j + 1, (j + 1) % RS);
        else rd(Hn, tap_off_n, Bn, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = ssa_mfma32(af[j % RS][mi], bfr[j % RS][ni], acc[mi][ni]);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      ssa_wait_vm_barrier<(D - 2) * NF + (t < D - 1 ? IT : 0), 4>();
      slot = sn;
    };
    tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
    tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
  }
  ssa_wait_vm_barrier<0, 0>();                  // the clamped prefetches of the last stages land before LDS is reused
  HaloTile k = {bx, nb0, b, y0, x0, wm, wn};
  halo_epilogue(a, k, acc, smem);
  }
};

template <class K>
int launch_halo(const ssa_conv_desc& d, size_t pipe, const void* x, const void* wfrag, const float* bias, void* y,
                double* stats, hipStream_t s) {
  constexpr size_t stage = (size_t)256 * (128 + 8) * 2 + 4 * 2 * 128 * sizeof(float);
  const size_t lds = pipe > stage ? pipe : stage;
  HaloArgs a;
  a.x = (const bf16_t*)x; a.wfrag = (const uint4*)wfrag; a.bias = bias; a.y = y; a.stats = stats;
  a.ldx = d.ldx; a.Cin = d.Cin; a.ldy = d.ldy; a.out_f32 = d.out_f32; a.B = d.B; a.H = d.H; a.W = d.W;
  a.Cout = d.Cout; a.nb_total = (d.Cout + 31) / 32;
  a.tiles_x = (d.W + 31) / 32; a.tiles_y = (d.H + 7) / 8;
  return ssa::submit<K>(a, a.tiles_x * a.tiles_y * d.B, (a.nb_total + 3) / 4, lds, s);
}

template <int CK, int RING>
int launch_halo3(const ssa_conv_desc& d, const void* x, const void* wfrag, const float* bias, void* y, double* stats,
                 hipStream_t s) {
  typedef ConvHaloGemm3<CK, RING> K;
  static_assert(K::PIPE_BYTES <= 160 * 1024, "does not fit in LDS");
  return launch_halo<K>(d, K::PIPE_BYTES, x, wfrag, bias, y, stats, s);
}

template <int CK>
int launch_halo1(const ssa_conv_desc& d, const void* x, const void* wfrag, const float* bias, void* y, double* stats,
                 hipStream_t s) {
  constexpr size_t halo = ((size_t)8 * 32 * (CK * 2 + 16) + 1023) / 1024 * 1024;
  constexpr size_t bst = (size_t)4 * (CK / 16) * 1024;
  return launch_halo<ConvHaloGemm1<CK>>(d, 2 * halo + 2 * bst, x, wfrag, bias, y, stats, s);
}

int pick_ck(int Cin) {
  if (Cin % 64 == 0) return 64;
  if (Cin % 48 == 0) return 48;
  return 0;
}

}  // namespace

extern "C" {

int ssa_conv2d_halo_supported(const ssa_conv_desc* d) {
  if (!d) return 0;
  if (d->KH != d->KW || (d->KH != 3 && d->KH != 1)) return 0;
  if (d->stride != 1 || d->dil != 1 || d->transposed || d->pad != d->KH / 2) return 0;
  if (d->Ho != d->H || d->Wo != d->W) return 0;
  if (d->ldx % 8 || (!d->out_f32 && (d->Cout % 8 || d->ldy % 8))) return 0;
  // small problems (the 192/384-channel trunk branches at <= 64x64: 32 workgroups) stay on
  // conv_tile / the K-pipelined igemm, which spread them over more workgroups
  if (d->Cin < 192 || d->Cout < 64 || d->W < 32 || (long)d->B * d->H * d->W < 16384) return 0;
  return pick_ck(d->Cin) != 0;
}

int ssa_conv2d_halo(const ssa_conv_desc* dp, const void* x, const void* w_frag, const float* bias,
                    void* y, double* stats, void* stream) {
  if (!dp || !x || !w_frag || !y) return SSA_EINVAL;
  if (!ssa_conv2d_halo_supported(dp)) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_frag)) & 15u)
    return SSA_EINVAL;
  if (stats && dp->out_f32) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  // the large 1x1 problems go to the 256 x 256 tile (conv_gemm_wide.hip); SSA_GEMM_WIDE=0: all stay here
  static const bool wide_on = !(getenv("SSA_GEMM_WIDE") && atoi(getenv("SSA_GEMM_WIDE")) == 0);
  if (wide_on && d.KH == 1 && ssa_conv2d_gemm_wide_supported(dp)) return ssa_conv2d_gemm_wide(dp, x, w_frag, bias, y, stats, stream);
  // the 3x3 problems go to the register-fed geometry (conv_halo_reg.hip); SSA_HALO3_REG=0: all stay here
  static const bool reg_on = !(getenv("SSA_HALO3_REG") && atoi(getenv("SSA_HALO3_REG")) == 0);
  if (reg_on && d.KH == 3 && ssa_conv2d_halo_reg_supported(dp)) return ssa_conv2d_halo_reg(dp, x, w_frag, bias, y, stats, stream);
  const int ck = pick_ck(d.Cin);
  if (d.KH == 3) {
    if (ck == 64) return launch_halo3<64, 4>(d, x, w_frag, bias, y, stats, s);
    return launch_halo3<48, 4>(d, x, w_frag, bias, y, stats, s);
  }
  if (ck == 64) return launch_halo1<64>(d, x, w_frag, bias, y, stats, s);
  return launch_halo1<48>(d, x, w_frag, bias, y, stats, s);
}

}  // extern "C"
