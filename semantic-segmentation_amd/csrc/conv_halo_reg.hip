// 3x3 convolutions of the OCR / attention heads, second geometry (gfx950 / MI355X): conv3x3_ocr 720->512 (28 % of the
// step's FLOPs), the scale-attention head 512->256 / 256->256, and their data gradients (network/ocrnet.py:54-58,
// network/utils.py:348-357).
//
// conv_halo_gemm.hip's ConvHaloGemm3 gives a wave 64 pixels x 64 channels and streams the filter through an LDS ring:
// every MFMA wants one ds_read_b128, every (chunk, tap) stage of 12-16 MFMAs per wave ends in a workgroup barrier,
// and the kernel sits at 0.43-0.46 of the MFMA peak with `mfma_busy` 0.49-0.58 (profiles/r05_pmc.txt).  Here
//   * a wave owns 128 pixels x 64 channels = 4 x 2 MFMA 32x32x16 tiles (128 accumulator registers): a 16-channel
//     k-step is 4 A-fragment reads for 8 MFMAs -- half an LDS read per MFMA;
//   * the FILTER NEVER TOUCHES LDS.  It lies in MFMA-fragment order [n-block][k-step][lane][8] (ssa_pack_filter mode
//     2 / 3): a wave's B fragment is one contiguous 1 KiB block, i.e. ONE global_load_dwordx4 per lane straight into
//     the registers the MFMA reads.  A tap's fragments (2 n-blocks x CK / 16 k-steps) are fetched one tap (1,500-2,000
//     clocks) ahead into the other half of a register double buffer; the compiler's own vmcnt bookkeeping orders them.
//     The four waves of a workgroup own four DIFFERENT 64-channel slices, so no filter byte is fetched twice by a
//     workgroup and nothing about the filter needs a barrier;
//   * the input HALO tile (6 x 34 pixels of a CK-channel chunk, XOR-swizzled pixel-major image as in conv_halo_gemm.hip)
//     arrives by LDS DMA, double buffered per chunk -- so the ONLY workgroup barrier of the main loop is the one per
//     chunk: 216-288 MFMAs per wave between barriers instead of 12-16;
//   * workgroup = 4 waves (1 x 4), tile 128 pixels (4 rows x 32) x 256 channels, 52 KiB of LDS for the pipeline: two
//     workgroups per CU with independent barriers (8 waves per CU, 256 registers each).
// Pieces outside the image are never fetched: both halo buffers are zeroed once and the DMA of such a piece is masked
// off (EXEC), so the slot keeps its zeros for every chunk.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"
#include <stdlib.h>
#include <type_traits>

#ifndef SSA_HREG_DRAIN      // experiment builds: -DSSA_HREG_DRAIN=0 drains the filter ring at every chunk barrier
#define SSA_HREG_DRAIN (-1)
#endif

namespace {

constexpr int kStatReplicasR = 8;   // must equal conv_tile.hip's kStatReplicas

struct HaloRegArgs {
  const bf16_t* x; const uint4* wfrag; const float* bias; bf16_t* y; double* stats;
  int ldx, Cin, ldy, B, H, W, Cout, nb_total, tiles_x, tiles_y;
};

template <int CK>
struct ConvHaloReg3 {
  typedef HaloRegArgs Args;
  static constexpr int NT = 256;
  static constexpr int WPE = 2;                        // two workgroups per CU: 256 registers per wave
  static constexpr int NB = 8, TW = 32, TH = 4, BM = TW * TH;
  static constexpr int HW_ = TW + 2, HH_ = TH + 2;
  static constexpr int CP = CK / 8;                    // 16-byte pieces of a pixel's chunk that carry data
  static constexpr int NPIECE = HH_ * HW_ * 8;         // LDS slots: 8 per pixel (128 bytes; CK = 48 leaves 2 unused)
  static constexpr int IT = (NPIECE + NT - 1) / NT;    // halo DMAs per thread and chunk
  static constexpr int HWI = (NPIECE + 63) / 64;       // wave instructions that land inside the halo image
  static constexpr int HALO_BYTES = HWI * 1024;
  static constexpr int CST = CK / 16;                  // k-steps per chunk and tap
  static constexpr int TAPS = 9;
  static constexpr size_t PIPE_BYTES = 2 * (size_t)HALO_BYTES;
  static constexpr int LDC = NB * 32 + 8;              // epilogue staging: [128][264] elements
  static constexpr size_t EPI_BYTES = (size_t)BM * LDC * 2;
  static constexpr size_t LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  static_assert(CK == 64 || CK == 48, "chunk width");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int /*gx*/) {
  const uint4* __restrict__ wfrag = a.wfrag;
  const int ldx = a.ldx, Cin = a.Cin, H = a.H, W = a.W, nb_total = a.nb_total;
  const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;
  SSA_DYN_LDS(unsigned char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);     // the wave's 64-channel slice
  int bid = bx;
  const int tx_i = bid % tiles_x; bid /= tiles_x;
  const int ty_i = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int nb0 = by * NB;
  const int x0 = tx_i * TW, y0 = ty_i * TH;
  const int nchunk = Cin / CK;
  const int csteps = Cin / 16;                  // k-steps per tap in the packed filter
  const int ksteps = TAPS * csteps;

  // ---- halo DMA: thread t owns pieces t, t + 256, ...: LDS slot (piece & 7) of halo pixel (piece >> 3), which holds
  // the channel piece q = slot ^ ((pixel >> 1) & 7) of the chunk.  Byte offset from the image's first element; pieces
  // outside the image / the chunk are masked off for good (bit i of `dma_ok` clear).
  const bf16_t* xb = a.x + (long)b * H * W * ldx;
  unsigned g_off[IT];
  unsigned dma_ok = 0;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int piece = tid + i * NT;
    const int pix = piece >> 3;
    const int hy = pix / HW_, hx = pix - hy * HW_;
    const int q = (piece & 7) ^ ((hx >> 1) & 7);
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = piece < NPIECE && q < CP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    g_off[i] = ok ? (unsigned)((((long)iy * W + ix) * ldx + q * 8) * 2) : 0u;
    dma_ok |= (ok ? 1u : 0u) << i;
  }
  const unsigned lds0 = ssa_lds_addr(smem);
  auto halo_dma = [&](int chunk, int buf) {
    const bf16_t* src = xb + chunk * CK;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      // wave instruction (i * 4 + wn) fills 64 consecutive slots
      if ((dma_ok >> i) & 1u) ssa_glds16_untracked_m0(src, g_off[i], lds0 + buf * HALO_BYTES + (i * 4 + wn) * 1024);
    }
  };

  // ---- zero both halo buffers once (the slots the DMA never writes are the conv's zero padding)
  for (int o = tid * 16; o < 2 * HALO_BYTES; o += NT * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- filter: this wave's two n-blocks, one 1 KiB fragment block per (n-block, k-step): wave-uniform base + 16 * lane
  const unsigned char* wq[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int nbg = min(nb0 + wn * 2 + ni, nb_total - 1);
    wq[ni] = reinterpret_cast<const unsigned char*>(wfrag + (long)nbg * ksteps * 64);
  }
  const unsigned lane16 = lane * 16;
  // Filter ring of R slots, one per k-step (two fragments each): the slot a k-step frees is reloaded at once with the
  // k-step R ahead in consumption order -- R - 1 k-steps (~2,500 clocks at R = 6) before its MFMAs.  A filter row
  // (3 taps x CST k-steps, the unrolled unit) is a multiple of R, so slot numbers are compile-time.
  constexpr int R = CST == 4 ? 6 : 3;
  constexpr int ROWK = 3 * CST;                 // k-steps per filter row
  static_assert(ROWK % R == 0 && R <= ROWK, "ring period");
  bf16x8_t bq[R][2];
  auto load_b = [&](const int slot, const int kstep) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
      bq[slot][ni] = *reinterpret_cast<const bf16x8_t*>(wq[ni] + (long)kstep * 1024 + lane16);
  };

  // A fragment of this lane: output pixel (row mi, column lane & 31) of the tile, k half lane >> 5
  const int a_col = lane & 31;
  const int a_half = (lane >> 5) << 4;
  f32x16_t acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // Halo image: pixel-major, 128 bytes per pixel, the 16-byte piece q of the pixel in halo column hx in slot
  // q ^ ((hx >> 1) & 7): the 16 lanes a ds_read_b128 services together read one q of 16 consecutive columns = 16 distinct
  // slots of the 256-byte bank row (HW_ is even: pixel parity = column parity).  The swizzle depends on the COLUMN only,
  // so a fragment address is  (one of three per-lane bases, by kw) ^ (k-step << 5)  +  the row offset.
  bf16x8_t af[4];
  int a_base[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int hx = a_col + kw;
    a_base[kw] = (hx << 7) | ((((hx >> 1) & 7) << 4) ^ a_half);
  }
  auto rd1 = [&](const unsigned char* Hrow, const int kw, const int j, const int mi) {
    af[mi] = *reinterpret_cast<const bf16x8_t*>(Hrow + mi * (HW_ * 128) + (a_base[kw] ^ (j << 5)));
  };

  // Main loop: chunk c, filter row kh are run-time loops; the 3 taps of a filter row x CST k-steps x 8 MFMAs are unrolled.
  // Position pos = kw * CST + j of filter row kh of chunk c is k-step (kh * 3 + kw) * csteps + c * CST + j of the packed filter.
  halo_dma(0, 0);
#pragma unroll
  for (int q = 0; q < R; ++q) load_b(q, (q / CST) * csteps + q % CST);
  ssa_wait_vm_barrier<0, 0>();
  for (int c = 0; c < nchunk; ++c) {
    const unsigned char* Hc = smem + (c & 1) * HALO_BYTES;
    if (c + 1 < nchunk) halo_dma(c + 1, (c + 1) & 1);          // the next chunk's halo image
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) rd1(Hc, 0, 0, mi);
    // (a real loop: unrolled, the compiler recognises that row mi + 1 of filter row kh IS row mi of filter row kh + 1 and
    // keeps those fragments in registers instead of reading them again -- 200 bytes per lane of spills)
#pragma nounroll
    for (int kh = 0; kh < 3; ++kh) {
      const unsigned char* Hrow = Hc + kh * (HW_ * 128);
      const int krow = kh * 3 * csteps + c * CST;
      // first k-step of the row after this one: the next filter row, or the next chunk's first (clamped at the very
      // end: fetched, never used)
      const int krow_n = kh < 2 ? krow + 3 * csteps : min(c + 1, nchunk - 1) * CST;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int j = 0; j < CST; ++j) {
          const int pos = kw * CST + j, slot = pos % R;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = ssa_mfma32(af[mi], bq[slot][ni], acc[mi][ni]);
            // this row block's fragment of the NEXT k-step takes the register its two MFMAs have just read (behind the
            // last filter row the read is of no use: the chunk barrier follows and the new chunk's first fragments are
            // read behind it; it stays inside the workgroup's LDS)
            if (j + 1 < CST) rd1(Hrow, kw, j + 1, mi);
            else if (kw < 2) rd1(Hrow, kw + 1, 0, mi);
            else rd1(Hrow + HW_ * 128, 0, 0, mi);
          }
          const int pn = pos + R;
          if (pn < ROWK) load_b(slot, krow + (pn / CST) * csteps + pn % CST);
          else load_b(slot, krow_n + ((pn - ROWK) / CST) * csteps + (pn - ROWK) % CST);
#ifndef SSA_EMU
          // the k-step is the scheduling unit: left alone, the scheduler sinks a slot's reload to just in front of its
          // next use (shorter live range), which is the opposite of a prefetch
          // inside it: two MFMAs, then the LDS read of the row block they have just used, four times over; the filter
          // reloads behind the first pair
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
    }
    // every wave is through with this chunk's halo buffer (the chunk after next lands in it) and the next chunk's
    // image -- issued a whole chunk ago -- has landed for every wave: memory operations retire in order and the 2 R
    // newest are the filter ring's reloads, which stay in flight across the barrier
    ssa_wait_vm_barrier<SSA_HREG_DRAIN < 0 ? 2 * R : SSA_HREG_DRAIN, 0>();
  }

  // ---- epilogue: bias, rounding to 16 bits, BatchNorm partial sums of the ROUNDED values (a wave holds all 128
  // pixels of its channels: the sums leave from registers), the tile staged through LDS for 16-byte row stores
  const float* __restrict__ bias = a.bias;
  double* __restrict__ stats = a.stats;
  const int ldy = a.ldy, Cout = a.Cout;
  bf16_t* Cs = reinterpret_cast<bf16_t*>(smem);
  double* st = stats ? stats + (long)(bx % kStatReplicasR) * 2 * Cout : nullptr;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = wn * 64 + ni * 32 + (lane & 31);
    const int n = nb0 * 32 + col;
    const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
    float sacc = 0.f, qacc = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
      if (lane < 32 && n < Cout) {
        atomicAdd(&st[n], (double)sacc);
        atomicAdd(&st[Cout + n], (double)qacc);
      }
    }
  }
  __syncthreads();
  bf16_t* yb = a.y + (long)b * H * W * ldy;
  constexpr int CPR = NB * 4;                    // 16-byte pieces per tile row
  for (int idx = tid; idx < BM * CPR; idx += NT) {
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
};

int pick_ck(int Cin) {
  if (Cin % 64 == 0) return 64;
  if (Cin % 48 == 0) return 48;
  return 0;
}

bool reg_shape_ok(const ssa_conv_desc* d) {
  if (!d || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dil != 1 || d->transposed || d->pad != 1) return false;
  if (d->Ho != d->H || d->Wo != d->W || d->out_f32) return false;
  if (d->ldx % 8 || d->Cout % 8 || d->ldy % 8 || pick_ck(d->Cin) == 0) return false;
  // halo pieces are addressed by 32-bit byte offsets from the image's first element
  return (long)d->H * d->W * d->ldx * 2 < (1L << 32);
}

}  // namespace

extern "C" {

int ssa_conv2d_halo_reg_supported(const ssa_conv_desc* d) { return reg_shape_ok(d) ? 1 : 0; }

int ssa_conv2d_halo_reg(const ssa_conv_desc* dp, const void* x, const void* w_frag, const float* bias,
                        void* y, double* stats, void* stream) {
  if (!dp || !x || !w_frag || !y) return SSA_EINVAL;
  if (!reg_shape_ok(dp)) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_frag)) & 15u)
    return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  HaloRegArgs a;
  a.x = (const bf16_t*)x; a.wfrag = (const uint4*)w_frag; a.bias = bias; a.y = (bf16_t*)y; a.stats = stats;
  a.ldx = d.ldx; a.Cin = d.Cin; a.ldy = d.ldy; a.B = d.B; a.H = d.H; a.W = d.W; a.Cout = d.Cout;
  a.nb_total = (d.Cout + 31) / 32;
  a.tiles_x = (d.W + 31) / 32; a.tiles_y = (d.H + 3) / 4;
  const int gx = a.tiles_x * a.tiles_y * d.B, gy = (a.nb_total + 7) / 8;
  if (pick_ck(d.Cin) == 64) {
    typedef ConvHaloReg3<64> K;
    return ssa::submit<K>(a, gx, gy, K::LDS_BYTES, (hipStream_t)stream);
  }
  typedef ConvHaloReg3<48> K;
  return ssa::submit<K>(a, gx, gy, K::LDS_BYTES, (hipStream_t)stream);
}

}  // extern "C"
