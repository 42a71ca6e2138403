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
//     what makes a 128-wide N tile affordable (A+B ~ 21 B/clk/CU at full MFMA
//     rate instead of ~47).  Double buffered; the next chunk's global loads are
//     held in registers during one tap's MFMAs.
//   * B operand: the filter in MFMA-fragment order ([n-block][k-step][lane][8],
//     ssa_pack_filter mode 2/3), streamed per (chunk, tap) straight into LDS with
//     global_load_lds_dwordx4 (1 KiB per wave instruction), double buffered.
//   * one barrier per (chunk, tap) stage: 12..16 MFMAs per wave between barriers,
//     two waves per SIMD to cover each other's LDS/DMA waits.
// Halo pixel stride is CK*2+16 bytes (an odd number of 16-byte slots): the 16
// lanes ds_read_b128 services together read distinct slots.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"

namespace {

constexpr int kStatReplicasG = 8;   // must equal conv_tile.hip's kStatReplicas

struct HaloArgs {
  const bf16_t* x; const uint4* wfrag; const float* bias; void* y; double* stats;
  int ldx, Cin, ldy, out_f32, B, H, W, Cout, nb_total, tiles_x, tiles_y;
};

template <int CK, int KS>
struct ConvHaloGemm {
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
  constexpr int NB = 4, TW = 32, TH = 8, BM = 256, R = KS / 2;
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

  // ---- epilogue
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
};

template <int CK, int KS>
int launch_halo(const ssa_conv_desc& d, const void* x, const void* wfrag, const float* bias, void* y,
                double* stats, hipStream_t s) {
  constexpr int R = KS / 2;
  constexpr size_t halo = ((size_t)(8 + 2 * R) * (32 + 2 * R) * (CK * 2 + 16) + 1023) / 1024 * 1024;
  constexpr size_t bst = (size_t)4 * (CK / 16) * 1024;
  constexpr size_t stage = (size_t)256 * (128 + 8) * 2 + 4 * 2 * 128 * sizeof(float);
  constexpr size_t pipe = 2 * halo + 2 * bst;
  constexpr size_t lds = pipe > stage ? pipe : stage;
  static_assert(lds <= 160 * 1024, "does not fit in LDS");
  HaloArgs a;
  a.x = (const bf16_t*)x; a.wfrag = (const uint4*)wfrag; a.bias = bias; a.y = y; a.stats = stats;
  a.ldx = d.ldx; a.Cin = d.Cin; a.ldy = d.ldy; a.out_f32 = d.out_f32; a.B = d.B; a.H = d.H; a.W = d.W;
  a.Cout = d.Cout; a.nb_total = (d.Cout + 31) / 32;
  a.tiles_x = (d.W + 31) / 32; a.tiles_y = (d.H + 7) / 8;
  return ssa::submit<ConvHaloGemm<CK, KS>>(a, a.tiles_x * a.tiles_y * d.B, (a.nb_total + 3) / 4, lds, s);
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
  const int ck = pick_ck(d.Cin);
  if (d.KH == 3) {
    if (ck == 64) return launch_halo<64, 3>(d, x, w_frag, bias, y, stats, s);
    return launch_halo<48, 3>(d, x, w_frag, bias, y, stats, s);
  }
  if (ck == 64) return launch_halo<64, 1>(d, x, w_frag, bias, y, stats, s);
  return launch_halo<48, 1>(d, x, w_frag, bias, y, stats, s);
}

}  // extern "C"
