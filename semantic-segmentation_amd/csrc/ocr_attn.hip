// Fused object attention of the OCR head (network/ocr_utils.py:100-113):
//
//     sim   = q k^T * key_channels^-0.5        [pixels, K]      K = object regions = classes (19; Mapillary 65)
//     probs = softmax(sim, over the K regions)
//     out   = probs v                           [pixels, 256]
//
// One kernel.  The reference (and rounds 1-4 here) ran it as matmul -> softmax -> matmul with `sim` (fp32) and
// `probs` (16 bit) making a round trip through HBM each, six launches per scale pass with the operand repacks;
// the op itself moves 2 x 512 bytes per pixel (read q, write out) and 0.64 GMAC -- it is HBM bound, so everything
// between the read of q and the write of out stays in registers:
//   * k and v (K x 256 each, 10-33 KB) are staged ONCE per workgroup in LDS, already in MFMA-fragment order
//     (a fragment = one contiguous 1 KB block, one conflict-free ds_read_b128 per lane);
//   * a wave owns 32 pixels.  q is read straight from HBM in B-fragment form (16 bytes per lane and k-step, all 16
//     loads of the tile in flight before the first MFMA), sim^T[region, pixel] = k q^T accumulates with the REGIONS
//     on the accumulator rows: the K logits of a pixel then sit in 16 registers of two lanes (l, l ^ 32) and the
//     softmax is register arithmetic plus one cross-lane exchange for the maximum and one for the sum;
//   * the normalised probabilities, rounded to the storage format, ARE the B fragments of the second product --
//     v is staged with its regions permuted into the order the accumulator rows come in, so no lane exchange
//     is needed: out^T[channel, pixel] = v^T probs^T, 4 channel blocks at a time;
//   * the epilogue packs to 16 bit in registers, completes 16-byte pieces with v_permlane32_swap and stores them.
// Backward (ocr_attn_bwd): the probabilities are recomputed the same way, dprobs^T = v dout^T is a third product
// of the same shape, dsim = scale * probs * (dprobs - <probs, dprobs>) is register arithmetic, dq^T = k^T dsim^T;
// probs and dsim leave as [pixels, Kpad] 16-bit matrices (64 bytes per pixel at K = 19) for the two pixel
// reductions dv = probs^T dout and dk = dsim^T q, which run on the weight-gradient kernels.
#include "common.h"
#include "group.h"
#include "../../include/semseg_hip.h"
#include <float.h>

namespace {

constexpr int D = 256;            // key / value channels (OCR.KEY_CHANNELS; network/ocr_utils.py:121-133)
constexpr int KS = D / 16;        // k-steps of a product over the channels

// lanes l and l + 32 exchange: afterwards (a, b) of a lane < 32 = (own a, partner's a), of a lane >= 32 =
// (partner's b, own b)   [v_permlane32_swap]
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
#ifdef SSA_EMU
  const unsigned pa = __shfl_xor(a, 32, 64), pb = __shfl_xor(b, 32, 64);
  if ((threadIdx.x & 63) < 32) b = pa; else a = pb;
#else
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
#endif
}

// region (class) held by accumulator register r of a lane in half h = lane >> 5, m-block mb
__device__ __forceinline__ int acc_row(int mb, int r, int h) { return mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h; }

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout;
  bf16_t* out;          // forward: out; backward: dq
  bf16_t* probs;        // backward only: [P][Kp]
  bf16_t* dsim;         // backward only: [P][Kp]
  long P;
  int ldq, ldo, lddo, K;
  float scale;
};

// ---- LDS staging of a [K][256] matrix in the two fragment orders
// "row" form (A operand of  m[region, pixel] = sum_d M[region, d] * x[pixel, d]):
//     block (mb, ks): lane l, j  <-  M[mb*32 + (l & 31)][ks*16 + 8*(l >> 5) + j]
// "perm" form (A operand of  o[channel, pixel] = sum_region M[region, channel] * p[region, pixel], the regions of
// k-step ks2 in accumulator-row order):
//     block (db, ks2): lane l, j  <-  M[acc_row(ks2 >> 1, 8*(ks2 & 1) + j, l >> 5)][db*32 + (l & 31)]
template <int MBK>
__device__ __forceinline__ void stage_row_form(const bf16_t* __restrict__ M, int K, bf16_t* __restrict__ dst) {
  // one 16-byte piece per (region, 8 channels): the piece IS a lane's fragment of block (mb, ks)
  for (int i = threadIdx.x; i < MBK * 32 * (D / 8); i += 256) {
    const int row = i / (D / 8), c8 = i - row * (D / 8);
    const int mb = row >> 5, ks = c8 >> 1, h = c8 & 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < K) v = *reinterpret_cast<const uint4*>(M + (long)row * D + c8 * 8);
    *reinterpret_cast<uint4*>(dst + ((long)(mb * KS + ks) * 64 + (row & 31) + 32 * h) * 8) = v;
  }
}

template <int MBK>
__device__ __forceinline__ void stage_perm_form(const bf16_t* __restrict__ M, int K, bf16_t* __restrict__ dst) {
  // lanes run along the channels of one region (64-byte runs of the source); the 2-byte LDS writes of a wave land
  // 16 bytes apart (4-way bank conflicts on 16 K elements per workgroup: noise next to the tile loop)
  for (int i = threadIdx.x; i < 8 * 2 * MBK * 64 * 8; i += 256) {
    const int l = i & 63, j = (i >> 6) & 7, blk = i >> 9;
    const int ks2 = blk % (2 * MBK), db = blk / (2 * MBK);
    const int row = acc_row(ks2 >> 1, 8 * (ks2 & 1) + j, l >> 5);
    dst[((long)blk * 64 + l) * 8 + j] = row < K ? M[(long)row * D + db * 32 + (l & 31)] : (bf16_t)0;
  }
}

// m^T[region, pixel] += sum over the 256 channels: A = a "row"-form LDS matrix, B = fragments xf of 32 pixels
template <int MBK>
__device__ __forceinline__ void product_rows(const bf16_t* __restrict__ frag, const bf16x8_t (&xf)[KS], int lane,
                                             f32x16_t (&acc)[MBK]) {
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(frag + ((long)(mb * KS + ks) * 64 + lane) * 8);
      acc[mb] = ssa_mfma32(a, xf[ks], acc[mb]);
    }
  }
}

// the 32 pixels' fragments of a [P][ld] matrix with 256 channels: lane l reads pixel l & 31, 8 channels at
// ks*16 + 8*(l >> 5).  Pixels beyond P read the last pixel (never stored).
__device__ __forceinline__ void load_pixel_frags(const bf16_t* __restrict__ x, int ld, long pix, int h,
                                                 bf16x8_t (&xf)[KS]) {
  const bf16_t* p = x + pix * ld + 8 * h;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8_t*>(p + ks * 16);
}

// softmax over the regions of sim^T (in place: acc <- probabilities, exactly 0 for the padding regions)
template <int MBK>
__device__ __forceinline__ void softmax_rows(f32x16_t (&acc)[MBK], int K, float scale, int h) {
  float mx = -FLT_MAX;
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s = acc_row(mb, r, h) < K ? acc[mb][r] * scale : -FLT_MAX;
      acc[mb][r] = s;
      mx = fmaxf(mx, s);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = acc_row(mb, r, h) < K ? __expf(acc[mb][r] - mx) : 0.f;
      acc[mb][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] *= inv;
}

// B fragments (regions in accumulator-row order) of a region-major accumulator set
template <int MBK>
__device__ __forceinline__ void rows_to_frags(const f32x16_t (&acc)[MBK], bf16x8_t (&pf)[2 * MBK]) {
#pragma unroll
  for (int ks2 = 0; ks2 < 2 * MBK; ++ks2) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = acc[ks2 >> 1][8 * (ks2 & 1) + j];
    const uint4 u = pack8(f);
    pf[ks2] = __builtin_bit_cast(bf16x8_t, u);
  }
}

// o^T[channel, pixel] = sum over regions of a "perm"-form matrix times the fragments pf; stored as [pixel][ld]
template <int MBK>
__device__ __forceinline__ void product_channels_store(const bf16_t* __restrict__ frag, const bf16x8_t (&pf)[2 * MBK],
                                                       int lane, bf16_t* __restrict__ orow, bool ok) {
  const int h = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x16_t acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
      const int db = half * 4 + b;
#pragma unroll
      for (int ks2 = 0; ks2 < 2 * MBK; ++ks2) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(frag + ((long)(db * 2 * MBK + ks2) * 64 + lane) * 8);
        acc[b] = ssa_mfma32(a, pf[ks2], acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned w0 = f2bf_pair(acc[b][8 * t + 0], acc[b][8 * t + 1]), w1 = f2bf_pair(acc[b][8 * t + 2], acc[b][8 * t + 3]);
        unsigned w2 = f2bf_pair(acc[b][8 * t + 4], acc[b][8 * t + 5]), w3 = f2bf_pair(acc[b][8 * t + 6], acc[b][8 * t + 7]);
        swap32(w0, w2);
        swap32(w1, w3);
        if (ok) *reinterpret_cast<uint4*>(orow + (half * 4 + b) * 32 + 16 * t + 8 * h) = make_uint4(w0, w1, w2, w3);
      }
  }
}

// a region-major accumulator set as rows of a [pixels][Kp] 16-bit matrix (Kp = MBK * 32)
template <int MBK>
__device__ __forceinline__ void store_rows(const f32x16_t (&acc)[MBK], int lane, bf16_t* __restrict__ row, bool ok) {
  const int h = lane >> 5;
#pragma unroll
  for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned w0 = f2bf_pair(acc[mb][8 * t + 0], acc[mb][8 * t + 1]), w1 = f2bf_pair(acc[mb][8 * t + 2], acc[mb][8 * t + 3]);
      unsigned w2 = f2bf_pair(acc[mb][8 * t + 4], acc[mb][8 * t + 5]), w3 = f2bf_pair(acc[mb][8 * t + 6], acc[mb][8 * t + 7]);
      swap32(w0, w2);
      swap32(w1, w3);
      if (ok) *reinterpret_cast<uint4*>(row + mb * 32 + 16 * t + 8 * h) = make_uint4(w0, w1, w2, w3);
    }
}

template <int MBK>
struct OcrAttnFwd {
  typedef AttnArgs Args;
  static constexpr int NT = 256;
  static constexpr int WPE = 2;         // <= 256 registers: two workgroups per CU keep 8 waves' loads in flight
  static constexpr size_t LDS = (size_t)2 * MBK * KS * 64 * 8 * sizeof(bf16_t);
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int gx) {
    SSA_DYN_LDS(bf16_t, smem);
    bf16_t* kfrag = smem;
    bf16_t* vfrag = smem + (size_t)MBK * KS * 64 * 8;
    stage_row_form<MBK>(a.k, a.K, kfrag);
    stage_perm_form<MBK>(a.v, a.K, vfrag);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const long tiles = (a.P + 127) / 128;
    for (long t = bx; t < tiles; t += gx) {
      // the fragment reads below are loop invariant: left alone the compiler hoists all of them (128 * MBK registers and
      // up to 0.9 KB of scratch per lane); re-reading 32 KB of LDS per 32 KB of HBM traffic costs nothing here
      asm volatile("" ::: "memory");
      const long pix = t * 128 + wave * 32 + (lane & 31);
      const bool ok = pix < a.P;
      const long pc = ok ? pix : a.P - 1;
      bf16x8_t qf[KS];
      load_pixel_frags(a.q, a.ldq, pc, h, qf);
      f32x16_t acc[MBK];
      product_rows<MBK>(kfrag, qf, lane, acc);
      softmax_rows<MBK>(acc, a.K, a.scale, h);
      bf16x8_t pf[2 * MBK];
      rows_to_frags<MBK>(acc, pf);
      product_channels_store<MBK>(vfrag, pf, lane, a.out + pc * a.ldo, ok);
    }
  }
};

template <int MBK>
struct OcrAttnBwd {
  typedef AttnArgs Args;
  static constexpr int NT = 256;
  static constexpr int WPE = 2;
  static constexpr size_t LDS = (size_t)3 * MBK * KS * 64 * 8 * sizeof(bf16_t);
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int gx) {
    SSA_DYN_LDS(bf16_t, smem);
    bf16_t* kfrag = smem;                                      // row form of k: sim
    bf16_t* vrow = smem + (size_t)MBK * KS * 64 * 8;           // row form of v: dprobs
    bf16_t* kperm = smem + (size_t)2 * MBK * KS * 64 * 8;      // perm form of k: dq
    stage_row_form<MBK>(a.k, a.K, kfrag);
    stage_row_form<MBK>(a.v, a.K, vrow);
    stage_perm_form<MBK>(a.k, a.K, kperm);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int Kp = MBK * 32;
    const long tiles = (a.P + 127) / 128;
    for (long t = bx; t < tiles; t += gx) {
      // the fragment reads below are loop invariant: left alone the compiler hoists all of them (128 * MBK registers and
      // up to 0.9 KB of scratch per lane); re-reading 32 KB of LDS per 32 KB of HBM traffic costs nothing here
      asm volatile("" ::: "memory");
      const long pix = t * 128 + wave * 32 + (lane & 31);
      const bool ok = pix < a.P;
      const long pc = ok ? pix : a.P - 1;
      f32x16_t pr[MBK], dp[MBK];
      {
        bf16x8_t xf[KS];
        load_pixel_frags(a.q, a.ldq, pc, h, xf);
        product_rows<MBK>(kfrag, xf, lane, pr);
      }
      {
        bf16x8_t xf[KS];
        load_pixel_frags(a.dout, a.lddo, pc, h, xf);
        product_rows<MBK>(vrow, xf, lane, dp);
      }
      softmax_rows<MBK>(pr, a.K, a.scale, h);
      // dsim = scale * probs * (dprobs - <probs, dprobs>)      (padding regions: probs = 0 -> dsim = 0)
      float dot = 0.f;
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dot += pr[mb][r] * dp[mb][r];
      dot += __shfl_xor(dot, 32, 64);
#pragma unroll
      for (int mb = 0; mb < MBK; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[mb][r] = a.scale * pr[mb][r] * (dp[mb][r] - dot);
      store_rows<MBK>(pr, lane, a.probs + pc * Kp, ok);
      store_rows<MBK>(dp, lane, a.dsim + pc * Kp, ok);
      bf16x8_t pf[2 * MBK];
      rows_to_frags<MBK>(dp, pf);
      product_channels_store<MBK>(kperm, pf, lane, a.out + pc * a.ldo, ok);
    }
  }
};

template <template <int> class KN>
int launch(int mbk, const AttnArgs& a, hipStream_t s) {
  long tiles = (a.P + 127) / 128;
  const int gx = (int)(tiles < 1024 ? tiles : 1024);          // 4 workgroups per CU at most: the staging is per workgroup
  switch (mbk) {
    case 1: return ssa::submit<KN<1>>(a, gx, 1, KN<1>::LDS, s);
    case 2: return ssa::submit<KN<2>>(a, gx, 1, KN<2>::LDS, s);
    case 3: return ssa::submit<KN<3>>(a, gx, 1, KN<3>::LDS, s);
    default: return SSA_EUNSUPPORTED;
  }
}

bool args_ok(const void* q, int ldq, const void* k, const void* v, long P, int K, int Dch) {
  return q && k && v && P > 0 && K > 0 && Dch == D && ldq % 8 == 0 &&
         ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
}

}  // namespace

extern "C" {

int ssa_ocr_attn_supported(int K, int Dch) { return Dch == D && K > 0 && K <= 96; }

int ssa_ocr_attn_fwd(const void* q, int ldq, const void* k, const void* v, long P, int K, int Dch, float scale,
                     void* out, int ldo, void* stream) {
  if (!ssa_ocr_attn_supported(K, Dch)) return SSA_EUNSUPPORTED;
  if (!args_ok(q, ldq, k, v, P, K, Dch) || !out || ldo % 8 || (reinterpret_cast<uintptr_t>(out) & 15u)) return SSA_EINVAL;
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)out;
  a.P = P; a.ldq = ldq; a.ldo = ldo; a.K = K; a.scale = scale;
  return launch<OcrAttnFwd>((K + 31) / 32, a, (hipStream_t)stream);
}

int ssa_ocr_attn_bwd(const void* q, int ldq, const void* k, const void* v, const void* dout, int lddo, long P, int K,
                     int Dch, float scale, void* dq, int lddq, void* probs, void* dsim, void* stream) {
  if (!ssa_ocr_attn_supported(K, Dch)) return SSA_EUNSUPPORTED;
  if (!args_ok(q, ldq, k, v, P, K, Dch) || !dout || !dq || !probs || !dsim || lddo % 8 || lddq % 8 ||
      ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) | reinterpret_cast<uintptr_t>(probs) |
        reinterpret_cast<uintptr_t>(dsim)) & 15u))
    return SSA_EINVAL;
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dout = (const bf16_t*)dout;
  a.out = (bf16_t*)dq; a.probs = (bf16_t*)probs; a.dsim = (bf16_t*)dsim;
  a.P = P; a.ldq = ldq; a.ldo = lddq; a.lddo = lddo; a.K = K; a.scale = scale;
  return launch<OcrAttnBwd>((K + 31) / 32, a, (hipStream_t)stream);
}

}  // extern "C"
