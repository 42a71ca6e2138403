// Persistent, software-pipelined halo-tile convolution for the 3x3 stride-1 convs of the HRNetV2 trunk
// (network/hrnetv2.py:31-66), forward and data gradient, with the neighbouring BatchNorm
// passes folded into its operand staging and its epilogue (gfx950 / MI355X).
//
// conv_tile.hip gives every 128-pixel tile a workgroup of its own: ~1,500 workgroups per trunk level,
// each of which lives ~10 us (load the halo, DMA 54 KB of filter from L2, barrier, 0.7-3 us of MFMA,
// LDS-staged epilogue, 128 fp64 atomics) -- 0.13 of the HBM roof (profiles/r02_trace_step.txt).  Here
//   * a workgroup is PERSISTENT over a strip of consecutive tiles of one (problem, n-block group):
//     the grid is ~2 workgroups per CU whatever the problem sizes (strip length from the level's total
//     work, ssa_conv_tile_strip);
//   * the 48-channel instantiation keeps its whole filter slice (54 KB) resident in LDS for the strip
//     -- one DMA burst per workgroup instead of one per tile (81 MB -> 27 MB of L2 traffic per level);
//     the 96-channel-chunk instantiation (96/192/384 channels) streams 3-tap filter stages through a
//     double buffer as ONE continuous DMA pipeline across chunks and tiles (stage s+1 is in flight during
//     the MFMAs of stage s, also over a tile boundary);
//   * the next tile's (or channel chunk's) halo is fetched into registers during the current MFMAs;
//   * the epilogue is wave-local: a wave owns one 32-pixel tile row x all output channels of the
//     workgroup, stages its accumulators through its own LDS slice and stores whole pixel rows -- no
//     block barrier between the MFMAs and the stores; BatchNorm statistics stay in registers over the
//     strip: one set of fp64 atomics per workgroup instead of one per tile.
//   * XCD-aware order: consecutive strips (which share halo rows) and the n-block groups of a strip
//     (which share the whole halo) go to the same XCD's L2.
//
// Folded BatchNorm (what network/hrnetv2.py:53-64 runs as separate passes over HBM):
//   XF 1 (forward): the staged input is  relu(scale[c] * x + shift[c])  -- bn1 + ReLU applied while conv2's
//         halo passes through registers; bn1's output is never written.
//   XF 2 (backward): the staged input is BatchNorm+ReLU's data gradient computed from (dz, x):
//         dy = A[c] * (m ? dz : 0) + B0[c] + C0[c] * x,  m = [ma[c] * x + mb[c] > 0]  (table from
//         ssa_bn_bwd_coef) -- bn1's backward apply while conv1's data-gradient halo is staged.
//   epilogue (as conv_tile.hip): BatchNorm batch statistics of the bf16 outputs; aux_mode 1: + residual
//         gradient; aux_mode 2: bn1's backward sums from (x tile, dz).
// Out-of-image halo pixels are zero AFTER the transform (the conv pads the transformed activation).
#include "common.h"
#include "group.h"
#include <stdlib.h>
#include "../../include/semseg_hip.h"

namespace {

constexpr int kStatReplicas = 8;   // as conv_tile.hip (ssa_bn_stat_replicas)

// -DSSA_TILE_TIMING (tools/tilebench.py --timing; never in the product build): wave 0 of workgroup 0 writes
// s_memtime stamps of every phase of its first iterations to the `coef` pointer (reinterpreted, XF 0 / aux 0 only)
#ifdef SSA_TILE_TIMING
// (stamps go to spare LDS during the loop -- a global store would sit in the vmcnt queue every barrier waits for -- and
// are copied out at the end)
#define SSA_STAMP(k) do { if (tdbg && it < 24) { tlds[it * 8 + (k)] = (long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SSA_STAMP(k) do { } while (0)
#endif

struct TilePArgs {
  const bf16_t* x; const bf16_t* x2; const float* xf;     // staged input; XF 2: the layer input x; transform table
  const uint4* wfrag; const float* bias; bf16_t* y; double* stats;
  const bf16_t* aux; const float* coef;                    // epilogue tile; [4][Cout] table of aux_mode 2
  int ldx, ldx2, Cin, ldy, H, W, Cout, nb_total, tiles_x, tiles_y, ldaux, aux_mode;
  int total_tiles, tiles_per_wg, ngroups, nwg;
  int variant;
};

// Bijective XCD-aware order (block b runs on XCD b % 8): XCD x gets one contiguous range of work items.
__device__ __forceinline__ int xcd_order(int v, int n) {
  const int q = n >> 3, r = n & 7, x = v & 7, k = v >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

template <int NB, int TPC, int CST>
__device__ __forceinline__ void stage_filter(const uint4* __restrict__ wfrag, int nb0, int nb_total,
                                             int ksteps_total, int csteps_total, int cc, int tap0,
                                             unsigned char* dst, int wave, int lane) {
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

template <int CK, int NB, int TPC, int XF, int AUXM>
struct ConvTileP {
  static constexpr bool AUX = AUXM != 0;
  typedef TilePArgs Args;
  static constexpr int NT = 256;
  static constexpr int TW = 32, TH = 4;
  static constexpr int HW_ = TW + 2, HH_ = TH + 2, NPIX = HH_ * HW_;
  static constexpr int PSB = CK * 2 + 16;            // halo pixel stride (bytes): an odd number of 16-byte slots
  static constexpr int CP = CK / 8;                  // 16-byte pieces per halo pixel
  static constexpr int NA = (NT / CP) * CP;          // staging threads: thread t always moves channel group t % CP
  static constexpr int RP = NA / CP;                 // halo pixels per staging pass
  static constexpr int IT = (NPIX + RP - 1) / RP;
  static constexpr int CST = CK / 16;
  static constexpr int TAPS = 9, NSTAGE = TAPS / TPC, STAGE_KS = TPC * CST;
  static constexpr int STAGE_BYTES = NB * STAGE_KS * 1024;
  static constexpr bool RESIDENT = NSTAGE == 1;      // the whole filter slice stays in LDS for the strip
  static constexpr int NBUF = RESIDENT ? 1 : 2;
  static constexpr int HALO_BYTES = (NPIX * PSB + 1023) / 1024 * 1024;
  static constexpr int LDC = NB * 32 + 8;            // wave-local output staging: row stride (elements)
  static constexpr int CPR = NB * 4;                 // 16-byte pieces per staged output row
  static constexpr int CS_WAVE = 32 * LDC * 2;       // bytes per wave
  static constexpr int EIT = 32 * CPR / 64;          // output pieces per lane and tile
  static constexpr size_t LDS = (size_t)HALO_BYTES + (size_t)NBUF * STAGE_BYTES;
  static_assert(NSTAGE * TPC == TAPS, "taps per stage must divide the tap count");
  static_assert(4 * CS_WAVE <= (RESIDENT ? HALO_BYTES : STAGE_BYTES), "output staging does not fit");
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");

  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int /*gx*/) {
    const bf16_t* __restrict__ x = a.x;
    const bf16_t* __restrict__ x2 = a.x2;
    const float* __restrict__ xf = a.xf;
    const uint4* __restrict__ wfrag = a.wfrag;
    const float* __restrict__ bias = a.bias;
    bf16_t* __restrict__ y = a.y;
    double* __restrict__ stats = a.stats;
    const bf16_t* __restrict__ aux = a.aux;
    const float* __restrict__ coef = a.coef;
    const int ldx = a.ldx, ldx2 = a.ldx2, Cin = a.Cin, ldy = a.ldy, H = a.H, W = a.W, Cout = a.Cout;
    const int nb_total = a.nb_total, tiles_x = a.tiles_x, tiles_y = a.tiles_y, ldaux = a.ldaux;
    SSA_DYN_LDS(unsigned char, smem);
    unsigned char* Bs = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_ = xcd_order(bx, a.nwg);
    const int strip = w_ / a.ngroups, grp = w_ - strip * a.ngroups;
    const int nb0 = grp * NB;
    const int t_begin = strip * a.tiles_per_wg;
    const int t_end = min(a.total_tiles, t_begin + a.tiles_per_wg);
    const int nchunk = Cin / CK;
    const int csteps_total = Cin / 16;
    const int ksteps_total = TAPS * csteps_total;
    const int n_iter = (t_end - t_begin) * nchunk;

    // ---- staging role of this thread: channel group cg of halo pixels prow, prow + RP, ...
    const bool stg = tid < NA;
    const int cg = tid % CP, prow = tid / CP;
    uint4 v[IT];
    uint4 v2[XF == 2 ? IT : 1];
    unsigned okmask = 0;                       // bit i: piece i lies inside the image

    // tile-invariant part of this thread's pieces: halo coordinates (hy << 8 | hx, 0xffff: no such piece) and the
    // element offset relative to the tile's first halo pixel; invalid lanes load the tile's own first pixel instead
    // (always inside the image) and are zeroed at staging time -- ten unconditional loads issue back to back
    int hyx[IT], rel[IT], rel2[XF == 2 ? IT : 1];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int pix = prow + i * RP;
      const int hy = pix / HW_, hx = pix - hy * HW_;
      const bool have = stg && pix < NPIX;
      hyx[i] = have ? ((hy << 8) | hx) : 0xffff;
      rel[i] = (hy * W + hx) * ldx;
      if constexpr (XF == 2) rel2[i] = (hy * W + hx) * ldx2;
    }
    const int rel_c = (W + 1) * ldx, rel2_c = (W + 1) * ldx2;     // halo pixel (1, 1) = output pixel (0, 0) of the tile
    // tile walk without divisions: (image, tile row, tile column) of the tile whose halo is fetched next
    int f_b, f_ty, f_tx;
    {
      f_tx = t_begin % tiles_x;
      const int r = t_begin / tiles_x;
      f_ty = r % tiles_y;
      f_b = r / tiles_y;
    }
    int c_b = f_b, c_ty = f_ty, c_tx = f_tx;                       // ... and of the tile being computed
    // global -> registers: the halo of (the fetch tile, channel chunk cc)
    auto fetch = [&](int cc) {
      const int x0 = f_tx * TW, y0 = f_ty * TH;
      const bf16_t* xb = x + ((long)f_b * H * W + (long)(y0 - 1) * W + (x0 - 1)) * ldx + cc * CK + cg * 8;
      const bf16_t* xb2 = XF == 2 ? x2 + ((long)f_b * H * W + (long)(y0 - 1) * W + (x0 - 1)) * ldx2 + cc * CK + cg * 8 : nullptr;
      okmask = 0;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int iy = y0 - 1 + (hyx[i] >> 8), ix = x0 - 1 + (hyx[i] & 255);
        const bool ok = hyx[i] != 0xffff && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        v[i] = *reinterpret_cast<const uint4*>(xb + (ok ? rel[i] : rel_c));
        if constexpr (XF == 2) v2[i] = *reinterpret_cast<const uint4*>(xb2 + (ok ? rel2[i] : rel2_c));
        okmask |= (ok ? 1u : 0u) << i;
      }
    };
    auto advance = [&](int* b, int* ty, int* tx) {
      if (++*tx == tiles_x) {
        *tx = 0;
        if (++*ty == tiles_y) { *ty = 0; ++*b; }
      }
    };
    // registers -> (transform) -> LDS halo image of channel chunk cc
    auto stage = [&](int cc) {
      if constexpr (XF == 1) {
        float sc[8], sh[8];
        if (stg) {
          const float* t0 = xf + cc * CK + cg * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc[j] = t0[j]; sh[j] = t0[Cin + j]; }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          if ((okmask >> i) & 1u) {
            float f[8];
            unpack8(v[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * sc[j] + sh[j], 0.f);
            v[i] = pack8(f);
          } else {
            v[i] = make_uint4(0, 0, 0, 0);
          }
        }
      } else if constexpr (XF == 0) {
#pragma unroll
        for (int i = 0; i < IT; ++i)
          if (!((okmask >> i) & 1u)) v[i] = make_uint4(0, 0, 0, 0);
      } else if constexpr (XF == 2) {
        // two passes of four channels: 20 coefficient registers live at a time instead of 40
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float cA[4], cB[4], cC[4], ma[4], mb[4];
          if (stg) {
            const float* t0 = xf + cc * CK + cg * 8 + h * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              cA[j] = t0[j]; cB[j] = t0[Cin + j]; cC[j] = t0[2 * Cin + j]; ma[j] = t0[3 * Cin + j]; mb[j] = t0[4 * Cin + j];
            }
          }
#pragma unroll
          for (int i = 0; i < IT; ++i) {
            if ((okmask >> i) & 1u) {
              const unsigned g0 = h ? v[i].z : v[i].x, g1 = h ? v[i].w : v[i].y;
              const unsigned x0_ = h ? v2[i].z : v2[i].x, x1_ = h ? v2[i].w : v2[i].y;
              float g[4] = {__uint_as_float(g0 << 16), __uint_as_float(g0 & 0xffff0000u),
                            __uint_as_float(g1 << 16), __uint_as_float(g1 & 0xffff0000u)};
              const float xv[4] = {__uint_as_float(x0_ << 16), __uint_as_float(x0_ & 0xffff0000u),
                                   __uint_as_float(x1_ << 16), __uint_as_float(x1_ & 0xffff0000u)};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float gm = (xv[j] * ma[j] + mb[j]) > 0.f ? g[j] : 0.f;
                g[j] = gm * cA[j] + (cB[j] + cC[j] * xv[j]);
              }
              const unsigned p0 = f2bf_pair(g[0], g[1]), p1 = f2bf_pair(g[2], g[3]);
              if (h) { v[i].z = p0; v[i].w = p1; } else { v[i].x = p0; v[i].y = p1; }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i)
          if (!((okmask >> i) & 1u)) v[i] = make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int pix = prow + i * RP;
        if (hyx[i] != 0xffff) *reinterpret_cast<uint4*>(smem + pix * PSB + cg * 16) = v[i];
      }
    };

    // ---- epilogue role of this lane: output piece (row idx / CPR of the wave's tile row, channel group lane % CPR)
    const int ecp = lane % CPR;
    const int en = nb0 * 32 + ecp * 8;                  // first output channel of this lane's pieces
    const bool en_ok = en < Cout;
    float S[AUXM == 1 ? 1 : 8], Q[AUXM == 1 ? 1 : 8];  // statistics of this lane's 8 channels over the strip
#pragma unroll
    for (int j = 0; j < (AUXM == 1 ? 1 : 8); ++j) { S[j] = 0.f; Q[j] = 0.f; }
    float ema[AUXM == 2 ? 8 : 1], emb[AUXM == 2 ? 8 : 1];
    if constexpr (AUXM == 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { ema[j] = 0.f; emb[j] = 0.f; }
      if (en_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { ema[j] = coef[en + j]; emb[j] = coef[Cout + en + j]; }
      }
    }
    uint4 auxv[AUX ? EIT : 1];

    const int a_off = (wave * HW_ + (lane & 31)) * PSB + (lane >> 5) * 16;   // MFMA A rows = the wave's tile row

    f32x16_t acc[NB];

    if (n_iter <= 0) return;
#ifdef SSA_TILE_TIMING
    long* tdbg = (AUXM == 0 && XF == 0 && bx == 0 && tid == 0) ? reinterpret_cast<long*>(const_cast<float*>(coef)) : nullptr;
    long* tlds = reinterpret_cast<long*>(smem + LDS);          // 1.5 KB past the kernel's own LDS (the timing build asks for it)
#endif
    // ---- prologue: first halo into registers, the filter (slice / first stage) on its way into LDS
    fetch(0);
    stage_filter<NB, TPC, CST>(wfrag, nb0, nb_total, ksteps_total, csteps_total, 0, 0, Bs, wave, lane);
    int s = 0;                                         // global filter-stage counter (streamed variant)
    int cc = 0;
    for (int it = 0; it < n_iter; ++it) {
      // everyone is past the barrier that ended the previous iteration: the halo image is free
      SSA_STAMP(0);
      stage(cc);
      SSA_STAMP(1);
      __syncthreads();                                 // halo image + filter stage s (slice) landed
      SSA_STAMP(2);
      const bool last_chunk = cc + 1 == nchunk;
      int ccn = cc + 1;
      if (last_chunk) { ccn = 0; advance(&f_b, &f_ty, &f_tx); }
      if (it + 1 < n_iter) fetch(ccn);                 // next halo: in flight during this iteration's MFMAs
      const int b_ = c_b, x0 = c_tx * TW, y0 = c_ty * TH;
      if constexpr (AUX) {
        if (last_chunk) {
          const bf16_t* ab = aux + (long)b_ * H * W * ldaux;
#pragma unroll
          for (int i = 0; i < EIT; ++i) {
            const int idx = lane + i * 64;
            const int row = idx / CPR;
            const int oy = y0 + wave, ox = x0 + row;
            auxv[i] = make_uint4(0, 0, 0, 0);
            if (oy < H && ox < W && en_ok)
              auxv[i] = *reinterpret_cast<const uint4*>(ab + ((long)oy * W + ox) * ldaux + en);
          }
        }
      }
      if (cc == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
      }
      SSA_STAMP(3);
#pragma unroll
      for (int st = 0; st < NSTAGE; ++st) {
        if constexpr (!RESIDENT) {
          // stage s+1 of the continuous filter stream: the next taps, the next chunk, or the next tile's first stage
          const bool more = st + 1 < NSTAGE || it + 1 < n_iter;
          if (more) {
            const int cc1 = st + 1 < NSTAGE ? cc : ccn;
            const int tap1 = st + 1 < NSTAGE ? (st + 1) * TPC : 0;
            stage_filter<NB, TPC, CST>(wfrag, nb0, nb_total, ksteps_total, csteps_total, cc1, tap1,
                                       Bs + ((s + 1) & 1) * STAGE_BYTES, wave, lane);
          }
        }
        const unsigned char* Bc = Bs + (RESIDENT ? 0 : (s & 1) * STAGE_BYTES) + lane * 16;
        // fragments of k-step ksl + RD - 1 are read while the MFMAs of k-step ksl run: a ring of RD register sets
        // (left to itself the compiler reads two k-steps, waits, multiplies, and only then reads the next two)
        constexpr int RD = 4;
        bf16x8_t ra[RD], rb[RD][NB];
        auto rd_frag = [&](int ksl) {
          const int tap = st * TPC + ksl / CST, cs = ksl % CST;
          const int kh = tap / 3, kw = tap - kh * 3;
          ra[ksl % RD] = *reinterpret_cast<const bf16x8_t*>(smem + a_off + (kh * HW_ + kw) * PSB + cs * 32);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            rb[ksl % RD][nb] = *reinterpret_cast<const bf16x8_t*>(Bc + (nb * STAGE_KS + ksl) * 1024);
        };
#pragma unroll
        for (int p = 0; p < RD - 1 && p < STAGE_KS; ++p) rd_frag(p);
        __builtin_amdgcn_sched_group_barrier(0x100, (RD - 1) * (1 + NB), 0);
#pragma unroll
        for (int ksl = 0; ksl < STAGE_KS; ++ksl) {
          if (ksl + RD - 1 < STAGE_KS) rd_frag(ksl + RD - 1);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nb] = ssa_mfma32(ra[ksl % RD], rb[ksl % RD][nb], acc[nb]);
          if (ksl + RD - 1 < STAGE_KS) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NB, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
        }
        if (st == 0) SSA_STAMP(4);
        __syncthreads();        // stage s+1 landed; buffer s & 1 (and, after the last stage, the halo image) is free
        if (st == 0) SSA_STAMP(5);
        if constexpr (!RESIDENT) ++s;
      }
      SSA_STAMP(6);
      if (last_chunk) {
        // ---- wave-local epilogue: (+bias) -> bf16 -> this wave's LDS slice -> whole pixel rows
        // slice: inside the halo image (resident filter) / inside the filter buffer just consumed (streamed)
        unsigned char* Cw = (RESIDENT ? smem : Bs + ((s - 1) & 1) * STAGE_BYTES) + wave * CS_WAVE;
        bf16_t* Cs = reinterpret_cast<bf16_t*>(Cw);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int col = nb * 32 + (lane & 31);
          const int n = nb0 * 32 + col;
          const float bv = (bias != nullptr && n < Cout) ? bias[n] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            Cs[row * LDC + col] = f2bf(acc[nb][r] + bv);
          }
        }
        ssa_wave_sync();
        const int oy = y0 + wave;
        bf16_t* yb = y + ((long)b_ * H * W + (long)oy * W) * ldy + en;
#pragma unroll
        for (int i = 0; i < EIT; ++i) {
          const int idx = lane + i * 64;
          const int row = idx / CPR;
          const int ox = x0 + row;
          const bool ok = oy < H && ox < W && en_ok;
          uint4 o = *reinterpret_cast<const uint4*>(Cw + row * (LDC * 2) + ecp * 16);
          if constexpr (AUX) {
            float f[8], xv[8];
            unpack8(o, f);
            unpack8(auxv[i], xv);
            if constexpr (AUXM == 1) {
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] += xv[j];
              o = pack8(f);
            } else if constexpr (AUXM == 2) {
              if (ok) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float gm = (xv[j] * ema[j] + emb[j]) > 0.f ? f[j] : 0.f;
                S[j] += gm;
                Q[j] += gm * xv[j];
              }
              }
            }
          } else {
            if (stats != nullptr && ok) {
              float f[8];
              unpack8(o, f);
#pragma unroll
              for (int j = 0; j < 8; ++j) { S[j] += f[j]; Q[j] += f[j] * f[j]; }
            }
          }
          if (ok) *reinterpret_cast<uint4*>(yb + (long)ox * ldy) = o;
        }
        if constexpr (RESIDENT) __syncthreads();       // the slices live in the halo image the next tile overwrites
      }
      SSA_STAMP(7);
      if (last_chunk) advance(&c_b, &c_ty, &c_tx);
      cc = ccn;
    }

#ifdef SSA_TILE_TIMING
    if (tdbg)
      for (int i = 0; i < 24 * 8; ++i) tdbg[i] = i < n_iter * 8 ? tlds[i] : 0;
#endif
    // ---- statistics of the strip: lanes -> wave -> workgroup -> one fp64 atomic per channel
    if constexpr (AUXM == 1) return;          // residual add: no statistics (stats is NULL by contract)
    if (stats != nullptr) {
#pragma unroll
      for (int off = CPR; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          S[j] += __shfl_xor(S[j], off, 64);
          Q[j] += __shfl_xor(Q[j], off, 64);
        }
      }
      __syncthreads();                                 // every wave is done with its LDS slice
      float* red = reinterpret_cast<float*>(smem);     // [4 waves][2][NB*32]
      if (lane < CPR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[(wave * 2 + 0) * NB * 32 + lane * 8 + j] = S[j];
          red[(wave * 2 + 1) * NB * 32 + lane * 8 + j] = Q[j];
        }
      }
      __syncthreads();
      double* st = stats + (long)(strip % kStatReplicas) * 2 * Cout;
      if (tid < NB * 32) {
        const int n = nb0 * 32 + tid;
        if (n < Cout) {
          float sv = 0.f, qv = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            sv += red[(w * 2 + 0) * NB * 32 + tid];
            qv += red[(w * 2 + 1) * NB * 32 + tid];
          }
          double sd = (double)sv, qd = (double)qv;
          if constexpr (AUX) {
            // aux_mode 2 accumulated sum(m*dz*x); the consumer wants sum(m*dz*xhat), xhat = (x - mean) * invstd
            if constexpr (AUXM == 2) qd = (double)coef[3 * Cout + n] * (qd - (double)coef[2 * Cout + n] * sd);
          }
          atomicAdd(&st[n], sd);
          atomicAdd(&st[Cout + n], qd);
        }
      }
    }
  }
};

// The two instantiations a trunk level uses behind ONE kernel (as conv_tile.hip's ConvTileAny): 48 channels
// (both n-blocks per workgroup, resident filter) and the streamed 96-channel-chunk one (96 / 192 / 384 channels).
template <int XF, int AUXM>
struct ConvTilePAny {
  typedef TilePArgs Args;
  static constexpr int NT = 256;
  typedef ConvTileP<48, 2, 9, XF, AUXM> V0;
  typedef ConvTileP<96, 1, 3, XF, AUXM> V1;
#ifdef SSA_TILE_TIMING
  static constexpr size_t LDS = (V0::LDS > V1::LDS ? V0::LDS : V1::LDS) + 1536;
#else
  static constexpr size_t LDS = V0::LDS > V1::LDS ? V0::LDS : V1::LDS;
#endif
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int gx) {
    if (a.variant == 0) V0::run(a, bx, by, gx);
    else V1::run(a, bx, by, gx);
  }
};

static thread_local int g_strip_units = 0;     // work units (54 MFMAs per wave) per workgroup, 0 = per problem

template <int XF, int AUXM>
int launch_p(const ssa_conv_desc& d, const TilePArgs& a0, hipStream_t s) {
  TilePArgs a = a0;
  const int nchunk = d.Cin == 48 ? 1 : d.Cin / 96;
  const int NB = d.Cin == 48 ? 2 : 1;
  a.variant = d.Cin == 48 ? 0 : 1;
  a.nb_total = (d.Cout + 31) / 32;
  a.tiles_x = (d.W + 31) / 32;
  a.tiles_y = (d.H + 3) / 4;
  a.total_tiles = d.B * a.tiles_x * a.tiles_y;
  a.ngroups = (a.nb_total + NB - 1) / NB;
  int units = g_strip_units;
  if (units <= 0) {
    // a launch of its own: ~2 workgroups per CU from this problem alone
    const long total = (long)a.total_tiles * a.ngroups * nchunk;
    units = (int)((total + 511) / 512);
  }
  if (units > 16) units = 16;
  int tpw = units / nchunk;
  if (tpw < 1) tpw = 1;
  const int nstrips = (a.total_tiles + tpw - 1) / tpw;
  a.tiles_per_wg = (a.total_tiles + nstrips - 1) / nstrips;
  a.nwg = ((a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg) * a.ngroups;
  return ssa::submit<ConvTilePAny<XF, AUXM>>(a, a.nwg, 1, ConvTilePAny<XF, AUXM>::LDS, s);
}

}  // namespace

extern "C" {

int ssa_conv2d_tile_p_supported(const ssa_conv_desc* d) {
  if (!ssa_conv2d_tile_supported(d)) return 0;
  if (d->W < 16) return 0;                    // 32-pixel-wide tiles only
  return d->Cin == 48 || d->Cin == 96 || d->Cin == 192 || d->Cin == 384;
}

int ssa_conv_tile_strip(int units) {
  g_strip_units = units < 0 ? 0 : units;
  return SSA_OK;
}

int ssa_conv2d_tile_p(const ssa_conv_desc* dp, const void* x, const void* x2, int ldx2, const float* xf,
                      int xf_mode, const void* w_frag, const float* bias, void* y, double* stats,
                      const void* aux, int ldaux, const float* coef, int aux_mode, void* stream) {
  if (!dp || !x || !w_frag || !y) return SSA_EINVAL;
  if (!ssa_conv2d_tile_p_supported(dp)) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_frag)) & 15u)
    return SSA_EINVAL;
  if (xf_mode < 0 || xf_mode > 2 || aux_mode < 0 || aux_mode > 2) return SSA_EINVAL;
  if (xf_mode && (!xf || (reinterpret_cast<uintptr_t>(xf) & 15u))) return SSA_EINVAL;
  if (xf_mode == 2 && (!x2 || ldx2 % 8 || (reinterpret_cast<uintptr_t>(x2) & 15u) ||
                       (long)dp->H * dp->W * ldx2 >= (1L << 31)))
    return SSA_EINVAL;
  if (aux_mode && (!aux || ldaux % 8 || (reinterpret_cast<uintptr_t>(aux) & 15u))) return SSA_EINVAL;
  if (aux_mode == 2 && (!coef || !stats)) return SSA_EINVAL;
  if (aux_mode == 1 && stats) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  TilePArgs a;
  a.x = (const bf16_t*)x; a.x2 = (const bf16_t*)x2; a.xf = xf; a.wfrag = (const uint4*)w_frag; a.bias = bias;
  a.y = (bf16_t*)y; a.stats = stats; a.aux = (const bf16_t*)aux; a.coef = coef;
  a.ldx = d.ldx; a.ldx2 = ldx2; a.Cin = d.Cin; a.ldy = d.ldy; a.H = d.H; a.W = d.W; a.Cout = d.Cout;
  a.ldaux = ldaux; a.aux_mode = aux_mode;
  a.nb_total = a.tiles_x = a.tiles_y = a.total_tiles = a.tiles_per_wg = a.ngroups = a.nwg = a.variant = 0;
  hipStream_t s = (hipStream_t)stream;
  // instantiated combinations: every transform without an epilogue tile; the residual add with and without the
  // BatchNorm-backward transform (conv1's data gradient); the BatchNorm-backward sums on a plain input (conv2's)
  switch (aux_mode * 3 + xf_mode) {
    case 0: return launch_p<0, 0>(d, a, s);
    case 1: return launch_p<1, 0>(d, a, s);
    case 2: return launch_p<2, 0>(d, a, s);
    case 3: return launch_p<0, 1>(d, a, s);
    case 5: return launch_p<2, 1>(d, a, s);
    case 6: return launch_p<0, 2>(d, a, s);
    default: return SSA_EUNSUPPORTED;
  }
}

}  // extern "C"
