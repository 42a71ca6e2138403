// Persistent, software-pipelined halo-tile convolution for the 3x3 stride-1 convs of the HRNetV2 trunk
// (network/hrnetv2.py:31-66), forward and data gradient (gfx950 / MI355X).
//
// A workgroup (4 waves) is PERSISTENT over a strip of consecutive 128-pixel tiles (4 rows x 32) of one (problem,
// 32-channel n-block); the input is processed in chunks of 48 channels: a "unit" = one (tile, chunk) = the 6x34-pixel
// halo image of 48 channels staged in LDS once + 27 MFMAs (32x32x16) per wave.
//   * the filter runs as ONE continuous LDS-DMA pipeline over stages of 3 taps (9 KB) through a ring of THREE
//     buffers: stage s issues the DMAs of stage s + 2 and ends in `s_waitcnt vmcnt(K) ; s_barrier` with K counted, so
//     two stages of filter are in flight while one is multiplied (round 3 had two buffers behind __syncthreads(),
//     whose vmcnt(0) made every stage wait one L2 round trip for its successor AND for the next halo's loads:
//     8,600 clocks per 54-MFMA unit against 2,200 of MFMA);
//   * the next unit's halo is fetched into registers behind the current unit's DMAs, and no barrier of the loop
//     waits for it (the barriers wait on LDS traffic only: `s_waitcnt lgkmcnt(0) ; s_barrier`);
//   * the MFMA operands are SWAPPED (filter fragment as A, pixel fragment as B): the accumulator of a lane then
//     holds one PIXEL's channels (4 consecutive channels per register quad), so the epilogue packs to bf16 in
//     registers, completes 16-byte pieces with v_permlane32_swap and stores straight to HBM -- no LDS staging, no
//     barrier between the MFMAs and the stores (round 3: 64 ds_write_b16 + re-read per lane and tile, 1,300-2,000
//     clocks per unit);
//   * BatchNorm statistics of the bf16-rounded outputs stay in registers over the strip (a lane owns 16 channels),
//     are reduced over the 32 pixel lanes with DPP adds once per strip and leave as one fp64 atomic per channel and
//     workgroup;
//   * 50 KB of LDS and <= 168 registers: THREE workgroups per CU (12 waves), which is what a level of <= 750 short
//     workgroups wants -- measured against the round's other geometry (two n-blocks per workgroup, the 48-channel
//     filter resident: 77 KB, two workgroups per CU), profiles/r04_notes.md calls B-E: that one wins the 8-problem level
//     in isolation (27.0 against 29.1 us) and loses the 4-problem levels (22.0 / 19.1) and the step (22.99 / 22.51 ms);
//   * XCD-aware order: consecutive strips (which share halo rows) and the n-blocks of a strip (which share the whole
//     halo) go to the same XCD's L2.
// What bounds it (phase stamps, -DSSA_TILE_TIMING): with one n-block per wave every MFMA needs 2 KB of LDS reads,
// i.e. the full 256 B/clk of the CU at the MFMA rate -- the k-loop runs at 62 clocks per MFMA whatever the fragment
// ring's depth (3 / 4 / 6: no difference) -- and a level is as long as its longest chain: a 384-channel tile is 8
// units of ~3,800 clocks.
// Epilogues (as conv_tile.hip): BatchNorm batch statistics; aux_mode 1: + residual gradient; aux_mode 2: bn1's
// backward sums from (x tile, dz).  (Round 3's BatchNorm-in-the-staging fold was measured slower and is gone.)
#include "common.h"
#include "group.h"
#include <stdlib.h>
#include "../../include/semseg_hip.h"

namespace {

constexpr int kStatReplicas = 8;   // as conv_tile.hip (ssa_bn_stat_replicas)

#ifdef SSA_TILE_TIMING
#define SSA_STAMP(k) do { if (tdbg && it < 24) { tlds[it * 8 + (k)] = (long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SSA_STAMP(k) do { } while (0)
#endif

struct TilePArgs {
  const bf16_t* x; const uint4* wfrag; bf16_t* y; double* stats;
  const bf16_t* aux; const float* coef;                    // epilogue tile; [4][Cout] table of aux_mode 2
  int ldx, Cin, ldy, H, W, Cout, nb_total, tiles_x, tiles_y, ldaux;
  int total_tiles, tiles_per_wg, ngroups, nwg;
  int relu;                                                 // aux_mode 3 / 4
};

// Bijective XCD-aware order (block b runs on XCD b % 8): XCD x gets one contiguous range of work items.
__device__ __forceinline__ int xcd_order(int v, int n) {
  const int q = n >> 3, r = n & 7, x = v & 7, k = v >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// workgroup barrier that waits for this wave's LDS traffic only (vector-memory loads / stores / DMAs stay in flight)
__device__ __forceinline__ void lds_barrier() {
#ifdef SSA_EMU
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// lanes l and l + 32 exchange: afterwards (a, b) of a lane < 32 = (own a, partner's a), of a lane >= 32 =
// (partner's b, own b)   [v_permlane32_swap: rows 2-3 of the first operand <-> rows 0-1 of the second]
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

// sum over the 32 lanes of this lane's half of the wave; valid in lanes 16-31 / 48-63 afterwards
__device__ __forceinline__ float half_sum32(float v) {
#ifdef SSA_EMU
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
#define SSA_DPP_ADD(ctrl, rmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))
  SSA_DPP_ADD(0xB1, 0xf);     // quad_perm [1,0,3,2]
  SSA_DPP_ADD(0x4E, 0xf);     // quad_perm [2,3,0,1]
  SSA_DPP_ADD(0x141, 0xf);    // row_half_mirror
  SSA_DPP_ADD(0x140, 0xf);    // row_mirror
  SSA_DPP_ADD(0x142, 0xa);    // row_bcast:15 into rows 1 and 3
#undef SSA_DPP_ADD
  return v;
#endif
}

template <int NB, int TPC, int AUXM>
struct ConvTileP {
  // aux_mode 3 / 4 (inference): the conv's BatchNorm as the epilogue -- z = act(scale * y + shift [+ residual]) with
  // y the 16-bit-rounded conv output, i.e. bit for bit what the separate apply pass computes from the stored y;
  // 3: no residual, 4: residual tile in `aux`.  The ReLU is a run-time flag.
  static constexpr bool AUX = AUXM == 1 || AUXM == 2 || AUXM == 4;
  static constexpr bool AFFINE = AUXM >= 3;
  typedef TilePArgs Args;
  static constexpr int NT = 256;
  static constexpr int CK = 48;                      // input channels per unit
  static constexpr int TW = 32, TH = 4;
  static constexpr int HW_ = TW + 2, HH_ = TH + 2, NPIX = HH_ * HW_;
  static constexpr int PSB = CK * 2 + 16;            // halo pixel stride (bytes): an odd number of 16-byte slots
  static constexpr int CP = CK / 8;                  // 16-byte pieces per halo pixel
  static constexpr int NA = (NT / CP) * CP;          // staging threads: thread t always moves channel group t % CP
  static constexpr int RP = NA / CP;                 // halo pixels per staging pass
  static constexpr int IT = (NPIX + RP - 1) / RP;    // halo loads per thread and unit
  static constexpr int CST = CK / 16;
  static constexpr int TAPS = 9, NSTAGE = TAPS / TPC, STAGE_KS = TPC * CST;
  static constexpr int NFRAG = NB * STAGE_KS;        // 1 KiB filter fragments per stage
  static constexpr int ND = (NFRAG + 3) / 4;         // DMAs per wave and stage (the last ones duplicated: every wave issues ND)
  static constexpr int STAGE_BYTES = NFRAG * 1024;
  static constexpr int NBUF = 3;
  static constexpr int HALO_RAW = NPIX * PSB;
  static constexpr int HALO_BYTES = (HALO_RAW + 1023) / 1024 * 1024;
  static constexpr int NAUX = NB * 2;                // 16-byte epilogue pieces per lane and tile
  static constexpr size_t LDS = (size_t)HALO_BYTES + (size_t)NBUF * STAGE_BYTES;
  static_assert(NSTAGE * TPC == TAPS, "taps per stage must divide the tap count");
  static_assert(HALO_BYTES - HALO_RAW >= 2 * NB * 32 * 4, "coefficient table of aux_mode 2 does not fit behind the halo image");
  static_assert(NSTAGE >= 2, "streamed filter");
  static_assert(LDS <= 53 * 1024, "three workgroups per CU");
  static_assert(4 * 2 * NB * 32 * 4 <= HALO_RAW, "statistics reduction does not fit");

  // DMA of one filter stage into dst: ND fragments per wave.  voff[f] = byte offset of this lane's piece of fragment
  // f of the wave relative to the stage's first k-step (tile- and stage-invariant, computed once per workgroup);
  // sbase = the filter + the stage's (chunk, first tap) k-step offset (wave-uniform)
  static __device__ __forceinline__ void stage_filter(const unsigned char* sbase, const unsigned (&voff)[ND],
                                                      unsigned dst, int wave) {      // dst: LDS address (ssa_lds_addr)
#pragma unroll
    for (int f = 0; f < ND; ++f) {
      const int fi = min(f * 4 + wave, NFRAG - 1);      // wave-uniform; the last fragments are issued twice
      ssa_glds16_untracked_m0(sbase, voff[f], dst + (unsigned)fi * 1024u);
    }
  }

  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int /*gx*/) {
    const bf16_t* __restrict__ x = a.x;
    const uint4* __restrict__ wfrag = a.wfrag;
    bf16_t* __restrict__ y = a.y;
    double* __restrict__ stats = a.stats;
    const bf16_t* __restrict__ aux = a.aux;
    const float* __restrict__ coef = a.coef;
    const int ldx = a.ldx, Cin = a.Cin, ldy = a.ldy, H = a.H, W = a.W, Cout = a.Cout;
    const int nb_total = a.nb_total, tiles_x = a.tiles_x, tiles_y = a.tiles_y, ldaux = a.ldaux;
    SSA_DYN_LDS(unsigned char, smem);
    unsigned char* Bs = smem + HALO_BYTES;
    float* ctab = reinterpret_cast<float*>(smem + HALO_RAW);      // aux_mode 2: [2][NB*32] mask scale / shift

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
    if (n_iter <= 0) return;

    // ---- staging role of this thread: channel group cg of halo pixels prow, prow + RP, ...
    const bool stg = tid < NA;
    const int cg = tid % CP, prow = tid / CP;
    uint4 v[IT];
    unsigned okmask = 0;                       // bit i: piece i lies inside the image
    // tile-invariant part of this thread's pieces: halo coordinates (hy << 8 | hx, 0xffff: no such piece) and the
    // element offset relative to the tile's first halo pixel; invalid lanes load the tile's own first pixel instead
    // (always inside the image) and are zeroed at staging time -- the loads issue back to back, unconditionally
    int hyx[IT];
    unsigned rel[IT];
    unsigned havemask = 0;
    const unsigned rel_c = (unsigned)((W + 1) * ldx + cg * 8);      // halo pixel (1, 1) = output pixel (0, 0) of the tile
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int pix = prow + i * RP;
      const int hy = pix / HW_, hx = pix - hy * HW_;
      const bool have = stg && pix < NPIX;
      hyx[i] = have ? ((hy << 8) | hx) : 0xffff;
      rel[i] = have ? (unsigned)((hy * W + hx) * ldx + cg * 8) : rel_c;
      havemask |= (have ? 1u : 0u) << i;
    }
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
    // (wave-uniform 64-bit base: the tile's halo origin, possibly outside the buffer and never dereferenced there)
    // + (per-lane 32-bit offset); a tile whose whole halo lies inside the image skips the per-piece bounds tests -- a
    // lone wave issues an instruction every 5-8 clocks, so the ~35 instructions are ~200 clocks per unit
    auto fetch = [&](int cc) {
      const int x0 = f_tx * TW, y0 = f_ty * TH;
      const bf16_t* xb = x + (((long)f_b * H + (y0 - 1)) * W + (x0 - 1)) * ldx + cc * CK;
      const bool interior = x0 >= 1 && y0 >= 1 && x0 + TW + 1 <= W && y0 + TH + 1 <= H;
      if (interior) {
        okmask = havemask;
#pragma unroll
        for (int i = 0; i < IT; ++i) v[i] = *reinterpret_cast<const uint4*>(xb + rel[i]);
      } else {
        okmask = 0;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          const int iy = y0 - 1 + (hyx[i] >> 8), ix = x0 - 1 + (hyx[i] & 255);
          const bool ok = hyx[i] != 0xffff && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
          v[i] = *reinterpret_cast<const uint4*>(xb + (ok ? rel[i] : rel_c));
          okmask |= (ok ? 1u : 0u) << i;
        }
      }
    };
    auto advance = [&](int* b, int* ty, int* tx) {
      if (++*tx == tiles_x) {
        *tx = 0;
        if (++*ty == tiles_y) { *ty = 0; ++*b; }
      }
    };
    // registers -> LDS halo image (pixels outside the image are zero).  Branch-free: a thread without an i-th piece
    // writes into the unused 16-byte pad behind a halo pixel's 96 bytes -- with the stores (and the vmcnt waits in
    // front of them) under per-piece branches the compiler loses track of which loads have landed and drains the
    // vector-memory queue (vmcnt(0): the filter DMAs just issued included) before it re-uses one of their registers
    const int pad_slot = (tid % NPIX) * PSB + CK * 2;
    auto stage = [&]() {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int pix = prow + i * RP;
        const uint4 o = ((okmask >> i) & 1u) ? v[i] : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(smem + (hyx[i] != 0xffff ? pix * PSB + cg * 16 : pad_slot)) = o;
      }
    };

    // ---- epilogue role of this lane: pixel (wave's tile row, column lane & 31); after the exchange it holds, per
    // n-block nb and m = 0, 1, the 16-byte piece of channels  nb0*32 + nb*32 + 16*m + 8*(lane >> 5) .. + 7
    const int eh = lane >> 5, epx = lane & 31;
    const int ech = nb0 * 32 + 8 * eh;                            // first channel of piece (nb 0, m 0)
    constexpr int NSTAT = (AUXM == 1 || AFFINE) ? 1 : NAUX * 8;
    float S[NSTAT], Q[NSTAT];  // statistics of this lane's channels over the strip
#pragma unroll
    for (int j = 0; j < NSTAT; ++j) { S[j] = 0.f; Q[j] = 0.f; }
    if constexpr (AUXM == 2 || AFFINE) {
      for (int i = tid; i < 2 * NB * 32; i += NT) {
        const int k = i / (NB * 32), c = nb0 * 32 + i - k * (NB * 32);
        ctab[i] = c < Cout ? coef[k * Cout + c] : 0.f;
      }
    }
    uint4 auxv[AUX ? NAUX : 1];

    const int a_off = (wave * HW_ + (lane & 31)) * PSB + (lane >> 5) * 16;   // pixel fragment rows = the wave's tile row

    f32x16_t acc[NB];     // (two accumulators for even / odd k-steps measured the same: the loop is LDS bound, call D)

#ifdef SSA_TILE_TIMING
    long* tdbg = (AUXM == 0 && bx == 0 && tid == 0) ? reinterpret_cast<long*>(const_cast<float*>(coef)) : nullptr;
    long* tlds = reinterpret_cast<long*>(smem + LDS);          // 1.5 KB past the kernel's own LDS (the timing build asks for it)
#endif
    // ---- prologue: the filter (slice / first two stages) on its way into LDS, first halo into registers
    unsigned voff[ND];
    {
      constexpr int PER_NB = TPC * CST;
#pragma unroll
      for (int f = 0; f < ND; ++f) {
        const int fi = min(f * 4 + wave, NFRAG - 1);
        const int nb = fi / PER_NB, rem = fi - nb * PER_NB;
        const int tl = rem / CST, j = rem - tl * CST;
        const int nbg = min(nb0 + nb, nb_total - 1);       // n-blocks past the end re-read the last one (never stored)
        voff[f] = (unsigned)(((nbg * ksteps_total + tl * csteps_total + j) * 64 + lane) * 16);
      }
    }
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(wfrag);
    auto stage_base = [&](int cc_, int tap0) { return wbytes + (long)(tap0 * csteps_total + cc_ * CST) * 1024; };
    const unsigned bs_lds = ssa_lds_addr(Bs);          // LDS address of the filter ring, taken once
    stage_filter(stage_base(0, 0), voff, bs_lds, wave);
    stage_filter(stage_base(0, TPC), voff, bs_lds + STAGE_BYTES, wave);
    fetch(0);
    int s = 0;                                         // global filter-stage counter (streamed variant)
    int cc = 0;
    for (int it = 0; it < n_iter; ++it) {
      // everyone is past the barrier that ended the previous unit's MFMAs: the halo image is free
      SSA_STAMP(0);
      stage();
      SSA_STAMP(1);
      if (it == 0) ssa_wait_vm_barrier<ND, 0>();       // + filter stage 0 landed
      else lds_barrier();                              // halo image visible
      SSA_STAMP(2);
      const bool last_chunk = cc + 1 == nchunk;
      int ccn = cc + 1;
      if (last_chunk) { ccn = 0; if (it + 1 < n_iter) advance(&f_b, &f_ty, &f_tx); }
      const int b_ = c_b, x0 = c_tx * TW, y0 = c_ty * TH;
      const int oy = y0 + wave, ox = x0 + epx;
      const bool pix_ok = oy < H && ox < W;
      const long opix = (long)b_ * H * W + (long)min(oy, H - 1) * W + min(ox, W - 1);
      if (cc == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
      }
      SSA_STAMP(3);
#pragma unroll
      for (int st = 0; st < NSTAGE; ++st) {
        {
          // stage s + 2 of the continuous filter stream (past the end of the strip: a stage nobody reads)
          const int cc2 = st + 2 < NSTAGE ? cc : ccn;
          const int tap2 = ((st + 2) % NSTAGE) * TPC;
          stage_filter(stage_base(cc2, tap2), voff, bs_lds + ((s + 2) % 3) * STAGE_BYTES, wave);
        }
        if (st == 0) {
          // next unit's halo (the last unit re-reads its own: the count of loads in flight stays fixed) and this
          // tile's epilogue operand: in flight during the MFMAs, behind the DMAs
          fetch(ccn);
          if constexpr (AUX) {
            if (last_chunk) {
              const bf16_t* ab = aux + opix * ldaux + ech;
#pragma unroll
              for (int p = 0; p < NAUX; ++p) {
                const int cb = ech + (p >> 1) * 32 + (p & 1) * 16;
                auxv[p] = *reinterpret_cast<const uint4*>(ab + (cb < Cout ? (p >> 1) * 32 + (p & 1) * 16 : 0));
              }
            }
          }
          // the loads are ISSUED here, ahead of the MFMAs (left alone the compiler sinks them below the k-loop to re-use
          // their registers for fragments)
          asm volatile("" ::: "memory");
        }
        const unsigned char* Bc = Bs + (s % 3) * STAGE_BYTES + lane * 16;
        // fragments of k-step ksl + RD - 1 are read while the MFMAs of k-step ksl run: a ring of RD register sets
        constexpr int RD = 4;                 // (3 and 6 measured the same, call E)
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
          for (int nb = 0; nb < NB; ++nb) acc[nb] = ssa_mfma32(rb[ksl % RD][nb], ra[ksl % RD], acc[nb]);   // D[channel][pixel]
          if (ksl + RD - 1 < STAGE_KS) __builtin_amdgcn_sched_group_barrier(0x100, 1 + NB, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
        }
        if (st == 0) SSA_STAMP(4);
        // stage s + 1 landed (everything issued before this stage's own DMAs / loads); buffer s % 3 and, after the
        // last stage, the halo image are free
        if (st == 0) {
          if (AUX && last_chunk) ssa_wait_vm_barrier<ND + IT + NAUX, 0>();
          else ssa_wait_vm_barrier<ND + IT, 0>();
        } else {
          ssa_wait_vm_barrier<ND, 0>();
        }
        ++s;
        if (st == 0) SSA_STAMP(5);
      }
      SSA_STAMP(6);
      if (last_chunk) {
        // ---- epilogue in registers: bf16 pairs -> 16-byte pieces by lane exchange -> HBM
        bf16_t* yb = y + opix * ldy + ech;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          unsigned w[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = acc[nb][g * 4 + j];
            w[g][0] = f2bf_pair(f[0], f[1]);
            w[g][1] = f2bf_pair(f[2], f[3]);
          }
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            swap32(w[2 * m][0], w[2 * m + 1][0]);
            swap32(w[2 * m][1], w[2 * m + 1][1]);
            uint4 o = make_uint4(w[2 * m][0], w[2 * m][1], w[2 * m + 1][0], w[2 * m + 1][1]);
            const int p = nb * 2 + m;
            const int coff = nb * 32 + m * 16;
            const bool ok = pix_ok && ech + coff < Cout;
            if constexpr (AUXM == 1) {
              float f[8], xv[8];
              unpack8(o, f);
              unpack8(auxv[p], xv);
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] += xv[j];
              o = pack8(f);
            } else if constexpr (AFFINE) {
              float f[8];
              unpack8(o, f);
              const float4 ma0 = *reinterpret_cast<const float4*>(ctab + coff + 8 * eh);
              const float4 ma1 = *reinterpret_cast<const float4*>(ctab + coff + 8 * eh + 4);
              const float4 mb0 = *reinterpret_cast<const float4*>(ctab + NB * 32 + coff + 8 * eh);
              const float4 mb1 = *reinterpret_cast<const float4*>(ctab + NB * 32 + coff + 8 * eh + 4);
              const float ma[8] = {ma0.x, ma0.y, ma0.z, ma0.w, ma1.x, ma1.y, ma1.z, ma1.w};
              const float mb[8] = {mb0.x, mb0.y, mb0.z, mb0.w, mb1.x, mb1.y, mb1.z, mb1.w};
#pragma unroll
              for (int j = 0; j < 8; ++j) f[j] = f[j] * ma[j] + mb[j];          // (the arithmetic of bn_apply_rows)
              if constexpr (AUXM == 4) {
                float xv[8];
                unpack8(auxv[p], xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += xv[j];
              }
              if (a.relu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
              }
              o = pack8(f);
            } else if constexpr (AUXM == 2) {
              float f[8], xv[8];
              unpack8(o, f);
              unpack8(auxv[p], xv);
              const float4 ma0 = *reinterpret_cast<const float4*>(ctab + coff + 8 * eh);
              const float4 ma1 = *reinterpret_cast<const float4*>(ctab + coff + 8 * eh + 4);
              const float4 mb0 = *reinterpret_cast<const float4*>(ctab + NB * 32 + coff + 8 * eh);
              const float4 mb1 = *reinterpret_cast<const float4*>(ctab + NB * 32 + coff + 8 * eh + 4);
              const float ma[8] = {ma0.x, ma0.y, ma0.z, ma0.w, ma1.x, ma1.y, ma1.z, ma1.w};
              const float mb[8] = {mb0.x, mb0.y, mb0.z, mb0.w, mb1.x, mb1.y, mb1.z, mb1.w};
              if (ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const float gm = (xv[j] * ma[j] + mb[j]) > 0.f ? f[j] : 0.f;
                  S[p * 8 + j] += gm;
                  Q[p * 8 + j] += gm * xv[j];
                }
              }
            } else {
              if (stats != nullptr && ok) {
                float f[8];
                unpack8(o, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { S[p * 8 + j] += f[j]; Q[p * 8 + j] += f[j] * f[j]; }
              }
            }
            if (ok) *reinterpret_cast<uint4*>(yb + coff) = o;
          }
        }
      }
      SSA_STAMP(7);
      if (last_chunk) advance(&c_b, &c_ty, &c_tx);
      cc = ccn;
    }

#ifdef SSA_TILE_TIMING
    if (tdbg)
      for (int i = 0; i < 24 * 8; ++i) tdbg[i] = i < n_iter * 8 ? tlds[i] : 0;
#endif
    // ---- statistics of the strip: 32 pixel lanes -> wave -> workgroup -> one fp64 atomic per channel
    if constexpr (AUXM == 1 || AFFINE) {
      return;          // residual add / inference epilogue: no statistics (stats is NULL by contract)
    } else {
      if (stats == nullptr) return;
#pragma unroll
      for (int j = 0; j < NAUX * 8; ++j) { S[j] = half_sum32(S[j]); Q[j] = half_sum32(Q[j]); }
      __syncthreads();                                 // every wave is done with the halo image / the filter DMAs
      float* red = reinterpret_cast<float*>(smem);     // [4 waves][2][NB*32]
      if (epx == 31) {
#pragma unroll
        for (int p = 0; p < NAUX; ++p)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = (p >> 1) * 32 + (p & 1) * 16 + 8 * eh + j;
            red[(wave * 2 + 0) * NB * 32 + c] = S[p * 8 + j];
            red[(wave * 2 + 1) * NB * 32 + c] = Q[p * 8 + j];
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
          // aux_mode 2 accumulated sum(m*dz*x); the consumer wants sum(m*dz*xhat), xhat = (x - mean) * invstd
          if constexpr (AUXM == 2) qd = (double)coef[3 * Cout + n] * (qd - (double)coef[2 * Cout + n] * sd);
          atomicAdd(&st[n], sd);
          atomicAdd(&st[Cout + n], qd);
        }
      }
    }
  }
};

template <int AUXM>
struct ConvTilePK {
  typedef TilePArgs Args;
  static constexpr int NT = 256;
  static constexpr int WPE = 3;                  // three workgroups per CU: <= 168 registers
  typedef ConvTileP<1, 3, AUXM> V;
#ifdef SSA_TILE_TIMING
  static constexpr size_t LDS = V::LDS + 1536;
#else
  static constexpr size_t LDS = V::LDS;
#endif
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int by, const int gx) { V::run(a, bx, by, gx); }
};

static thread_local int g_strip_units = 0;     // work units (27 MFMAs per wave) per workgroup, 0 = per problem

template <int AUXM>
int launch_p(const ssa_conv_desc& d, const TilePArgs& a0, hipStream_t s) {
  TilePArgs a = a0;
  const int nchunk = d.Cin / 48;
  a.nb_total = (d.Cout + 31) / 32;
  a.tiles_x = (d.W + 31) / 32;
  a.tiles_y = (d.H + 3) / 4;
  a.total_tiles = d.B * a.tiles_x * a.tiles_y;
  a.ngroups = a.nb_total;
  int units = g_strip_units;
  if (units <= 0) {
    // a launch of its own: ~3 workgroups per CU from this problem alone
    const long total = (long)a.total_tiles * a.ngroups * nchunk;
    units = (int)((total + 767) / 768);
  }
  if (units > 64) units = 64;
  int tpw = units / nchunk;
  if (tpw < 1) tpw = 1;
  const int nstrips = (a.total_tiles + tpw - 1) / tpw;
  a.tiles_per_wg = (a.total_tiles + nstrips - 1) / nstrips;
  a.nwg = ((a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg) * a.ngroups;
  return ssa::submit<ConvTilePK<AUXM>>(a, a.nwg, 1, ConvTilePK<AUXM>::LDS, s);
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

int ssa_conv2d_tile_p(const ssa_conv_desc* dp, const void* x, const void* w_frag, const float* bias, void* y,
                      double* stats, const void* aux, int ldaux, const float* coef, int aux_mode, void* stream) {
  if (!dp || !x || !w_frag || !y) return SSA_EINVAL;
  if (!ssa_conv2d_tile_p_supported(dp) || bias) return SSA_EUNSUPPORTED;   // the trunk convs have no bias (hrnetv2.py:31-34)
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_frag)) & 15u)
    return SSA_EINVAL;
  if (aux_mode < 0 || aux_mode > 4) return SSA_EINVAL;
  // 3 / 4: the conv's inference BatchNorm as the epilogue (coef = its [4][Cout] table), 4 with the ReLU; aux = the
  // residual tile or NULL
  const bool affine = aux_mode >= 3;
  if (affine && (!coef || stats)) return SSA_EINVAL;
  if ((aux_mode == 1 || aux_mode == 2 || (affine && aux)) && (!aux || ldaux % 8 || (reinterpret_cast<uintptr_t>(aux) & 15u))) return SSA_EINVAL;
  if (aux_mode == 2 && (!coef || !stats)) return SSA_EINVAL;
  if (aux_mode == 1 && stats) return SSA_EINVAL;
  if (dp->ldy % 8) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  TilePArgs a;
  a.x = (const bf16_t*)x; a.wfrag = (const uint4*)w_frag;
  a.y = (bf16_t*)y; a.stats = stats; a.aux = (const bf16_t*)aux; a.coef = coef;
  a.ldx = d.ldx; a.Cin = d.Cin; a.ldy = d.ldy; a.H = d.H; a.W = d.W; a.Cout = d.Cout;
  a.ldaux = ldaux;
  a.relu = aux_mode == 4;
  a.nb_total = a.tiles_x = a.tiles_y = a.total_tiles = a.tiles_per_wg = a.ngroups = a.nwg = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (aux_mode) {
    case 0: return launch_p<0>(d, a, s);
    case 1: return launch_p<1>(d, a, s);
    case 2: return launch_p<2>(d, a, s);
    default: return aux ? launch_p<4>(d, a, s) : launch_p<3>(d, a, s);
  }
}

}  // extern "C"
