// 48-channel-block geometry of the persistent halo-tile convolution for the 3x3 stride-1 convs of the HRNetV2-W48
// trunk (network/hrnetv2.py:31-66), forward and data gradient (gfx950 / MI355X).  OPT-IN this round
// (SSA_TILE_Q=1): checked on the CPU emulation and by the op-level tests, not yet the default -- conv_tile_p.hip is.
//
// Why another geometry (profiles/r04_notes.md, "what bounds the trunk conv now"): conv_tile_p.hip computes a
// 32-pixel x 32-channel block per wave with v_mfma_f32_32x32x16 -- 2 KB of LDS reads per MFMA, the CU's whole LDS
// bandwidth at the MFMA rate, 32-channel n-blocks that waste a quarter of the work on 48 channels, and a chain of 8
// dependent units for a 384-channel tile.  Every width of the W48 trunk (48 / 96 / 192 / 384) is a multiple of 48, so:
//   * a wave computes (PB x 16 pixels) x 48 output channels with v_mfma_f32_16x16x32: PB pixel fragments + 3 filter
//     fragments feed 3*PB MFMAs per k-step of 32 -- 0.58 KB of LDS reads per MFMA at PB = 4 (3.4x fewer bytes per
//     FLOP), no n-block waste;
//   * K runs flattened over (tap, channel) of a 48-channel chunk: 432 = 13.5 k-steps, padded to 14 with zero filter
//     rows (the pixel operand of the padding re-reads the lane's previous k-group: finite data, times zero);
//   * a workgroup = 4 waves = 16 x (4*PB) pixels x 48 channels; the four waves share the filter stage in LDS; a
//     "unit" = one (tile, 48-channel chunk) = 14 k-steps = 42*PB MFMAs per wave, two filter stages of 21 KB through
//     a ring of two LDS buffers behind counted barriers (three barriers per unit; conv_tile_p: four per 27 MFMAs);
//     the 48-channel layers' whole filter slice (42 KB) IS the ring: loaded once per workgroup;
//   * PB is chosen per PROBLEM at run time (one kernel, so a level's problems still share one grouped launch): 2 up to
//     192 channels, 1 for 384 (choose_pb: measured; 4 through SSA_TILE_Q_PB) -- the chain of dependent units per tile
//     stays at 1-4 x 84 and 8 x 42 MFMAs instead of growing with the channel count;
//   * the accumulator of a lane holds 4 consecutive channels of one pixel (filter as the A operand); the epilogue
//     completes 16-byte pieces with v_permlane16_swap (channel blocks 0 / 1 against each other, block 2 of two pixel
//     rows against each other) and stores from registers;
//   * 78 KB of LDS, <= 256 registers: two workgroups per CU.
// Epilogues as conv_tile_p.hip: BatchNorm batch statistics; aux_mode 1: + residual gradient; aux_mode 2: bn1's
// backward sums from (x tile, dz).  Filter layout: ssa_pack_filter mode 2 / 3 with the flag 8 ("Q fragments").
#include "common.h"
#include "group.h"
#include <stdlib.h>
#include "../../include/semseg_hip.h"

namespace {

constexpr int kStatReplicas = 8;   // as conv_tile.hip (ssa_bn_stat_replicas)

// -DSSA_TILE_TIMING (experiment build, tools/expbuild.sh): s_memtime stamps of wave 0 of workgroup 0 at eight points of
// each of its first 24 units, written through the `coef` pointer of a plain (aux_mode 0) launch (tools/tilebench.py --timing-q)
#ifdef SSA_TILE_TIMING
#define SSA_QSTAMP(k) do { if (tdbg && it < 24) { tlds[it * 8 + (k)] = (long)__builtin_amdgcn_s_memtime(); } } while (0)
#else
#define SSA_QSTAMP(k) do { } while (0)
#endif

struct TileQArgs {
  const bf16_t* x; const uint4* wfrag; bf16_t* y; double* stats;
  const bf16_t* aux; const float* coef;                    // epilogue tile; [4][Cout] table of aux_mode 2
  int ldx, Cin, ldy, H, W, Cout, nt_total, tiles_x, tiles_y, ldaux;
  int total_tiles, tiles_per_wg, nwg, pb;
};

// Bijective XCD-aware order (block b runs on XCD b % 8): XCD x gets one contiguous range of work items.
__device__ __forceinline__ int xcd_order_q(int v, int n) {
  const int q = n >> 3, r = n & 7, x = v & 7, k = v >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ void lds_barrier_q() {
#ifdef SSA_EMU
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// 16-lane rows r = lane >> 4: rows 2i and 2i + 1 exchange: afterwards (a, b) of an even row = (own a, partner's a), of
// an odd row = (partner's b, own b)   [v_permlane16_swap: odd rows of the first operand <-> even rows of the second;
// pinned on the device by ssa_probe_swap16]
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
#ifdef SSA_EMU
  const unsigned pa = __shfl_xor(a, 16, 64), pb = __shfl_xor(b, 16, 64);
  if (((threadIdx.x >> 4) & 1) == 0) b = pa; else a = pb;
#else
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
#endif
}

// sum over the 16 lanes of this lane's row; valid in every lane afterwards
__device__ __forceinline__ float row_sum16(float v) {
#ifdef SSA_EMU
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
#define SSA_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
  SSA_DPP_ADD(0xB1);      // quad_perm [1,0,3,2]
  SSA_DPP_ADD(0x4E);      // quad_perm [2,3,0,1]
  SSA_DPP_ADD(0x141);     // row_half_mirror
  SSA_DPP_ADD(0x140);     // row_mirror
#undef SSA_DPP_ADD
  return v;
#endif
}

// NST = filter stages per unit: 2 (7 + 7 k-steps: a ring of two 21 KB buffers, 64-78 KB of LDS, two workgroups per CU) or
// 3 (5 + 5 + 4 k-steps: a ring of two 15 KB buffers, <= 51 KB at PB <= 2: THREE workgroups per CU)
template <int PB, int AUXM, int NST = 2>
struct ConvTileQBody {
  static constexpr bool AUX = AUXM != 0;
  static constexpr int NT = 256;
  static constexpr int CK = 48;                      // input channels per unit = output channels per workgroup
  static constexpr int TW = 16, TH = 4 * PB;         // wave w computes rows w*PB .. w*PB + PB - 1 of the tile
  static constexpr int HW_ = TW + 2, HH_ = TH + 2, NPIX = HH_ * HW_;
  static constexpr int PSB = CK * 2 + 16;            // halo pixel stride (bytes): 7 slots of 16 bytes, conflict-free fragment reads
  static constexpr int CP = CK / 8;                  // 16-byte pieces per halo pixel
  static constexpr int NA = (NT / CP) * CP;          // staging threads: thread t always moves channel group t % CP
  static constexpr int RP = NA / CP;                 // halo pixels per staging pass
  static constexpr int IT = (NPIX + RP - 1) / RP;    // halo loads per thread and unit
  static constexpr int KS = 14;                      // k-steps of 32 per unit: 9 taps x 48 channels = 432, padded to 448
  static constexpr int KREAL = 9 * CK;
  static_assert(NST == 2 || NST == 3, "filter stages per unit");
  static constexpr int SKS = NST == 2 ? 7 : 5;       // k-steps of the longest filter stage
  // first k-step of stage st (st = NST: the end)
  static constexpr int k0(int st) { return NST == 2 ? 7 * st : (st == 3 ? 14 : 5 * st); }
  static constexpr int NFRAG = SKS * 3;              // 1 KiB filter fragments of the longest stage (3 channel blocks of 16 per k-step)
  static constexpr int ND = (NFRAG + 3) / 4;         // DMAs per wave and stage (the last ones duplicated: every wave issues ND)
  static constexpr int STAGE_BYTES = NFRAG * 1024;   // one ring buffer
  static constexpr int UNIT_BYTES = 14 * 3 * 1024;   // filter bytes per (48-channel n-tile, chunk)
  static_assert((ND - 1) * 4 <= (k0(NST) - k0(NST - 1)) * 3, "only the last DMA of a wave may be a duplicate");
  static constexpr int HALO_RAW = NPIX * PSB;
  static constexpr int HALO_BYTES = (HALO_RAW + 2 * CK * 4 + 1023) / 1024 * 1024;   // + the coefficient table of aux_mode 2
  static constexpr int NPA = PB;                     // 16-byte epilogue pieces per lane from channel blocks 0 / 1
  static constexpr int NPB = PB >= 2 ? PB / 2 : 1;   // ... from channel block 2
  static constexpr int NPC = NPA + NPB;
  static constexpr size_t LDS = (size_t)HALO_BYTES + 2 * (size_t)STAGE_BYTES;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  static_assert(NST == 2 || PB == 4 || LDS <= 53 * 1024, "three workgroups per CU");
  static_assert(4 * 2 * CK * 4 <= HALO_RAW, "statistics reduction does not fit");
  static_assert(IT <= 8, "halo pieces per thread");

  // nfrag = fragments of the stage being loaded; vlast = this lane's source offset of its last (possibly clamped) DMA
  static __device__ __forceinline__ void stage_filter(const unsigned char* sbase, const unsigned (&voff)[ND], unsigned vlast,
                                                      int nfrag, unsigned dst, int wave) {   // dst: LDS address (ssa_lds_addr)
#pragma unroll
    for (int f = 0; f < ND; ++f) {
      const int fi = min(f * 4 + wave, nfrag - 1);      // wave-uniform; the last fragments are issued twice
      ssa_glds16_untracked_m0(sbase, f == ND - 1 ? vlast : voff[f], dst + (unsigned)fi * 1024u);
    }
  }

  static __device__ __forceinline__ void run(const TileQArgs& a, const int bx) {
    const bf16_t* __restrict__ x = a.x;
    bf16_t* __restrict__ y = a.y;
    double* __restrict__ stats = a.stats;
    const bf16_t* __restrict__ aux = a.aux;
    const float* __restrict__ coef = a.coef;
    const int ldx = a.ldx, Cin = a.Cin, ldy = a.ldy, H = a.H, W = a.W, Cout = a.Cout;
    const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, ldaux = a.ldaux;
    SSA_DYN_LDS(unsigned char, smem);
    unsigned char* Bs = smem + HALO_BYTES;
    float* ctab = reinterpret_cast<float*>(smem + HALO_RAW);      // aux_mode 2: [2][48] mask scale / shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_ = xcd_order_q(bx, a.nwg);
    const int strip = w_ / a.nt_total, nt = w_ - strip * a.nt_total;
    const int cbase = nt * CK;                                     // first output channel of this workgroup
    const int t_begin = strip * a.tiles_per_wg;
    const int t_end = min(a.total_tiles, t_begin + a.tiles_per_wg);
    const int nchunk = Cin / CK;
    const int n_iter = (t_end - t_begin) * nchunk;
    if (n_iter <= 0) return;
    const bool resident = nchunk == 1 && NST == 2;                 // the ring holds the whole filter slice

    // ---- staging role of this thread: channel group cg of halo pixels prow, prow + RP, ...
    const bool stg = tid < NA;
    const int cg = tid % CP, prow = tid / CP;
    uint4 v[IT];
    unsigned okmask = 0;                       // bit i: piece i lies inside the image
    // per-thread constants of its pieces: halo coordinates (hy << 8 | hx, 0xffff: no such piece) and the 32-bit
    // element offset relative to the tile's halo origin; a thread without an i-th piece (and, at the image border, a
    // piece outside the image) loads the tile's own first pixel instead and is zeroed at staging time -- the loads
    // issue back to back, unconditionally, as (wave-uniform 64-bit base) + (per-lane 32-bit offset): the phase stamps
    // (profiles/r04_notes.md, call S) showed 1,000 clocks per unit of per-piece 64-bit address arithmetic here
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
    int f_b, f_ty, f_tx;
    {
      f_tx = t_begin % tiles_x;
      const int r = t_begin / tiles_x;
      f_ty = r % tiles_y;
      f_b = r / tiles_y;
    }
    int c_b = f_b, c_ty = f_ty, c_tx = f_tx;
    auto fetch = [&](int cc) {
      const int x0 = f_tx * TW, y0 = f_ty * TH;
      // wave-uniform: the tile's halo origin (one row above / one pixel left of the tile: possibly outside the buffer,
      // never dereferenced there) and whether the whole halo lies inside the image
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
    // vector-memory queue (vmcnt(0): the filter DMAs just issued included) before it re-uses a register
    const int pad_slot = (tid % NPIX) * PSB + CK * 2;
    auto stage = [&]() {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int pix = prow + i * RP;
        const uint4 o = ((okmask >> i) & 1u) ? v[i] : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(smem + (hyx[i] != 0xffff ? pix * PSB + cg * 16 : pad_slot)) = o;
      }
    };

    // ---- MFMA / epilogue role of this lane: pixel column px of the tile, k-group / channel group g
    const int px = lane & 15, g = lane >> 4;
    const int chA = (g & 1) * 16 + (g >> 1) * 8;      // channels (within the 48) of this lane's pieces of blocks 0 / 1
    const int chB = 32 + (g >> 1) * 8;                //                                          ... of block 2
    const bool b_lane = PB >= 2 || (g & 1) == 0;      // PB = 1: the odd rows hold copies of the even rows' block-2 pieces
    float SA[AUXM == 1 ? 1 : 8], QA[AUXM == 1 ? 1 : 8], SB[AUXM == 1 ? 1 : 8], QB[AUXM == 1 ? 1 : 8];
#pragma unroll
    for (int j = 0; j < (AUXM == 1 ? 1 : 8); ++j) { SA[j] = QA[j] = SB[j] = QB[j] = 0.f; }
    if constexpr (AUXM == 2) {
      for (int i = tid; i < 2 * CK; i += NT) {
        const int k = i / CK, c = i - k * CK;
        ctab[i] = coef[k * Cout + cbase + c];
      }
    }
    uint4 auxv[AUX ? NPC : 1];
    // tile row of piece p, and its element offsets in y / aux relative to the tile's first pixel (channel included)
    auto prow_of = [&](int p) { return wave * PB + (p < NPA ? p : (PB >= 2 ? 2 * (p - NPA) + (g & 1) : 0)); };
    unsigned poff[NPC], paoff[AUX ? NPC : 1];
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
      poff[p] = (unsigned)((prow_of(p) * W + px) * ldy + (p < NPA ? chA : chB));
      if constexpr (AUX) paoff[p] = (unsigned)((prow_of(p) * W + px) * ldaux + (p < NPA ? chA : chB));
    }

    // pixel fragment of k-step ks: 8 channels of (tap, channel) = flattened k-group ks*32 + 8*g of the halo pixel
    // (row + kh, px + kw); the padding groups (k >= 432) re-read the lane's k-group 16 below (their filter rows are
    // zero).  ks is a compile-time constant at every call: two constants and a select per k-step instead of a
    // register per k-step
    const int g8 = g * 8;
    auto offk = [&](const int ks) {
      const int kk0 = ks * 32, tap0 = kk0 / CK, c0 = kk0 - tap0 * CK;
      const int conA = ((tap0 / 3) * HW_ + tap0 % 3) * PSB + c0 * 2;
      if (kk0 + 24 >= KREAL) return conA + (g8 >= KREAL - kk0 ? g8 - 16 : g8) * 2;      // the padded last k-step
      const int tap1 = tap0 + 1;
      const int conB = ((tap1 / 3) * HW_ + tap1 % 3) * PSB + (c0 - CK) * 2;
      return (g8 >= CK - c0 ? conB : conA) + g8 * 2;
    };
    const int b_off = (wave * PB * HW_ + px) * PSB;

    f32x4_t acc[3][PB];

    unsigned voff[ND];
#pragma unroll
    for (int f = 0; f < ND; ++f) voff[f] = (unsigned)((min(f * 4 + wave, NFRAG - 1) * 64 + lane) * 16);
    // (the last stage of three is a k-step shorter: its last DMA clamps to fragment 11 instead of 14)
    constexpr int NFRAG_LAST = (k0(NST) - k0(NST - 1)) * 3;
    const unsigned voff_short = (unsigned)((min((ND - 1) * 4 + wave, NFRAG_LAST - 1) * 64 + lane) * 16);
    const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(a.wfrag) + (long)nt * nchunk * UNIT_BYTES;
    auto stage_base = [&](int cc_, int st_) { return wbytes + (long)cc_ * UNIT_BYTES + k0(st_) * 3 * 1024; };
    auto load_stage = [&](int cc_, int st_, unsigned dst) {        // st_ is a compile-time constant at every call
      const int nfrag = (k0(st_ + 1) - k0(st_)) * 3;
      stage_filter(stage_base(cc_, st_), voff, nfrag == NFRAG ? voff[ND - 1] : voff_short, nfrag, dst, wave);
    };

    // ---- prologue: filter stage 0 on its way into LDS, first halo into registers
    const unsigned bs_lds = ssa_lds_addr(Bs);          // LDS address of the filter ring, taken once
    load_stage(0, 0, bs_lds);
    fetch(0);
    int s = 0;                                         // global filter-stage counter: stage s lives in buffer s & 1
    int cc = 0;
#ifdef SSA_TILE_TIMING
    long* tdbg = (AUXM == 0 && bx == 0 && tid == 0) ? reinterpret_cast<long*>(const_cast<float*>(coef)) : nullptr;
    long* tlds = reinterpret_cast<long*>(smem + ConvTileQBody<4, AUXM, 2>::LDS);   // 1.5 KB past the launch's own LDS (the timing build asks for it)
#endif
    for (int it = 0; it < n_iter; ++it) {
      // everyone is past the barrier that ended the previous unit's MFMAs: the halo image is free
      SSA_QSTAMP(0);
      stage();
      SSA_QSTAMP(1);
      if (it == 0) ssa_wait_vm_barrier<0, 0>();        // halo image visible + filter stage 0 landed
      else lds_barrier_q();                            // halo image visible
      SSA_QSTAMP(2);
      const bool last_chunk = cc + 1 == nchunk;
      int ccn = cc + 1;
      if (last_chunk) { ccn = 0; if (it + 1 < n_iter) advance(&f_b, &f_ty, &f_tx); }
      // this lane's piece p (p < NPA: pixel row p of the wave, channels chA; else rows 2j / 2j + 1, channels chB):
      // (wave-uniform 64-bit element offset of the tile's first pixel) + (per-lane 32-bit offset, fixed for the kernel)
      const int x0 = c_tx * TW, y0 = c_ty * TH;
      const long tile0 = ((long)c_b * H + y0) * W + x0;
      auto piece_ok = [&](int p) { return y0 + prow_of(p) < H && x0 + px < W && (p < NPA || b_lane); };
      if (cc == 0) {
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mb][pb][r] = 0.f;
      }
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        // stage s + 1 of the continuous filter stream into the buffer stage s - 1 left (past the end of the strip: a
        // stage nobody reads); a resident slice is complete after the first two stages
        if (!resident || s == 0) {
          if (st + 1 < NST) load_stage(cc, st + 1, bs_lds + ((s + 1) & 1) * STAGE_BYTES);
          else load_stage(ccn, 0, bs_lds + ((s + 1) & 1) * STAGE_BYTES);
        }
        if (st == 0) {
          // next unit's halo (the last unit re-reads its own: the count of loads in flight stays fixed) and this
          // tile's epilogue operand: in flight during the MFMAs, behind the DMAs
          fetch(ccn);
          if constexpr (AUX) {
            if (last_chunk) {
              const bf16_t* ab = aux + tile0 * ldaux + cbase;      // (pieces outside the image re-read the tile's first pixel)
#pragma unroll
              for (int p = 0; p < NPC; ++p)
                auxv[p] = *reinterpret_cast<const uint4*>(ab + (piece_ok(p) ? paoff[p] : 0u));
            }
          }
          // the loads are ISSUED here, ahead of the MFMAs (left alone the compiler sinks them below the k-loop to re-use
          // their registers for fragments, and then waits vmcnt(0) -- i.e. for the DMAs just issued -- before the loop)
          asm volatile("" ::: "memory");
          SSA_QSTAMP(3);
        }
        const unsigned char* Ac = Bs + (s & 1) * STAGE_BYTES + lane * 16;
        const unsigned char* Bc = smem + b_off;
        // filter fragments double-buffered over k-steps.  Pixel fragments: PB = 4 re-loads a fragment for the next
        // k-step right after its last MFMA of this one (9 MFMAs = 144 clocks ahead of its next use; 16 registers
        // instead of 32 -- the AUX = 2 body would not fit otherwise); PB <= 2 double-buffers them too (3-6 MFMAs
        // would not cover the LDS latency)
        constexpr int NFB = PB == 4 ? 1 : 2;
        bf16x8_t fa[2][3], fb[NFB][PB];
        auto rd_a = [&](int ksl) {
#pragma unroll
          for (int mb = 0; mb < 3; ++mb)
            fa[ksl & 1][mb] = *reinterpret_cast<const bf16x8_t*>(Ac + (ksl * 3 + mb) * 1024);
        };
        const int nks = k0(st + 1) - k0(st);          // k-steps of this stage (a constant: st is unrolled)
        auto rd_b = [&](int ksl, int pb) {
          fb[ksl % NFB][pb] = *reinterpret_cast<const bf16x8_t*>(Bc + offk(k0(st) + ksl) + pb * HW_ * PSB);
        };
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) rd_b(0, pb);
        rd_a(0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 + PB, 0);
#pragma unroll
        for (int ksl = 0; ksl < SKS; ++ksl) {
          if (ksl >= nks) break;
          if (ksl + 1 < nks) {
            rd_a(ksl + 1);
            if constexpr (NFB == 2) {
#pragma unroll
              for (int pb = 0; pb < PB; ++pb) rd_b(ksl + 1, pb);
            }
          }
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
              acc[mb][pb] = ssa_mfma16(fa[ksl & 1][mb], fb[ksl % NFB][pb], acc[mb][pb]);     // D[channel][pixel]
            if constexpr (NFB == 1) {
              if (ksl + 1 < nks) rd_b(ksl + 1, pb);
            }
          }
          // issue order of the k-step: the next k-step's 3 + PB fragment reads spread between this one's 3*PB MFMAs
          // (left alone the compiler shortens the fragments' live ranges by reading each one right in front of its
          // first MFMA, i.e. it exposes the LDS latency every three MFMAs)
          if (ksl + 1 < nks) {
            if constexpr (PB == 4) {
#pragma unroll
              for (int r = 0; r < 3; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            } else if constexpr (PB == 2) {
#pragma unroll
              for (int r = 0; r < 5; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            } else {
#pragma unroll
              for (int r = 0; r < 3; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              }
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * PB, 0);
          }
        }
        // stage s + 1 landed (it was issued before this stage's loads); buffer s & 1 and, after the second stage,
        // the halo image are free
        if (st == 0) SSA_QSTAMP(4);
        if (st == 0) {
          if (AUX && last_chunk) ssa_wait_vm_barrier<IT + NPC, 0>();
          else ssa_wait_vm_barrier<IT, 0>();
        } else {
          ssa_wait_vm_barrier<0, 0>();     // (nothing was issued after this stage's DMAs)
        }
        if (st == 0) SSA_QSTAMP(5);
        ++s;
      }
      SSA_QSTAMP(6);
      if (last_chunk) {
        // ---- epilogue in registers: 4 channels per (block, pixel row) -> 16-byte pieces by row exchange -> HBM
        unsigned w[3][PB][2];
#pragma unroll
        for (int mb = 0; mb < 3; ++mb)
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
            w[mb][pb][0] = f2bf_pair(acc[mb][pb][0], acc[mb][pb][1]);
            w[mb][pb][1] = f2bf_pair(acc[mb][pb][2], acc[mb][pb][3]);
          }
        // one 16-byte piece: fused epilogue, statistics (S / Q = this lane's sums of the piece's 8 channels), store
        bf16_t* yb = y + tile0 * ldy + cbase;
        auto finish = [&](uint4 o, const int p, const int ch, float (&S)[AUXM == 1 ? 1 : 8], float (&Q)[AUXM == 1 ? 1 : 8]) {
          const bool ok = piece_ok(p);
          if constexpr (AUXM == 1) {
            float f[8], xv[8];
            unpack8(o, f);
            unpack8(auxv[p], xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += xv[j];
            o = pack8(f);
          } else if constexpr (AUXM == 2) {
            float f[8], xv[8];
            unpack8(o, f);
            unpack8(auxv[p], xv);
            const float4 ma0 = *reinterpret_cast<const float4*>(ctab + ch);
            const float4 ma1 = *reinterpret_cast<const float4*>(ctab + ch + 4);
            const float4 mb0 = *reinterpret_cast<const float4*>(ctab + CK + ch);
            const float4 mb1 = *reinterpret_cast<const float4*>(ctab + CK + ch + 4);
            const float ma[8] = {ma0.x, ma0.y, ma0.z, ma0.w, ma1.x, ma1.y, ma1.z, ma1.w};
            const float mb[8] = {mb0.x, mb0.y, mb0.z, mb0.w, mb1.x, mb1.y, mb1.z, mb1.w};
            if (ok) {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float gm = (xv[j] * ma[j] + mb[j]) > 0.f ? f[j] : 0.f;
                S[j] += gm;
                Q[j] += gm * xv[j];
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
          if (ok) *reinterpret_cast<uint4*>(yb + poff[p]) = o;
        };
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
          // blocks 0 and 1 of pixel row p: even rows end up with 8 channels of block 0, odd rows of block 1
          unsigned a0 = w[0][p][0], a1 = w[0][p][1], b0 = w[1][p][0], b1 = w[1][p][1];
          swap16(a0, b0);
          swap16(a1, b1);
          finish(make_uint4(a0, a1, b0, b1), p, chA, SA, QA);
        }
#pragma unroll
        for (int p2 = 0; p2 < NPB; ++p2) {
          // block 2 of pixel rows 2j (-> even lane rows) and 2j + 1 (-> odd lane rows); PB = 1: the row against itself
          const int r0 = PB >= 2 ? 2 * p2 : 0, r1 = PB >= 2 ? r0 + 1 : 0;
          unsigned a0 = w[2][r0][0], a1 = w[2][r0][1], b0 = w[2][r1][0], b1 = w[2][r1][1];
          swap16(a0, b0);
          swap16(a1, b1);
          finish(make_uint4(a0, a1, b0, b1), NPA + p2, chB, SB, QB);
        }
        advance(&c_b, &c_ty, &c_tx);
      }
      SSA_QSTAMP(7);
      cc = ccn;
    }
#ifdef SSA_TILE_TIMING
    if (tdbg)
      for (int i = 0; i < 24 * 8; ++i) tdbg[i] = i < n_iter * 8 ? tlds[i] : 0;
#endif

    // ---- statistics of the strip: 16 pixel lanes -> (block 2: the two lane rows of a channel group) -> wave ->
    // workgroup -> one fp64 atomic per channel
    if constexpr (AUXM == 1) {
      return;          // residual add: no statistics (stats is NULL by contract)
    } else {
      if (stats == nullptr) return;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        SA[j] = row_sum16(SA[j]);
        QA[j] = row_sum16(QA[j]);
        SB[j] = row_sum16(SB[j]);
        QB[j] = row_sum16(QB[j]);
        SB[j] += __shfl_xor(SB[j], 16, 64);
        QB[j] += __shfl_xor(QB[j], 16, 64);
      }
      __syncthreads();                                 // every wave is done with the halo image
      float* red = reinterpret_cast<float*>(smem);     // [4 waves][2][48]
      if (px == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[(wave * 2 + 0) * CK + chA + j] = SA[j];
          red[(wave * 2 + 1) * CK + chA + j] = QA[j];
          if ((g & 1) == 0) {
            red[(wave * 2 + 0) * CK + chB + j] = SB[j];
            red[(wave * 2 + 1) * CK + chB + j] = QB[j];
          }
        }
      }
      __syncthreads();
      double* st = stats + (long)(strip % kStatReplicas) * 2 * Cout;
      if (tid < CK) {
        const int n = cbase + tid;
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
          sv += red[(w2 * 2 + 0) * CK + tid];
          qv += red[(w2 * 2 + 1) * CK + tid];
        }
        double sd = (double)sv, qd = (double)qv;
        // aux_mode 2 accumulated sum(m*dz*x); the consumer wants sum(m*dz*xhat), xhat = (x - mean) * invstd
        if constexpr (AUXM == 2) qd = (double)coef[3 * Cout + n] * (qd - (double)coef[2 * Cout + n] * sd);
        atomicAdd(&st[n], sd);
        atomicAdd(&st[Cout + n], qd);
      }
    }
  }
};

// One kernel for the three wave shapes: a level's problems (48 ... 384 channels) share one grouped launch.
template <int AUXM>
struct ConvTileQK {
  typedef TileQArgs Args;
  static constexpr int NT = 256;
  static constexpr int WPE = 2;                  // two workgroups per CU: <= 256 registers
#ifdef SSA_TILE_TIMING
  static constexpr size_t LDS = ConvTileQBody<4, AUXM>::LDS + 1536;
#else
  static constexpr size_t LDS = ConvTileQBody<4, AUXM>::LDS;
#endif
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int /*gx*/) {
    // (aux_mode 2 carries 56 more registers -- the x tile and two more sets of sums -- and does not fit 256 at
    // pb = 4: its problems run at pb <= 2, see choose_pb)
    if constexpr (AUXM != 2) {
      if (a.pb == 4) { ConvTileQBody<4, AUXM>::run(a, bx); return; }
    }
    if (a.pb == 2) ConvTileQBody<2, AUXM>::run(a, bx);
    else ConvTileQBody<1, AUXM>::run(a, bx);
  }
};

// The three-workgroups-per-CU form (ssa_conv_tile_q_config(1)): the same bodies with the filter in three stages of
// 5 + 5 + 4 k-steps through two 15 KB buffers -- 51 KB of LDS at pb = 2 --, pb <= 2, <= 168 registers.  The 48-channel
// slice is no longer resident (42 KB > the ring).  NOT yet run on the device: built and checked on the CPU emulation
// at the end of round 4 (profiles/r04_notes.md: what the two-per-CU form's phase stamps ask for).
template <int AUXM>
struct ConvTileQ3K {
  typedef TileQArgs Args;
  static constexpr int NT = 256;
  // three workgroups per CU: <= 168 registers (aux_mode 2 -- the x tile and two more sets of sums -- needs 183: it is
  // compiled for two waves per SIMD and lives with the occupancy its registers allow)
  static constexpr int WPE = AUXM == 2 ? 2 : 3;
  static constexpr size_t LDS = ConvTileQBody<2, AUXM, 3>::LDS;
  static __device__ __forceinline__ void run(const Args& a, const int bx, const int /*by*/, const int /*gx*/) {
    if (a.pb == 2) ConvTileQBody<2, AUXM, 3>::run(a, bx);
    else ConvTileQBody<1, AUXM, 3>::run(a, bx);
  }
};

static thread_local int g_q_budget = 0;        // MFMA budget (units of 42 per wave) per workgroup, 0 = per problem
static int g_q_three = 0;                      // ssa_conv_tile_q_config: 1 = the three-workgroups-per-CU form

int choose_pb(const ssa_conv_desc& d, int aux_mode) {
  const char* env = getenv("SSA_TILE_Q_PB");          // read per launch: experiments and tests switch it inside a process
  const int forced = env ? atoi(env) : 0;
  const int nchunk = d.Cin / 48;
  // measured (profiles/r04_notes.md, call R): 16 x 8-pixel tiles (pb = 2) beat 16 x 16 (pb = 4) for 48 / 96 channels
  // too -- twice the workgroups for the small problems -- and 384 channels want the shortest chain (pb = 1); pb = 4 is
  // reachable through SSA_TILE_Q_PB
  int pb = nchunk <= 4 ? 2 : 1;
  if (forced == 1 || forced == 2 || forced == 4) pb = forced;
  if ((aux_mode == 2 || g_q_three) && pb > 2) pb = 2;
  while (pb > 1 && 2 * pb >= d.H) pb >>= 1;     // half the rows still cover the image
  return pb;
}

void plan_q(const ssa_conv_desc& d, int budget, int aux_mode, TileQArgs* a) {
  const int nchunk = d.Cin / 48;
  a->pb = choose_pb(d, aux_mode);
  a->nt_total = d.Cout / 48;
  a->tiles_x = (d.W + 15) / 16;
  a->tiles_y = (d.H + 4 * a->pb - 1) / (4 * a->pb);
  a->total_tiles = d.B * a->tiles_x * a->tiles_y;
  if (budget <= 0) {
    // a launch of its own: ~2 (3) workgroups per CU from this problem alone
    const long total = (long)a->total_tiles * a->nt_total * nchunk * a->pb;
    const int slots = g_q_three ? 768 : 512;
    budget = (int)((total + slots - 1) / slots);
  }
  int tpw = budget / (nchunk * a->pb);
  if (tpw < 1) tpw = 1;
  if (tpw > 64) tpw = 64;
  const int nstrips = (a->total_tiles + tpw - 1) / tpw;
  a->tiles_per_wg = (a->total_tiles + nstrips - 1) / nstrips;
  a->nwg = ((a->total_tiles + a->tiles_per_wg - 1) / a->tiles_per_wg) * a->nt_total;
}

template <int AUXM>
int launch_q(const ssa_conv_desc& d, const TileQArgs& a0, hipStream_t s) {
  TileQArgs a = a0;
  plan_q(d, g_q_budget, AUXM, &a);
  if (g_q_three) return ssa::submit<ConvTileQ3K<AUXM>>(a, a.nwg, 1, ConvTileQ3K<AUXM>::LDS, s);
  return ssa::submit<ConvTileQK<AUXM>>(a, a.nwg, 1, ConvTileQK<AUXM>::LDS, s);
}

}  // namespace

extern "C" {

int ssa_conv2d_tile_q_supported(const ssa_conv_desc* d) {
  if (!ssa_conv2d_tile_supported(d)) return 0;
  if (d->W < 16) return 0;                    // 16-pixel-wide tiles; narrower images stay on conv_tile.hip
  if (d->Cout % 48) return 0;
  return d->Cin == 48 || d->Cin == 96 || d->Cin == 192 || d->Cin == 384;
}

int ssa_conv_tile_q_config(int three_per_cu) {
  g_q_three = three_per_cu ? 1 : 0;
  return SSA_OK;
}

int ssa_conv_tile_q_strip(int budget) {
  g_q_budget = budget < 0 ? 0 : budget;
  return SSA_OK;
}

int ssa_conv_tile_q_wgs(const ssa_conv_desc* d, int budget, int aux_mode) {
  if (!d || !ssa_conv2d_tile_q_supported(d)) return 0;
  TileQArgs a;
  plan_q(*d, budget, aux_mode, &a);
  return a.nwg;
}

int ssa_conv2d_tile_q(const ssa_conv_desc* dp, const void* x, const void* w_frag, void* y, double* stats,
                      const void* aux, int ldaux, const float* coef, int aux_mode, void* stream) {
  if (!dp || !x || !w_frag || !y) return SSA_EINVAL;
  if (!ssa_conv2d_tile_q_supported(dp)) return SSA_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w_frag)) & 15u)
    return SSA_EINVAL;
  if (aux_mode < 0 || aux_mode > 2) return SSA_EINVAL;
  if (aux_mode && (!aux || ldaux % 8 || (reinterpret_cast<uintptr_t>(aux) & 15u))) return SSA_EINVAL;
  if (aux_mode == 2 && (!coef || !stats)) return SSA_EINVAL;
  if (aux_mode == 1 && stats) return SSA_EINVAL;
  if (dp->ldy % 8) return SSA_EINVAL;
  const ssa_conv_desc& d = *dp;
  TileQArgs a;
  a.x = (const bf16_t*)x; a.wfrag = (const uint4*)w_frag;
  a.y = (bf16_t*)y; a.stats = stats; a.aux = (const bf16_t*)aux; a.coef = coef;
  a.ldx = d.ldx; a.Cin = d.Cin; a.ldy = d.ldy; a.H = d.H; a.W = d.W; a.Cout = d.Cout;
  a.ldaux = ldaux;
  a.nt_total = a.tiles_x = a.tiles_y = a.total_tiles = a.tiles_per_wg = a.nwg = a.pb = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (aux_mode) {
    case 0: return launch_q<0>(d, a, s);
    case 1: return launch_q<1>(d, a, s);
    default: return launch_q<2>(d, a, s);
  }
}

}  // extern "C"
