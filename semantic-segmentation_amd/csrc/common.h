// Shared device helpers for the gfx950 (CDNA4) segmentation kernels.
// Activations are NHWC, 16-bit elements stored as raw words; all arithmetic is
// fp32 (fp64 for the BN / RMI statistics).
//
// The 16-bit STORAGE FORMAT is an instantiation parameter of the whole library: every conversion and the MFMA
// go through the handful of helpers below, and the build compiles the same sources twice --
//   libsemseg_hip.so       bf16 storage (default; v_mfma_f32_32x32x16_bf16)
//   libsemseg_hip_f16.so   fp16 storage (-DSSA_ELEM_F16; v_mfma_f32_32x32x16_f16, same rate) -- the reference's own
//                          reduced precision (apex O1 / --fp16, train.py:381), 8x finer than bf16
// (the names bf16_t / bf2f / f2bf keep their round-1 spelling: "the 16-bit activation element").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                   // raw bits of a 16-bit activation element
#ifdef SSA_ELEM_F16
typedef _Float16 bf16x8_t __attribute__((ext_vector_type(8)));   // MFMA A/B fragment
#else
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));     // MFMA A/B fragment
#endif
typedef float f32x16_t __attribute__((ext_vector_type(16)));     // 32x32 MFMA C/D
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// The three gfx950-only operations the kernels use go through these wrappers, and dynamic LDS through
// SSA_DYN_LDS, so that the SAME kernel sources also compile for the CPU emulation harness of the test-suite
// (tools/emu, -DSSA_EMU: index arithmetic and barrier structure checked without a GPU).  The product build
// never defines SSA_EMU.
#ifdef SSA_EMU
#define SSA_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(emu::dyn_lds())
#ifdef SSA_ELEM_F16
__device__ __forceinline__ f32x16_t ssa_mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) { return emu::mfma_32x32x16_f16(a, b, c); }
#else
__device__ __forceinline__ f32x16_t ssa_mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) { return emu::mfma_32x32x16_bf16(a, b, c); }
#endif
#ifdef SSA_ELEM_F16
__device__ __forceinline__ f32x4_t ssa_mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return emu::mfma_16x16x32_f16(a, b, c); }
#else
__device__ __forceinline__ f32x4_t ssa_mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return emu::mfma_16x16x32_bf16(a, b, c); }
#endif
__device__ __forceinline__ s16x4_t ssa_tr16_b64(const void* lds_ptr) { return emu::ds_read_tr16_b64(lds_ptr); }
__device__ __forceinline__ void ssa_glds16(const void* gsrc, void* lds_dst) { emu::global_load_lds16(gsrc, lds_dst); }
__device__ __forceinline__ void ssa_wave_sync() { emu::sync_wave(); }
template <int VM, int LGKM = 0> __device__ __forceinline__ void ssa_wait_vm_barrier() { __syncthreads(); }
__device__ __forceinline__ void ssa_glds16_untracked(const void* gsrc, void* lds_dst) { emu::global_load_lds16(gsrc, lds_dst); }
__device__ __forceinline__ void ssa_glds16_untracked_sv(const void* sbase, unsigned voff, void* lds_dst) {
  emu::global_load_lds16(reinterpret_cast<const unsigned char*>(sbase) + voff, lds_dst);
}
__device__ __forceinline__ unsigned ssa_lds_addr(const void* lds_ptr) {
  return (unsigned)(reinterpret_cast<const unsigned char*>(lds_ptr) - emu::dyn_lds());
}
__device__ __forceinline__ void ssa_glds16_untracked_m0(const void* sbase, unsigned voff, unsigned lds_addr) {
  emu::global_load_lds16(reinterpret_cast<const unsigned char*>(sbase) + voff, emu::dyn_lds() + lds_addr);
}
#else
#define SSA_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
// D = A(32x16) * B(16x32) + C on one wave; lane l holds row/column l & 31, k = 8 * (l >> 5) + j
__device__ __forceinline__ f32x16_t ssa_mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
#ifdef SSA_ELEM_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}
// D = A(16x32) * B(32x16) + C on one wave; lane l holds row/column l & 15, k-group l >> 4 (8 consecutive k); D: column
// l & 15, rows 4 * (l >> 4) + j  (lane map pinned by ssa_probe_mfma16)
__device__ __forceinline__ f32x4_t ssa_mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
#ifdef SSA_ELEM_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// ds_read_b64_tr_b16: 4x4 transposing LDS read (lane map: tests/test_kernels_gpu.py::test_probe_tr16)
__device__ __forceinline__ s16x4_t ssa_tr16_b64(const void* lds_ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(lds_ptr));
}
// global -> LDS DMA of 16 bytes per lane: lane l's piece lands at lds_dst (wave-uniform) + 16 * l
__device__ __forceinline__ void ssa_glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// The same DMA issued behind the compiler's back (inline assembly): the compiler's wait-count pass then knows nothing
// of it.  For kernels that order their DMAs themselves with ssa_wait_vm_barrier<>: a tracked LDS DMA makes the pass
// put  s_waitcnt vmcnt(0)  in front of every later ds_read_b64_tr_b16 (the transposing-read builtin counts as a
// possible LDS writer) and in front of every use of a register-returning load -- i.e. it drains the DMA pipeline the
// kernel is trying to keep full.  __syncthreads() does NOT wait for an untracked DMA.  Compiler-placed vmcnt waits for
// its own loads stay correct: unknown operations in flight only make them wait longer.  M0 is written without the
// compiler knowing (it rejects m0 as a clobber): a kernel uses EITHER this form OR ssa_glds16, never both -- the
// compiler's own M0 uses on gfx950 are its LDS-DMA builtins only.
__device__ __forceinline__ void ssa_glds16_untracked(const void* gsrc, void* lds_dst) {
  const unsigned base = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(base) : "memory");
}
// The same with the address as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): a kernel that
// issues many DMAs per stage keeps one VGPR per fragment and one scalar add per stage instead of a 64-bit scalar
// address computation per DMA (which the compiler hoists and then spills).
__device__ __forceinline__ void ssa_glds16_untracked_sv(const void* sbase, unsigned voff, void* lds_dst) {
  const unsigned base = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(base) : "memory");
}
// The same with the destination as a 32-bit LDS address (ssa_lds_addr(pointer), taken ONCE per kernel): the cast of a
// generic pointer to the LDS address space is a null test + select every time it is written out, ~6 scalar instructions
// per DMA, and a lone wave issues an instruction every 5-8 clocks (profiles/r04_notes.md, call T)
__device__ __forceinline__ unsigned ssa_lds_addr(const void* lds_ptr) {
  return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)lds_ptr);
}
__device__ __forceinline__ void ssa_glds16_untracked_m0(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// LDS written by some lanes of a wave, read by others of the SAME wave: the hardware runs a wave's LDS operations
// in order, so only the compiler must not move the reads above the writes (the emulation runs lanes as
// independent fibers and needs a real rendezvous here)
__device__ __forceinline__ void ssa_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// Workgroup barrier that lets the newest VM vector-memory operations of this wave stay in flight:
//   s_waitcnt vmcnt(VM) lgkmcnt(LGKM) ; s_barrier
// (__syncthreads() is a fence: it waits vmcnt(0), i.e. for every LDS DMA issued so far -- a pipeline that keeps
// more than one stage of global_load_lds in flight needs this form).  Memory operations retire in order, so after
// the barrier everything every wave issued BEFORE its newest VM operations is visible in LDS.  The caller must
// know VM exactly: every wave has to issue the same operations on every path to this point.  The emulation's
// DMA lands at the next barrier whatever VM is (tools/emu), which is the latest the hardware may deliver it.
// LGKM > 0 likewise leaves the newest LGKM LDS reads in flight (reads of data no DMA overwrites before the next
// barrier: a fragment read-ahead that runs across the barrier).
template <int VM, int LGKM = 0> __device__ __forceinline__ void ssa_wait_vm_barrier() {
  static_assert(VM >= 0 && VM < 64 && LGKM >= 0 && LGKM < 16, "vmcnt is a 6-bit counter, lgkmcnt a 4-bit one");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(%1)\n\ts_barrier" ::"n"(VM), "n"(LGKM) : "memory");
}
#endif

#define SSA_OK 0
#define SSA_EINVAL (-1)
#define SSA_EUNSUPPORTED (-2)

namespace ssa { void count_launches(int n); }   // group.hip: library-wide launch counter (ssa_launch_count)

#define SSA_LAUNCH_CHECK()                          \
  do {                                              \
    ssa::count_launches(1);                         \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

typedef float f32x2_t __attribute__((ext_vector_type(2)));
#ifdef SSA_ELEM_F16
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even (v_cvt_f16_f32 / v_cvt_pk_f16_f32 under the default rounding mode); values beyond 65504
// become inf -- the activations this library stores are BatchNorm outputs and conv outputs of normalised inputs
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t f2bf_pair(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, w[i]);
    f[2 * i] = (float)h[0];
    f[2 * i + 1] = (float)h[1];
  }
}
#else
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN kept quiet: v_cvt_pk_bf16_f32 on gfx950 (one instruction; the shift-and-add
// sequence it replaces was five VALU operations per element in every epilogue and staging transform)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t f2bf_pair(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
#endif
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = f2bf_pair(f[0], f[1]);
  v.y = f2bf_pair(f[2], f[3]);
  v.z = f2bf_pair(f[4], f[5]);
  v.w = f2bf_pair(f[6], f[7]);
  return v;
}

// generic scalar load/store by element type (bf16_t or float)
template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_from_f32<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 1 / sqrt(v): v_rsq_f32 (1 ulp) -- callers that want fp32-exact results follow it with one Newton step
__device__ __forceinline__ float ssa_rsqrt(float v) {
#ifdef SSA_EMU
  return 1.0f / sqrtf(v);
#else
  return __builtin_amdgcn_rsqf(v);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
