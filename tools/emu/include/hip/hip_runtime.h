// CPU emulation shim for the gfx950 kernels of semantic-segmentation_amd/csrc (TEST INFRASTRUCTURE, never shipped,
// never loaded by the product: semseg_amd/_lib.py only ever opens lib/libsemseg_hip.so).
//
// The kernel sources are compiled UNCHANGED by the host clang with -DSSA_EMU and this directory first on the
// include path, so `#include <hip/hip_runtime.h>` lands here.  A workgroup runs as NT cooperative fibers
// (ucontext) of one OS thread; __syncthreads() and the wave collectives (MFMA, ds_read_tr16, shuffles) are
// rendezvous points of the fiber scheduler (tools/emu/emu_runtime.cpp).  What this buys: the index arithmetic,
// LDS layouts, barrier structure and host-side planning of a new kernel are checked against the oracle HERE,
// without a GPU -- GPU minutes then go to measurement.  What it cannot see: timing, bank conflicts, memory-model
// races between workgroups (workgroups run one after the other).
//
// Fidelity notes
//  * MFMA 32x32x16 bf16: A lane l holds row l&31, k = 8*(l>>5)+j; B likewise by column; C/D row =
//    (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31 (pinned on the device by ssa_probe_mfma32).
//  * ds_read_b64_tr_b16: within a 16-lane group, element j of lane i = element i&3 of the 8-byte chunk
//    addressed by lane 4*j + (i>>2) (pinned on the device by ssa_probe_tr16: gpurun_out/probe_tr16.txt).
//  * global_load_lds (LDS DMA): the 16 bytes of lane l land at dst + 16*l.  The copy is DEFERRED to the issuing
//    fiber's next __syncthreads() -- as on the device, nothing may read the data before vmcnt + barrier, and a
//    DMA into a buffer other waves are still reading corrupts their reads here too (fibers of later waves run
//    after the flush).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };

namespace emu {
struct Ctx {
  dim3 tid, bid, bdim, gdim;
  int lane, wave;
};
extern thread_local Ctx* cur;
void sync_block();
void sync_wave();
unsigned char* wave_buf(int lane);        // 64 bytes of exchange space per lane of the calling fiber's wave
unsigned char* dyn_lds();                 // the workgroup's dynamic LDS (160 KiB, 16-byte aligned)
void defer_dma(void* dst, const void* src, int bytes);
void launch(dim3 grid, dim3 block, size_t lds, void (*tramp)(void*), void* closure);
template <class F> void tramp_fn(void* p) { (*static_cast<F*>(p))(); }
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                 \
  do {                                                                           \
    auto emu_fn__ = [&]() { kernel(__VA_ARGS__); };                             \
    emu::launch((grid), (block), (size_t)(lds), &emu::tramp_fn<decltype(emu_fn__)>, &emu_fn__); \
  } while (0)

static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

// ---- device intrinsics
static inline void __syncthreads() { emu::sync_block(); }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline float rsqrtf(float x) { return 1.f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.f / a; }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline long min(long a, int b) { return a < b ? a : (long)b; }
static inline long min(int a, long b) { return a < b ? (long)a : b; }
static inline long max(long a, int b) { return a > b ? a : (long)b; }
static inline long max(int a, long b) { return a > b ? (long)a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

// fibers are cooperative: a read-modify-write between two yield points is atomic by construction
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T unsafeAtomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }

template <class T> static inline T __shfl_xor(T v, int mask, int /*width*/ = 64) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  const int lane = emu::cur->lane;
  memcpy(emu::wave_buf(lane), &v, sizeof(T));
  emu::sync_wave();
  T r;
  memcpy(&r, emu::wave_buf(lane ^ mask), sizeof(T));
  emu::sync_wave();
  return r;
}
template <class T> static inline T __shfl_down(T v, int d, int /*width*/ = 64) {
  const int lane = emu::cur->lane;
  memcpy(emu::wave_buf(lane), &v, sizeof(T));
  emu::sync_wave();
  T r;
  memcpy(&r, emu::wave_buf(lane + d < 64 ? lane + d : lane), sizeof(T));
  emu::sync_wave();
  return r;
}
template <class T> static inline T __shfl(T v, int src, int /*width*/ = 64) {
  const int lane = emu::cur->lane;
  memcpy(emu::wave_buf(lane), &v, sizeof(T));
  emu::sync_wave();
  T r;
  memcpy(&r, emu::wave_buf(src & 63), sizeof(T));
  emu::sync_wave();
  return r;
}

#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)

namespace emu {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

static inline f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  const int lane = cur->lane;
  unsigned char* mine = wave_buf(lane);
  memcpy(mine, &a, 16);
  memcpy(mine + 16, &b, 16);
  sync_wave();
  const int col = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      unsigned short ua, ub;
      memcpy(&ua, wave_buf(row + 32 * (k >> 3)) + 2 * (k & 7), 2);
      memcpy(&ub, wave_buf(col + 32 * (k >> 3)) + 16 + 2 * (k & 7), 2);
      const unsigned ia = (unsigned)ua << 16, ib = (unsigned)ub << 16;
      float fa, fb;
      memcpy(&fa, &ia, 4);
      memcpy(&fb, &ib, 4);
      acc += fa * fb;
    }
    c[r] = acc;
  }
  sync_wave();
  return c;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static inline f32x16 mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
  const int lane = cur->lane;
  unsigned char* mine = wave_buf(lane);
  memcpy(mine, &a, 16);
  memcpy(mine + 16, &b, 16);
  sync_wave();
  const int col = lane & 31;
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float acc = c[r];
    for (int k = 0; k < 16; ++k) {
      _Float16 ha, hb;
      memcpy(&ha, wave_buf(row + 32 * (k >> 3)) + 2 * (k & 7), 2);
      memcpy(&hb, wave_buf(col + 32 * (k >> 3)) + 16 + 2 * (k & 7), 2);
      acc += (float)ha * (float)hb;
    }
    c[r] = acc;
  }
  sync_wave();
  return c;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// MFMA 16x16x32: A lane l holds row l&15, k = 8*(l>>4)+j; B likewise by column; D column l&15, rows 4*(l>>4)+r
// (pinned on the device by ssa_probe_mfma16)
template <class LD>
static inline f32x4 mfma_16x16x32_impl(const void* a, const void* b, f32x4 c, LD ld) {
  const int lane = cur->lane;
  unsigned char* mine = wave_buf(lane);
  memcpy(mine, a, 16);
  memcpy(mine + 16, b, 16);
  sync_wave();
  const int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 32; ++k)
      acc += ld(wave_buf(row + 16 * (k >> 3)) + 2 * (k & 7)) * ld(wave_buf(col + 16 * (k >> 3)) + 16 + 2 * (k & 7));
    c[r] = acc;
  }
  sync_wave();
  return c;
}
static inline f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return mfma_16x16x32_impl(&a, &b, c, [](const unsigned char* p) {
    unsigned short u; memcpy(&u, p, 2); const unsigned i = (unsigned)u << 16; float f; memcpy(&f, &i, 4); return f; });
}
static inline f32x4 mfma_16x16x32_f16(f16x8 a, f16x8 b, f32x4 c) {
  return mfma_16x16x32_impl(&a, &b, c, [](const unsigned char* p) { _Float16 h; memcpy(&h, p, 2); return (float)h; });
}

static inline s16x4 ds_read_tr16_b64(const void* p) {
  const int lane = cur->lane;
  memcpy(wave_buf(lane), p, 8);
  sync_wave();
  const int grp = lane & ~15, i = lane & 15;
  s16x4 r;
  for (int j = 0; j < 4; ++j) {
    short v;
    memcpy(&v, wave_buf(grp + 4 * j + (i >> 2)) + 2 * (i & 3), 2);
    r[j] = v;
  }
  sync_wave();
  return r;
}

// dst: the wave-uniform LDS base the device puts in M0; lane l's 16 bytes land at dst + 16*l
static inline void global_load_lds16(const void* gsrc, void* lds_dst) {
  defer_dma(static_cast<unsigned char*>(lds_dst) + 16 * cur->lane, gsrc, 16);
}
}  // namespace emu
