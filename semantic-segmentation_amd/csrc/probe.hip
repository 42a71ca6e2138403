// Hardware-layout probes.  The MFMA fragment maps and the ds_read_b64_tr_b16
// lane map are checked on the device by tests/test_probe_gpu.py rather than
// trusted from documentation.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

// One wave: C[32x32] = A[32x16] * B[16x32], operands given row-major in global
// memory as bf16 (A: [32][16], B stored as B^T: [32 n][16 k]).
__global__ void probe_mfma32_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ bt,
                                    float* __restrict__ c) {
  const int lane = threadIdx.x;
  const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(a + (lane & 31) * 16 + (lane >> 5) * 8);
  const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(bt + (lane & 31) * 16 + (lane >> 5) * 8);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = ssa_mfma32(af, bf, acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    c[row * 32 + (lane & 31)] = acc[r];
  }
}

// One wave: C[16x16] = A[16x32] * B[32x16] through v_mfma_f32_16x16x32 (A: [16][32], B stored as B^T: [16 n][32 k]).
__global__ void probe_mfma16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ bt,
                                    float* __restrict__ c) {
  const int lane = threadIdx.x;
  const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(a + (lane & 15) * 32 + (lane >> 4) * 8);
  const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(bt + (lane & 15) * 32 + (lane >> 4) * 8);
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  acc = ssa_mfma16(af, bf, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

// v_permlane16_swap of (a = lane, b = lane + 100): out[lane] = (a, b) afterwards
__global__ void probe_swap16_kernel(unsigned* __restrict__ out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
#ifdef SSA_EMU
  const unsigned pa = __shfl_xor(a, 16, 64), pb = __shfl_xor(b, 16, 64);
  if (((threadIdx.x >> 4) & 1) == 0) b = pa; else a = pb;
#else
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
#endif
  out[threadIdx.x * 2] = a;
  out[threadIdx.x * 2 + 1] = b;
}

// LDS holds lds[i] = i (u16).  mode 0: lane supplies address of element 4*lane
// (contiguous 8-byte pieces).  mode 1: lane l supplies element
// 64*(l>>4) + 16*((l&15)>>2) + 4*(l&3)   (row (l&15)>>2, piece l&3 of a [4][16] block)
// -- identical to mode 0; mode 2: lane l supplies 64*(l>>4) + 16*(l&3) + 4*((l&15)>>2).
__global__ void probe_tr16_kernel(unsigned short* __restrict__ out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int e;
  if (mode == 0) e = 4 * l;
  else if (mode == 1) e = 64 * (l >> 4) + 16 * ((l & 15) >> 2) + 4 * (l & 3);
  else e = 64 * (l >> 4) + 16 * (l & 3) + 4 * ((l & 15) >> 2);
  s16x4_t v = ssa_tr16_b64(lds + e);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

}  // namespace

extern "C" {
int ssa_probe_mfma32(const void* a, const void* b, float* c, void* stream) {
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (const bf16_t*)a, (const bf16_t*)b, c);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
int ssa_probe_mfma16(const void* a, const void* b, float* c, void* stream) {
  hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (const bf16_t*)a, (const bf16_t*)b, c);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
int ssa_probe_swap16(unsigned* out, void* stream) {
  hipLaunchKernelGGL(probe_swap16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
int ssa_probe_tr16(unsigned short* out, int mode, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, mode);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
}
}
