// Micro-benchmark (gfx950): G workgroups each add a K-float block into ONE destination with global_atomic_add_f32
// (no return), against G partial stores + a summing pass -- what replacing WgradReduceK by atomics would cost.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/atomicbench tools/micro/atomicbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void add_atomic(float* dst, int K, float v) {
  for (int k = threadIdx.x; k < K; k += blockDim.x) unsafeAtomicAdd(&dst[k], v + k * 1e-9f);
}
__global__ void add_atomic4(float* dst, int K, float v) {   // each lane 4 consecutive floats (what an accumulator row gives)
  for (int k = threadIdx.x * 4; k < K; k += blockDim.x * 4)
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(&dst[k + j], v + k * 1e-9f);
}
__global__ void store_partial(float* part, int K, float v) {
  float4* p = reinterpret_cast<float4*>(part + (long)blockIdx.x * K);
  for (int k = threadIdx.x; k < K / 4; k += blockDim.x) p[k] = make_float4(v, v + k, v, v);
}
__global__ void reduce_partial(const float* part, float* dst, int K, int G) {
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (k >= K) return;
  float4 a = make_float4(0, 0, 0, 0);
  for (int g = 0; g < G; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long)g * K + k);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float4* d = reinterpret_cast<float4*>(dst + k);
  float4 o = *d; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; *d = o;
}

int main() {
  const int Ks[] = {48 * 432, 96 * 864, 192 * 1728, 384 * 3456};
  const int Gs[] = {16, 64, 256};
  float *dst, *part;
  CK(hipMalloc(&dst, 384 * 3456 * 4));
  CK(hipMalloc(&part, (size_t)256 * 384 * 3456 * 4));
  CK(hipMemset(dst, 0, 384 * 3456 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int K : Ks) for (int G : Gs) {
    float ms[4];
    for (int mode = 0; mode < 3; ++mode) {
      const int reps = 20;
      for (int r = -2; r < reps; ++r) {
        if (r == 0) CK(hipEventRecord(e0));
        if (mode == 0) add_atomic<<<G, 256>>>(dst, K, 1.f);
        else if (mode == 1) add_atomic4<<<G, 256>>>(dst, K, 1.f);
        else { store_partial<<<G, 256>>>(part, K, 1.f); reduce_partial<<<(K / 4 + 255) / 256, 256>>>(part, dst, K, G); }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[mode], e0, e1));
      ms[mode] *= 1000.f / reps;
    }
    printf("K %8d floats (%6.2f MB)  G %3d : atomic %8.1f us   atomic x4/lane %8.1f us   partial+reduce %8.1f us\n", K, K * 4e-6, G, ms[0], ms[1], ms[2]);
  }
  // many problems at once: 32 destinations of 48x432, 256 adders each spread over them
  return 0;
}
