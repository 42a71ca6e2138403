// One-shot all-reduce (SUM, fp64) over peer-mapped device memory for the SyncBN exchange of the data-parallel step
// (apex.parallel.SyncBatchNorm of the reference: config.py:216-222, network/__init__.py:37-39).
//
// The exchange is a few thousand fp64 partial sums per BatchNorm level, ~240 times per training step, each on the
// critical path (the normalisation that follows needs the global statistics): a ring / tree collective pays several
// launch-and-handshake latencies for 20-200 KB.  On one node every GPU can write every other GPU's memory over xGMI, so
// here every rank WRITES its contribution into its slot of every peer's exchange buffer, publishes a sequence number
// behind it, waits until all peers' sequence numbers have arrived in its own buffer and sums the slots locally:
//   * one kernel, ONE workgroup (no grid-wide dependency: it can always become resident), no communicator, no proxy
//     thread -- a plain kernel node in the captured hipGraph of the step;
//   * exchange buffer per rank (uncached / fine-grained device memory; mapped by every peer from a POSIX file descriptor
//     of the allocation -- ssa_p2p_vmm_* below, HIP's virtual-memory API, the route that needs no ptrace rights -- or
//     opened through hipIpc where that is permitted):
//       [2 parities][world slots][slot_doubles]  +  flags [2 parities][world] (one 64-byte line each);
//   * the sequence number lives in device memory and is advanced by the kernel itself, so a replayed graph counts on;
//     collective k uses parity k & 1: a rank can only start collective k + 2 after every peer has published k + 1, i.e.
//     has finished reading the slots of collective k -- the two parities never overlap;
//   * ordering: data stores -> __threadfence_system() (every storing thread) -> workgroup barrier -> release store of
//     the flag at system scope; the reader polls the flags with system-scope acquire loads and reads the slots with
//     system-scope atomic loads (the buffer is fine-grained: nothing of it may be served from a stale cache line).
// A rank that has polled ~2 s gives up, counts itself (ssa_p2p_timeouts) and sums what is there: a wrong statistic the
// tests detect, never a hung GPU.  The host side (semseg_amd/p2p.py) falls back to RCCL when the buffers cannot be
// mapped or a message exceeds a slot.
#include "common.h"
#include <unistd.h>
#include <cstring>
#include <cstdint>
#include "../../include/semseg_hip.h"

namespace {

__device__ unsigned g_p2p_timeouts;

struct P2pArgs {
  double* data; long n;
  unsigned char* const* peers;        // device array: base address of every rank's exchange buffer (as mapped HERE)
  unsigned long long* seq;            // this rank's collective counter (device memory)
  long slot_doubles;
  int rank, world;
};

__device__ __forceinline__ double* slot_of(unsigned char* base, int parity, int world, long slot_doubles, int r) {
  return reinterpret_cast<double*>(base) + ((long)parity * world + r) * slot_doubles;
}
__device__ __forceinline__ unsigned long long* flag_of(unsigned char* base, int parity, int world, long slot_doubles, int r) {
  unsigned char* flags = base + 2L * world * slot_doubles * sizeof(double);
  return reinterpret_cast<unsigned long long*>(flags + ((long)parity * world + r) * 64);
}

__global__ __launch_bounds__(1024) void p2p_allreduce_f64_kernel(const P2pArgs a) {
#ifndef SSA_EMU
  __shared__ unsigned long long s_seq;
  const int t = threadIdx.x, nt = blockDim.x;
  if (t == 0) {
    s_seq = *a.seq + 1;
    *a.seq = s_seq;
  }
  __syncthreads();
  const unsigned long long seq = s_seq;
  const int parity = (int)(seq & 1);
  // ---- my contribution into my slot of every rank's buffer (my own included)
  for (int p = 0; p < a.world; ++p) {
    double* dst = slot_of(a.peers[p], parity, a.world, a.slot_doubles, a.rank);
    for (long i = t; i < a.n; i += nt) __hip_atomic_store(dst + i, a.data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if (t < a.world)
    __hip_atomic_store(flag_of(a.peers[t], parity, a.world, a.slot_doubles, a.rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // ---- wait for every rank's sequence number in MY buffer
  if (t < a.world) {
    const unsigned long long* f = flag_of(a.peers[a.rank], parity, a.world, a.slot_doubles, t);
    unsigned spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) { atomicAdd(&g_p2p_timeouts, 1u); break; }
    }
  }
  __syncthreads();
  // ---- sum the slots in rank order (every rank forms the same sum bit for bit)
  unsigned char* mine = a.peers[a.rank];
  for (long i = t; i < a.n; i += nt) {
    double s = 0.0;
    for (int r = 0; r < a.world; ++r)
      s += __hip_atomic_load(slot_of(mine, parity, a.world, a.slot_doubles, r) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a.data[i] = s;
  }
#endif
}

}  // namespace

extern "C" {

int ssa_p2p_buffer_bytes(int world, long slot_doubles, size_t* bytes) {
  if (world < 1 || slot_doubles < 1 || !bytes) return SSA_EINVAL;
  *bytes = 2 * (size_t)world * (size_t)slot_doubles * sizeof(double) + 2 * (size_t)world * 64;
  return SSA_OK;
}

int ssa_p2p_allreduce_f64(double* data, long n, void* const* peers_dev, int rank, int world,
                          unsigned long long* seq_dev, long slot_doubles, void* stream) {
  if (!data || !peers_dev || !seq_dev || n < 1 || world < 1 || world > 64 || rank < 0 || rank >= world) return SSA_EINVAL;
  if (n > slot_doubles) return SSA_EUNSUPPORTED;
#ifdef SSA_EMU
  return SSA_EUNSUPPORTED;
#else
  P2pArgs a{data, n, (unsigned char* const*)peers_dev, seq_dev, slot_doubles, rank, world};
  hipLaunchKernelGGL(p2p_allreduce_f64_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  SSA_LAUNCH_CHECK();
  return SSA_OK;
#endif
}

// ---- exchange buffers through the virtual-memory API: hipMemCreate(uncached, exportable as a POSIX fd) on the owner,
// hipMemImportFromShareableHandle + reserve + map + set-access on every peer.  The fd travels over a unix socket
// (SCM_RIGHTS, semseg_amd/p2p.py): unlike hipIpcOpenMemHandle in dmabuf mode (pidfd_getfd) this needs neither
// CAP_SYS_PTRACE nor a relaxed seccomp profile.
#ifndef SSA_EMU
namespace {
int vmm_prop(hipMemAllocationProp* prop, int uncached) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  memset(prop, 0, sizeof(*prop));
  prop->type = uncached ? hipMemAllocationTypeUncached : hipMemAllocationTypePinned;
  prop->requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  prop->location.type = hipMemLocationTypeDevice;
  prop->location.id = dev;
  return 0;
}
int vmm_map(hipMemGenericAllocationHandle_t h, size_t bytes, size_t gran, void** ptr) {
  void* p = nullptr;
  hipError_t e = hipMemAddressReserve(&p, bytes, gran, nullptr, 0);
  if (e != hipSuccess) return (int)e;
  e = hipMemMap(p, bytes, 0, h, 0);
  if (e != hipSuccess) { hipMemAddressFree(p, bytes); return (int)e; }
  int dev = 0;
  hipGetDevice(&dev);
  hipMemAccessDesc acc;
  memset(&acc, 0, sizeof(acc));
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(p, bytes, &acc, 1);
  if (e != hipSuccess) { hipMemUnmap(p, bytes); hipMemAddressFree(p, bytes); return (int)e; }
  *ptr = p;
  return 0;
}
}  // namespace
#endif

int ssa_p2p_vmm_alloc(size_t bytes, void** ptr, int* fd, size_t* mapped_bytes) {
  if (!bytes || !ptr || !fd || !mapped_bytes) return SSA_EINVAL;
#ifdef SSA_EMU
  return SSA_EUNSUPPORTED;
#else
  for (int uncached = 1; uncached >= 0; --uncached) {      // uncached = fine-grained; plain pinned if the type is refused
    hipMemAllocationProp prop;
    int rc = vmm_prop(&prop, uncached);
    if (rc) return rc;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) continue;
    const size_t nb = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, nb, &prop, 0) != hipSuccess) continue;
    int out_fd = -1;
    if (hipMemExportToShareableHandle(&out_fd, h, hipMemHandleTypePosixFileDescriptor, 0) != hipSuccess || out_fd < 0) {
      hipMemRelease(h);
      continue;
    }
    void* p = nullptr;
    rc = vmm_map(h, nb, gran, &p);
    hipMemRelease(h);                    // the mapping (and the exported descriptor) keep the memory alive
    if (rc) { close(out_fd); continue; }
    if (hipMemset(p, 0, nb) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      hipMemUnmap(p, nb); hipMemAddressFree(p, nb); close(out_fd);
      continue;
    }
    *ptr = p; *fd = out_fd; *mapped_bytes = nb;
    return SSA_OK;
  }
  (void)hipGetLastError();
  return SSA_EUNSUPPORTED;
#endif
}

int ssa_p2p_vmm_import(int fd, size_t mapped_bytes, void** ptr) {
  if (fd < 0 || !mapped_bytes || !ptr) return SSA_EINVAL;
#ifdef SSA_EMU
  return SSA_EUNSUPPORTED;
#else
  int version = 0;
  hipRuntimeGetVersion(&version);
  hipMemGenericAllocationHandle_t h;
  // HIP < 7.1 reads the descriptor THROUGH the pointer, later runtimes take it by value (as the CUDA driver API does)
  void* os_handle = version >= 70100000 ? reinterpret_cast<void*>(static_cast<uintptr_t>(fd)) : static_cast<void*>(&fd);
  hipError_t e = hipMemImportFromShareableHandle(&h, os_handle, hipMemHandleTypePosixFileDescriptor);
  if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
  hipMemAllocationProp prop;
  vmm_prop(&prop, 1);
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) gran = 2u << 20;
  int rc = vmm_map(h, mapped_bytes, gran, ptr);
  hipMemRelease(h);
  if (rc) (void)hipGetLastError();
  return rc;
#endif
}

int ssa_p2p_vmm_unmap(void* ptr, size_t mapped_bytes) {
  if (!ptr || !mapped_bytes) return SSA_EINVAL;
#ifdef SSA_EMU
  return SSA_EUNSUPPORTED;
#else
  hipError_t e = hipMemUnmap(ptr, mapped_bytes);
  hipError_t f = hipMemAddressFree(ptr, mapped_bytes);
  return e != hipSuccess ? (int)e : (f != hipSuccess ? (int)f : SSA_OK);
#endif
}

int ssa_p2p_timeouts(unsigned* out) {
  if (!out) return SSA_EINVAL;
#ifdef SSA_EMU
  *out = 0;
  return SSA_OK;
#else
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_p2p_timeouts), sizeof(unsigned));
  return e == hipSuccess ? SSA_OK : (int)e;
#endif
}

}  // extern "C"
