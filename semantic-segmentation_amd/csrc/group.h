// Multi-problem ("grouped") launches.
//
// A B=1 training step of HRNet-OCR-MScale is ~5,500 kernel launches of 5-20 us each
// (profiles/r01_bench_graph_summary.txt): the 2-4 resolution branches of a
// HighResolutionModule (network/hrnetv2.py:181-254) times the two scale passes of
// MscaleOCR.two_scale_forward (network/ocrnet.py:264-327) are up to 8 INDEPENDENT problems
// of 8..512 workgroups each, none of which fills 256 CUs.  The host glue therefore walks
// those problems in lockstep and brackets the launches of one depth level with
// ssa_group_begin() / ssa_group_end(stream): inside the bracket every group-aware entry
// point appends its (arguments, grid) to a per-thread list instead of launching, and
// ssa_group_end issues ONE launch per kernel instantiation whose grid is the concatenation
// of the problems' grids.  The job table travels as a kernel argument (<= 4 KB, so a
// captured hipGraph owns it); a workgroup finds its problem by a scalar scan over at most
// 16 prefix sums and then runs the unchanged kernel body with a virtual block index.
//
// Contract: launches submitted inside one bracket are mutually independent (no job reads
// what another job of the bracket writes).  Entry points that are not group-aware launch
// immediately, which is always correct under that contract.
//
// A kernel takes part by being written as
//     struct MyKernel { struct Args {...}; static constexpr int NT = 256;
//                       static __device__ void run(const Args& a, int bx, int by, int gx); };
// and launched with ssa::submit<MyKernel>(args, grid_x, grid_y, lds_bytes, stream).
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace ssa {

constexpr int kKernargBudget = 3840;   // bytes of kernel arguments a grouped launch may use

// A kernel struct may declare  static constexpr int MAXJOBS  = the most problems one of its grouped launches carries
// (default 16; the weight-gradient tile kernel takes 32: twice the layers per launch = strips twice as long at the same
// workgroup count = half the fp32 partials its reduce has to read back).
template <class K, class = void> struct MaxJobs { static constexpr int value = 16; };
template <class K> struct MaxJobs<K, std::void_t<decltype(K::MAXJOBS)>> { static constexpr int value = K::MAXJOBS; };

template <class K>
struct GroupLimits {
  static constexpr int per_job = (int)sizeof(typename K::Args) + 8;
  static constexpr int raw = (kKernargBudget - 16) / per_job;
  static constexpr int cap = MaxJobs<K>::value;
  static constexpr int jobs = raw > cap ? cap : (raw < 1 ? 1 : raw);
};

template <class K>
struct GroupTable {
  int n, pad_;
  int end[GroupLimits<K>::jobs];        // exclusive prefix sums of the problems' workgroup counts
  int gx[GroupLimits<K>::jobs];         // grid.x of each problem (virtual blockIdx.x = v % gx, .y = v / gx)
  typename K::Args a[GroupLimits<K>::jobs];
};

// A kernel struct may declare  static constexpr int WPE  = waves per SIMD it must fit (second launch-bounds
// parameter -> register budget 512 / WPE): two 8-wave workgroups per CU need WPE = 4, i.e. <= 128 registers.
template <class K, class = void> struct WavesPerEu { static constexpr int value = 1; };
template <class K> struct WavesPerEu<K, std::void_t<decltype(K::WPE)>> { static constexpr int value = K::WPE; };

template <class K>
__global__ __launch_bounds__(K::NT, WavesPerEu<K>::value) void k_single(const typename K::Args a) {
  K::run(a, blockIdx.x, blockIdx.y, gridDim.x);
}

// A workgroup finds its problem in TWO scalar-memory round trips: the prefix sums sit at constant kernel-argument
// offsets, so the unrolled count below compiles to a few wide s_load's issued together and two scalar instructions per
// entry (the host fills the entries from the last problem's on with INT_MAX: no test against n); the second round trip
// fetches end[j - 1], gx[j] and the problem's arguments together.  The rolled loop it replaces was
// s_load end[j] -> s_waitcnt -> compare  per problem, then dependent loads of end[j - 1], gx[j] and the arguments:
// 4 + j round trips before a workgroup issued its first data load -- on the critical path of every workgroup of the
// one-chunk-per-workgroup kernels (BatchNorm, sums, resampling), where the LAST problems of the table end the launch
// (round 6: a 2-workgroup BatchNorm launch took 7 us).
template <class K>
__global__ __launch_bounds__(K::NT, WavesPerEu<K>::value) void k_grouped(const GroupTable<K> t) {
  constexpr int J = GroupLimits<K>::jobs;
  const int bx = (int)blockIdx.x;
  int j = 0;
#pragma unroll
  for (int i = 0; i + 1 < J; ++i) j += bx >= t.end[i] ? 1 : 0;
  const int prev = t.end[j > 0 ? j - 1 : 0];
  const int gx = t.gx[j];
  const int v = bx - (j > 0 ? prev : 0);
  int vx = v, vy = 0;
  if (v >= gx) { vy = v / gx; vx = v - vy * gx; }   // grid.y == 1 (or the first row): no integer division
  K::run(t.a[j], vx, vy, gx);
}

struct Bucket {
  const void* key;                                   // identity of the kernel instantiation
  int (*flush)(Bucket&, hipStream_t);
  std::vector<unsigned char> args;
  std::vector<int> gx, gy;
  std::vector<double> flops, bytes;                  // profile notes of the queued jobs (ssa_profile_note)
  size_t lds;
};

// Per-launch timing (ssa_profile_begin/_end): every launch that goes through submit<> is
// bracketed by HIP events on its stream and keyed by the kernel instantiation's name, with the
// algorithmic flops/bytes the host attached to its jobs -- what bench.py's roofline leg reads.
bool profiling();
void profile_take_note(double* flops, double* bytes);          // note attached to the next job (cleared)
void* profile_open(const char* kernel, hipStream_t s);          // records the start event
void profile_close(void* h, hipStream_t s, int jobs, double flops, double bytes);

template <class K>
const char* kernel_name() { return __PRETTY_FUNCTION__; }

struct GroupState {
  int depth = 0;
  int error = 0;
  std::vector<Bucket> buckets;
};

GroupState& group_state();      // per host thread (forward: main thread, backward: autograd thread)
void count_launches(int n);     // library-wide launch counter (ssa_launch_count)


// The dynamic-LDS limit of a kernel is a property of (function, DEVICE): the raised limit is remembered per device
// (hipFuncSetAttribute applies to the current device's code object only; one process normally drives one GPU, but a
// process that touches two must raise the limit on each).
constexpr int kMaxDevices = 16;
struct LdsLimit { size_t set_to[kMaxDevices] = {}; };

inline int raise_lds_limit(const void* fn, size_t lds, size_t threshold, LdsLimit* lim) {
  if (lds <= threshold) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  if (lds > lim->set_to[dev]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lim->set_to[dev] = lds;
  }
  return 0;
}

template <class K>
int ensure_lds(const void* fn, size_t lds, LdsLimit* lim) {
  return raise_lds_limit(fn, lds, 64 * 1024, lim);
}

template <class K>
int flush_bucket(Bucket& b, hipStream_t s) {
  typedef typename K::Args Args;
  constexpr int J = GroupLimits<K>::jobs;
  static LdsLimit lds_single, lds_grouped;
  const int n = (int)b.gx.size();
  for (int j0 = 0; j0 < n; j0 += J) {
    const int cnt = n - j0 < J ? n - j0 : J;
    if (cnt == 1) {
      Args a;
      memcpy(&a, b.args.data() + (size_t)j0 * sizeof(Args), sizeof(Args));
      if (int e = ensure_lds<K>((const void*)k_single<K>, b.lds, &lds_single)) return e;
      void* ph = profiling() ? profile_open(kernel_name<K>(), s) : nullptr;
      hipLaunchKernelGGL(k_single<K>, dim3(b.gx[j0], b.gy[j0]), dim3(K::NT), b.lds, s, a);
      if (ph) profile_close(ph, s, 1, b.flops[j0], b.bytes[j0]);
    } else {
      GroupTable<K> t;
      memset(&t, 0, sizeof(t));
      t.n = cnt;
      int total = 0;
      for (int i = 0; i < cnt; ++i) {
        total += b.gx[j0 + i] * b.gy[j0 + i];
        t.end[i] = total;
        t.gx[i] = b.gx[j0 + i];
        memcpy(&t.a[i], b.args.data() + (size_t)(j0 + i) * sizeof(Args), sizeof(Args));
      }
      // the kernel's search counts the entries <= blockIdx.x without looking at n: no entry from the last problem's on
      // may compare true (blockIdx.x < total always holds; end[j - 1] is only read for j <= cnt - 1)
      for (int i = cnt - 1; i < J; ++i) t.end[i] = 0x7fffffff;
      if (int e = ensure_lds<K>((const void*)k_grouped<K>, b.lds, &lds_grouped)) return e;
      void* ph = profiling() ? profile_open(kernel_name<K>(), s) : nullptr;
      hipLaunchKernelGGL(k_grouped<K>, dim3(total), dim3(K::NT), b.lds, s, t);
      if (ph) {
        double fl = 0, by = 0;
        for (int i = 0; i < cnt; ++i) { fl += b.flops[j0 + i]; by += b.bytes[j0 + i]; }
        profile_close(ph, s, cnt, fl, by);
      }
    }
    count_launches(1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// Launch now, or -- inside an ssa_group_begin/ssa_group_end bracket -- queue for the grouped launch.
template <class K>
int submit(const typename K::Args& a, int gx, int gy, size_t lds, hipStream_t s) {
  typedef typename K::Args Args;
  static_assert(sizeof(GroupTable<K>) <= kKernargBudget + 64, "job table exceeds the kernel-argument budget");
  if (gx <= 0 || gy <= 0) return -1;
  GroupState& g = group_state();
  double note_f = 0, note_b = 0;
  const bool prof = profiling();
  if (prof) profile_take_note(&note_f, &note_b);
  if (g.depth == 0) {
    static LdsLimit lds_single;
    if (int e = ensure_lds<K>((const void*)k_single<K>, lds, &lds_single)) return e;
    void* ph = prof ? profile_open(kernel_name<K>(), s) : nullptr;
    hipLaunchKernelGGL(k_single<K>, dim3(gx, gy), dim3(K::NT), lds, s, a);
    if (ph) profile_close(ph, s, 1, note_f, note_b);
    count_launches(1);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
  }
  const void* key = (const void*)k_grouped<K>;
  Bucket* b = nullptr;
  for (Bucket& c : g.buckets)
    if (c.key == key) { b = &c; break; }
  if (!b) {
    g.buckets.push_back(Bucket{key, &flush_bucket<K>, {}, {}, {}, {}, {}, 0});
    b = &g.buckets.back();
  }
  const size_t off = b->args.size();
  b->args.resize(off + sizeof(Args));
  memcpy(b->args.data() + off, &a, sizeof(Args));
  b->gx.push_back(gx);
  b->gy.push_back(gy);
  b->flops.push_back(note_f);
  b->bytes.push_back(note_b);
  if (lds > b->lds) b->lds = lds;
  return 0;
}

}  // namespace ssa
