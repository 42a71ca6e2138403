// Host state of the grouped-launch facility (group.h) and its C entry points.
#include "group.h"
#include "common.h"
#include "../../include/semseg_hip.h"
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <stdlib.h>
#include <vector>

namespace ssa {

GroupState& group_state() {
  static thread_local GroupState st;
  return st;
}

static std::atomic<long> g_launches{0};
void count_launches(int n) { g_launches.fetch_add(n, std::memory_order_relaxed); }


// ---- per-launch profile
struct ProfRec { const char* kernel; hipEvent_t e0, e1; int jobs; double flops, bytes; };
static std::atomic<bool> g_profiling{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static thread_local double t_note_f = 0, t_note_b = 0;

bool profiling() { return g_profiling.load(std::memory_order_relaxed); }
void profile_take_note(double* f, double* b) { *f = t_note_f; *b = t_note_b; t_note_f = t_note_b = 0; }
void* profile_open(const char* kernel, hipStream_t s) {
  ProfRec* r = new ProfRec{kernel, nullptr, nullptr, 0, 0, 0};
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) { delete r; return nullptr; }
  (void)hipEventRecord(r->e0, s);
  return r;
}
void profile_close(void* h, hipStream_t s, int jobs, double flops, double bytes) {
  ProfRec* r = (ProfRec*)h;
  (void)hipEventRecord(r->e1, s);
  r->jobs = jobs; r->flops = flops; r->bytes = bytes;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(*r);
  delete r;
}

// ---- side streams of a bracket (SSA_GROUP_STREAMS = total streams a bracket may use; 1 = off)
struct SideStreams { std::vector<hipStream_t> streams; std::vector<hipEvent_t> join; hipEvent_t fork; };
static SideStreams* side_streams() {
  static thread_local SideStreams* ss = nullptr;
  static thread_local bool tried = false;
  if (tried) return ss;
  tried = true;
  const char* e = getenv("SSA_GROUP_STREAMS");
  const int n = e ? atoi(e) : 1;
  if (n <= 1) return nullptr;
  SideStreams* t = new SideStreams;
  if (hipEventCreateWithFlags(&t->fork, hipEventDisableTiming) != hipSuccess) { delete t; return nullptr; }
  for (int i = 0; i < n - 1 && i < 7; ++i) {
    hipStream_t s; hipEvent_t ev;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) break;
    t->streams.push_back(s);
    t->join.push_back(ev);
  }
  if (t->streams.empty()) { delete t; return nullptr; }
  ss = t;
  return ss;
}

}  // namespace ssa

extern "C" {

// 0: bf16 storage (libsemseg_hip.so), 1: fp16 storage (libsemseg_hip_f16.so, -DSSA_ELEM_F16) -- the loader checks it
int ssa_elem_type(void) {
#ifdef SSA_ELEM_F16
  return 1;
#else
  return 0;
#endif
}

// first 16 hex digits of sha256 over the kernel sources this library was built from (csrc/Makefile); the loader
// recomputes it over the sources next to the binary and refuses a stale build
#ifndef SSA_SOURCE_SHA
#define SSA_SOURCE_SHA "unknown"
#endif
const char* ssa_source_sha(void) { return SSA_SOURCE_SHA; }


int ssa_group_begin(void) {
  ssa::GroupState& g = ssa::group_state();
  if (g.depth == 0) {
    g.buckets.clear();
    g.error = 0;
  }
  ++g.depth;
  return SSA_OK;
}

int ssa_group_end(void* stream) {
  ssa::GroupState& g = ssa::group_state();
  if (g.depth <= 0) return SSA_EINVAL;
  if (--g.depth > 0) return SSA_OK;          // nested brackets flush with the outermost one
  int rc = g.error;
  hipStream_t main = (hipStream_t)stream;
  int nonempty = 0;
  for (ssa::Bucket& b : g.buckets) nonempty += !b.gx.empty();
  ssa::SideStreams* ss = (rc == 0 && nonempty > 1 && !ssa::profiling()) ? ssa::side_streams() : nullptr;
  if (ss) {
    // The launches of a bracket are independent of one another by contract (the bracket reorders
    // them by kernel instantiation already): launch k > 0 goes to a side stream forked from `stream`
    // here and joined below -- in a captured step these are parallel branches of the hipGraph, so the
    // 48/96/192/384-channel launches of one depth level, none of which fills 256 CUs, overlap.
    if (hipEventRecord(ss->fork, main) != hipSuccess) ss = nullptr;
  }
  int k = 0, used = 0;
  for (ssa::Bucket& b : g.buckets) {
    if (rc != 0 || b.gx.empty()) continue;
    hipStream_t s = main;
    if (ss && k > 0) {
      const int i = (k - 1) % (int)ss->streams.size();
      s = ss->streams[i];
      if (i >= used) {
        if (hipStreamWaitEvent(s, ss->fork, 0) != hipSuccess) rc = SSA_EINVAL;
        used = i + 1;
      }
    }
    if (rc == 0) rc = b.flush(b, s);
    ++k;
  }
  for (int i = 0; ss && i < used; ++i) {
    if (hipEventRecord(ss->join[i], ss->streams[i]) != hipSuccess ||
        hipStreamWaitEvent(main, ss->join[i], 0) != hipSuccess) rc = rc ? rc : SSA_EINVAL;
  }
  g.buckets.clear();
  g.error = 0;
  return rc;
}

int ssa_group_abort(void) {
  ssa::GroupState& g = ssa::group_state();
  g.depth = 0;
  g.buckets.clear();
  g.error = 0;
  return SSA_OK;
}

int ssa_profile_begin(void) {
  std::lock_guard<std::mutex> lk(ssa::g_prof_mu);
  ssa::g_prof.clear();
  ssa::g_profiling.store(true);
  return SSA_OK;
}

int ssa_profile_note(double flops, double bytes) {
  ssa::t_note_f += flops;
  ssa::t_note_b += bytes;
  return SSA_OK;
}

// Ends profiling, waits for the device, aggregates per kernel instantiation.  Returns the
// number of distinct kernels (<= max_recs written to out).
int ssa_profile_end(ssa_profile_rec* out, int max_recs) {
  ssa::g_profiling.store(false);
  if (hipDeviceSynchronize() != hipSuccess) return SSA_EINVAL;
  std::lock_guard<std::mutex> lk(ssa::g_prof_mu);
  std::map<std::string, ssa_profile_rec> agg;
  for (ssa::ProfRec& r : ssa::g_prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.e0, r.e1);
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    std::string name(r.kernel);
    const size_t p = name.find("[K = ");
    if (p != std::string::npos) name = name.substr(p + 5, name.size() - p - 6);
    const std::string anon = "(anonymous namespace)::";
    for (size_t q; (q = name.find(anon)) != std::string::npos;) name.erase(q, anon.size());
    ssa_profile_rec& a = agg[name];
    if (a.launches == 0) {
      memset(&a, 0, sizeof(a));
      strncpy(a.kernel, name.c_str(), sizeof(a.kernel) - 1);
    }
    a.launches += 1;
    a.jobs += r.jobs;
    a.total_us += (double)ms * 1e3;
    a.flops += r.flops;
    a.bytes += r.bytes;
  }
  ssa::g_prof.clear();
  int n = 0;
  for (auto& kv : agg) {
    if (out && n < max_recs) out[n] = kv.second;
    ++n;
  }
  return n;
}

long ssa_launch_count(int reset) {
  const long v = ssa::g_launches.load(std::memory_order_relaxed);
  if (reset) ssa::g_launches.store(0, std::memory_order_relaxed);
  return v;
}

}  // extern "C"
