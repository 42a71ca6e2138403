// Fiber scheduler of the CPU emulation shim (tools/emu/include/hip/hip_runtime.h).  TEST INFRASTRUCTURE.
//
// One workgroup at a time: its NT threads are ucontext fibers of the calling OS thread, resumed round-robin
// (EMU_SCHED=rev: in reverse order, EMU_SCHED=rand: a fresh random order every pass -- to shake out code that
// depends on the order in which waves reach a barrier).  A fiber runs until it blocks at a rendezvous
// (__syncthreads(), a wave collective) or returns.  Fibers that have returned no longer count at a barrier
// (s_barrier semantics).  A pass over all fibers without progress = deadlock = abort with a message.
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <random>
#include <vector>

namespace emu {

thread_local Ctx* cur = nullptr;

namespace {

constexpr size_t kStack = 512 * 1024;
constexpr size_t kLds = 160 * 1024;

struct Rendezvous {
  int arrived = 0, alive = 0;
  unsigned gen = 0;
};

struct Dma { void* dst; unsigned char data[16]; int bytes; };

struct Fiber {
  ucontext_t ctx;
  Ctx c;
  bool done = false, started = false;
  Rendezvous* wait_on = nullptr;
  unsigned wait_gen = 0;
  std::vector<Dma> dma;
};

struct Block {
  std::vector<Fiber> f;
  Rendezvous block_rv;
  std::vector<Rendezvous> wave_rv;
  std::vector<unsigned char> wavebuf;      // [wave][64 lanes][64 bytes]
  ucontext_t sched;
  Fiber* running = nullptr;
  void (*tramp)(void*) = nullptr;
  void* closure = nullptr;
  unsigned char* lds = nullptr;
};

thread_local Block* B = nullptr;
thread_local unsigned char* g_stacks = nullptr;
thread_local size_t g_stacks_n = 0;
thread_local unsigned char* g_lds = nullptr;

void fiber_main() {
  Block* b = B;
  Fiber* me = b->running;
  b->tramp(b->closure);
  // flush DMAs nobody waited for (a kernel may end without a barrier after its last DMA; harmless)
  for (Dma& d : me->dma) memcpy(d.dst, d.data, d.bytes);
  me->dma.clear();
  me->done = true;
  // a returned thread no longer takes part in barriers
  Rendezvous* rvs[2] = {&b->block_rv, &b->wave_rv[me->c.wave]};
  for (Rendezvous* rv : rvs) {
    rv->alive -= 1;
    if (rv->alive > 0 && rv->arrived == rv->alive) { rv->arrived = 0; rv->gen += 1; }
  }
  swapcontext(&me->ctx, &b->sched);
}

void rendezvous(Rendezvous* rv) {
  Block* b = B;
  Fiber* me = b->running;
  const unsigned g = rv->gen;
  rv->arrived += 1;
  if (rv->arrived == rv->alive) {
    rv->arrived = 0;
    rv->gen += 1;
    return;
  }
  me->wait_on = rv;
  me->wait_gen = g;
  swapcontext(&me->ctx, &b->sched);
  me->wait_on = nullptr;
}

}  // namespace

void sync_block() {
  Fiber* me = B->running;
  // LDS DMAs of this thread become visible with the barrier (vmcnt(0) + s_barrier on the device)
  for (Dma& d : me->dma) memcpy(d.dst, d.data, d.bytes);
  me->dma.clear();
  rendezvous(&B->block_rv);
}

void sync_wave() { rendezvous(&B->wave_rv[B->running->c.wave]); }

unsigned char* wave_buf(int lane) {
  return B->wavebuf.data() + ((size_t)B->running->c.wave * 64 + (size_t)(lane & 63)) * 64;
}

unsigned char* dyn_lds() { return B->lds; }

void defer_dma(void* dst, const void* src, int bytes) {
  Dma d;
  d.dst = dst;
  d.bytes = bytes;
  memcpy(d.data, src, bytes);
  B->running->dma.push_back(d);
}

void launch(dim3 grid, dim3 block, size_t lds, void (*tramp)(void*), void* closure) {
  const int nt = (int)(block.x * block.y * block.z);
  if (nt <= 0 || nt > 1024 || lds > kLds) {
    fprintf(stderr, "emu::launch: bad configuration nt=%d lds=%zu\n", nt, lds);
    abort();
  }
  if (g_stacks_n < (size_t)nt) {
    if (g_stacks) munmap(g_stacks, g_stacks_n * kStack);
    g_stacks = (unsigned char*)mmap(nullptr, (size_t)nt * kStack, PROT_READ | PROT_WRITE,
                                    MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_stacks == MAP_FAILED) { perror("mmap"); abort(); }
    g_stacks_n = nt;
  }
  if (!g_lds) {
    if (posix_memalign((void**)&g_lds, 256, kLds)) abort();
  }
  static const char* sched_env = getenv("EMU_SCHED");
  const int sched_mode = !sched_env ? 0 : (sched_env[0] == 'r' && sched_env[1] == 'e') ? 1 : (sched_env[0] == 'r' ? 2 : 0);
  static thread_local std::mt19937 rng(12345);
  const int nwave = (nt + 63) / 64;
  Block blk;
  Block* saved_B = B;
  Ctx* saved_cur = cur;
  B = &blk;
  blk.tramp = tramp;
  blk.closure = closure;
  blk.lds = g_lds;
  blk.wavebuf.assign((size_t)nwave * 64 * 64, 0);
  std::vector<int> order(nt);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        // poison the LDS: a kernel that reads what it never wrote gets NaN-ish garbage, not zeros
        memset(g_lds, 0x7f, kLds);
        blk.f.clear();
        blk.f.resize(nt);
        blk.block_rv = Rendezvous();
        blk.block_rv.alive = nt;
        blk.wave_rv.assign(nwave, Rendezvous());
        for (int t = 0; t < nt; ++t) {
          Fiber& f = blk.f[t];
          f.c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          f.c.bid = dim3(bx, by, bz);
          f.c.bdim = block;
          f.c.gdim = grid;
          f.c.lane = t & 63;
          f.c.wave = t >> 6;
          blk.wave_rv[t >> 6].alive += 1;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * kStack;
          f.ctx.uc_stack.ss_size = kStack;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, fiber_main, 0);
        }
        int remaining = nt;
        for (int t = 0; t < nt; ++t) order[t] = t;
        while (remaining > 0) {
          if (sched_mode == 1) { for (int t = 0; t < nt; ++t) order[t] = nt - 1 - t; }
          else if (sched_mode == 2) std::shuffle(order.begin(), order.end(), rng);
          bool progress = false;
          for (int oi = 0; oi < nt; ++oi) {
            Fiber& f = blk.f[order[oi]];
            if (f.done) continue;
            if (f.wait_on && f.wait_on->gen == f.wait_gen) continue;    // still blocked
            blk.running = &f;
            cur = &f.c;
            progress = true;
            swapcontext(&blk.sched, &f.ctx);
            if (f.done) --remaining;
          }
          if (!progress) {
            fprintf(stderr, "emu: DEADLOCK in block (%u,%u,%u): %d threads blocked (a barrier or a wave collective "
                            "not reached by every live thread)\n", bx, by, bz, remaining);
            abort();
          }
        }
      }
  B = saved_B;
  cur = saved_cur;
}

}  // namespace emu
