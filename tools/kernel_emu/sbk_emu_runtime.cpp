// Fiber runtime behind tools/kernel_emu/sbk_device.h  --  TEST TOOLING ONLY.
//
// A launch walks the grid with a pool of OS threads; each OS thread executes
// one workgroup at a time as `blockDim` cooperative fibers (hand-rolled x86-64
// context switch, no syscalls).  __syncthreads / wave rendezvous are
// generation barriers that yield to the next fiber of the same workgroup.
#include <sbk_device.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void sbk_emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl sbk_emu_ctx_switch
.type sbk_emu_ctx_switch,@function
sbk_emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size sbk_emu_ctx_switch, .-sbk_emu_ctx_switch
)");

namespace sbk_emu {
namespace {

constexpr size_t kStack = 96 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  ThreadCtx ctx;
};

struct BlockRun {
  std::vector<Fiber> fibers;
  int n = 0;
  int current = 0;
  int live = 0;
  void* sched_sp = nullptr;
  const std::function<void()>* body = nullptr;
  // block barrier
  int bar_count = 0;
  unsigned bar_gen = 0;
  // wave barriers
  int wave_count[kMaxThreads / 64] = {0};
  unsigned wave_gen[kMaxThreads / 64] = {0};
  int wave_live[kMaxThreads / 64] = {0};
  float wbuf[kMaxThreads / 64][2][512];
  std::vector<char> dyn;
  unsigned long progress = 0;  // bumped on every barrier arrival / fiber exit (deadlock detection)
  ~BlockRun() {
    for (auto& f : fibers) free(f.stack);
  }
};

thread_local BlockRun* g_run = nullptr;
thread_local ThreadCtx g_dummy;

void yield_to_sched() {
  BlockRun* r = g_run;
  Fiber& f = r->fibers[r->current];
  sbk_emu_ctx_switch(&f.sp, r->sched_sp);
}

void release_block_barrier_if_complete(BlockRun* r) {
  if (r->live > 0 && r->bar_count >= r->live) {
    r->bar_count = 0;
    r->bar_gen++;
  }
}
void release_wave_barrier_if_complete(BlockRun* r, int w) {
  if (r->wave_live[w] > 0 && r->wave_count[w] >= r->wave_live[w]) {
    r->wave_count[w] = 0;
    r->wave_gen[w]++;
  }
}

void fiber_entry() {
  BlockRun* r = g_run;
  Fiber& f = r->fibers[r->current];
  (*r->body)();
  f.done = true;
  r->progress++;
  r->live--;
  r->wave_live[f.ctx.wave]--;
  // a thread that left the kernel no longer takes part in barriers
  release_block_barrier_if_complete(r);
  release_wave_barrier_if_complete(r, f.ctx.wave);
  yield_to_sched();
  abort();  // never resumed
}

void run_block(BlockRun& r, dim3 bid, dim3 grid, dim3 block, const std::function<void()>& body, size_t lds) {
  const int n = block.x * block.y * block.z;
  r.n = n;
  r.live = n;
  r.body = &body;
  r.bar_count = 0;
  if ((int)r.fibers.size() < n) r.fibers.resize(n);
  if (r.dyn.size() < lds + 64) r.dyn.resize(lds + 64);
  memset(r.wave_count, 0, sizeof(r.wave_count));
  memset(r.wave_live, 0, sizeof(r.wave_live));
  for (int i = 0; i < n; ++i) {
    Fiber& f = r.fibers[i];
    if (!f.stack) f.stack = (char*)aligned_alloc(64, kStack);
    f.done = false;
    f.ctx.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    f.ctx.bid = bid;
    f.ctx.bdim = block;
    f.ctx.gdim = grid;
    f.ctx.lin = i;
    f.ctx.lane = i & 63;
    f.ctx.wave = i >> 6;
    r.wave_live[i >> 6]++;
    // initial frame: 6 callee-saved slots, entry address, fake return address
    uintptr_t top = ((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *(--sp) = nullptr;              // fake return address for fiber_entry (keeps rsp%16==8 at entry)
    *(--sp) = (void*)&fiber_entry;  // popped by `ret`
    for (int k = 0; k < 6; ++k) *(--sp) = nullptr;
    f.sp = sp;
  }
  g_run = &r;
  int remaining = n;
  while (remaining > 0) {
    const unsigned long before = r.progress;
    for (int i = 0; i < n; ++i) {
      Fiber& f = r.fibers[i];
      if (f.done) continue;
      r.current = i;
      sbk_emu_ctx_switch(&r.sched_sp, f.sp);
      if (f.done) remaining--;
    }
    if (remaining > 0 && r.progress == before) {
      fprintf(stderr, "sbk_emu: deadlock in block (%u,%u,%u): %d threads wait on a barrier that cannot complete\n",
              bid.x, bid.y, bid.z, remaining);
      abort();
    }
  }
  g_run = nullptr;
}

}  // namespace

ThreadCtx& cur() {
  BlockRun* r = g_run;
  if (!r) return g_dummy;
  return r->fibers[r->current].ctx;
}

void block_barrier() {
  BlockRun* r = g_run;
  const unsigned gen = r->bar_gen;
  r->progress++;
  r->bar_count++;
  release_block_barrier_if_complete(r);
  while (r->bar_gen == gen) yield_to_sched();
}

void wave_barrier() {
  BlockRun* r = g_run;
  const int w = r->fibers[r->current].ctx.wave;
  const unsigned gen = r->wave_gen[w];
  r->progress++;
  r->wave_count[w]++;
  release_wave_barrier_if_complete(r, w);
  while (r->wave_gen[w] == gen) yield_to_sched();
}

float* wave_buf(int which) {
  BlockRun* r = g_run;
  return r->wbuf[r->fibers[r->current].ctx.wave][which];
}

void* dyn_lds() {
  BlockRun* r = g_run;
  uintptr_t p = (uintptr_t)r->dyn.data();
  return (void*)((p + 63) & ~(uintptr_t)63);
}

namespace {
struct Pool {
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  const std::function<void()>* body = nullptr;
  dim3 grid, block;
  size_t lds = 0;
  long nblocks = 0;
  std::atomic<long> next{0};
  unsigned long epoch = 0;
  int busy = 0;

  void drain() {
    static thread_local BlockRun run;
    for (;;) {
      const long b = next.fetch_add(1);
      if (b >= nblocks) break;
      dim3 bid(b % grid.x, (b / grid.x) % grid.y, b / ((long)grid.x * grid.y));
      run_block(run, bid, grid, block, *body, lds);
    }
  }
  void worker_main() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        cv_work.wait(lk, [&] { return epoch != seen; });
        seen = epoch;
      }
      drain();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--busy == 0) cv_done.notify_all();
      }
    }
  }
  void run(const std::function<void()>& b, dim3 g, dim3 blk, size_t l, long nb, int nworkers) {
    body = &b; grid = g; block = blk; lds = l; nblocks = nb;
    next.store(0);
    const int extra = (int)std::min<long>(nworkers - 1, nb - 1);
    if (extra > 0) {
      while ((int)threads.size() < nworkers - 1) {
        threads.emplace_back([this] { worker_main(); });
        threads.back().detach();
      }
      {
        std::lock_guard<std::mutex> lk(m);
        busy = (int)threads.size();
        epoch++;
      }
      cv_work.notify_all();
    }
    drain();
    if (extra > 0) {
      std::unique_lock<std::mutex> lk(m);
      cv_done.wait(lk, [&] { return busy == 0; });
    }
  }
};
}  // namespace

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  const int nthreads_blk = block.x * block.y * block.z;
  if (nblocks <= 0 || nthreads_blk <= 0) return;
  if (nthreads_blk > kMaxThreads) {
    fprintf(stderr, "sbk_emu: block of %d threads\n", nthreads_blk);
    abort();
  }
  static int nworkers = [] {
    const char* e = getenv("SBK_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n;
  }();
  static Pool* pool = new Pool();  // leaked on purpose: workers are detached
  pool->run(body, grid, block, lds_bytes, nblocks, nworkers);
}

// A cooperative launch: one OS thread per workgroup, all alive together, so that a workgroup may wait for the others.
void launch_coop(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes) {
  const long nblocks = (long)grid.x * grid.y * grid.z;
  const int nthreads_blk = block.x * block.y * block.z;
  if (nblocks <= 0 || nthreads_blk <= 0) return;
  if (nthreads_blk > kMaxThreads || nblocks > 64) {
    fprintf(stderr, "sbk_emu: cooperative launch of %ld blocks x %d threads\n", nblocks, nthreads_blk);
    abort();
  }
  std::vector<std::thread> ts;
  for (long b = 0; b < nblocks; ++b) {
    ts.emplace_back([&, b] {
      BlockRun* run = new BlockRun();  // (not the pool's thread_local: this thread dies with the launch)
      dim3 bid(b % grid.x, (b / grid.x) % grid.y, b / ((long)grid.x * grid.y));
      run_block(*run, bid, grid, block, body, lds_bytes);
      delete run;
    });
  }
  for (auto& t : ts) t.join();
}

void grid_barrier_wait(int* ctr, int target) {
  while (__atomic_load_n(ctr, __ATOMIC_SEQ_CST) - target < 0) std::this_thread::yield();
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
}

}  // namespace sbk_emu
