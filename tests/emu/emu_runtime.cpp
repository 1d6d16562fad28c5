// TEST INFRASTRUCTURE ONLY: the fiber scheduler behind tests/emu/hip/hip_runtime.h (see there).  One OS thread; the threads of
// a workgroup are fibers run round-robin; a fiber gives up the processor only at __syncthreads and at wave operations.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <string>
#include <vector>

namespace emu {

struct Wave {
  unsigned char deposit[64 * 64];      // operands of the lanes currently waiting in a wave operation (one 64-byte slot per lane)
  unsigned char snap[2][64 * 64];      // what a released group reads: the deposit area at the moment of its release (two of them:
                                       // a released lane may deposit for its NEXT operation while a slower lane of the group still reads)
  unsigned long long snap_mask[2] = {0, 0};   // lanes that took part in that release (EXEC mask of the operation)
  long long gen = 0;                   // releases so far
  int waiting = 0;                     // lanes currently waiting in a wave operation (not yet released)
  int alive = 0;
  unsigned long long live_mask = 0;
};

struct Dma { void* dst; const void* src; int bytes; };

// Context switch between fibers: callee-saved registers + stack pointer, no signal-mask system calls (ucontext's swapcontext
// makes two per switch: a kernel of 10^5 threads spent its time there).  x86-64 System V only -- this is test infrastructure
// for the build container.
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch,.-emu_switch
)");

struct Fiber {
  void* sp = nullptr;                  // saved stack pointer while the fiber is not running
  uint3 tid;
  int lane = 0, wave = 0;
  bool done = false;
  int wait = 0;                        // 0 runnable, 1 block barrier, 2 wave operation (not yet released)
  long long wait_gen = 0;              // barrier generation waited for
  const void* site = nullptr;          // wave operation: the call site this lane waits at
  int snap = 0;                        // ... and, once released, which snapshot it reads
  std::vector<Dma> dma;                // lazy mode: LDS-DMA pieces issued and not yet retired (oldest first)
};

Fiber* g_cur = nullptr;
dim3 g_block_idx, g_block_dim, g_grid_dim;
static void* g_sched_sp = nullptr;
static std::vector<Fiber> g_fibers;
static std::vector<Wave> g_waves;
static long long g_bar_gen = 0;
static int g_bar_arrived = 0, g_alive = 0;
static const std::function<void()>* g_body = nullptr;
static unsigned char* g_stacks = nullptr;
static size_t g_stack_cap = 0;
static const size_t STACK = 192 * 1024;

const uint3& tid() { return g_cur->tid; }
int lane() { return g_cur->lane; }

static void yield() { emu_switch(&g_cur->sp, g_sched_sp); }

static double g_count[3] = {0, 0, 0};        // per-LANE events: MFMA, LDS-DMA piece, barrier (divide by 64 for wave instructions)
void count_mfma() { g_count[0] += 1; }
static int g_dma_lazy = 0;
void dma_issue(void* dst, const void* src, int bytes) {
  g_count[1] += 1;
  if (!g_dma_lazy) {
    memcpy(dst, src, (size_t)bytes);
    return;
  }
  g_cur->dma.push_back(Dma{dst, src, bytes});
}
void waitcnt_vm(int n) {
  std::vector<Dma>& q = g_cur->dma;
  const size_t keep = n < 0 ? 0 : (size_t)n;
  if (q.size() <= keep) return;
  const size_t retire = q.size() - keep;
  for (size_t i = 0; i < retire; ++i) memcpy(q[i].dst, q[i].src, (size_t)q[i].bytes);
  q.erase(q.begin(), q.begin() + (long)retire);
}

void block_barrier() {
  Fiber* f = g_cur;
  g_count[2] += 1;
  if (++g_bar_arrived == g_alive) {      // last one in: release everybody
    g_bar_arrived = 0;
    ++g_bar_gen;
    return;
  }
  f->wait = 1;
  f->wait_gen = g_bar_gen;
  yield();
}

// Wave operations under divergence.  The hardware runs a wave in lockstep under an EXEC mask: lanes that skipped a branch do not
// take part in the shuffles inside it (a quad reduction inside `if (active)` is legal when the four lanes of a quad agree).  Here
// every lane is a fiber, so the lanes inside the branch wait at ITS call site while the others run ahead to a later one.  Rule:
// when every live lane of the wave waits at the same site, they are released together (the common case); when every live lane
// is blocked but at DIFFERENT sites, the group at the lowest code address goes first with only its lanes active (structured
// code lays the body of a branch / loop out before what follows it) -- reads from a lane outside the group return zeros.
static void release_group(Wave& w, int wave_index, const void* site) {
  const int idx = (int)(w.gen & 1);
  memcpy(w.snap[idx], w.deposit, sizeof w.deposit);
  unsigned long long mask = 0;
  for (int l = 0; l < 64; ++l) {
    const size_t t = (size_t)wave_index * 64 + (size_t)l;
    if (t >= g_fibers.size()) break;
    Fiber& f = g_fibers[t];
    if (!f.done && f.wait == 2 && f.site == site) {
      f.wait = 0;
      f.snap = idx;
      mask |= 1ull << l;
      --w.waiting;
    }
  }
  for (int l = 0; l < 64; ++l)
    if (!((mask >> l) & 1ull)) memset(w.snap[idx] + 64 * l, 0, 64);
  w.snap_mask[idx] = mask;
  ++w.gen;
}

// all = true: only if every live lane of the wave waits in a wave operation; false (the scheduler found nothing runnable): also
// when the other live lanes are stuck elsewhere.  Returns whether a group was released.
static bool try_release(int wave_index, bool all) {
  Wave& w = g_waves[(size_t)wave_index];
  const void* lowest = nullptr;
  int waiting = 0;
  for (int l = 0; l < 64; ++l) {
    const size_t t = (size_t)wave_index * 64 + (size_t)l;
    if (t >= g_fibers.size()) break;
    const Fiber& f = g_fibers[t];
    if (f.done || f.wait != 2) continue;
    ++waiting;
    if (lowest == nullptr || (uintptr_t)f.site < (uintptr_t)lowest) lowest = f.site;
  }
  if (waiting == 0 || (all && waiting != w.alive)) return false;
  release_group(w, wave_index, lowest);
  return true;
}

WaveView wave_exchange(const void* site, const void* mine, int bytes) {
  Fiber* f = g_cur;
  Wave& w = g_waves[(size_t)f->wave];
  memcpy(w.deposit + 64 * f->lane, mine, (size_t)bytes);
  f->wait = 2;
  f->site = site;
  if (++w.waiting == w.alive) try_release(f->wave, true);      // (the scan over the lanes only when everybody has arrived)
  if (f->wait == 2) yield();
  return WaveView{w.snap[f->snap], w.snap_mask[f->snap]};
}

static void trampoline() {
  (*g_body)();
  waitcnt_vm(0);                        // (s_endpgm waits for outstanding memory operations)
  Fiber* f = g_cur;
  f->done = true;
  // a thread that has left no longer takes part in barriers / wave operations: release what now only waited for it
  --g_alive;
  Wave& w = g_waves[f->wave];
  --w.alive;
  w.live_mask &= ~(1ull << f->lane);
  if (g_alive > 0 && g_bar_arrived == g_alive) {
    g_bar_arrived = 0;
    ++g_bar_gen;
  }
  if (w.alive > 0 && w.waiting == w.alive) try_release(f->wave, true);
  emu_switch(&f->sp, g_sched_sp);
  abort();                             // (a finished fiber is never resumed)
}

static bool runnable(const Fiber& f) {
  if (f.done) return false;
  if (f.wait == 0) return true;
  if (f.wait == 1) return g_bar_gen > f.wait_gen;
  return false;                        // (a lane in a wave operation is made runnable by release_group)
}

static std::string g_log;
static std::vector<unsigned char> g_dyn;
// Order in which the runnable threads of a workgroup get the processor between synchronisation points: 0 wave-major ascending
// (fast: development / audit runs), 1 descending and 2 random over the whole workgroup (the test suite).  A kernel whose result depends on it is missing a barrier (or races on LDS / global memory inside a workgroup):
// the tests run every kernel under several orders and require identical results.
static int g_order = 0;
static unsigned long long g_rng = 1;
static std::vector<int> g_perm;
void* dyn_shared() { return g_dyn.data(); }

void launch(const char* name, dim3 grid, dim3 block, size_t dyn_shared_bytes, const std::function<void()>& body) {
  if (g_dyn.size() < dyn_shared_bytes + 64) g_dyn.resize(dyn_shared_bytes + 64);
  const int nt = (int)(block.x * block.y * block.z);
  if (nt <= 0 || grid.x * (size_t)grid.y * grid.z == 0) return;
  if (g_log.size() < (1u << 20)) {
    char tmp[96];
    snprintf(tmp, sizeof tmp, " grid=(%u,%u,%u) block=%d\n", grid.x, grid.y, grid.z, nt);
    g_log += name;
    g_log += tmp;
  }
  if (g_cur != nullptr) {
    fprintf(stderr, "emu: nested launch\n");
    abort();
  }
  if ((size_t)nt * STACK > g_stack_cap) {
    if (g_stacks) munmap(g_stacks, g_stack_cap);
    g_stack_cap = (size_t)nt * STACK;
    g_stacks = (unsigned char*)mmap(nullptr, g_stack_cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_stacks == MAP_FAILED) {
      perror("emu: mmap");
      abort();
    }
    for (int t = 0; t < nt; ++t)        // a guard page at the low end of every stack: an overflow faults instead of corrupting a neighbour
      mprotect(g_stacks + (size_t)t * STACK, 4096, PROT_NONE);
  }
  g_block_dim = block;
  g_grid_dim = grid;
  g_body = &body;
  const int nw = (nt + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_block_idx = dim3(bx, by, bz);
        g_fibers.assign((size_t)nt, Fiber());
        g_perm.resize((size_t)nt);
        for (int t = 0; t < nt; ++t) g_perm[(size_t)t] = t;
        g_waves.assign((size_t)nw, Wave());
        g_bar_gen = 0;
        g_bar_arrived = 0;
        g_alive = nt;
        for (int t = 0; t < nt; ++t) {
          Fiber& f = g_fibers[(size_t)t];
          f.tid = uint3{(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
          f.lane = t & 63;
          f.wave = t >> 6;
          g_waves[(size_t)f.wave].alive++;
          g_waves[(size_t)f.wave].live_mask |= 1ull << f.lane;
          // initial frame: six callee-saved slots, then the address emu_switch "returns" to; the entry point then sees the
          // stack as after a call (rsp = 8 mod 16)
          void** top = (void**)(g_stacks + (size_t)(t + 1) * STACK);
          top[-1] = nullptr;
          top[-2] = (void*)&trampoline;
          for (int i = 3; i <= 8; ++i) top[-i] = nullptr;
          f.sp = (void*)(top - 8);
        }
        int done = 0;
        while (done < nt) {
          bool progressed = false;
          if (g_order == 2) {                                      // a fresh random order of the fibers in every scheduling round
            for (int i = nt - 1; i > 0; --i) {
              g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull;
              std::swap(g_perm[(size_t)i], g_perm[(size_t)((g_rng >> 33) % (unsigned long long)(i + 1))]);
            }
          }
          if (g_order == 0) {
            // wave-major: a wave keeps the processor until all of its lanes are blocked at a workgroup barrier (or done) -- a
            // wave operation is released by its last arriving lane, so the wave's next pass continues right behind it; the other
            // waves' fibers are not even looked at in between (4 x fewer checks per MFMA in a 4-wave workgroup)
            for (int wv = 0; wv < nw; ++wv) {
              const int t0 = wv * 64, t1 = t0 + 64 < nt ? t0 + 64 : nt;
              for (bool again = true; again;) {
                again = false;
                for (int t = t0; t < t1; ++t) {
                  Fiber& f = g_fibers[(size_t)t];
                  if (!runnable(f)) continue;
                  f.wait = 0;
                  g_cur = &f;
                  emu_switch(&g_sched_sp, f.sp);
                  g_cur = nullptr;
                  again = progressed = true;
                  if (f.done) ++done;
                }
              }
            }
          } else {
            for (int q = 0; q < nt; ++q) {
              const int t = g_order == 1 ? nt - 1 - q : g_perm[(size_t)q];
              Fiber& f = g_fibers[(size_t)t];
              if (!runnable(f)) continue;
              f.wait = 0;
              g_cur = &f;
              emu_switch(&g_sched_sp, f.sp);
              g_cur = nullptr;
              progressed = true;
              if (f.done) ++done;
            }
          }
          if (!progressed) {                                       // divergent wave operations: let the earliest group go
            for (int wv = 0; wv < nw; ++wv) progressed = try_release(wv, false) || progressed;
            if (progressed) continue;
            fprintf(stderr, "emu: deadlock in block (%u, %u, %u): %d of %d threads finished; the rest wait at a barrier / wave "
                    "operation the others never reach (divergent __syncthreads or a partial-wave shuffle)\n", bx, by, bz, done, nt);
            abort();
          }
        }
      }
  g_body = nullptr;
}

}  // namespace emu

// wave-level work of the launches since the last call: out[0] MFMA instructions, out[1] LDS-DMA instructions (1 KiB pieces),
// out[2] barrier arrivals of waves -- the static work model of a kernel (staging pieces per MFMA)
extern "C" void es_emu_take_counters(double* out) {
  for (int i = 0; i < 3; ++i) {
    out[i] = emu::g_count[i] / 64.0;
    emu::g_count[i] = 0;
  }
}

extern "C" void es_emu_set_dma_mode(int lazy) { emu::g_dma_lazy = lazy; }

extern "C" void es_emu_set_schedule(int order, unsigned long long seed) {
  emu::g_order = order;
  emu::g_rng = seed * 2 + 1;
}

// the launches since the last call, one per line ("<kernel expression> grid=(x,y,z) block=n"): lets a test assert WHICH kernel
// a dispatch function chose; returns the number of bytes written (truncated to cap - 1) and clears the log
extern "C" int es_emu_take_launch_log(char* buf, int cap) {
  int n = (int)emu::g_log.size();
  if (n > cap - 1) n = cap - 1;
  if (n > 0) memcpy(buf, emu::g_log.data(), (size_t)n);
  if (cap > 0) buf[n > 0 ? n : 0] = 0;
  emu::g_log.clear();
  return n;
}
