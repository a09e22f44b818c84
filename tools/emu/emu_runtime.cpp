// tools/emu/emu_runtime.cpp — the fiber scheduler and the runtime API behind tools/emu/hip/hip_runtime.h (development aid; see there).
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <utility>
#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

namespace emu {

thread_local Ctx* cur = nullptr;
thread_local uintptr_t lds_base = 0;

namespace {

// ---- context switch: callee-saved registers on the fiber's own stack ----------------------------------------------------------
#if defined(__x86_64__)
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
#else
#error "tools/emu: the fiber switch is written for x86-64"
#endif

// ThreadSanitizer has to be told about the stack switches (tools/emu/tsan_host.cpp: races in the HOST layer — lanes, worker pool, mailbox —
// with the kernels running on this runtime); a switch also orders the two fibers, which is what a barrier or cross-lane operation does
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
}
#endif
#endif

constexpr size_t kStack = 128 << 10;     // per work-item
constexpr unsigned kMaxItems = 1024;

struct Fiber {
  void* sp = nullptr;
  Ctx ctx;
  bool done = false;
  const unsigned long long* wait_gen = nullptr;    // blocked until *wait_gen != wait_val
  unsigned long long wait_val = 0;
  void* tsan = nullptr;
};

struct Sched {
  char* stacks = nullptr;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  Group group;
  void* main_sp = nullptr;
  Fiber* running = nullptr;
  Launch* launch = nullptr;
  unsigned nitems = 0;
  void* main_tsan = nullptr;
};
thread_local Sched* g_sched = nullptr;
std::recursive_mutex g_launch_mutex;      // one launch at a time: __shared__ variables are function-local statics

void to_scheduler() {
  Sched* s = g_sched;
  Fiber* f = s->running;
#ifdef EMU_TSAN
  __tsan_switch_to_fiber(s->main_tsan, 0);
#endif
  emu_switch(&f->sp, s->main_sp);
}

void leave(Fiber* f) {       // a work-item has returned from the kernel
  Wave* w = f->ctx.wave;
  Group* g = f->ctx.group;
  w->live--;
  w->live_mask &= ~(1ull << f->ctx.lane);
  if (w->live > 0 && w->arrived == w->live) { w->arrived = 0; w->ballot[(w->gen + 1) & 1] = 0; w->gen++; }
  g->live--;
  if (g->live > 0 && g->arrived == g->live) { g->arrived = 0; g->gen++; }
  f->done = true;
}

extern "C" void emu_fiber_main() {
  Sched* s = g_sched;
  Fiber* f = s->running;
  s->launch->run();
  leave(f);
  to_scheduler();
  std::fprintf(stderr, "emu: a finished work-item was resumed\n");
  std::abort();
}

void prepare(Sched* s, unsigned i) {
  Fiber& f = s->fibers[i];
  char* top = s->stacks + (size_t)(i + 1) * kStack;
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                  // keeps the entry's frame 16-byte aligned as after a call
  *--sp = reinterpret_cast<void*>(&emu_fiber_main);  // `ret` of emu_switch jumps here
  for (int k = 0; k < 6; k++) *--sp = nullptr;      // rbp rbx r12 r13 r14 r15
  f.sp = sp;
  f.done = false;
  f.wait_gen = nullptr;
}

// The order in which the work-items of a workgroup get their turn: 0 = ascending (default), 1 = descending, otherwise a new pseudo-random
// order every pass seeded by the value (TM_EMU_ORDER in the environment).  The hardware runs the lanes of a wavefront in lockstep: between
// two synchronisation points a lane may see what a lane that ran before it has written, but nothing may DEPEND on who ran first, so the
// tests must give the same results under every order (a kernel that passes only under one of them has a race the device would decide
// its own way).
unsigned order_mode() {
  static const unsigned m = [] { const char* e = getenv("TM_EMU_ORDER"); return e ? (unsigned)strtoul(e, nullptr, 10) : 0u; }();
  return m;
}

void run_group(Sched* s) {
  unsigned live = s->nitems;
  const unsigned mode = order_mode();
  static thread_local std::vector<unsigned> perm;
  static thread_local unsigned long long rng = 0;
  if (mode > 1) {
    if (!rng) rng = 0x9E3779B97F4A7C15ull * mode;
    perm.resize(s->nitems);
    for (unsigned i = 0; i < s->nitems; i++) perm[i] = i;
  }
  while (live > 0) {
    bool progressed = false;
    if (mode > 1)
      for (unsigned i = s->nitems; i > 1; i--) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; std::swap(perm[i - 1], perm[rng % i]); }
    for (unsigned k = 0; k < s->nitems; k++) {
      const unsigned i = mode == 0 ? k : mode == 1 ? s->nitems - 1 - k : perm[k];
      Fiber& f = s->fibers[i];
      if (f.done) continue;
      if (f.wait_gen && *f.wait_gen == f.wait_val) continue;
      f.wait_gen = nullptr;
      s->running = &f;
      cur = &f.ctx;
#ifdef EMU_TSAN
      if (!f.tsan) f.tsan = __tsan_create_fiber(0);
      s->main_tsan = __tsan_get_current_fiber();
      __tsan_switch_to_fiber(f.tsan, 0);
#endif
      emu_switch(&s->main_sp, f.sp);
      progressed = true;
      if (f.done) live--;
    }
    if (!progressed) {
      std::fprintf(stderr, "emu: DEADLOCK in workgroup (%u,%u,%u): %u work-items wait for lanes that never arrive "
                           "(a cross-lane operation or barrier inside divergent control flow?)\n",
                   s->fibers[0].ctx.bid.x, s->fibers[0].ctx.bid.y, s->fibers[0].ctx.bid.z, live);
      for (unsigned i = 0; i < s->nitems && i < 256; i++)
        if (!s->fibers[i].done) std::fprintf(stderr, "  item %u waits on %s\n", i, s->fibers[i].wait_gen == &s->group.gen ? "__syncthreads" : "its wavefront");
      std::abort();
    }
  }
}

}  // namespace

static void block_on(const unsigned long long* gen, unsigned long long val) {
  Fiber* f = g_sched->running;
  while (*gen == val) {
    f->wait_gen = gen;
    f->wait_val = val;
    to_scheduler();
  }
}

void wave_sync() {
  Wave* w = cur->wave;
  const unsigned long long my = w->gen;
  if (++w->arrived == w->live) {
    w->arrived = 0;
    w->ballot[(my + 1) & 1] = 0;       // every lane has read the result of collective my-1 before it arrived here
    w->gen = my + 1;
    return;
  }
  block_on(&w->gen, my);
}

void group_sync() {
  Group* g = cur->group;
  const unsigned long long my = g->gen;
  if (++g->arrived == g->live) { g->arrived = 0; g->gen = my + 1; return; }
  block_on(&g->gen, my);
}

unsigned long long ballot(bool p) {
  Wave* w = cur->wave;
  const unsigned slot = (unsigned)(w->gen & 1);
  if (p) w->ballot[slot] |= 1ull << cur->lane;
  wave_sync();
  return w->ballot[slot];
}

unsigned long long exchange(unsigned long long v, unsigned src) {
  Wave* w = cur->wave;
  const unsigned slot = (unsigned)(w->gen & 1);
  w->xch[slot][cur->lane] = v;
  wave_sync();
  return w->xch[slot][src & 63u];
}

unsigned long long first_lane(unsigned long long v) {
  Wave* w = cur->wave;
  const unsigned slot = (unsigned)(w->gen & 1);
  w->xch[slot][cur->lane] = v;
  const unsigned long long mask = w->live_mask;     // (a lane that leaves later cannot change who was first when the lanes met)
  wave_sync();
  return w->xch[slot][__builtin_ctzll(mask)];
}

void lds_objects(const void* a, size_t na, const void* b, size_t nb) {
  const uintptr_t lo = (uintptr_t)a < (uintptr_t)b ? (uintptr_t)a : (uintptr_t)b;
  const uintptr_t hi = (uintptr_t)a + na > (uintptr_t)b + nb ? (uintptr_t)a + na : (uintptr_t)b + nb;
  if (hi - lo + 16 >= 32768) {
    std::fprintf(stderr, "emu: the LDS objects of this kernel lie %zu bytes apart in the host image; their addresses do not fit 15 bits\n", (size_t)(hi - lo));
    std::abort();
  }
  lds_base = lo - 16;
}

void launch_impl(dim3 grid, dim3 block, Launch& l) {
  std::lock_guard<std::recursive_mutex> lock(g_launch_mutex);
  const unsigned n = block.x * block.y * block.z;
  if (n == 0 || n > kMaxItems) { std::fprintf(stderr, "emu: workgroup of %u work-items\n", n); std::abort(); }
  if (g_sched && g_sched->running) { std::fprintf(stderr, "emu: launch from inside a kernel\n"); std::abort(); }
  static thread_local Sched sched;
  Sched* s = &sched;
  if (!s->stacks) {
    void* m = mmap(nullptr, (size_t)kMaxItems * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) { std::perror("emu: mmap of the fiber stacks"); std::abort(); }
    s->stacks = static_cast<char*>(m);
    s->fibers.resize(kMaxItems);
  }
  g_sched = s;
  s->launch = &l;
  s->nitems = n;
  const unsigned nwaves = (n + 63) / 64;
  s->waves.assign(nwaves, Wave());
  static char anchor;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        lds_base = ((uintptr_t)&anchor) - (1u << 30);      // default: statics of the image are within 1 GiB above this
        for (unsigned w = 0; w < nwaves; w++) {
          Wave& wv = s->waves[w];
          wv = Wave();
          wv.live = n - w * 64 < 64 ? n - w * 64 : 64;
          wv.live_mask = wv.live == 64 ? ~0ull : ((1ull << wv.live) - 1);
        }
        s->group = Group();
        s->group.live = n;
        for (unsigned i = 0; i < n; i++) {
          Fiber& f = s->fibers[i];
          f.ctx.tid = Idx{i % block.x, (i / block.x) % block.y, i / (block.x * block.y)};
          f.ctx.bid = Idx{bx, by, bz};
          f.ctx.bdim = Idx{block.x, block.y, block.z};
          f.ctx.gdim = Idx{grid.x, grid.y, grid.z};
          f.ctx.lane = i & 63u;
          f.ctx.wave = &s->waves[i >> 6];
          f.ctx.group = &s->group;
          prepare(s, i);
        }
        run_group(s);
      }
  s->running = nullptr;
  cur = nullptr;
}

}  // namespace emu

// ---- runtime API -----------------------------------------------------------------------------------------------------------------
namespace {
std::mutex g_mem_mutex;
struct Block { size_t map_bytes; void* map; };
std::map<uintptr_t, Block> g_blocks;             // device + pinned allocations by start address
std::map<uintptr_t, size_t> g_pinned;            // pinned / registered host ranges
size_t page() { static const size_t p = (size_t)sysconf(_SC_PAGESIZE); return p; }

// the end of the allocation lies (up to 255 bytes of alignment slack) against an inaccessible page
void* guarded_alloc(size_t n) {
  const size_t pg = page();
  const size_t need = ((n ? n : 1) + 255) & ~(size_t)255;
  const size_t body = (need + pg - 1) / pg * pg;
  void* m = mmap(nullptr, body + pg, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return nullptr;
  mprotect(static_cast<char*>(m) + body, pg, PROT_NONE);
  char* start = static_cast<char*>(m) + body - need;
  // device memory is not zero when it is handed out: a kernel that reads what nobody has written must not get away with it here
  // (TM_EMU_POISON=0: the zero pages of the mapping, as before round 4)
  static const bool poison = [] { const char* e = getenv("TM_EMU_POISON"); return !e || atoi(e) != 0; }();
  if (poison) memset(start, 0xA5, need);
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  g_blocks[(uintptr_t)start] = Block{body + pg, m};
  return start;
}
bool guarded_free(void* p) {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  auto it = g_blocks.find((uintptr_t)p);
  if (it == g_blocks.end()) return false;
  munmap(it->second.map, it->second.map_bytes);
  g_blocks.erase(it);
  return true;
}
}  // namespace

struct emuStream { int id; };
struct emuEvent { std::chrono::steady_clock::time_point t; };

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceGetPCIBusId(char*, int, int) { return hipErrorInvalidDevice; }      // (no PCI device behind the emulated one: the library then leaves thread placement alone)
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }    // "compute units": few persistent workgroups
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated HIP runtime error"; }
hipError_t hipMalloc(void** p, size_t n) { *p = guarded_alloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { if (!p) return hipSuccess; return guarded_free(p) ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
  *p = guarded_alloc(n);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  g_pinned[(uintptr_t)*p] = n;
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  if (!p) return hipSuccess;
  { std::lock_guard<std::mutex> lk(g_mem_mutex); g_pinned.erase((uintptr_t)p); }
  return guarded_free(p) ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipHostRegister(void* p, size_t n, unsigned) { std::lock_guard<std::mutex> lk(g_mem_mutex); g_pinned[(uintptr_t)p] = n; return hipSuccess; }
hipError_t hipHostUnregister(void* p) { std::lock_guard<std::mutex> lk(g_mem_mutex); return g_pinned.erase((uintptr_t)p) ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  auto it = g_pinned.upper_bound((uintptr_t)p);
  if (it != g_pinned.begin()) {
    --it;
    if ((uintptr_t)p < it->first + it->second) { a->type = hipMemoryTypeHost; a->device = 0; a->devicePointer = a->hostPointer = const_cast<void*>(p); return hipSuccess; }
  }
  auto ib = g_blocks.upper_bound((uintptr_t)p);
  if (ib != g_blocks.begin()) {
    --ib;
    if ((uintptr_t)p < (uintptr_t)ib->second.map + ib->second.map_bytes) { a->type = hipMemoryTypeDevice; a->device = 0; a->devicePointer = const_cast<void*>(p); a->hostPointer = nullptr; return hipSuccess; }
  }
  return hipErrorInvalidValue;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) { if (n) std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyPeer(void* dst, int, const void* src, int, size_t n) { if (n) std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* dst, int, const void* src, int, size_t n, hipStream_t) { if (n) std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 0; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipErrorInvalidDevice; }
hipError_t hipMemset(void* p, int v, size_t n) { if (n) std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { if (n) std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n) { std::memcpy(sym, src, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n) { std::memcpy(dst, sym, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { static std::atomic<int> ids{0}; *s = new emuStream{++ids}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
