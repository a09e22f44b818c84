// tm_host.hip — the host-buffer entry points of the C ABI (include/tokenmonster_hip.h): what a Go caller binds.
//
// The reference fans documents out over goroutines, each calling Vocab.tokenize (training/tokenmonsterserver.go:363-378,
// :773-787), and is reentrant: many callers tokenize at once on one read-only Vocab.  Here every call borrows a LANE of the
// vocabulary: a HIP stream + a grow-only device workspace (tm_batch) + grow-only pinned staging.  Steady state does no
// hipMalloc / hipFree and never touches the NULL stream; concurrent callers (cgo calls run on distinct OS threads) take
// different lanes and overlap on the device.  tm_tokenize_pipeline is the large-batch form: the corpus is cut into chunks that
// run H2D | normalize + tokenize (+ serialize) | D2H on several lanes at once, so that PCIe moves chunk k+1 in and chunk k-1 out
// while chunk k computes — the host-to-host number of bench.py.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tm_pipeline.h"

#ifndef TM_EMU
// The HIP runtime spreads the streams of a process over GPU_MAX_HW_QUEUES hardware queues - four unless the environment says otherwise - and
// two streams that share one run their commands in the order they were submitted in: a copy stream's 32 MiB transfer in front of a compute
// stream's kernels holds those back for its 0.6 ms.  The ring (below) has four streams of its own beside the lanes' and the caller's; with
// four queues it ran 1 GiB in 36.2 ms, with eight in 28.1 (profiles/r06_h2h.txt; the lanes' form 34.3 -> 31.9 with sixteen).  So the library
// asks for eight when it is loaded - before the runtime reads the variable at its first call - unless the environment has set it already.
__attribute__((constructor)) static void tm_runtime_defaults() { (void)setenv("GPU_MAX_HW_QUEUES", "8", 0); }
#endif

namespace tmh {

struct Lane {
  hipStream_t stream = nullptr;
  tm_batch* ws = nullptr;
  uint8_t* h_stage = nullptr;     // pinned staging (pageable input of a one-shot call, pageable output), grow-only
  uint64_t h_cap = 0;
  uint8_t* h_stage_in = nullptr;  // pinned staging of the pipeline's pageable input (filled for the next chunk while h_stage still holds output)
  uint64_t h_in_cap = 0;
  hipStream_t up_stream = nullptr;   // tm_tokenize_pipeline: the next chunk's raw text is uploaded here while the current chunk computes
  hipEvent_t up_done = nullptr;
  uint8_t* d_bytes = nullptr;     // serialized ids on the device, grow-only
  uint64_t d_bytes_cap = 0;
  uint8_t* d_dec_a = nullptr;     // tm_decode_batch: ids, lengths, offsets (grow-only) ...
  uint64_t d_dec_a_cap = 0;
  uint8_t* d_dec_b = nullptr;     // ... and the decoded bytes, before and after capcode decoding (grow-only)
  uint64_t d_dec_b_cap = 0;
  bool busy = false;
};

// ---- the host-to-host ring (tm_tokenize_pipeline on page-locked buffers) ---------------------------------------------------------------------
// A slot = one chunk in flight: a workspace, the packed ids on the device, and a page-locked block for what goes to and comes from the host
// beside the bulk data (the chunk's verdict, raw offsets in, id offsets and missing counts out).
#if defined(__x86_64__) || defined(__i386__)
#define TM_CPU_RELAX() __builtin_ia32_pause()
#else
#define TM_CPU_RELAX() ((void)0)
#endif
struct RingSlot {
  tm_batch* ws = nullptr;
  uint8_t* d_bytes = nullptr;
  uint64_t d_bytes_cap = 0;
  uint8_t* h_pin = nullptr;
  uint64_t h_pin_cap = 0, docs_cap = 0;
  hipEvent_t up_done = nullptr, comp_done = nullptr, dl_done = nullptr;
  const uint8_t* ids_at = nullptr;   // where the packed ids of the chunk in the slot lie (d_bytes, or the workspace's id buffer)
  uint64_t* h_status() const { return reinterpret_cast<uint64_t*>(h_pin); }
  uint64_t* h_roff() const { return reinterpret_cast<uint64_t*>(h_pin + 64); }
  uint64_t* h_toff() const { return reinterpret_cast<uint64_t*>(h_pin + 64 + (docs_cap + 2) * 8); }
  uint32_t* h_missing() const { return reinterpret_cast<uint32_t*>(h_pin + 64 + 2 * (docs_cap + 2) * 8); }
};
struct Ring {
  hipStream_t up = nullptr, down = nullptr;
  std::vector<hipStream_t> comp;
  std::vector<RingSlot> slots;
  bool busy = false;                 // one call at a time drives the ring (LanePool::mu); a second caller takes the lanes
};

struct LanePool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Lane*> lanes;
  size_t max_lanes = 8;
  Ring ring;
};

static void lane_destroy(Lane* l) {
  if (!l) return;
  tm_batch_free(l->ws);
  if (l->stream) (void)hipStreamDestroy(l->stream);
  if (l->up_stream) (void)hipStreamDestroy(l->up_stream);
  if (l->up_done) (void)hipEventDestroy(l->up_done);
  (void)hipHostFree(l->h_stage);
  (void)hipHostFree(l->h_stage_in);
  (void)hipFree(l->d_bytes);
  (void)hipFree(l->d_dec_a);
  (void)hipFree(l->d_dec_b);
  delete l;
}

void pool_destroy(LanePool* p) {
  if (!p) return;
  for (Lane* l : p->lanes) lane_destroy(l);
  Ring& r = p->ring;
  for (RingSlot& s : r.slots) {
    tm_batch_free(s.ws);
    (void)hipFree(s.d_bytes);
    (void)hipHostFree(s.h_pin);
    for (hipEvent_t ev : {s.up_done, s.comp_done, s.dl_done}) if (ev) (void)hipEventDestroy(ev);
  }
  for (hipStream_t st : r.comp) if (st) (void)hipStreamDestroy(st);
  if (r.up) (void)hipStreamDestroy(r.up);
  if (r.down) (void)hipStreamDestroy(r.down);
  delete p;
}

static LanePool* pool_of(const tm_vocab* v) {
  static std::mutex create_mu;
  std::lock_guard<std::mutex> g(create_mu);
  if (!v->pool) {
    v->pool = new LanePool();
    if (const char* e = getenv("TM_LANES")) { int n = atoi(e); if (n >= 1 && n <= 64) v->pool->max_lanes = (size_t)n; }
  }
  return v->pool;
}

// borrow a lane (blocks while all max_lanes are busy); *out is returned with busy == true
static int lane_acquire(const tm_vocab* v, Lane** out) {
  int rc = enter_device(v);
  if (rc != TM_OK) return rc;
  LanePool* p = pool_of(v);
  std::unique_lock<std::mutex> lk(p->mu);
  for (;;) {
    for (Lane* l : p->lanes) if (!l->busy) { l->busy = true; *out = l; return TM_OK; }
    if (p->lanes.size() < p->max_lanes) {
      Lane* l = new Lane();
      hipError_t e = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking);
      if (e != hipSuccess) { delete l; return hip_fail(e, "hipStreamCreate (lane)"); }
      l->busy = true;
      p->lanes.push_back(l);
      *out = l;
      return TM_OK;
    }
    p->cv.wait(lk);
  }
}

// ---- the GPU's NUMA node ------------------------------------------------------------------------------------------------------------------
// A two-socket host has the GPU's PCIe root under ONE of its sockets.  Page-locked memory from hipHostMalloc already comes from the NUMA node
// nearest to the current device (the runtime's default; hipHostMallocNumaUser would switch that off), but the THREADS that feed a lane -
// command submission, the memcpy through pinned staging for pageable buffers, the waits - run wherever the scheduler put them, and from the
// far socket every doorbell and every completion crosses the inter-socket link: the box-to-box spread of the host-to-host rate in rounds
// 3 and 4 (27 - 34 GB/s for one library).  The workers of tm_tokenize_pipeline therefore run on the CPUs of the device's node for the length
// of the call (the calling thread gets its own mask back); TM_NUMA=0 in the environment switches that off.
struct NumaNear { int node = -1; cpu_set_t cpus; bool pin = false; };      // cpus: the node's own list, as the kernel gives it; pin: TM_NUMA has not switched the moving of threads off
static const NumaNear& numa_near(int device) {
  static std::mutex mu;
  static NumaNear table[64];
  static bool done[64] = {};
  std::lock_guard<std::mutex> g(mu);
  NumaNear& n = table[device < 0 || device >= 64 ? 0 : device];
  if (device < 0 || device >= 64 || done[device]) return n;
  done[device] = true;
  CPU_ZERO(&n.cpus);
  // (the node is read whatever TM_NUMA says: tm_device_numa_node reports it to callers that pin their own feeding threads; TM_NUMA=0 only
  // keeps THIS library from moving threads)
  const char* off = getenv("TM_NUMA");
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return n; }
  for (char* c = bdf; *c; c++) *c = (char)tolower((unsigned char)*c);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return n;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  if (node < 0) return n;                        // (a one-node host, or a kernel that does not say)
  n.node = node;
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  if (!(f = fopen(path, "r"))) return n;
  char list[4096] = {0};
  const bool got = fgets(list, sizeof list, f) != nullptr;
  fclose(f);
  if (!got) return n;
  int count = 0;
  for (char* p = list; *p && *p != '\n';) {       // "0-63,128-191"
    char* end = nullptr;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, &n.cpus); count++; }
    p = *end == ',' ? end + 1 : end;
  }
  n.pin = count > 0 && !(off && atoi(off) == 0);
  return n;
}
// the calling thread on the CPUs near `device` while the object lives - those of them the thread may use at all: the node's list is
// intersected with the thread's OWN mask at every call (a container's cpuset, a caller's pinning of this thread: what one thread was
// allowed at the first call says nothing about the next), and a thread that already runs on the node only (a caller's pinned feeder
// thread) is left exactly where it is
struct NearDevice {
  cpu_set_t before;
  bool changed = false;
  explicit NearDevice(int device) {
    const NumaNear& n = numa_near(device);
    if (!n.pin || sched_getaffinity(0, sizeof before, &before) != 0) return;
    cpu_set_t want;
    CPU_AND(&want, &n.cpus, &before);
    if (CPU_COUNT(&want) == 0 || CPU_EQUAL(&want, &before)) return;       // nothing of the node is allowed / the thread is on the node already
    changed = sched_setaffinity(0, sizeof want, &want) == 0;
  }
  ~NearDevice() { if (changed) (void)sched_setaffinity(0, sizeof before, &before); }
  NearDevice(const NearDevice&) = delete;
  NearDevice& operator=(const NearDevice&) = delete;
};

static void lane_release(const tm_vocab* v, Lane* l) {
  LanePool* p = v->pool;
  { std::lock_guard<std::mutex> g(p->mu); l->busy = false; }
  p->cv.notify_one();
}

// grow-only: a workspace that is too small is replaced by one with headroom, so that a server converges to zero allocations
static int lane_workspace(Lane* l, const tm_vocab* v, uint64_t bytes, uint32_t docs) {
  if (l->ws && l->ws->vocab == v && l->ws->max_bytes >= bytes && l->ws->max_docs >= docs) return TM_OK;
  uint64_t want_b = std::max<uint64_t>(bytes + bytes / 4 + (1u << 20), l->ws ? l->ws->max_bytes : 0);
  uint64_t want_d = std::max<uint64_t>((uint64_t)docs + docs / 4 + 64, l->ws ? l->ws->max_docs : 0);
  if (l->ws) trace_grow("lane workspace", want_b);
  tm_batch_free(l->ws);
  l->ws = nullptr;
  return tm_batch_create(v, want_b, (uint32_t)std::min<uint64_t>(want_d, 0xFFFFFFF0ull), &l->ws);
}

static int stage_grow(uint8_t** buf, uint64_t* cap, uint64_t bytes) {
  if (*cap >= bytes) return TM_OK;
  if (*cap) trace_grow("lane staging (pinned)", bytes);
  (void)hipHostFree(*buf);
  *buf = nullptr;
  *cap = bytes + bytes / 4 + 4096;
  hipError_t e = hipHostMalloc((void**)buf, *cap, hipHostMallocDefault);
  if (e != hipSuccess) { *cap = 0; return hip_fail(e, "hipHostMalloc (lane staging)"); }
  return TM_OK;
}
static int lane_stage(Lane* l, uint64_t bytes) { return stage_grow(&l->h_stage, &l->h_cap, bytes); }

static int lane_dbytes(Lane* l, uint64_t bytes) {
  if (l->d_bytes_cap >= bytes) return TM_OK;
  if (l->d_bytes_cap) trace_grow("lane serialized ids", bytes);
  (void)hipFree(l->d_bytes);
  l->d_bytes = nullptr;
  l->d_bytes_cap = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc((void**)&l->d_bytes, l->d_bytes_cap);
  if (e != hipSuccess) { l->d_bytes_cap = 0; return hip_fail(e, "hipMalloc (serialized ids)"); }
  return TM_OK;
}

static int dev_grow(uint8_t** buf, uint64_t* cap, uint64_t bytes, const char* what) {
  if (*cap >= bytes) return TM_OK;
  if (*cap) trace_grow(what, bytes);
  (void)hipFree(*buf);
  *buf = nullptr;
  *cap = bytes + bytes / 4 + 4096;
  hipError_t e = hipMalloc((void**)buf, *cap);
  if (e != hipSuccess) { *cap = 0; return hip_fail(e, what); }
  return TM_OK;
}

static bool is_pinned(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

struct RunOut { uint64_t total_tokens = 0; };

// upload + run one batch on a lane; the ids stay on the device (l->ws->d_out)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// raw text grows under capcode (about 1.1x; worst case every byte a capital: "DC x" = 4x)
static uint64_t lane_need(bool raw, uint64_t nbytes) { return raw ? nbytes + nbytes / 2 + 4096 : nbytes; }
// H2D of one batch into the lane's workspace on stream `st` (the workspace must be idle, or — raw text only — past its normalizer pass)
static int lane_upload(Lane* l, const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, bool raw, hipStream_t st) {
  const uint64_t nbytes = ndocs ? offsets[ndocs] : 0;
  int rc = lane_workspace(l, v, lane_need(raw, nbytes), ndocs);
  if (rc != TM_OK) return rc;
  return raw ? batch_upload_raw_on(l->ws, text, offsets, ndocs, st) : batch_upload_on(l->ws, text, offsets, ndocs, st);
}
// normalize (raw text) + the tokenizer pipeline on what lane_upload has put into the workspace; the ids stay on the device (l->ws->d_out)
static int lane_compute(Lane* l, const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, bool raw, bool emit, RunOut* ro,
                        const std::function<int()>& after_normalize = nullptr) {
  static const bool trace = getenv("TM_TRACE") != nullptr;
  const double t0 = trace ? now_ms() : 0;
  double t1 = 0, t2 = 0;
  const uint64_t nbytes = ndocs ? offsets[ndocs] : 0;
  int rc = TM_OK;
  tm_batch* b = l->ws;
  if (raw) {
    rc = tm_batch_normalize(b, l->stream);
    if (rc == TM_E_LIMIT) {      // capital-heavy text: retry once with the worst-case workspace
      if ((rc = lane_workspace(l, v, 4 * nbytes + 4096, ndocs)) != TM_OK) return rc;
      b = l->ws;
      if ((rc = batch_upload_raw_on(b, text, offsets, ndocs, l->stream)) != TM_OK) return rc;
      rc = tm_batch_normalize(b, l->stream);
    }
  }
  if (rc != TM_OK) return rc;
  if (after_normalize && (rc = after_normalize()) != TM_OK) return rc;     // (the raw text buffer is free from here on)
  if (trace) t1 = now_ms();
  if ((rc = run_pipeline(b, l->stream, false, nullptr, emit)) != TM_OK) return rc;
  if (trace) t2 = now_ms();
  if (emit) {
    if ((rc = ensure_output(b)) != TM_OK) return rc;
  } else {
    hipError_t e = hipStreamSynchronize(l->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  if (ro) {
    if (!emit) {                       // (ensure_output has read them already otherwise)
      if ((rc = small_d2h(b, b->last_totals, b->d_totals, sizeof b->last_totals, l->stream)) != TM_OK || (rc = small_sync(b, l->stream)) != TM_OK) return rc;
    }
    ro->total_tokens = ndocs ? b->last_totals[1] : 0;
  }
  if (trace) fprintf(stderr, "[lane %p] %llu bytes: %s %.2f ms, launch %.2f ms, wait %.2f ms\n", (void*)l, (unsigned long long)nbytes, raw ? "normalize" : "-", t1 - t0, t2 - t1, now_ms() - t2);
  return TM_OK;
}
static int lane_run(Lane* l, const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, bool raw, bool emit, RunOut* ro) {
  int rc = lane_upload(l, v, text, offsets, ndocs, raw, l->stream);
  return rc == TM_OK ? lane_compute(l, v, text, offsets, ndocs, raw, emit, ro) : rc;
}

static int d2h(void* dst, const void* src, uint64_t bytes, hipStream_t st, const char* what) {
  if (!bytes) return TM_OK;
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  return e == hipSuccess ? TM_OK : hip_fail(e, what);
}

}  // namespace tmh

using namespace tmh;

extern "C" {

int tm_device_numa_node(int device) {
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || device < 0 || device >= have) { (void)hipGetLastError(); return -1; }
  return numa_near(device).node;
}

void* tm_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void tm_host_free(void* p) { if (p) (void)hipHostFree(p); }
int tm_host_register(void* p, size_t bytes) {
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  return e == hipSuccess ? TM_OK : hip_fail(e, "hipHostRegister");
}
int tm_host_unregister(void* p) {
  hipError_t e = hipHostUnregister(p);
  return e == hipSuccess ? TM_OK : hip_fail(e, "hipHostUnregister");
}

int tm_tokenize_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t* tokens_out,
                      uint64_t tokens_cap, uint64_t* tok_offsets, uint32_t* missing) {
  if (!v || (ndocs && !offsets)) return set_error(TM_E_INVALID, "null argument");
  Lane* l = nullptr;
  int rc = lane_acquire(v, &l);
  if (rc != TM_OK) return rc;
  RunOut ro;
  rc = lane_run(l, v, text, offsets, ndocs, false, true, &ro);
  if (rc == TM_OK) {
    tm_batch* b = l->ws;
    if (tok_offsets) rc = small_d2h(b, tok_offsets, b->d_tok_offsets, ((uint64_t)ndocs + 1) * 8, l->stream);
    if (rc == TM_OK && missing && ndocs) rc = small_d2h(b, missing, b->d_doc_missing, (uint64_t)ndocs * 4, l->stream);
    if (rc == TM_OK && ro.total_tokens > tokens_cap) {
      (void)small_sync(b, l->stream);
      rc = set_error(TM_E_NOSPACE, "tokens_cap %llu < %llu required", (unsigned long long)tokens_cap, (unsigned long long)ro.total_tokens);
    }
    if (rc == TM_OK) rc = small_d2h(b, tokens_out, b->d_out, ro.total_tokens * 4, l->stream);
    if (rc == TM_OK) rc = small_sync(b, l->stream); else (void)small_sync(b, l->stream);
    if (ndocs == 0 && tok_offsets) tok_offsets[0] = 0;
  }
  lane_release(v, l);
  return rc;
}

static int count_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, bool raw, uint64_t* counts, uint32_t* missing) {
  if (!v || (ndocs && !offsets)) return set_error(TM_E_INVALID, "null argument");
  Lane* l = nullptr;
  int rc = lane_acquire(v, &l);
  if (rc != TM_OK) return rc;
  rc = lane_run(l, v, text, offsets, ndocs, raw, false, nullptr);
  if (rc == TM_OK && ndocs) {
    tm_batch* b = l->ws;
    std::vector<uint32_t> ev(ndocs);
    uint32_t err = 0;
    rc = small_d2h(b, &err, b->d_error, 4, l->stream);
    if (rc == TM_OK) rc = small_d2h(b, ev.data(), b->d_doc_events, (uint64_t)ndocs * 4, l->stream);
    if (rc == TM_OK && missing) rc = small_d2h(b, missing, b->d_doc_missing, (uint64_t)ndocs * 4, l->stream);
    if (rc == TM_OK) rc = small_sync(b, l->stream); else (void)small_sync(b, l->stream);
    if (rc == TM_OK && err) rc = error_from_flag(err);
    if (rc == TM_OK && counts) for (uint32_t d = 0; d < ndocs; d++) counts[d] = ev[d];
  }
  lane_release(v, l);
  return rc;
}

int tm_count_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint64_t* counts, uint32_t* missing) {
  return count_batch(v, text, offsets, ndocs, false, counts, missing);
}
int tm_count_batch_raw(const tm_vocab* v, const uint8_t* raw, const uint64_t* offsets, uint32_t ndocs, uint64_t* counts, uint32_t* missing) {
  return count_batch(v, raw, offsets, ndocs, true, counts, missing);
}

int tm_tokenize_batch_serialized(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t encoding_length,
                                 uint8_t* bytes_out, uint64_t bytes_cap, uint64_t* byte_offsets, uint32_t* missing, uint32_t* encoding_length_used) {
  if (!v || (ndocs && !offsets)) return set_error(TM_E_INVALID, "null argument");
  if (encoding_length <= 1) encoding_length = v->host.n_ids <= 65536 ? 2 : 3;          // go :990-996
  if (encoding_length < 2 || encoding_length > 4) return set_error(TM_E_INVALID, "Invalid encoding length");   // go :1012
  if (encoding_length_used) *encoding_length_used = encoding_length;
  Lane* l = nullptr;
  int rc = lane_acquire(v, &l);
  if (rc != TM_OK) return rc;
  RunOut ro;
  rc = lane_run(l, v, text, offsets, ndocs, false, true, &ro);
  if (rc == TM_OK) {
    tm_batch* b = l->ws;
    const uint64_t nb = ro.total_tokens * encoding_length;
    std::vector<uint64_t> offs((size_t)ndocs + 1, 0);
    if (ndocs) rc = small_d2h(b, offs.data(), b->d_tok_offsets, offs.size() * 8, l->stream);
    if (rc == TM_OK && missing && ndocs) rc = small_d2h(b, missing, b->d_doc_missing, (uint64_t)ndocs * 4, l->stream);
    if (rc == TM_OK && nb <= bytes_cap && nb) {
      if ((rc = lane_dbytes(l, nb)) == TM_OK) {
        launch_serialize(b->d_out, ro.total_tokens, encoding_length, l->d_bytes, l->stream);
        rc = small_d2h(b, bytes_out, l->d_bytes, nb, l->stream);
      }
    }
    { int rs = small_sync(b, l->stream); if (rc == TM_OK) rc = rs; }
    if (rc == TM_OK && byte_offsets) for (size_t d = 0; d <= ndocs; d++) byte_offsets[d] = offs[d] * encoding_length;
    if (rc == TM_OK && nb > bytes_cap) rc = set_error(TM_E_NOSPACE, "bytes_cap too small");
  }
  lane_release(v, l);
  return rc;
}

// ---- large batches, host to host -------------------------------------------------------------------------------------------------
int tm_tokenize_pipeline(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, int raw, uint32_t encoding_length,
                         uint64_t chunk_bytes, uint32_t lanes, uint8_t* bytes_out, uint64_t bytes_cap, uint64_t* byte_offsets, uint32_t* missing,
                         uint32_t* encoding_length_used, tm_pipeline_stats* stats) {
  return tmh::tokenize_pipeline_on(&v, 1, text, offsets, ndocs, raw, encoding_length, chunk_bytes, lanes, bytes_out, bytes_cap, byte_offsets, missing,
                                   encoding_length_used, stats);
}

}  // extern "C"
namespace tmh {
// ---- tm_tokenize_pipeline ------------------------------------------------------------------------------------------------------------------
// What both forms of the pipeline share: the call's arguments, the chunk list, and the chain of id counts - chunk k's ids go behind those of
// chunks 0..k-1 wherever and whenever they were computed.
struct PipeCall {
  const tm_vocab* const* vs; uint32_t nv;
  const uint8_t* text; const uint64_t* offsets; uint32_t ndocs; int raw; uint32_t enc;
  uint64_t chunk_bytes; uint32_t lanes;
  uint8_t* bytes_out; uint64_t bytes_cap; uint64_t* byte_offsets; uint32_t* missing; tm_pipeline_stats* stats;
  bool in_pinned = false, out_pinned = false;
  std::vector<uint32_t> first;     // first document of every chunk, + ndocs
  size_t nchunks = 0;
  std::vector<uint64_t> tok_base;  // ids before chunk k
  std::vector<char> known;
  std::mutex mu;
  std::condition_variable cv;
  int first_error = TM_OK;
  std::string first_msg;
  std::atomic<size_t> next{0};
  double t0 = 0;
  void fail(int rc) {
    { std::lock_guard<std::mutex> g(mu); if (first_error == TM_OK) { first_error = rc; first_msg = last_error(); } }
    cv.notify_all();
  }
  bool failed() { std::lock_guard<std::mutex> g(mu); return first_error != TM_OK; }
  // publish chunk k's count, learn where its ids go; false: another chunk has failed
  bool order(size_t k, uint64_t ntok, uint64_t* base) {
    { std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return known[k] || first_error != TM_OK; });
      if (first_error != TM_OK) return false;
      tok_base[k + 1] = tok_base[k] + ntok;
      known[k + 1] = 1;
      *base = tok_base[k]; }
    cv.notify_all();
    return true;
  }
};

// One chunk through the exact path on a borrowed lane, start to finish (the ring's way out for a chunk its one-pass form does not take; no
// prefetching, every wait in line)
static int chunk_exact(PipeCall& c, const tm_vocab* v, size_t k) {
  const uint32_t d0 = c.first[k], nd = c.first[k + 1] - d0;
  const uint64_t b0 = c.offsets[d0];
  Lane* l = nullptr;
  int rc = lane_acquire(v, &l);
  if (rc != TM_OK) return rc;
  std::vector<uint64_t> lo((size_t)nd + 1), toff((size_t)nd + 1);
  for (uint32_t d = 0; d <= nd; d++) lo[d] = c.offsets[d0 + d] - b0;
  const uint8_t* src = c.text + b0;
  do {
    if (!c.in_pinned) {
      if ((rc = stage_grow(&l->h_stage_in, &l->h_in_cap, lo[nd])) != TM_OK) break;
      std::memcpy(l->h_stage_in, src, lo[nd]);
      src = l->h_stage_in;
    }
    RunOut ro;
    if ((rc = lane_run(l, v, src, lo.data(), nd, c.raw != 0, true, &ro)) != TM_OK) break;
    uint64_t base = 0;
    if (!c.order(k, ro.total_tokens, &base)) break;
    tm_batch* b = l->ws;
    const uint64_t out_b = ro.total_tokens * c.enc;
    if ((rc = small_d2h(b, toff.data(), b->d_tok_offsets, toff.size() * 8, l->stream)) != TM_OK) break;
    if (c.missing && nd && (rc = small_d2h(b, c.missing + d0, b->d_doc_missing, (uint64_t)nd * 4, l->stream)) != TM_OK) break;
    const bool fits = (base + ro.total_tokens) * c.enc <= c.bytes_cap && c.bytes_out;
    if (fits && out_b) {
      if ((rc = lane_dbytes(l, out_b)) != TM_OK) break;
      launch_serialize(b->d_out, ro.total_tokens, c.enc, l->d_bytes, l->stream);
      if (c.out_pinned) rc = d2h(c.bytes_out + base * c.enc, l->d_bytes, out_b, l->stream, "D2H ids");
      else if ((rc = lane_stage(l, out_b)) == TM_OK) rc = d2h(l->h_stage, l->d_bytes, out_b, l->stream, "D2H ids");
      if (rc != TM_OK) break;
    }
    if ((rc = small_sync(b, l->stream)) != TM_OK) break;
    if (fits && out_b && !c.out_pinned) std::memcpy(c.bytes_out + base * c.enc, l->h_stage, out_b);
    for (uint32_t d = 1; d <= nd; d++) c.byte_offsets[d0 + d] = (base + toff[d]) * c.enc;
    if (c.stats) { std::lock_guard<std::mutex> g(c.mu); c.stats->host_fallback_docs += c.raw ? b->host_fallback_docs : 0; c.stats->normalized_bytes += b->nbytes; }
  } while (false);
  if (rc != TM_OK) (void)hipStreamSynchronize(l->stream);
  lane_release(v, l);
  return rc;
}

// ---- the ring --------------------------------------------------------------------------------------------------------------------------------
// Raw text in page-locked memory -> packed ids in page-locked memory, with NO host round trip inside a chunk.  The lanes' form below waits
// for the device three times per chunk (the normalizer's counts, the id count, the end of the download), and every wait drains the lane's
// stream: four lanes' streams on the runtime's four hardware queues kept the device 85 % busy at best (profiles/r06_h2h.txt).  Here ONE
// thread per device (the issuer) enqueues chunk after chunk - upload on the copy stream `up`, the normalizer pass and K0 .. K4 + the packing
// of the ids on one of two compute streams, alternating, so that the thin kernels of one chunk run beside the wide ones of the next - and never
// waits for the device: grids behind the normalizer pass are launched over a bound and the kernels look the counts up (tm_batch::d_ctl).
// A second thread (the finisher) waits for a chunk's end, reads its verdict and id count from the slot's page-locked block, takes the chunk's
// place in the output from the chain of counts, and enqueues the download on the copy stream `down`.  A chunk the one-pass form does not
// take (documents for the host normalizer, a long document, a piece whose margins could not tell ...) costs its kernels nothing behind the
// normalizer pass (ctl[0] == 0) and is run through the exact path by the finisher (chunk_exact).
static int ring_slot_size(RingSlot& s, const tm_vocab* v, uint64_t need_bytes, uint32_t need_docs, uint32_t enc) {
  hipError_t e;
  if (!s.ws || s.ws->vocab != v || s.ws->max_bytes < need_bytes || s.ws->max_docs < need_docs) {
    const uint64_t wb = std::max<uint64_t>(need_bytes + need_bytes / 8 + (1u << 20), s.ws ? s.ws->max_bytes : 0);
    const uint64_t wd = std::max<uint64_t>((uint64_t)need_docs + need_docs / 4 + 64, s.ws ? s.ws->max_docs : 0);
    if (s.ws) trace_grow("ring workspace", wb);
    tm_batch_free(s.ws);
    s.ws = nullptr;
    int rc = tm_batch_create(v, wb, (uint32_t)std::min<uint64_t>(wd, 0xFFFFFFF0ull), &s.ws);
    if (rc != TM_OK) return rc;
  }
  const uint64_t nb = s.ws->out_cap * 4;          // (room for the widest form: the slot outlives the call)
  (void)enc;
  if (s.d_bytes_cap < nb) {
    (void)hipFree(s.d_bytes);
    s.d_bytes = nullptr;
    if ((e = hipMalloc((void**)&s.d_bytes, nb)) != hipSuccess) { s.d_bytes_cap = 0; return hip_fail(e, "hipMalloc (ring ids)"); }
    s.d_bytes_cap = nb;
  }
  const uint64_t dc = s.ws->max_docs;
  const uint64_t hp = 64 + 2 * (dc + 2) * 8 + (dc + 2) * 4;
  if (s.h_pin_cap < hp || s.docs_cap != dc) {
    (void)hipHostFree(s.h_pin);
    s.h_pin = nullptr;
    if ((e = hipHostMalloc((void**)&s.h_pin, hp, hipHostMallocDefault)) != hipSuccess) { s.h_pin_cap = 0; return hip_fail(e, "hipHostMalloc (ring slot)"); }
    s.h_pin_cap = hp;
    s.docs_cap = dc;
  }
  for (hipEvent_t* ev : {&s.up_done, &s.comp_done, &s.dl_done})
    if (!*ev && (e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return hip_fail(e, "hipEventCreate (ring)");
  return TM_OK;
}

static int pipeline_ring(PipeCall& c, const std::vector<Ring*>& rings) {
  static const bool trace = getenv("TM_TRACE") != nullptr;
  static const uint32_t slack_pct = [] { const char* e = getenv("TM_RING_SLACK"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 100 && v <= 400 ? v : 125); }();
  // per device: a queue of issued chunks for the finisher, and the slots' states
  struct Dev {
    const tm_vocab* v; Ring* r;
    std::mutex mu; std::condition_variable cv;
    std::vector<std::pair<size_t, int>> queue;       // (chunk, slot) in issue order; slot -1: a chunk for the exact path as it is
    size_t qhead = 0;
    bool issuer_done = false;
    std::vector<char> slot_free;
  };
  std::vector<std::unique_ptr<Dev>> devs;
  for (uint32_t i = 0; i < c.nv; i++) { auto d = std::make_unique<Dev>(); d->v = c.vs[i]; d->r = rings[i]; d->slot_free.assign(rings[i]->slots.size(), 1); devs.push_back(std::move(d)); }

  auto issuer = [&](Dev& dv) {
    const tm_vocab* const v = dv.v;
    Ring& r = *dv.r;
    NearDevice near_gpu(v->device);
    int rc = enter_device(v);
    size_t issued = 0;
    while (rc == TM_OK && !c.failed()) {
      const size_t k = c.next.fetch_add(1);
      if (k >= c.nchunks) break;
      const uint32_t d0 = c.first[k], nd = c.first[k + 1] - d0;
      const uint64_t b0 = c.offsets[d0], nb = c.offsets[d0 + nd] - b0;
      int si = -1;
      if (nb > 0) {
        // a free slot (the finisher hands them back in order)
        std::unique_lock<std::mutex> lk(dv.mu);
        dv.cv.wait(lk, [&] { for (char f : dv.slot_free) if (f) return true; return false; });
        for (size_t j = 0; j < dv.slot_free.size(); j++) { const size_t q = (issued + j) % dv.slot_free.size(); if (dv.slot_free[q]) { si = (int)q; break; } }
        dv.slot_free[si] = 0;
      }
      if (si >= 0) {
        RingSlot& s = r.slots[si];
        tm_batch* b = s.ws;
        uint64_t* lo = s.h_roff();
        uint64_t npieces = 0;
        for (uint32_t d = 0; d <= nd; d++) lo[d] = c.offsets[d0 + d] - b0;
        const uint64_t grows = (v->host.norm_flag & 64u) ? 1u : 0u;       // (leadingspace: as batch_upload_raw_on counts them)
        for (uint32_t d = 0; d < nd; d++) npieces += (lo[d + 1] - lo[d] + grows + 1023) / 1024;
        hipStream_t cs = r.comp[issued % r.comp.size()];
        hipError_t e = hipSuccess;
        const double ti0 = trace ? now_ms() : 0;
        if ((rc = raw_prepare(b, nb, nd, npieces, cs)) != TM_OK) break;
        if ((e = hipMemcpyAsync(b->d_raw, c.text + b0, nb, hipMemcpyHostToDevice, r.up)) != hipSuccess ||
            (e = hipMemcpyAsync(b->d_raw_off, lo, ((uint64_t)nd + 1) * 8, hipMemcpyHostToDevice, r.up)) != hipSuccess ||
            (e = hipEventRecord(s.up_done, r.up)) != hipSuccess || (e = hipStreamWaitEvent(cs, s.up_done, 0)) != hipSuccess) { rc = hip_fail(e, "ring upload"); break; }
        const uint64_t seg_bound = (nb / 100 * slack_pct + 99) / SEG + nd + 1;
        s.h_status()[0] = ~0ull;
        if ((rc = ring_enqueue_normalize(b, cs, seg_bound, lo)) != TM_OK) break;
        if ((rc = ring_enqueue_tokenize(b, cs, c.enc, s.d_bytes, s.d_bytes_cap, s.h_status(), &s.ids_at)) != TM_OK) break;
        if ((e = hipEventRecord(s.comp_done, cs)) != hipSuccess) { rc = hip_fail(e, "hipEventRecord"); break; }
        if (trace) fprintf(stderr, "[ring] issue chunk %3zu (%5.1f MiB, %u docs) slot %d at %7.2f ms, %.3f ms of launches\n", k, nb / 1048576.0, nd, si, ti0 - c.t0, now_ms() - ti0);
        issued++;
      }
      { std::lock_guard<std::mutex> g(dv.mu); dv.queue.emplace_back(k, si); }
      dv.cv.notify_all();
    }
    if (rc != TM_OK) c.fail(rc);
    { std::lock_guard<std::mutex> g(dv.mu); dv.issuer_done = true; }
    dv.cv.notify_all();
  };

  auto finisher = [&](Dev& dv) {
    const tm_vocab* const v = dv.v;
    Ring& r = *dv.r;
    NearDevice near_gpu(v->device);
    int rc = enter_device(v);
    struct Pending { size_t k; int si; uint64_t base, ntok; };
    Pending prev{0, -1, 0, 0};
    // the downloads of a chunk are through: its offsets and counts to the caller, the slot back to the issuer
    auto complete = [&](const Pending& p) -> int {
      RingSlot& s = r.slots[p.si];
      hipError_t e = hipEventSynchronize(s.dl_done);
      if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize (ring download)");
      const uint32_t d0 = c.first[p.k], nd = c.first[p.k + 1] - d0;
      const uint64_t* toff = s.h_toff();
      for (uint32_t d = 1; d <= nd; d++) c.byte_offsets[d0 + d] = (p.base + toff[d]) * c.enc;
      if (c.missing && nd) std::memcpy(c.missing + d0, s.h_missing(), (size_t)nd * 4);
      if (c.stats) { std::lock_guard<std::mutex> g(c.mu); c.stats->normalized_bytes += s.h_status()[2]; }
      { std::lock_guard<std::mutex> g(dv.mu); dv.slot_free[p.si] = 1; }
      dv.cv.notify_all();
      if (trace) fprintf(stderr, "[ring] chunk %3zu complete at %7.2f ms\n", p.k, now_ms() - c.t0);
      return TM_OK;
    };
    for (;;) {
      std::pair<size_t, int> item;
      { std::unique_lock<std::mutex> lk(dv.mu);
        dv.cv.wait(lk, [&] { return dv.qhead < dv.queue.size() || dv.issuer_done; });
        if (dv.qhead >= dv.queue.size()) break;
        item = dv.queue[dv.qhead++]; }
      const size_t k = item.first;
      const int si = item.second;
      if (rc != TM_OK || c.failed()) {              // (drain: whatever is in flight ends before the call returns)
        if (si >= 0) (void)hipEventSynchronize(r.slots[si].comp_done);
        continue;
      }
      bool exact = si < 0;
      uint64_t ntok = 0;
      if (si >= 0) {
        RingSlot& s = r.slots[si];
        // The chunk's verdict is a word in page-locked memory that its last kernel writes (the issuer has set it to ~0): the finisher WATCHES it
        // instead of sleeping in hipEventSynchronize - on some boxes every other call lost 3.2 ms right there, in the wait for its first chunk
        // (gpurun_out/r06_probe25: every chunk of a 30.5 ms call lies 3.2 ms behind its place in a 26.9 ms call, from chunk 0 on).  The event
        // is still asked now and then (a stream that has failed must not be waited for for ever), and the download stream waits for it on the device.
        hipError_t e = hipSuccess;
        {
          const volatile uint64_t* flag = s.h_status();
          for (uint32_t spin = 1;; spin++) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != ~0ull) break;
            if ((spin & 0xFFFu) == 0) {
              const hipError_t q = hipEventQuery(s.comp_done);
              if (q == hipSuccess) { if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ~0ull) e = hipEventSynchronize(s.comp_done); break; }
              if (q != hipErrorNotReady) { e = q; break; }
            }
            TM_CPU_RELAX();
          }
        }
        if (e != hipSuccess) { rc = hip_fail(e, "ring chunk"); c.fail(rc); continue; }
        const uint64_t st = s.h_status()[0];
        if (st == ~0ull) { rc = set_error(TM_E_INTERNAL, "a chunk of the ring ended without its verdict"); c.fail(rc); continue; }
        if (trace) fprintf(stderr, "[ring] chunk %3zu computed at %7.2f ms: status %llu, %llu ids, %llu segments\n", k, now_ms() - c.t0, (unsigned long long)st,
                           (unsigned long long)s.h_status()[1], (unsigned long long)s.h_status()[3]);
        if (st != 0) exact = true; else ntok = s.h_status()[1];
      }
      if (exact) {
        if (si >= 0) { std::lock_guard<std::mutex> g(dv.mu); dv.slot_free[si] = 1; }
        if (si >= 0) dv.cv.notify_all();
        if (c.stats && si >= 0) { std::lock_guard<std::mutex> g(c.mu); c.stats->ring_exact_chunks++; }
        if ((rc = chunk_exact(c, v, k)) != TM_OK) c.fail(rc);
        continue;
      }
      RingSlot& s = r.slots[si];
      uint64_t base = 0;
      if (!c.order(k, ntok, &base)) { rc = TM_E_INTERNAL; continue; }
      const uint32_t nd = c.first[k + 1] - c.first[k];
      const uint64_t out_b = ntok * c.enc;
      const bool fits = (base + ntok) * c.enc <= c.bytes_cap && c.bytes_out;
      hipError_t e = hipStreamWaitEvent(r.down, s.comp_done, 0);
      if (e == hipSuccess && fits && out_b) e = hipMemcpyAsync(c.bytes_out + base * c.enc, s.ids_at, out_b, hipMemcpyDeviceToHost, r.down);
      if (e == hipSuccess) e = hipMemcpyAsync(s.h_toff(), s.ws->d_tok_offsets, ((uint64_t)nd + 1) * 8, hipMemcpyDeviceToHost, r.down);
      if (e == hipSuccess && c.missing) e = hipMemcpyAsync(s.h_missing(), s.ws->d_doc_missing, (uint64_t)nd * 4, hipMemcpyDeviceToHost, r.down);
      if (e == hipSuccess) e = hipEventRecord(s.dl_done, r.down);
      if (e != hipSuccess) { rc = hip_fail(e, "ring download"); c.fail(rc); continue; }
      if (prev.si >= 0 && (rc = complete(prev)) != TM_OK) { c.fail(rc); prev.si = -1; continue; }
      prev = Pending{k, si, base, ntok};
    }
    if (prev.si >= 0) { if (rc == TM_OK && !c.failed()) { if ((rc = complete(prev)) != TM_OK) c.fail(rc); } else (void)hipEventSynchronize(r.slots[prev.si].dl_done); }
    (void)hipStreamSynchronize(r.down);
  };

  std::vector<std::thread> th;
  for (uint32_t i = 0; i < c.nv; i++) {
    th.emplace_back(finisher, std::ref(*devs[i]));
    if (i > 0) th.emplace_back(issuer, std::ref(*devs[i]));
  }
  issuer(*devs[0]);
  for (auto& t : th) t.join();
  return c.first_error;
}

// borrow the rings of the replicas (all or none) and size their slots for this call; false: somebody else drives one of them
static int rings_acquire(PipeCall& c, uint32_t nslots, uint32_t nstreams, std::vector<Ring*>& rings, bool* got) {
  *got = false;
  uint64_t max_nb = 0; uint32_t max_nd = 0;
  for (size_t k = 0; k < c.nchunks; k++) {
    max_nb = std::max<uint64_t>(max_nb, c.offsets[c.first[k + 1]] - c.offsets[c.first[k]]);
    max_nd = std::max<uint32_t>(max_nd, c.first[k + 1] - c.first[k]);
  }
  for (uint32_t i = 0; i < c.nv; i++) {
    LanePool* p = pool_of(c.vs[i]);
    std::lock_guard<std::mutex> g(p->mu);
    if (p->ring.busy) { for (Ring* r : rings) r->busy = false; rings.clear(); return TM_OK; }      // (a ring's busy flag is only read under its pool's lock; clearing ours without it is safe: nobody else sets it while true)
    p->ring.busy = true;
    rings.push_back(&p->ring);
  }
  int rc = TM_OK;
  for (uint32_t i = 0; i < c.nv && rc == TM_OK; i++) {
    Ring& r = *rings[i];
    if ((rc = enter_device(c.vs[i])) != TM_OK) break;
    hipError_t e = hipSuccess;
    if (!r.up && (e = hipStreamCreateWithFlags(&r.up, hipStreamNonBlocking)) != hipSuccess) { rc = hip_fail(e, "hipStreamCreate (ring)"); break; }
    if (!r.down && (e = hipStreamCreateWithFlags(&r.down, hipStreamNonBlocking)) != hipSuccess) { rc = hip_fail(e, "hipStreamCreate (ring)"); break; }
    while (r.comp.size() < nstreams) {
      hipStream_t st = nullptr;
      if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) { rc = hip_fail(e, "hipStreamCreate (ring)"); break; }
      r.comp.push_back(st);
    }
    if (rc != TM_OK) break;
    if (r.slots.size() < nslots) r.slots.resize(nslots);
    for (RingSlot& s : r.slots) if ((rc = ring_slot_size(s, c.vs[i], lane_need(true, max_nb), max_nd, c.enc)) != TM_OK) break;
  }
  if (rc != TM_OK) { for (Ring* r : rings) r->busy = false; rings.clear(); return rc; }
  *got = true;
  return TM_OK;
}

static int pipeline_lanes(PipeCall& c);

// The pipeline over the lanes of ONE vocabulary or of its replicas on several devices (tm_tokenize_pipeline_multi, tm_multi.hip): worker w
// borrows a lane of replica w % nv, so the chunks — handed out from one counter — go to whichever lane of whichever device is free, and the
// ids of chunk k land behind those of chunks 0..k-1 wherever they were computed.  `lanes` = lanes per replica.
int tokenize_pipeline_on(const tm_vocab* const* vs, uint32_t nv, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, int raw, uint32_t encoding_length,
                         uint64_t chunk_bytes, uint32_t lanes, uint8_t* bytes_out, uint64_t bytes_cap, uint64_t* byte_offsets, uint32_t* missing,
                         uint32_t* encoding_length_used, tm_pipeline_stats* stats) {
  if (!vs || nv == 0 || !vs[0] || (ndocs && (!offsets || !text)) || !byte_offsets) return set_error(TM_E_INVALID, "null argument");
  const tm_vocab* const v = vs[0];          // (what the replicas have in common: id width)
  if (encoding_length <= 1) encoding_length = v->host.n_ids <= 65536 ? 2 : 3;
  if (encoding_length < 2 || encoding_length > 4) return set_error(TM_E_INVALID, "Invalid encoding length");
  if (encoding_length_used) *encoding_length_used = encoding_length;
  if (ndocs && offsets[0] != 0) return set_error(TM_E_INVALID, "offsets[0] must be 0");
  const bool chunk_default = chunk_bytes == 0;
  if (chunk_default) chunk_bytes = 32ull << 20;
  if (lanes == 0) lanes = 4;
  lanes = std::min<uint32_t>(lanes, 8);
  PipeCall c{vs, nv, text, offsets, ndocs, raw, encoding_length, chunk_bytes, lanes, bytes_out, bytes_cap, byte_offsets, missing, stats};
  c.in_pinned = is_pinned(text); c.out_pinned = is_pinned(bytes_out);
  // the ring: raw text, page-locked buffers on both sides, and a normalizer pass in the one-pass form
  static const int ring_env = [] { const char* e = getenv("TM_RING"); return e ? atoi(e) : 1; }();
  static const uint32_t ring_slots = [] { const char* e = getenv("TM_RING_SLOTS"); const int n = e ? atoi(e) : 0; return (uint32_t)(n >= 2 && n <= 16 ? n : 4); }();
  static const uint32_t ring_streams = [] { const char* e = getenv("TM_RING_STREAMS"); const int n = e ? atoi(e) : 0; return (uint32_t)(n >= 1 && n <= 4 ? n : 2); }();
  bool use_ring = ring_env != 0 && raw && c.in_pinned && c.out_pinned && ndocs > 0;
  for (uint32_t i = 0; i < nv && use_ring; i++) use_ring = ring_supported(vs[i]);
  // (the ring's chunks: 48 MiB unless the caller says otherwise - every launch of the match kernel has a ramp and a tail of its own, 1 GiB in
  // 27.0 ms against 27.6 with 32 MiB chunks and 28.9 with 24; beyond 48 the first results come later for nothing, profiles/r06_h2h.txt)
  if (use_ring && chunk_default) c.chunk_bytes = chunk_bytes = 48ull << 20;
  // chunks = maximal runs of whole documents of at most `limit` bytes (a longer document is a chunk of its own)
  std::vector<uint32_t>& first = c.first;
  for (uint32_t d = 0; d < ndocs; d++) if (offsets[d + 1] < offsets[d]) return set_error(TM_E_INVALID, "offsets not monotone at document %u", d);
  const double t_entry = now_ms();
  const size_t nramp = (size_t)lanes * nv;                          // the lanes' form: workers, if there are chunks enough
  // The steady state of the pipeline runs near the device-resident rate; what it loses it loses at the two ends.
  // Lanes' form: W workers that all begin with full chunks upload W chunks at once and then run their kernels in lock step, so the first W
  // chunks grow geometrically - chunk_bytes / 2^W ... chunk_bytes / 2 - and the last ones shrink by 1 / W of what is left each.
  // Ring: a chunk's kernels cannot begin before its upload has ended, and the link is only a fifth faster than the kernels (56 against 46 GB/s):
  // behind a chunk of c bytes the next may have c x RAMP / 100 (default 1.5: 2, 3, 4.5 ... MiB) or the device waits for it; the tail halves
  // (... 32, 16, 8, 4 MiB: what the end costs is the last chunk's kernels and download with nothing beside them).
  static const uint64_t ramp_pct = [] { const char* e = getenv("TM_RING_RAMP"); const int v = e ? atoi(e) : 0; return (uint64_t)(v >= 110 && v <= 400 ? v : 150); }();
  static const uint64_t ramp_first = [] { const char* e = getenv("TM_RING_FIRST_KIB"); const int v = e ? atoi(e) : 0; return (uint64_t)(v >= 64 && v <= (1 << 20) ? v : 2048) << 10; }();
  uint64_t ring_limit = ramp_first;
  for (uint32_t d = 0; d < ndocs;) {
    first.push_back(d);
    const size_t ci = first.size() - 1;                              // this chunk's number
    const uint64_t left = offsets[ndocs] - offsets[d];
    uint64_t limit = chunk_bytes;
    if (use_ring) {
      limit = std::min(chunk_bytes, ring_limit);
      ring_limit = std::min<uint64_t>(chunk_bytes, ring_limit / 100 * ramp_pct);
      if (left < 2 * (uint64_t)nv * chunk_bytes) limit = std::min(limit, std::max<uint64_t>(left / (2 * (uint64_t)nv), std::min<uint64_t>(chunk_bytes, 4u << 20)));
    } else {
      // (the shift count is clamped: 8 lanes on 8 devices - or on the virtual devices of a test - make nramp 64 and more)
      if (ci < nramp) limit = std::max<uint64_t>(chunk_bytes >> std::min<size_t>(nramp - ci, 63), std::min<uint64_t>(chunk_bytes, 1u << 20));
      // (the drain's shape - a floor of 2 / 4 / 8 / 16 MiB, a third or half of what is left instead of a quarter - was swept in round 5: every
      // setting 30 - 37 ms per call, the same as this one: profiles/r05_h2h.txt)
      if (left < (uint64_t)nramp * chunk_bytes) limit = std::min(limit, std::max<uint64_t>(left / nramp, std::min<uint64_t>(chunk_bytes, 2u << 20)));
    }
    // the last document that still ends within `limit` bytes of the chunk's first byte (at least one document)
    const uint64_t* const lo = offsets + d + 1;
    const uint64_t* const hi = std::upper_bound(lo, offsets + ndocs + 1, offsets[d] + limit);
    d = std::max<uint32_t>(d + 1, (uint32_t)(hi - offsets) - 1);
  }
  first.push_back(ndocs);
  c.nchunks = first.size() - 1;
  byte_offsets[0] = 0;
  if (stats) *stats = tm_pipeline_stats{};
  if (c.nchunks == 0) return TM_OK;
  c.tok_base.assign(c.nchunks + 1, 0);
  c.known.assign(c.nchunks + 1, 0);
  c.known[0] = 1;
  c.t0 = now_ms();
  { static const bool trace = getenv("TM_TRACE") != nullptr; if (trace) fprintf(stderr, "[pipe] %zu chunks laid out in %.3f ms (%s)\n", c.nchunks, c.t0 - t_entry, use_ring ? "ring" : "lanes"); }
  if (c.nchunks < 2) use_ring = false;          // (one chunk: the lanes' form is the shorter way)
  int rc = TM_OK;
  uint32_t workers = 0;
  if (use_ring) {
    std::vector<Ring*> rings;
    bool got = false;
    if ((rc = rings_acquire(c, ring_slots, ring_streams, rings, &got)) != TM_OK) return rc;
    if (got) {
      rc = pipeline_ring(c, rings);
      for (uint32_t i = 0; i < nv; i++) { LanePool* p = pool_of(vs[i]); std::lock_guard<std::mutex> g(p->mu); p->ring.busy = false; }
      workers = nv;
      if (stats) stats->ring = 1;
    } else use_ring = false;
  }
  if (!use_ring) { rc = pipeline_lanes(c); workers = (uint32_t)std::min<size_t>((size_t)lanes * nv, c.nchunks); }
  if (rc != TM_OK || c.first_error != TM_OK) return set_error(c.first_error != TM_OK ? c.first_error : rc, "%s", c.first_msg.c_str());
  if (stats) { stats->chunks = (uint32_t)c.nchunks; stats->lanes = workers; stats->input_pinned = c.in_pinned; stats->output_pinned = c.out_pinned; }
  if (byte_offsets[ndocs] > bytes_cap || !bytes_out) return set_error(TM_E_NOSPACE, "bytes_cap %llu < %llu required", (unsigned long long)bytes_cap, (unsigned long long)byte_offsets[ndocs]);
  return TM_OK;
}

// The lanes' form: every worker thread borrows a lane and takes chunk after chunk through it - upload, tm_batch_normalize (one wait), the
// tokenizer (one wait for the id count), the download (one wait).  For pageable buffers, already-normalized text, vocabularies whose
// normalizer is not the one-pass form, and callers that find the ring taken.
static int pipeline_lanes(PipeCall& c) {
  const tm_vocab* const* vs = c.vs; const uint32_t nv = c.nv; const uint8_t* text = c.text; const uint64_t* offsets = c.offsets; const int raw = c.raw;
  const uint32_t encoding_length = c.enc; uint8_t* bytes_out = c.bytes_out; const uint64_t bytes_cap = c.bytes_cap; uint64_t* byte_offsets = c.byte_offsets;
  uint32_t* missing = c.missing; tm_pipeline_stats* stats = c.stats;
  const std::vector<uint32_t>& first = c.first;
  const size_t nchunks = c.nchunks;
  const bool in_pinned = c.in_pinned, out_pinned = c.out_pinned;
  std::atomic<size_t>& next = c.next;
  const uint32_t nworkers = (uint32_t)std::min<size_t>((size_t)c.lanes * nv, nchunks);
  // A lane works on one chunk at a time, but the raw text of its NEXT chunk is uploaded (on the lane's second stream) as soon as the
  // normalizer pass of the current one is through with the raw buffer: the H2D of chunk k+1 hides behind the tokenizer kernels of chunk k.
  static const bool trace = getenv("TM_TRACE") != nullptr;
  const double t_pipe0 = c.t0;
  auto worker = [&](uint32_t wi) {
    const tm_vocab* const v = vs[wi % nv];
    NearDevice near_gpu(v->device);             // (worker 0 is the calling thread: it gets its affinity back when the call returns)
    Lane* l = nullptr;
    int rc = lane_acquire(v, &l);
    std::vector<uint64_t> loc, loc_next, toff;
    if (rc == TM_OK && !l->up_stream) {
      hipError_t e = hipStreamCreateWithFlags(&l->up_stream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&l->up_done, hipEventDisableTiming);
      if (e != hipSuccess) rc = hip_fail(e, "hipStreamCreate (lane upload)");
    }
    // chunk k: offsets relative to its first byte into `lo`; uploads it on `st` (through the pinned input staging when the caller's
    // buffer is pageable, so that the H2D runs at link speed); *srcp = where the text was read from (for a retry)
    auto upload = [&](size_t k, std::vector<uint64_t>& lo, hipStream_t st, const uint8_t** srcp) -> int {
      const uint32_t d0 = first[k], nd = first[k + 1] - d0;
      const uint64_t b0 = offsets[d0], nb = offsets[d0 + nd] - b0;
      lo.resize((size_t)nd + 1);
      for (uint32_t d = 0; d <= nd; d++) lo[d] = offsets[d0 + d] - b0;
      const uint8_t* src = text + b0;
      if (!in_pinned) {
        int r = stage_grow(&l->h_stage_in, &l->h_in_cap, nb);
        if (r != TM_OK) return r;
        std::memcpy(l->h_stage_in, src, nb);
        src = l->h_stage_in;
      }
      *srcp = src;
      return lane_upload(l, v, src, lo.data(), nd, raw != 0, st);
    };
    size_t k = rc == TM_OK ? next.fetch_add(1) : nchunks;
    bool prefetched = false;
    const uint8_t* src = nullptr;
    while (rc == TM_OK && k < nchunks) {
      if (c.failed()) break;
      const uint32_t d0 = first[k], d1 = first[k + 1], nd = d1 - d0;
      const double tr0 = trace ? now_ms() : 0;
      double tr1 = 0, tr2 = 0, tr3 = 0;
      if (prefetched) {
        hipError_t e = hipStreamWaitEvent(l->stream, l->up_done, 0);
        if (e != hipSuccess) { rc = hip_fail(e, "hipStreamWaitEvent"); break; }
      } else if ((rc = upload(k, loc, l->stream, &src)) != TM_OK) break;
      size_t k_next = nchunks;
      bool next_up = false;
      const uint8_t* src_next = nullptr;
      auto prefetch = [&]() -> int {
        k_next = next.fetch_add(1);
        if (k_next >= nchunks || !raw) return TM_OK;
        const uint32_t n0 = first[k_next], nn = first[k_next + 1] - n0;
        const uint64_t need = lane_need(true, offsets[n0 + nn] - offsets[n0]);
        // (a chunk that would replace the workspace, or the per-document arrays the current chunk still reads, is uploaded later)
        if (!l->ws || l->ws->max_bytes < need || l->ws->max_docs < nn || (uint64_t)nn + 2 > l->ws->raw_docs_cap) return TM_OK;
        loc_next.resize((size_t)nn + 1);
        for (uint32_t d = 0; d <= nn; d++) loc_next[d] = offsets[n0 + d] - offsets[n0];
        if (raw_upload_replaces_buffers(l->ws, loc_next.data(), nn)) return TM_OK;       // (the slabs and piece offsets of the current chunk are read by its match kernel)
        int r = upload(k_next, loc_next, l->up_stream, &src_next);
        if (r != TM_OK) return r;
        hipError_t e = hipEventRecord(l->up_done, l->up_stream);
        if (e != hipSuccess) return hip_fail(e, "hipEventRecord");
        next_up = true;
        return TM_OK;
      };
      RunOut ro;
      if (trace) tr1 = now_ms();
      if ((rc = lane_compute(l, v, src, loc.data(), nd, raw != 0, true, &ro, prefetch)) != TM_OK) break;
      if (trace) tr2 = now_ms();
      if (k_next == nchunks && !next_up) k_next = next.fetch_add(1);          // (already-normalized input: nothing was prefetched)
      uint64_t base = 0;
      if (!c.order(k, ro.total_tokens, &base)) break;          // publish this chunk's count, learn where its ids go
      if (trace) tr3 = now_ms();
      tm_batch* b = l->ws;
      const uint64_t out_b = ro.total_tokens * encoding_length;
      toff.resize((size_t)nd + 1);
      if ((rc = small_d2h(b, toff.data(), b->d_tok_offsets, toff.size() * 8, l->stream)) != TM_OK) break;
      if (missing && nd && (rc = small_d2h(b, missing + d0, b->d_doc_missing, (uint64_t)nd * 4, l->stream)) != TM_OK) break;
      const bool fits = (base + ro.total_tokens) * encoding_length <= bytes_cap && bytes_out;
      if (fits && out_b) {
        uint8_t* dst = bytes_out + base * encoding_length;
        // (Writing the ids straight into the caller's page-locked buffer from the serialize kernel - no device staging, no copy command - was
        // built and timed in round 5: 42 - 45 ms per GiB call against 32 - 36 with the copy engine, profiles/r05_h2h.txt: stores of a kernel to
        // fine-grained host memory cross PCIe far below the engine's rate.  Removed.)
        if ((rc = lane_dbytes(l, out_b)) != TM_OK) break;
        launch_serialize(b->d_out, ro.total_tokens, encoding_length, l->d_bytes, l->stream);
        if (out_pinned) rc = d2h(dst, l->d_bytes, out_b, l->stream, "D2H ids");
        else {
          if ((rc = lane_stage(l, out_b)) != TM_OK) break;
          rc = d2h(l->h_stage, l->d_bytes, out_b, l->stream, "D2H ids");
        }
        if (rc != TM_OK) break;
      }
      if ((rc = small_sync(b, l->stream)) != TM_OK) break;
      if (fits && out_b && !out_pinned) std::memcpy(bytes_out + base * encoding_length, l->h_stage, out_b);
      if (trace) fprintf(stderr, "[pipe] worker %u chunk %3zu (%5.1f MiB): start %7.2f  upload/wait %5.2f  compute %5.2f  order-wait %5.2f  download %5.2f  -> end %7.2f ms\n", wi, k,
                         (offsets[d1] - offsets[d0]) / 1048576.0, tr0 - t_pipe0, tr1 - tr0, tr2 - tr1, tr3 - tr2, now_ms() - tr3, now_ms() - t_pipe0);
      for (uint32_t d = 1; d <= nd; d++) byte_offsets[d0 + d] = (base + toff[d]) * encoding_length;
      if (stats) { std::lock_guard<std::mutex> g(c.mu); stats->host_fallback_docs += raw ? b->host_fallback_docs : 0; stats->normalized_bytes += b->nbytes; }
      k = k_next;
      prefetched = next_up;
      if (next_up) { loc.swap(loc_next); src = src_next; }
    }
    if (l && l->up_stream) (void)hipStreamSynchronize(l->up_stream);          // (an error may leave an upload in flight)
    if (rc != TM_OK) c.fail(rc);
    if (l) lane_release(v, l);
  };
  std::vector<std::thread> th;
  for (uint32_t t = 1; t < nworkers; t++) th.emplace_back(worker, t);
  worker(0);
  for (auto& t : th) t.join();
  return c.first_error;
}
}  // namespace tmh
extern "C" {

// Decode / decode_raw (go/tokenmonster.go:445-550; tokenmonster.cpp:1404-1425) on a lane like the tokenize entry points: the lane's
// stream, grow-only device arenas and pinned staging — steady state allocates nothing and stays off the NULL stream, so decode jobs of a
// server (tokenmonsterserver jobs 2-9) do not stall the tokenize jobs running beside them (a hipFree synchronizes the whole device).
static thread_local uint32_t g_decode_host_docs = 0;
uint32_t tm_decode_host_docs(void) { return g_decode_host_docs; }

int tm_decode_batch(const tm_vocab* v, const uint32_t* tokens, const uint64_t* tok_offsets, uint32_t ndocs, int raw,
                    uint8_t* out, uint64_t out_cap, uint64_t* out_offsets) {
  g_decode_host_docs = 0;
  if (!v || !tok_offsets || !out_offsets) return set_error(TM_E_INVALID, "null argument");
  const uint64_t n = tok_offsets[ndocs];
  if (n && !tokens) return set_error(TM_E_INVALID, "null argument");
  if (tok_offsets[0] != 0) return set_error(TM_E_INVALID, "tok_offsets[0] must be 0");
  for (uint32_t d = 0; d < ndocs; d++) if (tok_offsets[d + 1] < tok_offsets[d]) return set_error(TM_E_INVALID, "tok_offsets not monotone");
  Lane* l = nullptr;
  int rc = lane_acquire(v, &l);
  if (rc != TM_OK) return rc;
  hipStream_t st = l->stream;
  hipError_t e = hipSuccess;
  const bool dev_capcode = !raw && v->host.capcode == 2 && v->host.charset == 1 && ndocs > 0;
  // arena A: ids | document offsets | the decode's own words (dec_arena: tiles of ids, byte offset / decoded length of every document)
  auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
  const uint64_t o_tok = 0, o_toff = o_tok + up((n + 1) * 4);
  const DecArena a = dec_arena(o_toff + up(((uint64_t)ndocs + 1) * 8), n, ndocs);
  const uint64_t a_bytes = a.bytes;
  std::vector<uint64_t> doff((size_t)ndocs + 1, 0), declen;
  uint64_t total = 0;
  bool need_raw = !dev_capcode;
  const uint8_t *h_dec = nullptr, *h_raw = nullptr;          // decoded / raw bytes in the lane's pinned staging
  do {
    if ((rc = dev_grow(&l->d_dec_a, &l->d_dec_a_cap, a_bytes, "hipMalloc (decode)")) != TM_OK) break;
    uint8_t* A = l->d_dec_a;
    uint32_t* d_tok = (uint32_t*)(A + o_tok); uint64_t* d_toff = (uint64_t*)(A + o_toff);
    uint64_t* d_total = (uint64_t*)(A + a.o_total); uint64_t* d_doff = (uint64_t*)(A + a.o_doff); uint64_t* d_declen = (uint64_t*)(A + a.o_declen);
    // ids and offsets through pinned staging (the caller's buffers are pageable Go / Python memory)
    const uint64_t in_bytes = n * 4 + ((uint64_t)ndocs + 1) * 8;
    if ((rc = stage_grow(&l->h_stage_in, &l->h_in_cap, in_bytes)) != TM_OK) break;
    std::memcpy(l->h_stage_in, tok_offsets, ((uint64_t)ndocs + 1) * 8);
    if (n) std::memcpy(l->h_stage_in + ((uint64_t)ndocs + 1) * 8, tokens, n * 4);
    if ((e = hipMemcpyAsync(d_toff, l->h_stage_in, ((uint64_t)ndocs + 1) * 8, hipMemcpyHostToDevice, st)) != hipSuccess ||
        (n && (e = hipMemcpyAsync(d_tok, l->h_stage_in + ((uint64_t)ndocs + 1) * 8, n * 4, hipMemcpyHostToDevice, st)) != hipSuccess)) { rc = hip_fail(e, "H2D tokens"); break; }
    launch_decode_lengths(v, d_tok, n, d_toff, ndocs, a, A, st);
    if ((rc = lane_stage(l, 8)) != TM_OK) break;
    if ((rc = d2h(l->h_stage, d_total, 8, st, "decode lengths")) != TM_OK) break;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize"); break; }
    std::memcpy(&total, l->h_stage, 8);
    // arena B: the gathered bytes | the same after capcode decoding (out of place: the host decoder needs the others as they were)
    const uint64_t o_dec = up(total + 16);
    if ((rc = dev_grow(&l->d_dec_b, &l->d_dec_b_cap, o_dec + up(total + 16), "hipMalloc (decode output)")) != TM_OK) break;
    uint8_t* d_out = l->d_dec_b; uint8_t* d_dec = l->d_dec_b + o_dec;
    launch_decode_copy(v, d_tok, n, d_toff, ndocs, a, A, d_out, st);
    if (dev_capcode) {
      if ((rc = launch_decode_capcode(v, d_out, d_doff, ndocs, d_dec, d_declen, d_total + 8, st)) != TM_OK) break;
      declen.resize(ndocs);
    }
    // staging: the documents' byte offsets (they come out of the gather) | decoded lengths | decoded text | gathered bytes
    const uint64_t s_doff = 0, s_declen = up(((uint64_t)ndocs + 1) * 8), s_text = s_declen + up((uint64_t)ndocs * 8 + 8), s_raw = s_text + up(total + 16);
    if ((rc = lane_stage(l, s_raw + up(total + 16))) != TM_OK) break;
    if ((rc = d2h(l->h_stage + s_doff, d_doff, ((uint64_t)ndocs + 1) * 8, st, "decode offsets")) != TM_OK) break;
    if (dev_capcode) {
      uint8_t* hp = l->h_stage + s_text;
      if ((rc = d2h(l->h_stage + s_declen, d_declen, (uint64_t)ndocs * 8, st, "D2H decoded text")) != TM_OK || (rc = d2h(hp, d_dec, total, st, "D2H decoded text")) != TM_OK) break;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize"); break; }
      std::memcpy(declen.data(), l->h_stage + s_declen, (uint64_t)ndocs * 8);
      h_dec = hp;
      for (uint32_t d = 0; d < ndocs && !need_raw; d++) need_raw = declen[d] == DEC_HOST;
      if (need_raw) {
        uint8_t* hr = l->h_stage + s_raw;
        if ((rc = d2h(hr, d_out, total, st, "D2H decoded bytes")) != TM_OK) break;
        if ((e = hipStreamSynchronize(st)) != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize"); break; }
        h_raw = hr;
      }
    } else {
      if ((rc = d2h(l->h_stage + s_raw, d_out, total, st, "D2H decoded bytes")) != TM_OK) break;
      if ((e = hipStreamSynchronize(st)) != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize"); break; }
      h_raw = l->h_stage + s_raw;
    }
    std::memcpy(doff.data(), l->h_stage + s_doff, doff.size() * 8);
  } while (false);
  if (rc != TM_OK) (void)hipStreamSynchronize(st);      // nothing of this call may still be in flight when the lane goes back
  if (rc == TM_OK) {
    if (raw || v->host.capcode == 0) {
      std::memcpy(out_offsets, doff.data(), doff.size() * 8);
      if (total > out_cap) rc = set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)total);
      else if (total) std::memcpy(out, h_raw, total);
    } else {
      // the documents the device left alone (anything beyond ASCII; every document of a capcode-1 or UTF-16 vocabulary) go through the host decoder
      std::vector<uint32_t> todo;
      for (uint32_t d = 0; d < ndocs; d++) if (!dev_capcode || declen[d] == DEC_HOST) todo.push_back(d);
      g_decode_host_docs = (uint32_t)todo.size();
      std::vector<std::vector<uint8_t>> touts;
      if (!todo.empty()) {
        std::vector<uint64_t> toff(todo.size() + 1, 0);
        std::vector<uint8_t> tbytes;
        for (size_t k = 0; k < todo.size(); k++) {
          tbytes.insert(tbytes.end(), h_raw + doff[todo[k]], h_raw + doff[todo[k] + 1]);
          toff[k + 1] = tbytes.size();
        }
        capcode_decode_batch(tbytes.data(), toff.data(), (uint32_t)todo.size(), v->host.capcode, 0, touts);
      }
      std::vector<uint64_t> hostlen((size_t)ndocs, DEC_HOST);
      for (size_t k = 0; k < todo.size(); k++) hostlen[todo[k]] = k;
      uint64_t o = 0;
      for (uint32_t d = 0; d < ndocs; d++) { out_offsets[d] = o; o += hostlen[d] != DEC_HOST ? touts[hostlen[d]].size() : declen[d]; }
      out_offsets[ndocs] = o;
      if (o > out_cap) rc = set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)o);
      else for (uint32_t d = 0; d < ndocs; d++) {
        if (hostlen[d] != DEC_HOST) { const auto& t = touts[hostlen[d]]; if (!t.empty()) std::memcpy(out + out_offsets[d], t.data(), t.size()); }
        else if (declen[d]) std::memcpy(out + out_offsets[d], h_dec + doff[d], declen[d]);
      }
    }
  }
  lane_release(v, l);
  return rc;
}

}  // extern "C"
