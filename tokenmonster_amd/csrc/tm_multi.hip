// tm_multi.hip — several GPUs of one node behind ONE handle, driven from inside the library (include/tokenmonster_hip.h, "several devices").
//
// The reference's parallelism lives inside the host process: trainvocab launches `workers` goroutines over one dataset
// (training/trainvocab.go:1827-1829; after "midway" each walks the dataset as ONE strip, :909-922) and the server fans the documents
// of a job out over goroutines (training/tokenmonsterserver.go:363-378, :773-787).  A Go host that binds this library stays ONE process,
// so the multi-GPU half of the path has to be reachable through the C ABI as well: a tm_devices handle names the GPUs, a
// tm_vocab_set holds one replica of a vocabulary's device block per GPU, and
//   tm_tokenize_pipeline_multi   hands chunks of whole documents to lanes of EVERY device (no collective: documents are independent),
//   tm_score_multi               scores ONE whole-buffer walk cut into one byte range per device (halo of 128 bytes, 80 exit states per
//                                range chained on the host, tm_score_begin / tm_score_finish) and merges the histograms with ONE
//                                ncclAllReduce(sum, uint32) over xGMI — the only collective of the path.
// RCCL is bound at run time (dlopen of librccl.so.1 at the first collective): the library is 570 MB, and the single-GPU entry points —
// everything else in this library — must not pay for mapping it.  Members of a tm_devices that share a physical device ("virtual
// devices": the way these code paths are tested on a one-GPU box) cannot form an RCCL communicator (RCCL refuses two
// ranks on one device); there, and wherever librccl cannot be loaded, member 0 sums the histograms itself (peer copy + add kernel).
#include <hip/hip_runtime.h>
#ifndef TM_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tm_pipeline.h"
#include "tm_build.h"

using namespace tmh;

namespace tmh {

#ifndef TM_EMU
// the handful of RCCL entry points the path needs, resolved once
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
static Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // RCCL has to come from the SAME ROCm installation as the HIP runtime this process runs on: a process may hold a second one (a Python
    // host that has imported torch carries torch's own librccl / libhsa-runtime64 under the same sonames), and a librccl that talks to
    // the other installation's HSA runtime finds it uninitialised ("no ROCm-capable device is detected").  So: first the librccl that lies
    // beside the libamdhip64 in use (by path: a soname look-up would hand back whichever copy was loaded first), then the usual names.
    std::vector<std::string> names;
    Dl_info di;
    if (dladdr((const void*)&hipGetDeviceCount, &di) && di.dli_fname) {
      std::string dir(di.dli_fname);
      const size_t cut = dir.rfind('/');
      if (cut != std::string::npos) { dir.resize(cut + 1); names.push_back(dir + "librccl.so.1"); names.push_back(dir + "librccl.so"); }
    }
    for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(n);
    for (const std::string& name : names) if ((r.so = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.so) return;
    auto sym = [&](const char* n) { return dlsym(r.so, n); };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString;
  });
  return r;
}
#endif

// the calling thread's current device, put back when an entry point that had to visit the members' devices returns: the one-device entry
// points (tm_vocab_load, tm_dataset_upload, ...) use the current device implicitly, and a caller must find it where it left it
struct DeviceKeeper {
  int dev = -1;
  DeviceKeeper() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
  ~DeviceKeeper() { if (dev >= 0) (void)hipSetDevice(dev); }
  DeviceKeeper(const DeviceKeeper&) = delete;
  DeviceKeeper& operator=(const DeviceKeeper&) = delete;
};

// The entry state of member `member`'s byte range from the exit maps of all ranges (ENT bytes each, in member order): the chain the host used
// to walk between tm_score_begin and tm_score_finish, on the device (round 6) - a member's stream goes from its match kernel through the
// all-gather of the maps and this kernel straight into its histogram walk, no trip to the host in between.  error bit 2: a range that
// cannot be entered in the state the walk reaches it in.
__global__ void k_chain_entry(const uint8_t* __restrict__ all_exits, int member, uint8_t* __restrict__ entry_out, uint32_t* __restrict__ error_word) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t e = 0;
  for (int k = 0; k < member; k++) {
    e = all_exits[(size_t)k * ENT + e];
    if (e >= (uint32_t)ENT) { atomicOr(error_word, 4u); e = 0; break; }
  }
  entry_out[0] = (uint8_t)e;
}

__global__ void k_hist_add(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

}  // namespace tmh

struct tm_devices {
  std::vector<int> dev;                 // HIP device of every member (the same device may appear more than once: virtual devices)
  bool distinct = true;                 // no device appears twice: the members can form an RCCL communicator
  std::vector<hipStream_t> stream;      // one stream per member, for the scoring passes
  std::vector<hipEvent_t> done;         // per member: its half of a pass has been enqueued up to here
  std::vector<hipEvent_t> begun;        // per member: its match kernel and exit map have been enqueued (the no-RCCL gather of the maps waits for these)
  std::vector<uint8_t*> d_all_exits;    // per member, on its device: the exit maps of all members' ranges (ENT bytes each)
#ifndef TM_EMU
  std::vector<ncclComm_t> comm;         // created at the first collective (rccl_ready)
#endif
  int rccl_state = 0;                   // 0 not tried, 1 communicator ready, -1 not available (virtual devices, TM_RCCL=0, librccl missing)
  int rccl_ranks = 0;
  std::string rccl_note;                // why RCCL is not used, for tm_devices_rccl_ranks' caller
  uint32_t* d_scratch = nullptr;        // member 0: a peer's histogram on its way into the sum (no-RCCL path)
  uint64_t scratch_words = 0;
  std::mutex mu;                        // one collective pass at a time per handle
};

struct tm_vocab_set {
  tm_devices* devs = nullptr;
  std::vector<tm_vocab*> v;             // v[0] was loaded from the file bytes (it has the host tables); the others are imported replicas of its device block
};

struct tm_dataset_set {
  tm_devices* devs = nullptr;
  std::vector<tm_dataset*> part;        // per member: its byte range followed by the halo (an empty dataset for a member without bytes: its pass is the identity)
  std::vector<uint64_t> own;            // bytes of the range
  std::vector<int> continues;           // tm_score_begin's `continues` of the range
  uint64_t n = 0;
};

namespace tmh {

static int open_list(const int* devices, int n, tm_devices** out) {
  if (!out || n < 1 || !devices) return set_error(TM_E_INVALID, "tm_devices: empty device list");
  *out = nullptr;
  int have = 0;
  hipError_t e = hipGetDeviceCount(&have);
  if (e != hipSuccess || have < 1) return set_error(TM_E_NODEVICE, "no HIP device visible");
  DeviceKeeper keep;
  auto* g = new tm_devices();
  for (int i = 0; i < n; i++) {
    if (devices[i] < 0 || devices[i] >= have) { delete g; return set_error(TM_E_INVALID, "device %d of %d visible", devices[i], have); }
    for (int d : g->dev) if (d == devices[i]) g->distinct = false;
    g->dev.push_back(devices[i]);
  }
  g->stream.assign(n, nullptr);
  g->done.assign(n, nullptr);
  g->begun.assign(n, nullptr);
  g->d_all_exits.assign(n, nullptr);
  for (int i = 0; i < n; i++) {
    if ((e = hipSetDevice(g->dev[i])) != hipSuccess || (e = hipStreamCreateWithFlags(&g->stream[i], hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming)) != hipSuccess || (e = hipEventCreateWithFlags(&g->begun[i], hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc((void**)&g->d_all_exits[i], (size_t)n * ENT + 16)) != hipSuccess) {
      tm_devices_close(g);
      return hip_fail(e, "tm_devices: stream of a member");
    }
  }
  // peer access among the physical devices, where the node offers it: the replication of a vocabulary block and the no-RCCL sum then run over xGMI
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      int can = 0;
      if (g->dev[i] == g->dev[j] || hipDeviceCanAccessPeer(&can, g->dev[i], g->dev[j]) != hipSuccess || !can) continue;
      (void)hipSetDevice(g->dev[i]);
      const hipError_t pe = hipDeviceEnablePeerAccess(g->dev[j], 0);
      if (pe != hipSuccess) (void)hipGetLastError();        // (already enabled: fine)
    }
  *out = g;
  return TM_OK;
}

// the communicator of the handle's devices, made at the first collective; false: sum without RCCL
static bool rccl_ready(tm_devices* g) {
  if (g->rccl_state != 0) return g->rccl_state > 0;
  g->rccl_state = -1;
#ifdef TM_EMU
  g->rccl_note = "emulated device";
  return false;
#else
  const char* env = getenv("TM_RCCL");
  const int n = (int)g->dev.size();
  if (env && atoi(env) == 0) { g->rccl_note = "TM_RCCL=0"; return false; }
  if (!g->distinct) { g->rccl_note = "members share a device (RCCL refuses two ranks on one device)"; return false; }
  if (n < 2 && !(env && atoi(env) > 0)) { g->rccl_note = "one member: nothing to reduce (TM_RCCL=1 runs the collective anyway)"; return false; }
  Rccl& r = rccl();
  if (!r.ok) { g->rccl_note = "librccl.so.1 could not be loaded"; return false; }
  DeviceKeeper keep;                      // (ncclCommInitAll leaves the thread on a device of its choosing)
  g->comm.assign(n, nullptr);
  // (RCCL looks at the runtime's sticky "last error" between its own calls: one left behind by an unrelated earlier call - a probe of a
  // pageable pointer, say - would be reported as RCCL's failure)
  for (int i = 0; i < n; i++) { (void)hipSetDevice(g->dev[i]); (void)hipGetLastError(); }
  const ncclResult_t rc = r.CommInitAll(g->comm.data(), n, g->dev.data());
  if (rc != ncclSuccess) { g->rccl_note = std::string("ncclCommInitAll: ") + r.GetErrorString(rc); g->comm.clear(); return false; }
  g->rccl_ranks = n;
  if (r.CommCount) { int c = 0; if (r.CommCount(g->comm[0], &c) == ncclSuccess) g->rccl_ranks = c; }
  g->rccl_state = 1;
  return true;
#endif
}

// `work(member)` on one host thread per member (member 0 on the calling thread), each with its device current; the first failure wins
static int on_members(const tm_devices* g, const std::function<int(int)>& work) {
  const int n = (int)g->dev.size();
  DeviceKeeper keep;                      // (member 0 runs on the calling thread)
  std::vector<int> rc(n, TM_OK);
  std::vector<std::string> msg(n);
  auto run = [&](int i) {
    const hipError_t e = hipSetDevice(g->dev[i]);
    rc[i] = e == hipSuccess ? work(i) : hip_fail(e, "hipSetDevice (member)");
    if (rc[i] != TM_OK) msg[i] = last_error();
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n; i++) th.emplace_back(run, i);
  run(0);
  for (auto& t : th) t.join();
  for (int i = 0; i < n; i++) if (rc[i] != TM_OK) return set_error(rc[i], "member %d (device %d): %s", i, g->dev[i], msg[i].c_str());
  return TM_OK;
}

// a barrier of the members' host threads that also carries the verdict "somebody failed": nobody waits for a member that has given up
struct Meet {
  std::mutex mu;
  std::condition_variable cv;
  int n, arrived = 0, round = 0;
  bool failed = false, verdict = true;
  explicit Meet(int members) : n(members) {}
  // Returns false if any member had failed when the LAST member of this round arrived - the same answer for every member of the round.
  // (A member that fails right after this barrier and reaches the next one before a slower member has woken up from this one must not
  // change what the slower member is told here: it would leave, never reach the next barrier, and the failing member would wait there
  // forever.  `verdict` is written when a round completes and can only be written again when the next one does, which needs every
  // member - the slow one included - to have arrived there, i.e. to have read it.)
  bool wait(bool ok) {
    std::unique_lock<std::mutex> lk(mu);
    if (!ok) failed = true;
    const int my = round;
    if (++arrived == n) { arrived = 0; verdict = !failed; round++; cv.notify_all(); }
    else cv.wait(lk, [&] { return round != my; });
    return verdict;
  }
};

}  // namespace tmh

extern "C" {

int tm_devices_open(int max_devices, tm_devices** out) {
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have < 1) { (void)hipGetLastError(); return set_error(TM_E_NODEVICE, "no HIP device visible"); }
  std::vector<int> list;
  if (const char* e = hooks_armed() ? getenv("TM_VIRTUAL_DEVICES") : nullptr) {       // N members on device 0: the multi-device code paths on a one-GPU box (test processes only: TM_TEST_HOOKS)
    const int n = atoi(e);
    if (n >= 1 && n <= 64) list.assign((size_t)n, 0);
  }
  if (list.empty()) for (int d = 0; d < have && (max_devices <= 0 || d < max_devices); d++) list.push_back(d);
  if (max_devices > 0 && (int)list.size() > max_devices) list.resize((size_t)max_devices);
  return open_list(list.data(), (int)list.size(), out);
}

int tm_devices_open_list(const int* devices, int n, tm_devices** out) { return open_list(devices, n, out); }

int tm_devices_count(const tm_devices* g) { return g ? (int)g->dev.size() : 0; }
int tm_devices_device(const tm_devices* g, int member) { return g && member >= 0 && member < (int)g->dev.size() ? g->dev[member] : -1; }

int tm_devices_rccl_ranks(tm_devices* g, const char** why_not) {
  if (!g) return 0;
  std::lock_guard<std::mutex> lk(g->mu);
  const bool on = rccl_ready(g);
  if (why_not) *why_not = on ? "" : g->rccl_note.c_str();
  return on ? g->rccl_ranks : 0;
}

void tm_devices_close(tm_devices* g) {
  if (!g) return;
  DeviceKeeper keep;
#ifndef TM_EMU
  for (size_t i = 0; i < g->comm.size(); i++) if (g->comm[i]) { (void)hipSetDevice(g->dev[i]); (void)rccl().CommDestroy(g->comm[i]); }
#endif
  for (size_t i = 0; i < g->dev.size(); i++) {
    (void)hipSetDevice(g->dev[i]);
    if (g->stream[i]) (void)hipStreamDestroy(g->stream[i]);
    if (g->done[i]) (void)hipEventDestroy(g->done[i]);
    if (i < g->begun.size() && g->begun[i]) (void)hipEventDestroy(g->begun[i]);
    if (i < g->d_all_exits.size() && g->d_all_exits[i]) (void)hipFree(g->d_all_exits[i]);
  }
  if (g->d_scratch) { (void)hipSetDevice(g->dev[0]); (void)hipFree(g->d_scratch); }
  delete g;
}

// ---- one vocabulary on every device ---------------------------------------------------------------------------------------------------
static int replicate(tm_devices* g, tm_vocab* first, tm_vocab_set** out);
int tm_vocab_load_all(tm_devices* g, const uint8_t* vocab_file, size_t n, tm_vocab_set** out) {
  if (!g || !out) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  // the tables are built ONCE (parse, trie, double array, links: tm_vocab_load) and uploaded to member 0; the finished block then goes
  // device to device (xGMI where the devices are peers) into an imported vocabulary of the same shape on every other member
  tm_vocab* first = nullptr;
  const int rc = tm_vocab_load_on(vocab_file, n, g->dev[0], &first);
  return rc == TM_OK ? replicate(g, first, out) : rc;
}

// the set around a vocabulary that already lives on member 0: replicas of its device block on every other member
static int replicate(tm_devices* g, tm_vocab* first, tm_vocab_set** out) {
  DeviceKeeper keep;
  auto* s = new tm_vocab_set();
  s->devs = g;
  s->v.assign(g->dev.size(), nullptr);
  s->v[0] = first;
  tm_vocab_block meta;
  void* src = nullptr;
  int rc = tm_vocab_block_export(first, &meta, &src);
  for (size_t i = 1; i < g->dev.size() && rc == TM_OK; i++) {
    void* dst = nullptr;
    if ((rc = tm_vocab_block_import(&meta, g->dev[i], &s->v[i], &dst)) != TM_OK) break;
    const hipError_t e = g->dev[i] == g->dev[0] ? hipMemcpy(dst, src, meta.bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, g->dev[i], src, g->dev[0], meta.bytes);
    if (e != hipSuccess) rc = hip_fail(e, "replicating the vocabulary block");
  }
  if (rc != TM_OK) { const std::string keep = last_error(); tm_vocab_set_free(s); return set_error(rc, "%s", keep.c_str()); }
  *out = s;
  return TM_OK;
}

int tm_vocab_build_all(tm_devices* g, const uint8_t* blob, const uint32_t* off, uint32_t n_tokens, const uint8_t* special, uint32_t capcode, uint32_t charset,
                       uint32_t norm_flag, uint32_t level, int with_unk, tm_vocab_set** out) {
  if (!g || !out) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  tm_vocab* first = nullptr;
  const int rc = tm_vocab_build(blob, off, n_tokens, special, capcode, charset, norm_flag, level, with_unk, g->dev[0], &first);
  return rc == TM_OK ? replicate(g, first, out) : rc;
}

// tm_vocab_tune for a set: member 0 (the one with the records) lays its tables out again, the others take over its block as they did at first
extern "C++" void tmh_vocab_quiesce(tm_vocab* v);                                   // (tm_vocab.hip)
extern "C++" int tmh_vocab_adopt(tm_vocab* v, const tm_vocab_block* m);
int tm_vocab_set_tune(tm_vocab_set* s, const uint8_t* normalized_sample, uint64_t n) {
  if (!s || s->v.empty() || !s->v[0]) return set_error(TM_E_INVALID, "null argument");
  DeviceKeeper keep;
  int rc = tm_vocab_tune(s->v[0], normalized_sample, n);
  tm_vocab_block meta;
  void* src = nullptr;
  if (rc == TM_OK) rc = tm_vocab_block_export(s->v[0], &meta, &src);
  const tm_devices* g = s->devs;
  for (size_t i = 1; i < s->v.size() && rc == TM_OK; i++) {
    tm_vocab* w = s->v[i];
    if ((rc = tm_set_device(g->dev[i])) != TM_OK) break;
    tmh_vocab_quiesce(w);
    void* dst = nullptr;
    tm_vocab_block mine;
    if ((rc = tm_vocab_block_export(w, &mine, &dst)) != TM_OK) break;
    const hipError_t e = g->dev[i] == g->dev[0] ? hipMemcpy(dst, src, meta.bytes, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, g->dev[i], src, g->dev[0], meta.bytes);
    if (e != hipSuccess) { rc = hip_fail(e, "replicating the vocabulary block"); break; }
    rc = tmh_vocab_adopt(w, &meta);
  }
  return rc;
}

const tm_vocab* tm_vocab_set_member(const tm_vocab_set* s, int member) { return s && member >= 0 && member < (int)s->v.size() ? s->v[member] : nullptr; }
int tm_vocab_set_count(const tm_vocab_set* s) { return s ? (int)s->v.size() : 0; }

void tm_vocab_set_free(tm_vocab_set* s) {
  if (!s) return;
  DeviceKeeper keep;
  for (tm_vocab* v : s->v) tm_vocab_free(v);
  delete s;
}

// ---- batch tokenize over every device: chunks of whole documents go to whichever lane of whichever device is free (tm_host.hip) ---------
int tm_tokenize_pipeline_multi(const tm_vocab_set* s, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, int raw, uint32_t encoding_length,
                               uint64_t chunk_bytes, uint32_t lanes_per_device, uint8_t* bytes_out, uint64_t bytes_cap, uint64_t* byte_offsets,
                               uint32_t* missing, uint32_t* encoding_length_used, tm_pipeline_stats* stats) {
  if (!s || s->v.empty()) return set_error(TM_E_INVALID, "null argument");
  return tokenize_pipeline_on(s->v.data(), (uint32_t)s->v.size(), text, offsets, ndocs, raw, encoding_length, chunk_bytes, lanes_per_device, bytes_out, bytes_cap,
                              byte_offsets, missing, encoding_length_used, stats);
}

// ---- the scoring pass over every device ---------------------------------------------------------------------------------------------------
int tm_dataset_upload_sharded(tm_devices* g, const uint8_t* normalized, uint64_t n, tm_dataset_set** out) {
  if (!g || !out || (n && !normalized)) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  auto* s = new tm_dataset_set();
  s->devs = g;
  s->n = n;
  const int nd = (int)g->dev.size();
  s->part.assign(nd, nullptr); s->own.assign(nd, 0); s->continues.assign(nd, 0);
  // one contiguous byte range per member, cut on multiples of 4 as trainvocab cuts its strips (trainvocab.go:1674); a dataset too small
  // to give every member a range worth a launch (and the 64 bytes tm_score_begin asks of a range that is followed by text) uses fewer
  constexpr uint64_t kHalo = 128, kMinRange = 4096;
  const int active = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)nd, n / kMinRange));
  const uint64_t per = (n / (uint64_t)active) / 4 * 4;
  std::vector<uint64_t> lo(nd, n), hi(nd, n);
  for (int i = 0; i < active; i++) { lo[i] = per * (uint64_t)i; hi[i] = i + 1 == active ? n : per * (uint64_t)(i + 1); }
  const int rc = on_members(g, [&](int i) -> int {
    const uint64_t halo = std::min<uint64_t>(kHalo, n - hi[i]);
    s->own[i] = hi[i] - lo[i];
    s->continues[i] = hi[i] == n ? 0 : (halo < kHalo ? 2 : 1);
    return tm_dataset_upload_on(normalized + lo[i], s->own[i] + halo, g->dev[i], &s->part[i]);
  });
  if (rc != TM_OK) { const std::string keep = last_error(); tm_dataset_set_free(s); return set_error(rc, "%s", keep.c_str()); }
  *out = s;
  return TM_OK;
}

void tm_dataset_set_free(tm_dataset_set* s) {
  if (!s) return;
  DeviceKeeper keep;
  for (size_t i = 0; i < s->part.size(); i++) if (s->part[i]) { (void)hipSetDevice(s->devs->dev[i]); tm_dataset_free(s->part[i]); }
  delete s;
}

uint64_t tm_dataset_set_range(const tm_dataset_set* s, int member, uint64_t* halo_bytes) {
  if (!s || member < 0 || member >= (int)s->part.size()) return 0;
  if (halo_bytes) *halo_bytes = s->part[member] ? s->part[member]->n - s->own[member] : 0;
  return s->own[member];
}

int tm_score_multi(const tm_vocab_set* vs, tm_dataset_set* ds, uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]) {
  if (!vs || !ds) return set_error(TM_E_INVALID, "null argument");
  tm_devices* g = vs->devs;
  if (ds->devs != g) return set_error(TM_E_INVALID, "vocabulary set and dataset set belong to different tm_devices handles");
  const int nd = (int)g->dev.size();
  DeviceKeeper keep;
  std::lock_guard<std::mutex> pass(g->mu);
  const bool use_rccl = rccl_ready(g);
  const uint64_t words = (uint64_t)vs->v[0]->host.n_ids + 4 + 256;
  std::vector<uint32_t> h0(words);        // member 0 reads the summed histogram on its own stream, behind the collective
  uint32_t err0 = 0;
  Meet meet(nd);
  // Every member's host thread ENQUEUES its whole half of the pass and waits once (round 6; the 80 exit states used to make a trip to the host
  // and back between (1) and (3)): (1) match kernel over its range -> the range's exit state for all 80 entry states; (2) the maps of all
  // ranges gathered on every device - ncclAllGather of 80 bytes per member, or peer copies where there is no communicator - and chained to
  // the member's true entry state by k_chain_entry (dist.resolve_entry is the Python form of the same rule); (3) the histogram walk from that
  // state; (4) ONE all-reduce(sum) of the n_ids + 4 + 256 uint32 words (tm_score.hip's layout: every word is a plain sum over ranges), and
  // behind it on member 0's stream the copy of the sum to the host.  The members meet (host threads only, nothing is waited for on the
  // devices) before each collective, so that a member that failed keeps the others out of it.  A member without bytes (tiny dataset) runs
  // the same calls on an empty range: identity map, zero histogram.
  const int rc = on_members(g, [&](int i) -> int {
    tm_dataset* d = ds->part[i];
    const tm_vocab* v = vs->v[i];
    hipStream_t st = g->stream[i];
    hipError_t e;
    int r = score_begin_device(v, d, 0, ds->own[i], ds->continues[i], st);
    if (r == TM_OK && (e = hipEventRecord(g->begun[i], st)) != hipSuccess) r = hip_fail(e, "hipEventRecord");
    if (!meet.wait(r == TM_OK)) return r;
    // (test hook 14: the last member gives up between the first and the second meeting - every member must come back with its error,
    // nobody may be left waiting at a meeting the others never reach)
    if ((debug_flags() & 16384) && i == nd - 1) r = set_error(TM_E_INPUT, "test hook 14: member %d gives up after the first meeting", i);
    if (r == TM_OK) {
#ifndef TM_EMU
      if (use_rccl) {
        (void)hipGetLastError();
        const ncclResult_t nr = rccl().AllGather(score_exits_device(d), g->d_all_exits[i], (size_t)ENT, ncclUint8, g->comm[i], st);
        if (nr != ncclSuccess) r = set_error(TM_E_HIP, "ncclAllGather: %s", rccl().GetErrorString(nr));
      } else
#endif
      for (int k = 0; k < nd && r == TM_OK; k++) {
        if (k != i && (e = hipStreamWaitEvent(st, g->begun[k], 0)) != hipSuccess) { r = hip_fail(e, "hipStreamWaitEvent"); break; }
        e = g->dev[k] == g->dev[i] ? hipMemcpyAsync(g->d_all_exits[i] + (size_t)k * ENT, score_exits_device(ds->part[k]), ENT, hipMemcpyDeviceToDevice, st)
                                   : hipMemcpyPeerAsync(g->d_all_exits[i] + (size_t)k * ENT, g->dev[i], score_exits_device(ds->part[k]), g->dev[k], ENT, st);
        if (e != hipSuccess) r = hip_fail(e, "copy of an exit map");
      }
    }
    if (r == TM_OK) {
      TM_LAUNCH(k_chain_entry, 1, 64, 0, st, g->d_all_exits[i], i, score_entry_device(d), score_error_device(d));
      r = score_finish_device(v, d, st);
    }
    if (r == TM_OK && (e = hipEventRecord(g->done[i], st)) != hipSuccess) r = hip_fail(e, "hipEventRecord");
    if (!meet.wait(r == TM_OK)) return r;
#ifndef TM_EMU
    if (use_rccl) {
      // in place, on the member's own stream behind its histogram kernel; the members call from their own threads (RCCL's one-thread-per-device mode)
      (void)hipGetLastError();
      const ncclResult_t nr = rccl().AllReduce(d->d_hist, d->d_hist, (size_t)words, ncclUint32, ncclSum, g->comm[i], st);
      if (nr != ncclSuccess) r = set_error(TM_E_HIP, "ncclAllReduce: %s", rccl().GetErrorString(nr));
    } else
#endif
    if (i == 0) {
      // no communicator (virtual devices / TM_RCCL=0 / no librccl): member 0 pulls every other histogram over and adds it
      if (g->scratch_words < words) {
        (void)hipFree(g->d_scratch); g->d_scratch = nullptr; g->scratch_words = 0;
        if ((e = hipMalloc((void**)&g->d_scratch, (words + words / 4) * 4)) != hipSuccess) r = hip_fail(e, "hipMalloc (histogram scratch)");
        else g->scratch_words = words + words / 4;
      }
      for (int k = 1; k < nd && r == TM_OK; k++) {
        if ((e = hipStreamWaitEvent(st, g->done[k], 0)) != hipSuccess) { r = hip_fail(e, "hipStreamWaitEvent"); break; }
        e = g->dev[k] == g->dev[0] ? hipMemcpyAsync(g->d_scratch, ds->part[k]->d_hist, words * 4, hipMemcpyDeviceToDevice, st)
                                   : hipMemcpyPeerAsync(g->d_scratch, g->dev[0], ds->part[k]->d_hist, g->dev[k], words * 4, st);
        if (e != hipSuccess) { r = hip_fail(e, "peer copy of a histogram"); break; }
        TM_LAUNCH(k_hist_add, (uint32_t)((words + 255) / 256), 256, 0, st, d->d_hist, g->d_scratch, words);
      }
    }
    // the sum goes to the host behind the collective, on the same stream: ONE wait per member and pass
    if (r == TM_OK && i == 0) { r = small_d2h(d->ws, h0.data(), d->d_hist, words * 4, st); if (r == TM_OK) r = small_d2h(d->ws, &err0, d->ws->d_error, 4, st); }
    if (r == TM_OK) r = i == 0 ? small_sync(d->ws, st) : ((e = hipStreamSynchronize(st)) != hipSuccess ? hip_fail(e, "hipStreamSynchronize (scoring pass)") : TM_OK);
    if (r == TM_OK) r = i == 0 ? error_from_flag(err0) : score_check(d);
    // nobody frees or reuses a histogram before member 0 has read them all
    meet.wait(r == TM_OK);
    return r;
  });
  if (rc != TM_OK) return rc;
  const uint32_t n_ids = vs->v[0]->host.n_ids;
  if (scores) std::memcpy(scores, h0.data(), (size_t)n_ids * 4);
  if (tokens_in_text) {
    uint64_t t = 0;
    for (int k = 0; k < 4; k++) t += (uint64_t)h0[n_ids + k] << (16 * k);
    *tokens_in_text = t;
  }
  if (missing_set) {
    std::memset(missing_set, 0, 32);
    for (int k = 0; k < 256; k++) if (h0[n_ids + 4 + k]) missing_set[k >> 3] |= (uint8_t)(1u << (k & 7));
  }
  return TM_OK;
}

}  // extern "C"
