// tm_score.hip — the trainvocab candidate-scoring pass behind tm_dataset_upload / tm_score* (include/tokenmonster_hip.h).
//
// Replaces the worker loop of training/trainvocab.go:925-1176: the SAME walk as tokenization (tm_kernels.hip: match_branch,
// resolve) over strips of one device-resident normalized dataset, but instead of emitting ids the chain kernel accumulates
// scores[id] += bytes covered, scores[deleteToken] += 1 per forward delete, tokensInText and the set of bytes that had no
// token (trainvocab.go:1105-1174).  The histogram lives in HBM as plain uint32 sums so that data-parallel ranks merge it
// with ONE RCCL all-reduce (tokenmonster_amd/dist.py).
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "tm_pipeline.h"

using namespace tmh;

// (struct tm_dataset: tm_pipeline.h - tm_multi.hip drives one dataset per device)

// One scoring pass in two halves.  score_prepare: the strips become the documents of the workspace, K0 + K1 run (and the group maps
// of long strips).  score_complete: K3 from the strips' entry states + the histogram walk.  Between the two a caller that scores
// ONE byte range of a whole-buffer walk (tm_score_begin / tm_score_finish) reads the range's exit map and learns its entry state.
static int score_prepare(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
                         bool continues, hipStream_t st) {
  if (!v || !d) return set_error(TM_E_INVALID, "null argument");
  { int rc = enter_device(v); if (rc != TM_OK) return rc; }
  if (d->device != v->device) return set_error(TM_E_INVALID, "dataset lives on device %d, vocabulary on device %d", d->device, v->device);
  d->prepared = false;
  std::vector<uint64_t> be;
  uint64_t whole_off = 0, whole_len = d->n;
  if (n_strips == 0) { strip_off = &whole_off; strip_len = &whole_len; n_strips = 1; }
  be.resize(3ull * n_strips);
  uint64_t nseg = 0;
  for (uint32_t k = 0; k < n_strips; k++) {
    if (strip_off[k] > d->n || strip_len[k] > d->n - strip_off[k]) return set_error(TM_E_INVALID, "strip %u outside the dataset", k);
    be[k] = strip_off[k];
    be[n_strips + k] = strip_off[k] + strip_len[k];
    be[2ull * n_strips + k] = continues ? d->n : strip_off[k] + strip_len[k];     // how far the strip may look (k_match_branch)
    nseg += (strip_len[k] + SEG - 1) / SEG;
  }
  // Strips must be disjoint: the workspace (max_bytes / SEG + n_strips + 1 segments) is sized for strips that together cover
  // the dataset at most once, and a byte that two strips share would be counted twice.  (The trainvocab worker's strips are disjoint: trainvocab.go:1668-1695.)
  if (n_strips > 1) {
    std::vector<std::pair<uint64_t, uint64_t>> iv;
    iv.reserve(n_strips);
    for (uint32_t k = 0; k < n_strips; k++) if (strip_len[k]) iv.emplace_back(strip_off[k], strip_off[k] + strip_len[k]);
    std::sort(iv.begin(), iv.end());
    for (size_t k = 1; k < iv.size(); k++)
      if (iv[k].first < iv[k - 1].second) return set_error(TM_E_INVALID, "strips overlap at dataset byte %llu", (unsigned long long)iv[k].first);
  }
  hipError_t e;
  // (the workspace does not depend on the vocabulary: a new candidate reuses it as it is)
  if (d->ws && d->ws_docs < n_strips) { tm_batch_free(d->ws); d->ws = nullptr; }
  if (!d->ws) {
    int rc = make_workspace(v, d->n, n_strips, false, false, &d->ws);
    if (rc != TM_OK) return rc;
    d->ws_docs = n_strips;
    d->ws->d_text = d->d_text;
  }
  tm_batch* b = d->ws;
  b->vocab = v;
  if (nseg > b->max_segs) return set_error(TM_E_LIMIT, "%llu segments, workspace holds %llu", (unsigned long long)nseg, (unsigned long long)b->max_segs);
  const uint64_t words = (uint64_t)v->host.n_ids + 4 + 256;
  if (d->hist_cap < words) {                 // grow-only: candidate vocabularies differ in size, and hipFree synchronizes the device
    (void)hipFree(d->d_hist);
    d->d_hist = nullptr;
    d->hist_cap = 0;
    if ((e = hipMalloc((void**)&d->d_hist, (words + words / 4) * 4)) != hipSuccess) return hip_fail(e, "hipMalloc histogram");
    d->hist_cap = words + words / 4;
  }
  d->hist_words = words;
  if (d->strip_cap < n_strips) {
    (void)hipFree(d->d_vis); (void)hipFree(d->d_entry); (void)hipFree(d->d_exits);
    d->d_vis = nullptr; d->d_entry = nullptr; d->d_exits = nullptr;
    d->strip_cap = n_strips + 16;
    if ((e = hipMalloc((void**)&d->d_vis, (size_t)d->strip_cap * 8)) != hipSuccess || (e = hipMalloc((void**)&d->d_entry, d->strip_cap)) != hipSuccess ||
        (e = hipMalloc((void**)&d->d_exits, (size_t)d->strip_cap * ENT)) != hipSuccess) { d->strip_cap = 0; return hip_fail(e, "hipMalloc (strips)"); }
  }
  // The strips of a training run do not change from pass to pass (one strip: the whole dataset; a member's byte range of tm_score_multi): what
  // they put on the device - offsets, how far each may look, the group tree of the long ones - is still there from the last pass, and a pass
  // that finds them unchanged starts its kernels at once (the upload of the tree ends in a wait of its own: 0.1 ms of the 3.7 a member's
  // share of the 8-GPU pass takes, profiles/r06_score_rank_protocol.json).
  const bool same_strips = d->strips_valid && d->strips_key == be && d->ws_strips_owner == b;
  if (!same_strips) {
    d->strips_valid = false;
    int rc = small_h2d(b, b->d_offsets, be.data(), 2ull * n_strips * 8, st);          // (through the pinned mailbox: no pageable copies on this path)
    if (rc == TM_OK) rc = small_h2d(b, d->d_vis, be.data() + 2ull * n_strips, (uint64_t)n_strips * 8, st);
    if (rc != TM_OK) return rc; }
  b->d_doc_begin = b->d_offsets;
  b->text_in_slabs = false;
  b->d_doc_end = b->d_offsets + n_strips;
  b->d_doc_vis = d->d_vis;
  b->d_doc_entry = nullptr;
  b->ndocs = n_strips;
  b->nbytes = d->n;
  b->nseg = nseg;
  if (!same_strips) {
    int grc = build_groups(b, be.data(), be.data() + n_strips, n_strips, st);
    if (grc != TM_OK) return grc;
    d->strips_key = be; d->ws_strips_owner = b; d->strips_valid = true;
  }
  int rc = pipeline_match(b, st, nullptr, true);
  if (rc == TM_OK) d->prepared = true;
  return rc;
}

static int score_complete(const tm_vocab* v, tm_dataset* d, const uint8_t* entry_states, hipStream_t st) {
  if (!d->prepared) return set_error(TM_E_INVALID, "tm_score_finish without tm_score_begin");
  d->prepared = false;
  tm_batch* b = d->ws;
  hipError_t e;
  if (entry_states) {
    if ((e = hipMemcpyAsync(d->d_entry, entry_states, b->ndocs, hipMemcpyHostToDevice, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess)
      return hip_fail(e, "H2D entry states");
    b->d_doc_entry = d->d_entry;
  }
  const uint64_t words = d->hist_words;
  (void)hipMemsetAsync(d->d_hist, 0, words * 4, st);
  (void)hipMemsetAsync(d->d_tokens, 0, 8, st);
  (void)hipMemsetAsync(d->d_missing_bits, 0, 32, st);
  int rc = pipeline_resolve(b, st, nullptr, 0);
  if (rc != TM_OK) return rc;
  launch_chain_hist(b, v->tables.has_delete ? v->tables.delete_id : 0, d->n_cu, d->d_hist, d->d_tokens, d->d_missing_bits, v->host.n_ids, st);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "kernel launch");
  return TM_OK;
}

static int score_run(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips, hipStream_t st) {
  int rc = score_prepare(v, d, strip_off, strip_len, n_strips, false, st);
  return rc == TM_OK ? score_complete(v, d, nullptr, st) : rc;
}

namespace tmh {
// The two halves of a byte range's pass WITHOUT a trip to the host between them (tm_multi.hip: tm_score_multi): begin = tm_score_begin up to the
// exit map on the device; the caller puts the range's entry state into score_entry_device(d) with a kernel on the same stream; finish = the
// histogram walk from there.  The caller holds no lock of the dataset between the two: tm_score_multi owns its datasets for the pass (tm_devices::mu).
int score_begin_device(const tm_vocab* v, tm_dataset* d, uint64_t off, uint64_t len, int continues, hipStream_t st) {
  if (!v || !d) return set_error(TM_E_INVALID, "null argument");
  if (off > d->n || len > d->n - off) return set_error(TM_E_INVALID, "byte range [%llu, +%llu) outside the dataset of %llu bytes", (unsigned long long)off, (unsigned long long)len, (unsigned long long)d->n);
  if (continues && len < 64) return set_error(TM_E_INVALID, "a byte range that is followed by more text must be at least 64 bytes long");
  if (continues == 1 && d->n - (off + len) < 128) return set_error(TM_E_INVALID, "a byte range that is followed by more text needs >= 128 bytes of it behind the range");
  std::lock_guard<std::mutex> g(d->mu);
  int rc = score_prepare(v, d, &off, &len, 1, continues != 0, st);
  if (rc != TM_OK) return rc;
  launch_doc_exits(d->ws, d->d_exits, st);
  return TM_OK;
}
const uint8_t* score_exits_device(const tm_dataset* d) { return d->d_exits; }
uint8_t* score_entry_device(tm_dataset* d) { return d->d_entry; }
uint32_t* score_error_device(tm_dataset* d) { return d->ws->d_error; }
int score_finish_device(const tm_vocab* v, tm_dataset* d, hipStream_t st) {
  if (!v || !d) return set_error(TM_E_INVALID, "null argument");
  std::lock_guard<std::mutex> g(d->mu);
  if (!d->prepared) return set_error(TM_E_INVALID, "score_finish_device without score_begin_device");
  d->ws->d_doc_entry = d->d_entry;
  return score_complete(v, d, nullptr, st);
}
int score_check(tm_dataset* d) {
  if (!d || !d->ws) return TM_OK;
  uint32_t err = 0;
  int rc = small_d2h(d->ws, &err, d->ws->d_error, 4, d->ws->last_stream);
  if (rc == TM_OK) rc = small_sync(d->ws, d->ws->last_stream);
  return rc == TM_OK ? error_from_flag(err) : rc;
}
}  // namespace tmh

extern "C" {

int tm_dataset_upload_on(const uint8_t* normalized, uint64_t n, int device, tm_dataset** out) {
  const int rc = tm_set_device(device);
  return rc == TM_OK ? tm_dataset_upload(normalized, n, out) : rc;
}
int tm_dataset_device(const tm_dataset* d) { return d ? d->device : -1; }

int tm_dataset_upload(const uint8_t* normalized, uint64_t n, tm_dataset** out) {
  if (!out || (n && !normalized)) return set_error(TM_E_INVALID, "null argument");
  auto* d = new tm_dataset();
  hipError_t e;
  if ((e = hipMalloc((void**)&d->d_text, n + 256)) != hipSuccess || (e = hipMalloc((void**)&d->d_tokens, 8)) != hipSuccess ||
      (e = hipMalloc((void**)&d->d_missing_bits, 32)) != hipSuccess ||
      (n && (e = hipMemcpy(d->d_text, normalized, n, hipMemcpyHostToDevice)) != hipSuccess)) {
    tm_dataset_free(d);
    return hip_fail(e, "dataset upload");
  }
  d->n = n;
  (void)hipGetDevice(&d->device);
  { int dev = 0, cu = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) d->n_cu = cu; }
  *out = d;
  return TM_OK;
}

void tm_dataset_free(tm_dataset* d) {
  if (!d) return;
  tm_batch_free(d->ws);
  (void)hipFree(d->d_text); (void)hipFree(d->d_hist); (void)hipFree(d->d_tokens); (void)hipFree(d->d_missing_bits);
  (void)hipFree(d->d_vis); (void)hipFree(d->d_entry); (void)hipFree(d->d_exits);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int tm_score_device(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
                    void* stream, uint32_t** dev_hist, uint64_t* n_words) {
  if (!d) return set_error(TM_E_INVALID, "null argument");
  std::lock_guard<std::mutex> g(d->mu);
  int rc = score_run(v, d, strip_off, strip_len, n_strips, (hipStream_t)stream);
  if (rc != TM_OK) return rc;
  if (dev_hist) *dev_hist = d->d_hist;
  if (n_words) *n_words = d->hist_words;
  return TM_OK;
}

int tm_score_device_into(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
                         void* stream, uint32_t* dst_device, uint64_t dst_words) {
  if (!dst_device || !d) return set_error(TM_E_INVALID, "null argument");
  std::lock_guard<std::mutex> g(d->mu);
  int rc = score_run(v, d, strip_off, strip_len, n_strips, (hipStream_t)stream);
  if (rc != TM_OK) return rc;
  if (dst_words < d->hist_words) return set_error(TM_E_NOSPACE, "destination holds %llu words, histogram has %llu", (unsigned long long)dst_words, (unsigned long long)d->hist_words);
  hipError_t e = hipMemcpyAsync(dst_device, d->d_hist, d->hist_words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "D2D histogram");
  return TM_OK;
}

int tm_score_begin(const tm_vocab* v, tm_dataset* d, uint64_t off, uint64_t len, int continues, void* stream, uint8_t* exits) {
  if (!exits || !d) return set_error(TM_E_INVALID, "null argument");
  if (off > d->n || len > d->n - off) return set_error(TM_E_INVALID, "byte range [%llu, +%llu) outside the dataset of %llu bytes", (unsigned long long)off, (unsigned long long)len, (unsigned long long)d->n);
  if (continues && len < 64) return set_error(TM_E_INVALID, "a byte range that is followed by more text must be at least 64 bytes long");
  // tokens that begin in the range may end behind it, and the look-ahead of the last ones reaches further still: without the halo the match
  // kernel would take the text to end early and the exit states would silently differ from the whole-buffer walk's
  if (continues == 1 && d->n - (off + len) < 128) return set_error(TM_E_INVALID, "a byte range that is followed by more text needs >= 128 bytes of it behind the range (the dataset holds %llu; continues = 2 if that is all the text there is)", (unsigned long long)(d->n - (off + len)));
  std::lock_guard<std::mutex> g(d->mu);      // (the two halves of a pass are two calls: one caller per dataset between them; the lock keeps a concurrent tm_score out of each half)
  hipStream_t st = (hipStream_t)stream;
  int rc = score_prepare(v, d, &off, &len, 1, continues != 0, st);
  if (rc != TM_OK) return rc;
  launch_doc_exits(d->ws, d->d_exits, st);
  hipError_t e;
  if ((e = hipMemcpyAsync(exits, d->d_exits, ENT, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) {
    d->prepared = false;
    return hip_fail(e, "D2H exit map");
  }
  return TM_OK;
}

int tm_score_finish(const tm_vocab* v, tm_dataset* d, uint32_t entry_state, void* stream, uint32_t* dst_device, uint64_t dst_words) {
  if (!v || !d) return set_error(TM_E_INVALID, "null argument");
  if (entry_state >= (uint32_t)ENT) return set_error(TM_E_INVALID, "entry state %u out of range", entry_state);
  const uint8_t es = (uint8_t)entry_state;
  std::lock_guard<std::mutex> g(d->mu);
  int rc = score_complete(v, d, &es, (hipStream_t)stream);
  if (rc != TM_OK || !dst_device) return rc;
  if (dst_words < d->hist_words) return set_error(TM_E_NOSPACE, "destination holds %llu words, histogram has %llu", (unsigned long long)dst_words, (unsigned long long)d->hist_words);
  hipError_t e = hipMemcpyAsync(dst_device, d->d_hist, d->hist_words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return e == hipSuccess ? TM_OK : hip_fail(e, "D2D histogram");
}

int tm_score_read(const tm_vocab* v, tm_dataset* d, uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]) {
  if (!v || !d || !d->d_hist || !d->ws) return set_error(TM_E_INVALID, "no scoring pass to read");
  hipError_t e;
  std::vector<uint32_t> h(d->hist_words);
  uint32_t err = 0;
  {
    hipStream_t st = d->ws->last_stream;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    int rc = small_d2h(d->ws, h.data(), d->d_hist, h.size() * 4, st);
    if (rc == TM_OK) rc = small_d2h(d->ws, &err, d->ws->d_error, 4, st);
    if (rc == TM_OK) rc = small_sync(d->ws, st);
    if (rc != TM_OK) return rc;
  }
  if (err) return error_from_flag(err);
  const uint32_t n_ids = v->host.n_ids;
  if (scores) std::memcpy(scores, h.data(), (size_t)n_ids * 4);
  if (tokens_in_text) {
    uint64_t t = 0;
    for (int k = 0; k < 4; k++) t += (uint64_t)h[n_ids + k] << (16 * k);
    *tokens_in_text = t;
  }
  if (missing_set) {
    std::memset(missing_set, 0, 32);
    for (int k = 0; k < 256; k++) if (h[n_ids + 4 + k]) missing_set[k >> 3] |= (uint8_t)(1u << (k & 7));
  }
  return TM_OK;
}

int tm_score(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
             uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]) {
  if (!d) return set_error(TM_E_INVALID, "null argument");
  static const bool trace = getenv("TM_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = trace ? now() : 0;
  std::lock_guard<std::mutex> g(d->mu);
  const double t1 = trace ? now() : 0;
  hipStream_t st = nullptr;
  if (!d->stream && hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) d->stream = nullptr;
  st = d->stream;          // (not the NULL stream: the table uploads of other threads' tm_vocab_load must not order behind this pass)
  int rc = score_run(v, d, strip_off, strip_len, n_strips, st);
  const double t2 = trace ? now() : 0;
  if (rc == TM_OK) { hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize"); }
  const double t3 = trace ? now() : 0;
  rc = rc == TM_OK ? tm_score_read(v, d, scores, tokens_in_text, missing_set) : rc;
  if (trace) fprintf(stderr, "[tm_score] lock wait %.2f ms, launch %.2f ms, kernels %.2f ms, read %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
  return rc;
}

}  // extern "C"
