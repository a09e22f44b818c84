// tm_decode.hip — Decode / decode_raw behind tm_decode_batch (include/tokenmonster_hip.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "tm_pipeline.h"

using namespace tmh;

// ================================================================================================
// Decode (go/tokenmonster.go:445-550 Decode; tokenmonster.cpp:1404-1425): ids -> bytes
// ================================================================================================
// reverse[id] lengths -> exclusive scan -> copy.  Ids >= n_ids are skipped like the reference does.  Capcode decoding
// (a per-document state machine, javascript/tokenmonster.js:1007-1065) runs on the host after the gather.
namespace tmh {
__global__ void k_dec_len(const uint32_t* __restrict__ tokens, uint64_t n, const uint32_t* __restrict__ rev_off, uint32_t n_ids,
                          uint32_t* __restrict__ tok_len) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = tokens[i];
  tok_len[i] = id < n_ids ? rev_off[id + 1] - rev_off[id] : 0u;
}
__global__ void k_dec_copy(const uint32_t* __restrict__ tokens, uint64_t n, const uint32_t* __restrict__ rev_off, const uint8_t* __restrict__ rev_bytes,
                           uint32_t n_ids, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = tokens[i];
  if (id >= n_ids) return;
  const uint32_t s = rev_off[id], l = rev_off[id + 1] - s;
  uint8_t* o = out + out_off[i];
  for (uint32_t j = 0; j < l; j++) o[j] = rev_bytes[s + j];
}
__global__ void k_dec_doc_off(const uint64_t* __restrict__ out_off, const uint64_t* __restrict__ tok_offsets, uint32_t ndocs, uint64_t* __restrict__ doc_off) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d <= ndocs) doc_off[d] = out_off[tok_offsets[d]];
}
}  // namespace tmh

extern "C" int tm_decode_batch(const tm_vocab* v, const uint32_t* tokens, const uint64_t* tok_offsets, uint32_t ndocs, int raw,
                               uint8_t* out, uint64_t out_cap, uint64_t* out_offsets) {
  if (!v || !tok_offsets || !out_offsets) return set_error(TM_E_INVALID, "null argument");
  const uint64_t n = tok_offsets[ndocs];
  if (n && !tokens) return set_error(TM_E_INVALID, "null argument");
  if (tok_offsets[0] != 0) return set_error(TM_E_INVALID, "tok_offsets[0] must be 0");
  for (uint32_t d = 0; d < ndocs; d++) if (tok_offsets[d + 1] < tok_offsets[d]) return set_error(TM_E_INVALID, "tok_offsets not monotone");
  hipError_t e = hipSuccess;
  uint32_t *d_tok = nullptr, *d_len = nullptr;
  uint64_t *d_off = nullptr, *d_sums = nullptr, *d_total = nullptr, *d_toff = nullptr, *d_doff = nullptr;
  uint8_t* d_out = nullptr;
  int rc = TM_OK;
  const uint32_t sblocks = (uint32_t)((n + 1 + SCAN_CH - 1) / SCAN_CH) + 2;
  std::vector<uint64_t> doff((size_t)ndocs + 1, 0);
  std::vector<uint8_t> rawbytes;
  if ((e = hipMalloc((void**)&d_tok, (n + 1) * 4)) != hipSuccess || (e = hipMalloc((void**)&d_len, (n + 1) * 4)) != hipSuccess ||
      (e = hipMalloc((void**)&d_off, (n + 2) * 8)) != hipSuccess || (e = hipMalloc((void**)&d_sums, (uint64_t)sblocks * 8)) != hipSuccess ||
      (e = hipMalloc((void**)&d_total, 8)) != hipSuccess || (e = hipMalloc((void**)&d_toff, ((uint64_t)ndocs + 1) * 8)) != hipSuccess ||
      (e = hipMalloc((void**)&d_doff, ((uint64_t)ndocs + 1) * 8)) != hipSuccess) rc = hip_fail(e, "hipMalloc (decode)");
  if (rc == TM_OK && ((n && (e = hipMemcpy(d_tok, tokens, n * 4, hipMemcpyHostToDevice)) != hipSuccess) ||
                      (e = hipMemcpy(d_toff, tok_offsets, ((uint64_t)ndocs + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess)) rc = hip_fail(e, "H2D tokens");
  uint64_t total = 0;
  if (rc == TM_OK) {
    if (n) TM_LAUNCH(k_dec_len, (uint32_t)((n + 255) / 256), 256, 0, 0, d_tok, n, v->d_rev_off, v->host.n_ids, d_len);
    scan_u32(d_len, n, d_sums, d_total, d_off, nullptr);
    TM_LAUNCH(k_dec_doc_off, (ndocs + 256) / 256, 256, 0, 0, d_off, d_toff, ndocs, d_doff);
    if ((e = hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(doff.data(), d_doff, doff.size() * 8, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "decode lengths");
  }
  if (rc == TM_OK && (e = hipMalloc((void**)&d_out, total + 16)) != hipSuccess) rc = hip_fail(e, "hipMalloc (decode output)");
  if (rc == TM_OK) {
    if (n) TM_LAUNCH(k_dec_copy, (uint32_t)((n + 255) / 256), 256, 0, 0, d_tok, n, v->d_rev_off, v->d_rev_bytes, v->host.n_ids, d_off, d_out);
    rawbytes.resize(total);
    if (total && (e = hipMemcpy(rawbytes.data(), d_out, total, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H decoded bytes");
  }
  void* frees[] = {d_tok, d_len, d_off, d_sums, d_total, d_toff, d_doff, d_out};
  for (void* q : frees) (void)hipFree(q);
  if (rc != TM_OK) return rc;
  if (raw || v->host.capcode == 0) {
    std::memcpy(out_offsets, doff.data(), doff.size() * 8);
    if (total > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)total);
    if (total) std::memcpy(out, rawbytes.data(), total);
    return TM_OK;
  }
  std::vector<std::vector<uint8_t>> outs;
  capcode_decode_batch(rawbytes.data(), doff.data(), ndocs, v->host.capcode, 0, outs);
  uint64_t o = 0;
  for (uint32_t d = 0; d < ndocs; d++) { out_offsets[d] = o; o += outs[d].size(); }
  out_offsets[ndocs] = o;
  if (o > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)o);
  for (uint32_t d = 0; d < ndocs; d++) if (!outs[d].empty()) std::memcpy(out + out_offsets[d], outs[d].data(), outs[d].size());
  return TM_OK;
}

