// tm_decode.hip — Decode / decode_raw behind tm_decode_batch (include/tokenmonster_hip.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "tm_pipeline.h"

using namespace tmh;

// ================================================================================================
// Decode (go/tokenmonster.go:445-550 Decode; tokenmonster.cpp:1404-1425): ids -> bytes
// ================================================================================================
// reverse[id] lengths -> exclusive scan -> copy.  Ids >= n_ids are skipped like the reference does.  Capcode decoding (a per-document
// state machine, javascript/tokenmonster.js:1007-1065) follows: on the device for the pure-ASCII documents of a capcode-2 UTF-8
// vocabulary (k_dec_capcode), on the host for the rest (Unicode case and character classes need ICU).
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

// Capcode level 2 decoding (javascript/tokenmonster.js:1007-1065; the host form is capcode_decode_stream, tm_normalize.cpp) of documents
// that are pure ASCII — there NFD and Unicode case play no part and the decoder is a four-bit state machine over bytes: 'D' deletes the
// next character, 'C' capitalises the next one that is not a (kept) space, 'W' capitalises letters until the word ends, and a space
// straight after 'W' does not end it.  One wavefront per document walks it 64 bytes at a time; inside a chunk each flag is a flood fill on
// the ballots of the byte classes, done with the carry chain of ONE 64-bit addition: with P the positions a flag survives, S where it is
// set (S inside P) and the flag's value on entry as carry-in, (P + S + carry) ^ P has a one from every start up to and INCLUDING the
// first position outside P above it — the position that sees the flag and consumes or clears it — and the carry out of bit 63 is the flag's
// value for the next chunk.  Documents with any byte >= 0x80 are left to the host decoder (dec_len = DEC_HOST).
constexpr uint64_t DEC_HOST = ~0ull;
__device__ __forceinline__ unsigned long long dec_fill(unsigned long long P, unsigned long long S, unsigned& carry) {
  const unsigned long long t = P + S, u = t + carry;
  carry = (t < P) | (u < t);
  return u ^ P;
}
__global__ __launch_bounds__(256) void k_dec_capcode(const uint8_t* __restrict__ in, const uint64_t* __restrict__ doc_off, uint32_t ndocs,
                                                      uint8_t* __restrict__ out, uint64_t* __restrict__ dec_len) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t d = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (d >= ndocs) return;
  const uint64_t b = doc_off[d], e = doc_off[d + 1];
  const unsigned long long below = (1ull << lane) - 1ull;
  uint64_t o = b;                                        // decoded bytes of the document go to out[b ..): never more than it had
  unsigned c_del = 0, c_char = 0, c_word = 0, c_ign = 0;   // the decoder's state (tm_internal.h CapcodeState) between chunks
  bool host = false;
  for (uint64_t pos = b; pos < e; pos += 64) {
    const bool valid = pos + lane < e;
    const uint32_t c = valid ? in[pos + lane] : 0u;
    if (__any(c >= 0x80u)) { host = true; break; }
    const unsigned long long V = __ballot(valid);
    const unsigned long long mC = __ballot(c == 'C'), mW = __ballot(c == 'W'), mD = __ballot(c == 'D');
    const unsigned long long M = mC | mW | mD, N = V & ~M;
    const unsigned long long SP = __ballot(c == ' ');
    const bool lower = c - 'a' < 26u;
    const unsigned long long LET = __ballot(lower || c - 'A' < 26u) & N;
    const unsigned long long WC = __ballot(c - '0' < 10u || c == '\'');          // what keeps a capitalised word going besides letters
    const unsigned long long deleted = N & dec_fill(M, mD, c_del);               // a 'D' since the last character: this one goes
    const unsigned long long ign = N & dec_fill(M, mW, c_ign);                   // a 'W' since the last character
    const unsigned long long kept = N & ~deleted;
    const unsigned long long K = kept & ~SP;                                      // the characters that use up a pending 'C'
    const unsigned long long in_char = K & dec_fill(~(mW | K), mC, c_char);
    const unsigned long long ends_word = mC | (SP & kept & ~ign) | (K & ~LET & ~WC);
    const unsigned long long in_word = dec_fill(~ends_word, mW, c_word);
    const bool cap = lower && (((in_char | (in_word & LET & K)) >> lane) & 1ull);
    if ((kept >> lane) & 1ull) out[o + (uint64_t)__popcll(kept & below)] = (uint8_t)(cap ? c - 32u : c);
    o += (uint64_t)__popcll(kept);
  }
  if (lane == 0) dec_len[d] = host ? DEC_HOST : o - b;
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
  uint8_t *d_out = nullptr, *d_dec = nullptr;
  uint64_t* d_declen = nullptr;
  const bool dev_capcode = !raw && v->host.capcode == 2 && v->host.charset == 1 && ndocs > 0;
  std::vector<uint64_t> declen;
  std::vector<uint8_t> decbytes;
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
    bool need_raw = !dev_capcode;
    if (dev_capcode) {
      // capcode decoding of the ASCII documents where the bytes are (out of place: the host decoder needs the others as they were)
      if ((e = hipMalloc((void**)&d_dec, total + 16)) != hipSuccess || (e = hipMalloc((void**)&d_declen, (uint64_t)ndocs * 8)) != hipSuccess) rc = hip_fail(e, "hipMalloc (capcode decode)");
      if (rc == TM_OK) {
        TM_LAUNCH(k_dec_capcode, (ndocs + 3) / 4, 256, 0, 0, d_out, d_doff, ndocs, d_dec, d_declen);
        declen.resize(ndocs);
        decbytes.resize(total);
        if ((e = hipMemcpy(declen.data(), d_declen, (uint64_t)ndocs * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
            (total && (e = hipMemcpy(decbytes.data(), d_dec, total, hipMemcpyDeviceToHost)) != hipSuccess)) rc = hip_fail(e, "D2H decoded text");
        for (uint32_t d = 0; d < ndocs && !need_raw; d++) need_raw = declen[d] == DEC_HOST;
      }
    }
    if (rc == TM_OK && need_raw) {
      rawbytes.resize(total);
      if (total && (e = hipMemcpy(rawbytes.data(), d_out, total, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H decoded bytes");
    }
  }
  void* frees[] = {d_tok, d_len, d_off, d_sums, d_total, d_toff, d_doff, d_out, d_dec, d_declen};
  for (void* q : frees) (void)hipFree(q);
  if (rc != TM_OK) return rc;
  if (raw || v->host.capcode == 0) {
    std::memcpy(out_offsets, doff.data(), doff.size() * 8);
    if (total > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)total);
    if (total) std::memcpy(out, rawbytes.data(), total);
    return TM_OK;
  }
  std::vector<std::vector<uint8_t>> outs;
  if (dev_capcode) {
    // the documents the device left alone (anything beyond ASCII) go through the host decoder, the others are where the kernel put them
    std::vector<uint32_t> todo;
    for (uint32_t d = 0; d < ndocs; d++) if (declen[d] == DEC_HOST) todo.push_back(d);
    outs.assign(ndocs, {});
    if (!todo.empty()) {
      std::vector<uint64_t> toff(todo.size() + 1, 0);
      std::vector<uint8_t> tbytes;
      for (size_t k = 0; k < todo.size(); k++) {
        tbytes.insert(tbytes.end(), rawbytes.begin() + (ptrdiff_t)doff[todo[k]], rawbytes.begin() + (ptrdiff_t)doff[todo[k] + 1]);
        toff[k + 1] = tbytes.size();
      }
      std::vector<std::vector<uint8_t>> touts;
      capcode_decode_batch(tbytes.data(), toff.data(), (uint32_t)todo.size(), 2, 0, touts);
      for (size_t k = 0; k < todo.size(); k++) outs[todo[k]].swap(touts[k]);
    }
    uint64_t o = 0;
    for (uint32_t d = 0; d < ndocs; d++) { out_offsets[d] = o; o += declen[d] == DEC_HOST ? outs[d].size() : declen[d]; }
    out_offsets[ndocs] = o;
    if (o > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)o);
    for (uint32_t d = 0; d < ndocs; d++) {
      if (declen[d] == DEC_HOST) { if (!outs[d].empty()) std::memcpy(out + out_offsets[d], outs[d].data(), outs[d].size()); }
      else if (declen[d]) std::memcpy(out + out_offsets[d], decbytes.data() + doff[d], declen[d]);
    }
    return TM_OK;
  }
  capcode_decode_batch(rawbytes.data(), doff.data(), ndocs, v->host.capcode, 0, outs);
  uint64_t o = 0;
  for (uint32_t d = 0; d < ndocs; d++) { out_offsets[d] = o; o += outs[d].size(); }
  out_offsets[ndocs] = o;
  if (o > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)o);
  for (uint32_t d = 0; d < ndocs; d++) if (!outs[d].empty()) std::memcpy(out + out_offsets[d], outs[d].data(), outs[d].size());
  return TM_OK;
}

