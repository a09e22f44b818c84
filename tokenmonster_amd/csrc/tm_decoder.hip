// tm_decoder.hip — the streaming Decoder of the reference (go/tokenmonster.go:552-700; tokenmonster-cpp/src/tokenmonster.cpp:1509-1721)
// behind tm_decoder_* (include/tokenmonster_hip.h).  A Decoder is per-connection HOST state: the bytes of a character that a call
// ended in the middle of (remainder) and the four flags of the capcode decoder; its calls carry a handful of ids each (a language
// model streams one token at a time), so the gather of reverse[id] runs on the host from the vocabulary's host copy of the
// reverse table — there is nothing for a GPU to do.  tm_decode_batch is the batch form that does run on the device.
#include <cstring>
#include <vector>

#include "tm_device.h"

using namespace tmh;

struct tm_decoder {
  const tm_vocab* v = nullptr;
  std::vector<uint8_t> remainder;   // tail of the last call that does not yet form a whole character
  std::vector<uint8_t> pending;     // decoded bytes not yet handed to the caller (a call whose buffer was too small)
  CapcodeState cc;
};

namespace {

// tokenmonster.cpp:83-110: how many bytes at the end of `b` belong to a character that is not complete yet
int incomplete_utf8(const std::vector<uint8_t>& b) {
  const int n = (int)b.size();
  if (n == 0 || (b[n - 1] & 0x80) == 0) return 0;
  int s = n - 1;
  while (s >= 0 && (b[s] & 0xC0) == 0x80) s--;
  if (s == -1) return n;
  const uint8_t f = b[s];
  int len;
  if ((f & 0x80) == 0) len = 1;
  else if ((f & 0xE0) == 0xC0) len = 2;
  else if ((f & 0xF0) == 0xE0) len = 3;
  else if ((f & 0xF8) == 0xF0) len = 4;
  else return n - s;
  if (n - s < len) return len - (n - s);      // (as the reference has it: the number of bytes still MISSING, tokenmonster.cpp:105-107)
  if (len == 1 && (b[s] & 0xC0) != 0) return n;
  return 0;
}
// tokenmonster.cpp:112-129
int incomplete_utf16(const std::vector<uint8_t>& b) {
  const int n = (int)b.size();
  if (n == 0) return 0;
  auto u16 = [&](int i) { return (uint16_t)(b[i] | (b[i + 1] << 8)); };
  if (n % 2 != 0) {
    if (n >= 3) { const uint16_t t = u16(n - 3); if (t >= 0xD800 && t <= 0xDBFF) return 3; }
    return 1;
  }
  const uint16_t last = u16(n - 2);
  if (last >= 0xD800 && last <= 0xDBFF) return 2;
  const uint16_t first = u16(0);
  if (first >= 0xDC00 && first <= 0xDFFF) return 2;
  return 0;
}

// the body shared by decode and decode_serialized (tokenmonster.cpp:1509-1540): remainder + reverse[id]..., cut at the last whole
// character, capcode-decode what is whole
template <typename NextId>
void decode_ids(tm_decoder* d, uint64_t n, NextId next_id) {
  const HostVocab& hv = d->v->host;
  std::vector<uint8_t> data;
  if (hv.charset == 0) {                        // no charset: bytes as they are, nothing is held back (:1511)
    for (uint64_t k = 0; k < n; k++) {
      const uint32_t id = next_id(k);
      if (id < hv.n_ids) data.insert(data.end(), hv.rev_bytes.begin() + hv.rev_off[id], hv.rev_bytes.begin() + hv.rev_off[id + 1]);
    }
    d->pending.insert(d->pending.end(), data.begin(), data.end());
    return;
  }
  data.swap(d->remainder);
  for (uint64_t k = 0; k < n; k++) {
    const uint32_t id = next_id(k);
    if (id < hv.n_ids) data.insert(data.end(), hv.rev_bytes.begin() + hv.rev_off[id], hv.rev_bytes.begin() + hv.rev_off[id + 1]);
  }
  const int inc = hv.charset == 1 ? incomplete_utf8(data) : incomplete_utf16(data);
  const size_t keep = data.size() - (size_t)std::min<size_t>((size_t)inc, data.size());
  d->remainder.assign(data.begin() + (ptrdiff_t)keep, data.end());
  if (hv.capcode == 2) capcode_decode_stream(d->cc, data.data(), keep, d->pending);
  else if (hv.capcode == 1) nocapcode_decode_stream(d->cc, data.data(), keep, d->pending);
  else d->pending.insert(d->pending.end(), data.begin(), data.begin() + (ptrdiff_t)keep);
}

int hand_over(tm_decoder* d, uint8_t* out, uint64_t out_cap, uint64_t* out_len) {
  if (out_len) *out_len = d->pending.size();
  if (d->pending.size() > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %zu bytes decoded (call again with no ids and a larger buffer)", (unsigned long long)out_cap, d->pending.size());
  if (!d->pending.empty()) std::memcpy(out, d->pending.data(), d->pending.size());
  d->pending.clear();
  return TM_OK;
}

}  // namespace

extern "C" {

int tm_decoder_new(const tm_vocab* v, tm_decoder** out) {
  if (!v || !out) return set_error(TM_E_INVALID, "null argument");
  if (v->host.rev_off.empty()) return set_error(TM_E_INVALID, "an imported vocabulary (tm_vocab_block_import) has no host tables: no streaming decoder");
  auto* d = new tm_decoder();
  d->v = v;
  *out = d;
  return TM_OK;
}
void tm_decoder_free(tm_decoder* d) { delete d; }

int tm_decoder_decode(tm_decoder* d, const uint32_t* tokens, uint64_t n, uint8_t* out, uint64_t out_cap, uint64_t* out_len) {
  if (!d || (n && !tokens)) return set_error(TM_E_INVALID, "null argument");
  if (n) decode_ids(d, n, [&](uint64_t k) { return tokens[k]; });
  return hand_over(d, out, out_cap, out_len);
}

int tm_decoder_decode_serialized(tm_decoder* d, const uint8_t* data, uint64_t nbytes, uint32_t encoding_length, uint8_t* out, uint64_t out_cap,
                                 uint64_t* out_len) {
  if (!d || (nbytes && !data)) return set_error(TM_E_INVALID, "null argument");
  if (encoding_length <= 1) encoding_length = d->v->host.n_ids <= 65536 ? 2 : 3;      // tokenmonster.cpp:1545-1551
  if (encoding_length < 2 || encoding_length > 4) return set_error(TM_E_INVALID, "Invalid encoding length");
  const uint64_t n = nbytes / encoding_length;                                            // a trailing partial id is ignored, as in the reference's loops
  if (n) decode_ids(d, n, [&](uint64_t k) {
    const uint8_t* p = data + k * encoding_length;
    uint32_t id = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
    if (encoding_length >= 3) id |= (uint32_t)p[2] << 16;
    if (encoding_length == 4) id |= (uint32_t)p[3] << 24;
    return id;
  });
  return hand_over(d, out, out_cap, out_len);
}

int tm_decoder_flush(tm_decoder* d, uint8_t* out, uint64_t out_cap, uint64_t* out_len) {      // tokenmonster.cpp:1717-1721
  if (!d) return set_error(TM_E_INVALID, "null argument");
  d->pending.insert(d->pending.end(), d->remainder.begin(), d->remainder.end());
  d->remainder.clear();
  return hand_over(d, out, out_cap, out_len);
}

}  // extern "C"
