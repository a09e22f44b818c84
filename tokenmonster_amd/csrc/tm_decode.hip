// tm_decode.hip — the kernels of Decode / decode_raw (tm_decode_batch, include/tokenmonster_hip.h; the entry point itself borrows a lane: tm_host.hip).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "tm_internal.h"
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
// made of ASCII, the two-byte characters U+0080..U+07FF (Latin, the combining marks, Greek, Cyrillic, Armenian, Hebrew, Arabic ...) and the
// three-byte characters without case (punctuation, CJK, kana, symbols ...) — there the decoder is a four-bit state machine over CHARACTERS: 'D' deletes the
// next character, 'C' capitalises the next one that is not a (kept) space, 'W' capitalises letters until the word ends, and a space
// straight after 'W' does not end it.  One wavefront per document walks it 64 bytes at a time; inside a chunk each flag is a flood fill on
// the ballots of the byte classes, done with the carry chain of ONE 64-bit addition: with P the positions a flag survives, S where it is
// set (S inside P) and the flag's value on entry as carry-in, (P + S + carry) ^ P has a one from every start up to and INCLUDING the
// first position outside P above it — the position that sees the flag and consumes or clears it — and the carry out of bit 63 is the flag's
// value for the next chunk.  A character is decided at its first byte; its other bytes are transparent to every flag and take the decision
// (kept / capitalised) of the first, also across the end of a chunk.  Capitalising a two-byte letter replaces its two bytes by those of
// its upper-case form (`tab`, built by the host from the host decoder's own functions: tm_normalize.cpp build_dec_tables; р D1 80 -> Р D0 A0
// changes the lead byte too); four-byte characters whose block of 64 code points is caseless throughout (emoji, symbols, the ideographs of
// plane 2 ...) are passed on like the three-byte ones; a character whose upper-case form has another length, a three- or four-byte letter
// with case, and any byte sequence that is not well-formed UTF-8 leave the document to the host decoder (dec_len = DEC_HOST).
__device__ __forceinline__ unsigned long long dec_fill(unsigned long long P, unsigned long long S, unsigned& carry) {
  const unsigned long long t = P + S, u = t + carry;
  carry = (t < P) | (u < t);
  return u ^ P;
}
__device__ __forceinline__ uint32_t dec_tab_index(uint32_t lead, uint32_t second) {      // lead in C2..DF
  return ((lead - 0xC2u) << 6) | (second & 63u);
}
// code of the three-byte character lead b1 b2 (tm_internal.h: 0 host, 1 passed on, 2 digit or mark)
__device__ __forceinline__ uint32_t dec_three_code(const uint32_t* __restrict__ tab, uint32_t lead, uint32_t b1, uint32_t b2) {
  const uint32_t cp = ((lead & 15u) << 12) | ((b1 & 63u) << 6) | (b2 & 63u);
  const uint32_t bc = (tab[DEC_TWO + (cp >> 10)] >> (2u * ((cp >> 6) & 15u))) & 3u;
  return bc != 3u ? bc : ((tab[DEC_TWO + DEC_BLK_WORDS + (cp >> 4)] >> (2u * (cp & 15u))) & 3u);
}
// the same for the four-byte character lead b1 b2 b3: a code per block of 64 code points of the planes 1..16 (0 also for what is not one)
__device__ __forceinline__ uint32_t dec_four_code(const uint32_t* __restrict__ tab, uint32_t lead, uint32_t b1, uint32_t b2, uint32_t b3) {
  const uint32_t cp = ((lead & 7u) << 18) | ((b1 & 63u) << 12) | ((b2 & 63u) << 6) | (b3 & 63u);
  if (cp - 0x10000u >= 0x100000u) return 0u;
  const uint32_t blk = (cp - 0x10000u) >> 6;
  const uint32_t code = (tab[DEC_TWO + DEC_BLK_WORDS + DEC_CP_WORDS + (blk >> 4)] >> (2u * (blk & 15u))) & 3u;
  return code == 3u ? 0u : code;
}
__global__ __launch_bounds__(256) void k_dec_capcode(const uint8_t* __restrict__ in, const uint64_t* __restrict__ doc_off, uint32_t ndocs,
                                                      uint8_t* __restrict__ out, uint64_t* __restrict__ dec_len, const uint32_t* __restrict__ tab) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t d = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (d >= ndocs) return;
  const uint64_t b = doc_off[d], e = doc_off[d + 1];
  const unsigned long long below = (1ull << lane) - 1ull;
  uint64_t o = b;                                        // decoded bytes of the document go to out[b ..): never more than it had
  unsigned c_del = 0, c_char = 0, c_word = 0, c_ign = 0;   // the decoder's state (tm_internal.h CapcodeState) between chunks
  unsigned long long kept_in = 0, cap_in = 0;              // bytes at the start of this chunk that continue a character of the chunk before: kept / capitalised
  bool host = false;
  for (uint64_t pos = b; pos < e; pos += 64) {
    const uint64_t at = pos + lane;
    const bool valid = at < e;
    const uint32_t c = valid ? in[at] : 0u;
    const bool hi = __any(c >= 0x80u);
    uint32_t cn = 0, cnn = 0, cn3 = 0, cp = 0, cpp = 0, cp3 = 0;   // the bytes around it (inside the document), only looked at when the chunk is not pure ASCII
    if (hi) {
      cn = at + 1 < e ? in[at + 1] : 0u; cnn = at + 2 < e ? in[at + 2] : 0u; cn3 = at + 3 < e ? in[at + 3] : 0u;
      cp = valid && at >= b + 1 ? in[at - 1] : 0u; cpp = valid && at >= b + 2 ? in[at - 2] : 0u; cp3 = valid && at >= b + 3 ? in[at - 3] : 0u;
    }
    auto is_lead2 = [](uint32_t x) { return x - 0xC2u < 30u; };
    auto is_lead3 = [](uint32_t x) { return (x & 0xF0u) == 0xE0u; };
    auto is_cont = [](uint32_t x) { return (x & 0xC0u) == 0x80u; };
    auto is_lead4 = [](uint32_t x) { return x - 0xF0u < 5u; };
    const bool ascii = c < 0x80u;
    const bool lead2 = valid && is_lead2(c), lead3 = valid && is_lead3(c), lead4 = valid && is_lead4(c);
    uint32_t te = 0, t3 = 0;                                // table entry of the two-byte character this lane starts or ends / code of the three- or four-byte character it starts
    if (lead2 && is_cont(cn)) te = tab[dec_tab_index(c, cn)];
    const bool tail2 = valid && is_cont(c) && is_lead2(cp);
    if (tail2) te = tab[dec_tab_index(cp, c)];
    if (lead3 && is_cont(cn) && is_cont(cnn)) t3 = dec_three_code(tab, c, cn, cnn);
    if (lead4 && is_cont(cn) && is_cont(cnn) && is_cont(cn3)) t3 = dec_four_code(tab, c, cn, cnn, cn3);
    // (the other bytes of a three- or four-byte character: the lane of its first byte vouches for it)
    const bool tail3a = valid && is_cont(c) && is_lead3(cp), tail3b = valid && is_cont(c) && is_cont(cp) && is_lead3(cpp);
    const bool tail4 = valid && is_cont(c) && (is_lead4(cp) || (is_cont(cp) && (is_lead4(cpp) || (is_cont(cpp) && is_lead4(cp3)))));
    const bool ok = !valid || ascii || (lead2 && (te & 1u)) || ((lead3 || lead4) && t3 != 0u) || (tail2 && (te & 1u)) || tail3a || tail3b || tail4;
    if (__any(!ok)) { host = true; break; }
    const unsigned long long V = __ballot(valid);
    const unsigned long long L2 = __ballot(lead2), L4 = __ballot(lead4), L3 = __ballot(lead3) | L4;      // (L3: characters of three bytes or more)
    const unsigned long long START = __ballot(valid && (ascii || lead2 || lead3 || lead4));
    const unsigned long long mC = __ballot(c == 'C'), mW = __ballot(c == 'W'), mD = __ballot(c == 'D');
    const unsigned long long M = mC | mW | mD, N = START & ~M;
    const unsigned long long SP = __ballot(c == ' ');
    const bool lower = c - 'a' < 26u;
    const unsigned long long LET = __ballot(lower || c - 'A' < 26u || (lead2 && (te & 2u))) & N;      // upper- or lower-case letters
    // what keeps a capitalised word going besides letters: digits, the apostrophe and U+2019, marks
    const unsigned long long WC = __ballot(c - '0' < 10u || c == '\'' || (lead2 && (te & 4u)) || ((lead3 || lead4) && (t3 == 2u || (c == 0xE2u && cn == 0x80u && cnn == 0x99u))));
    const unsigned long long deleted = N & dec_fill(M, mD, c_del);               // a 'D' since the last character: this one goes
    const unsigned long long ign = N & dec_fill(M, mW, c_ign);                   // a 'W' since the last character
    const unsigned long long kept = N & ~deleted;
    const unsigned long long K = kept & ~SP;                                      // the characters that use up a pending 'C'
    const unsigned long long in_char = K & dec_fill(~(mW | K), mC, c_char);
    const unsigned long long ends_word = mC | (SP & kept & ~ign) | (K & ~LET & ~WC);
    const unsigned long long in_word = dec_fill(~ends_word, mW, c_word);
    const unsigned long long capS = in_char | (in_word & LET & K);                // characters that come out in upper case
    // the other bytes of a character: kept / capitalised like its first byte
    const unsigned long long k23 = kept & (L2 | L3), k3 = kept & L3, k4 = kept & L4, cap2 = capS & L2;
    const unsigned long long kept_all = kept | (k23 << 1) | (k3 << 2) | (k4 << 3) | kept_in;
    const unsigned long long cap_tail = (cap2 << 1) | cap_in;
    kept_in = (k23 >> 63) | (k3 >> 62) | (k4 >> 61);          // what the shifts above push beyond bit 63: the first bytes of the next chunk
    cap_in = cap2 >> 63;
    uint32_t oc = c;
    if (lower && ((capS >> lane) & 1ull)) oc = c - 32u;
    if (lead2 && ((capS >> lane) & 1ull)) oc = (te >> 8) & 0xFFu;
    if (tail2 && ((cap_tail >> lane) & 1ull)) oc = (te >> 16) & 0xFFu;
    if ((kept_all >> lane) & 1ull) out[o + (uint64_t)__popcll(kept_all & below)] = (uint8_t)oc;
    o += (uint64_t)__popcll(kept_all);
  }
  if (lane == 0) dec_len[d] = host ? DEC_HOST : o - b;
}
// out[0] = bytes of the documents the device decoded, out[1] = documents it left to the host decoder (one workgroup)
__global__ void k_dec_sum(const uint64_t* __restrict__ dec_len, uint32_t ndocs, uint64_t* __restrict__ out) {
  __shared__ unsigned long long s_bytes, s_host;
  if (threadIdx.x == 0) { s_bytes = 0; s_host = 0; }
  __syncthreads();
  unsigned long long bytes = 0, host = 0;
  for (uint32_t d = threadIdx.x; d < ndocs; d += blockDim.x) { const uint64_t l = dec_len[d]; if (l == DEC_HOST) host++; else bytes += l; }
  atomicAdd(&s_bytes, bytes); atomicAdd(&s_host, host);
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = s_bytes; out[1] = s_host; }
}
}  // namespace tmh

namespace tmh {
// the three stages of a decode on a stream, in buffers of the caller (tm_host.hip: a lane's grow-only arenas)
void launch_decode_lengths(const tm_vocab* v, const uint32_t* d_tok, uint64_t n, const uint64_t* d_toff, uint32_t ndocs, uint32_t* d_len, uint64_t* d_off,
                           uint64_t* d_sums, uint64_t* d_total, uint64_t* d_doff, hipStream_t st) {
  if (n) TM_LAUNCH(k_dec_len, (uint32_t)((n + 255) / 256), 256, 0, st, d_tok, n, v->d_rev_off, v->host.n_ids, d_len);
  note_table_use(v, st);
  scan_u32(d_len, n, d_sums, d_total, d_off, st);
  TM_LAUNCH(k_dec_doc_off, (ndocs + 256) / 256, 256, 0, st, d_off, d_toff, ndocs, d_doff);
}
void launch_decode_copy(const tm_vocab* v, const uint32_t* d_tok, uint64_t n, const uint64_t* d_off, uint8_t* d_out, hipStream_t st) {
  if (n) TM_LAUNCH(k_dec_copy, (uint32_t)((n + 255) / 256), 256, 0, st, d_tok, n, v->d_rev_off, v->d_rev_bytes, v->host.n_ids, d_off, d_out);
  note_table_use(v, st);
}
// the decoder's table of two-byte characters (vocabulary-independent): one copy per device, made on first use and kept
static const uint32_t* dec_table(int device) {
  static std::mutex mu;
  static const uint32_t* tabs[64] = {};
  static std::vector<uint32_t> h;
  std::lock_guard<std::mutex> g(mu);
  if (device < 0 || device >= 64) return nullptr;
  if (!tabs[device]) {
    if (h.empty()) { h.resize(DEC_TABLE_WORDS); build_dec_tables(h.data(), h.data() + DEC_TWO, h.data() + DEC_TWO + DEC_BLK_WORDS, h.data() + DEC_TWO + DEC_BLK_WORDS + DEC_CP_WORDS); }
    uint32_t* dp = nullptr;
    if (hipMalloc((void**)&dp, h.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(dp, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dp); return nullptr; }
    tabs[device] = dp;
  }
  return tabs[device];
}
int launch_decode_capcode(const tm_vocab* v, const uint8_t* d_out, const uint64_t* d_doff, uint32_t ndocs, uint8_t* d_dec, uint64_t* d_declen, hipStream_t st) {
  const uint32_t* tab = dec_table(v->device);
  if (!tab) return set_error(TM_E_HIP, "the decoder's character table could not be placed on device %d", v->device);
  if (ndocs) TM_LAUNCH(k_dec_capcode, (ndocs + 3) / 4, 256, 0, st, d_out, d_doff, ndocs, d_dec, d_declen, tab);
  return TM_OK;
}
}  // namespace tmh

// ---- device-resident decode of the ids a batch holds -------------------------------------------------------------------------------------------
// What tm_decode_batch (tm_host.hip) does between its upload and its download, on buffers of the batch: ids in HBM -> text in HBM.  The
// decode leg of bench.py times this; a caller that keeps token streams on the device (a detokenizing server beside the tokenizing one)
// gets its text without the ids ever crossing the host link.
extern "C" {

static int batch_decode(tm_batch* b, int raw, void* stream, uint64_t* decoded_bytes, uint32_t* host_docs, float* ms);
int tm_batch_decode(tm_batch* b, int raw, void* stream, uint64_t* decoded_bytes, uint32_t* host_docs) { return batch_decode(b, raw, stream, decoded_bytes, host_docs, nullptr); }
int tm_batch_decode_timed(tm_batch* b, int raw, void* stream, uint64_t* decoded_bytes, uint32_t* host_docs, float* ms) {
  if (!ms) return set_error(TM_E_INVALID, "null argument");
  return batch_decode(b, raw, stream, decoded_bytes, host_docs, ms);
}
// ms (may be null): HIP events on `stream` around [0] lengths + scan + document offsets, [1] the gather (k_dec_copy), [2] capcode decoding (k_dec_capcode)
static int batch_decode(tm_batch* b, int raw, void* stream, uint64_t* decoded_bytes, uint32_t* host_docs, float* ms) {
  if (!b) return set_error(TM_E_INVALID, "null argument");
  const tm_vocab* v = b->vocab;
  { int rc = enter_device(v); if (rc != TM_OK) return rc; }
  { int rc = ensure_output(b); if (rc != TM_OK) return rc; }          // (the ids are all there: the emit stage is repeated if its buffer was too small)
  hipStream_t st = (hipStream_t)stream;
  const uint32_t nd = b->ndocs;
  const uint64_t n = nd ? b->last_totals[1] : 0;
  b->dec_ndocs = nd; b->dec_total = 0; b->dec_capcode = false; b->dec_raw = raw != 0;
  if (decoded_bytes) *decoded_bytes = 0;
  if (host_docs) *host_docs = 0;
  if (ms) ms[0] = ms[1] = ms[2] = 0.f;
  if (nd == 0) return TM_OK;
  hipError_t e;
  if (ms && !b->have_events) {
    for (auto& ev : b->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return hip_fail(e, "hipEventCreate");
    b->have_events = true;
  }
  auto mark = [&](int k) { if (ms) (void)hipEventRecord(b->ev[k], st); };
  auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
  auto grow = [&](uint8_t** p, uint64_t* cap, uint64_t need) -> hipError_t {
    if (*cap >= need) return hipSuccess;
    if (*cap) trace_grow("decode arena", need);
    (void)hipFree(*p);
    *p = nullptr;
    *cap = need + need / 8 + 4096;
    return hipMalloc((void**)p, *cap);
  };
  const uint64_t sblocks = (n + 1 + SCAN_CH - 1) / SCAN_CH + 2;
  const uint64_t o_len = 0, o_off = o_len + up((n + 1) * 4), o_sums = o_off + up((n + 2) * 8), o_total = o_sums + up(sblocks * 8), o_doff = o_total + 256,
                 o_declen = o_doff + up(((uint64_t)nd + 1) * 8), a_bytes = o_declen + up((uint64_t)nd * 8 + 8);
  if ((e = grow(&b->d_dec_a, &b->dec_a_cap, a_bytes)) != hipSuccess) { b->dec_a_cap = 0; return hip_fail(e, "hipMalloc (decode)"); }
  uint8_t* A = b->d_dec_a;
  uint32_t* d_len = (uint32_t*)(A + o_len); uint64_t* d_off = (uint64_t*)(A + o_off); uint64_t* d_sums = (uint64_t*)(A + o_sums); uint64_t* d_total = (uint64_t*)(A + o_total);
  uint64_t* d_doff = (uint64_t*)(A + o_doff); uint64_t* d_declen = (uint64_t*)(A + o_declen);
  mark(0);
  launch_decode_lengths(v, b->d_out, n, b->d_tok_offsets, nd, d_len, d_off, d_sums, d_total, d_doff, st);
  mark(1);
  uint64_t total = 0;
  { int rc = small_d2h(b, &total, d_total, 8, st); if (rc == TM_OK) rc = small_sync(b, st); if (rc != TM_OK) return rc; }
  const uint64_t o_dec = up(total + 16);
  if ((e = grow(&b->d_dec_b, &b->dec_b_cap, o_dec + up(total + 16))) != hipSuccess) { b->dec_b_cap = 0; return hip_fail(e, "hipMalloc (decode output)"); }
  mark(2);
  launch_decode_copy(v, b->d_out, n, d_off, b->d_dec_b, st);
  mark(3);
  const bool dev_capcode = !raw && v->host.capcode == 2 && v->host.charset == 1;
  uint64_t sum[2] = {total, 0};
  if (dev_capcode) {
    int rc = launch_decode_capcode(v, b->d_dec_b, d_doff, nd, b->d_dec_b + o_dec, d_declen, st);
    if (rc != TM_OK) return rc;
    mark(4);
    TM_LAUNCH(k_dec_sum, 1, 256, 0, st, d_declen, nd, d_total);            // bytes the device decoded, documents it left to the host
    if ((rc = small_d2h(b, sum, d_total, 16, st)) != TM_OK) return rc;
  }
  { int rc = small_sync(b, st); if (rc != TM_OK) return rc; }
  if (ms) {
    (void)hipEventElapsedTime(&ms[0], b->ev[0], b->ev[1]);
    (void)hipEventElapsedTime(&ms[1], b->ev[2], b->ev[3]);
    if (dev_capcode) (void)hipEventElapsedTime(&ms[2], b->ev[3], b->ev[4]);
  }
  b->dec_total = total; b->dec_o_doff = o_doff; b->dec_o_declen = o_declen; b->dec_o_dec = o_dec; b->dec_capcode = dev_capcode;
  // (a capcode-1 or UTF-16 vocabulary, or one without capcode that was not asked for the raw form: every document is the host decoder's)
  const bool all_host = !raw && v->host.capcode != 0 && !dev_capcode;
  if (decoded_bytes) *decoded_bytes = all_host ? 0 : sum[0];
  if (host_docs) *host_docs = all_host ? nd : (uint32_t)sum[1];
  return TM_OK;
}

int tm_batch_decoded_download(tm_batch* b, uint8_t* out, uint64_t out_cap, uint64_t* out_offsets) {
  if (!b || !out_offsets) return set_error(TM_E_INVALID, "null argument");
  const tm_vocab* v = b->vocab;
  { int rc = enter_device(v); if (rc != TM_OK) return rc; }
  const uint32_t nd = b->dec_ndocs;
  out_offsets[0] = 0;
  if (nd == 0) return TM_OK;
  hipError_t e;
  const uint64_t total = b->dec_total;
  std::vector<uint64_t> doff((size_t)nd + 1), declen(nd, 0);
  std::vector<uint8_t> enc(total + 16), dec(b->dec_capcode ? total + 16 : 0);
  if ((e = hipMemcpy(doff.data(), b->d_dec_a + b->dec_o_doff, doff.size() * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
      (total && (e = hipMemcpy(enc.data(), b->d_dec_b, total, hipMemcpyDeviceToHost)) != hipSuccess)) return hip_fail(e, "D2H decoded bytes");
  if (b->dec_capcode && ((e = hipMemcpy(declen.data(), b->d_dec_a + b->dec_o_declen, (size_t)nd * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
                         (total && (e = hipMemcpy(dec.data(), b->d_dec_b + b->dec_o_dec, total, hipMemcpyDeviceToHost)) != hipSuccess))) return hip_fail(e, "D2H decoded text");
  const bool plain = b->dec_raw || v->host.capcode == 0;      // no capcode to undo: the gathered bytes are the text
  // the documents the device left alone go through the host decoder, as in tm_decode_batch
  std::vector<uint32_t> todo;
  if (!plain) for (uint32_t d = 0; d < nd; d++) if (!b->dec_capcode || declen[d] == DEC_HOST) todo.push_back(d);
  std::vector<std::vector<uint8_t>> touts;
  if (!todo.empty()) {
    std::vector<uint64_t> toff(todo.size() + 1, 0);
    std::vector<uint8_t> tbytes;
    for (size_t k = 0; k < todo.size(); k++) { tbytes.insert(tbytes.end(), enc.begin() + doff[todo[k]], enc.begin() + doff[todo[k] + 1]); toff[k + 1] = tbytes.size(); }
    capcode_decode_batch(tbytes.data(), toff.data(), (uint32_t)todo.size(), v->host.capcode, 0, touts);
  }
  std::vector<uint64_t> slot((size_t)nd, DEC_HOST);
  for (size_t k = 0; k < todo.size(); k++) slot[todo[k]] = k;
  uint64_t o = 0;
  for (uint32_t d = 0; d < nd; d++) { out_offsets[d] = o; o += slot[d] != DEC_HOST ? touts[slot[d]].size() : (plain ? doff[d + 1] - doff[d] : declen[d]); }
  out_offsets[nd] = o;
  if (o > out_cap) return set_error(TM_E_NOSPACE, "out_cap %llu < %llu required", (unsigned long long)out_cap, (unsigned long long)o);
  for (uint32_t d = 0; d < nd; d++) {
    const uint64_t len = out_offsets[d + 1] - out_offsets[d];
    if (!len) continue;
    if (slot[d] != DEC_HOST) std::memcpy(out + out_offsets[d], touts[slot[d]].data(), len);
    else std::memcpy(out + out_offsets[d], (plain ? enc.data() : dec.data()) + doff[d], len);
  }
  return TM_OK;
}

}  // extern "C"
