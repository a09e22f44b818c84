// tm_decode.hip — the kernels of Decode / decode_raw (tm_decode_batch, include/tokenmonster_hip.h; the entry point itself borrows a lane: tm_host.hip).
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
void launch_decode_capcode(const uint8_t* d_out, const uint64_t* d_doff, uint32_t ndocs, uint8_t* d_dec, uint64_t* d_declen, hipStream_t st) {
  if (ndocs) TM_LAUNCH(k_dec_capcode, (ndocs + 3) / 4, 256, 0, st, d_out, d_doff, ndocs, d_dec, d_declen);
}
}  // namespace tmh
