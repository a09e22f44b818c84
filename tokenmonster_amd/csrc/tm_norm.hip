// tm_norm.hip — GPU normalizer behind tm_batch_upload_raw / tm_batch_normalize (include/tokenmonster_hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "tm_pipeline.h"
#include "tm_build.h"
#include "tm_norm_masks.h"

using namespace tmh;

// ================================================================================================
// GPU normalizer: the pre-step of Tokenize (go/tokenmonster.go:242-253: norm.Normalize then capcode.Encode)
// ================================================================================================
// Handles, entirely on the device, documents made of ASCII plus the NFD-stable General Punctuation block
// (U+2010..U+2027, U+2030..U+205E: curly quotes, dashes, ellipsis ...; U+2019 counts as an apostrophe exactly as in
// javascript/tokenmonster.js:878).  For these NFD is the identity (tokenmonster.cpp:190-198) and capcode level 2
// (javascript/tokenmonster.js:900-1005) becomes a LOCAL function of each character once three facts about its
// "capital run" are known.  A run is what the encoder's inWord state covers: it starts at the first capital of a
// maximal block of {capital, digit, apostrophe} characters and extends to the end of that block.  Needed per char:
//   ub   capitals before it in its block (0 = it is the run's first letter; >= 1 = inside the run)
//   ua   capitals after it in its block       (every later capital of a run is inside a run of several capitals)
//   tL   the character right after the block is a lowercase letter ('C' mode: "Hello"; otherwise 'W' mode: "HELLO")
// Documents are cut into 1 KiB pieces, one wavefront each (text staged in LDS):
//   pass 1  k_norm_summary  per piece: leading / trailing block-run summary, unsupported-byte flag
//   pass 2  k_norm_carry    per document: carries ub / ua / tL across piece boundaries (a few bits per piece)
//   pass 3  k_norm_emit<0>  per piece: output length            -> scan -> where every piece goes
//   pass 4  k_norm_emit<1>  per piece: the normalized bytes
// Any other document (other non-ASCII bytes: needs ICU for NFD / Unicode case) is normalized by the host normalizer
// (tm_normalize.cpp) and appended after the device-normalized documents; documents are (begin, end) ranges, so the
// layout of the device part never waits for the host.
namespace tmh {

constexpr int PIECE = 1024, PMARGIN = 68, PLDS = PIECE + 2 * PMARGIN;      // 64 bytes of margin either side (k_norm_emit2<false> takes its carries from them) + 4: the classifier looks three bytes either way
// character classes NC_* and flag bits NF_*: tm_norm_masks.h
// piece summary bits
constexpr uint32_t PS_WHOLE = 1u, PS_LEADU_SHIFT = 1, PS_LEADTL = 8u, PS_TRAILU_SHIFT = 5, PS_FIRSTBLOCK = 128u, PS_FIRSTL = 256u, PS_BAD = 512u;

__device__ __forceinline__ uint32_t ncls_ascii(uint32_t c, bool lower_all) {
  if (c - 'a' < 26u) return NC_L;
  if (c - 'A' < 26u) return lower_all ? NC_L : NC_U;
  if (c - '0' < 10u) return NC_N;
  if (c == '\'') return NC_AP;
  if (c == ' ') return NC_SP;
  return NC_O;
}
__device__ __forceinline__ bool nblock(uint32_t cls) { return (cls & NF_BLOCK) != 0; }       // capital, digit, apostrophe, combining mark

// (chg[c]: chunk c of the piece has a byte of a character that the pass CHANGES - a two-byte character, which takes its bytes from the table, or one
// NFD splits; set by norm_load_piece, looked at by k_norm_emit2, which leaves emit_high_byte alone for a chunk of nothing but inert characters)
struct PieceLds { alignas(16) uint8_t raw[PLDS]; alignas(4) uint8_t f[PLDS]; uint8_t chg[32]; };
// The normalizer's tables in ONE device buffer (tm_norm_masks.h): NmTwo[NM_TWO_SIZE] | block codes [NM_BLK_WORDS] | code-point codes
// [NM_CP_WORDS] | block codes of the four-byte characters [NM_BLK4_WORDS] | one word of switches (NM_MISC_*; 16 bytes with its padding).  The 256 work-items of a workgroup stage the entries of U+0080..U+017F (2 KB) and the block codes (256 bytes) in LDS.
constexpr size_t NM_TABLE_BYTES = NM_TWO_SIZE * sizeof(NmTwo) + (NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS + 4) * sizeof(uint32_t) + NM_LEA_SIZE * sizeof(NmLea) + NM_KANA_SIZE * sizeof(uint16_t) + NM_CCC_SIZE + (NM_DEC3_SIZE + NM_DEC3_THIRDS) * sizeof(uint32_t);      // (... | the characters of Latin Extended Additional, read where they lie)
struct TabLds { NmTwo two[NM_TWO_FAST]; uint32_t blk[NM_BLK_WORDS]; uint32_t misc; };
__device__ __forceinline__ NmTabs stage_tabs(TabLds& s, const NmTwo* __restrict__ two) {
  static_assert(NM_TWO_FAST == 256 && NM_BLK_WORDS <= 256, "one entry per work-item");
  const uint32_t* blk = reinterpret_cast<const uint32_t*>(two + NM_TWO_SIZE);
  s.two[threadIdx.x] = two[threadIdx.x];
  if (threadIdx.x < NM_BLK_WORDS) s.blk[threadIdx.x] = blk[threadIdx.x];
  const uint32_t misc = (uint32_t)__builtin_amdgcn_readfirstlane((int)blk[NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS]);
  if (threadIdx.x == 0) s.misc = misc;
  return NmTabs{s.two, two, s.blk, blk + NM_BLK_WORDS, blk + NM_BLK_WORDS + NM_CP_WORDS, misc,
                reinterpret_cast<const NmLea*>(blk + NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS + 4)};
}
// The out-of-line functions below get the tables as TWO POINTERS - the staged part in LDS, the whole in global memory - and put the NmTabs
// together themselves: as a 56-byte argument by value it went through scratch memory (a copy per call and lane), and on the device - not on
// the emulated one - a kernel whose out-of-line callee took a few bytes more of arguments faulted at address 0.
__device__ __forceinline__ NmTabs tabs_from(const TabLds* s, const NmTwo* __restrict__ two) {
  const uint32_t* blk = reinterpret_cast<const uint32_t*>(two + NM_TWO_SIZE);
  return NmTabs{s->two, two, s->blk, blk + NM_BLK_WORDS, blk + NM_BLK_WORDS + NM_CP_WORDS, s->misc, reinterpret_cast<const NmLea*>(blk + NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS + 4)};
}

// Bytes beyond ASCII are the exception (a dword of plain ASCII never gets here), and what they need is long: kept OUT of line, so that the
// unrolled loops of the kernels stay short enough for the instruction cache (inlined into every copy of a loop body, the classifier and the
// two-byte / Hangul output code made k_norm_emit2 57 % larger and 7 % slower on text that has none of these characters).
// the character that begins at raw_x, byte x of a staged range of `n` bytes (nm_classify_char; what lies outside the range reads as 0)
__device__ __noinline__ uint32_t classify_high_char(const uint8_t* raw_x, int x, int n, const TabLds* ts, const NmTwo* two) {
  const NmTabs tabs = tabs_from(ts, two);
  return nm_classify_char(raw_x[0], x >= 1 ? raw_x[-1] : 0u, x >= 2 ? raw_x[-2] : 0u, x >= 3 ? raw_x[-3] : 0u, x + 1 < n ? raw_x[1] : 0u, x + 2 < n ? raw_x[2] : 0u,
                          x + 3 < n ? raw_x[3] : 0u, tabs);
}
// what the lane of a byte >= 0x80 emits in k_norm_emit2 (tm_norm_masks.h): the bytes of a two-byte character come from the table - the lead
// lane its first byte, or the ASCII letter the character decomposes into; the second lane its second byte, or the two bytes of the combining
// mark -; every lane of a Hangul syllable (NFD) emits the three bytes of one of its jamo, or - the third lane of a syllable without a final
// consonant - nothing at all.  In: r = the byte in LDS, fl its class byte, code its rule-table entry; o = {o3, ysp, m3, len1} as the rule
// table left them.  Returns the same four, len1 = 0xFF for "no byte at all".
struct HighOut { uint32_t o3, ysp, m3, len1; };
__device__ __noinline__ HighOut emit_high_byte(const uint8_t* r, uint32_t fl, uint32_t code, HighOut o, const TabLds* ts, const NmTwo* two) {
  const NmTabs tabs = tabs_from(ts, two);
  const uint32_t b = r[0], bm1 = r[-1], bp1 = r[1];
  const bool lead2 = nm_two_lead(b), cont2 = nm_cont_byte(b) && nm_two_lead(bm1);
  if (lead2 || cont2) {
    const NmTwo e = nm_two_get(tabs, lead2 ? nm_two_index(b, bp1) : nm_two_index(bm1, b));
    uint32_t y = 0, mm = 0;
    const uint32_t extra = nm_two_out(e, cont2, (code & 4u) != 0, true, &o.o3, &y, &mm);
    if (extra >= 1u) { o.len1 = extra; o.ysp = y; o.m3 = mm; }
  } else {
    // a byte of a three-byte character that changes under NFD (one decoding of the character, then by its range - the characters that come here
    // for nothing, general punctuation and ideographs, leave after four compares)
    uint32_t role, cp;
    if (fl != NF_BAD && (tabs.misc & (NM_MISC_LEA | NM_MISC_KANA | NM_MISC_DEC3 | NM_MISC_HANGUL)) && nm_three_role(b, bm1, r[-2], bp1, r[2], &role, &cp)) {
      if ((tabs.misc & NM_MISC_LEA) && cp - 0x1E00u < (uint32_t)NM_LEA_SIZE) {
        // a letter of Latin Extended Additional: the letter, its first mark, its second mark or nothing
        const NmLea e = tabs.lea[cp - 0x1E00u];
        if (e.a & NT_OK) {
          if (role == 0u) o.o3 = (code & 4u) ? ((e.a >> 16) & 0xFFu) : ((e.a >> 8) & 0xFFu);
          else if (role == 1u) { o.len1 = 1u; o.ysp = e.b & 0xFFu; o.o3 = (e.b >> 8) & 0xFFu; }
          else if (((e.a >> 24) & 3u) == 2u) { o.len1 = 1u; o.ysp = (e.b >> 16) & 0xFFu; o.o3 = e.b >> 24; }
          else o.len1 = 0xFFu;
        }
      } else if ((tabs.misc & NM_MISC_KANA) && cp - 0x3040u < (uint32_t)NM_KANA_SIZE) {
        // a voiced kana: the base kana, the mark, nothing
        const uint32_t e = nm_kana_tab(tabs)[cp - 0x3040u];
        if (e & NK_OK) o.len1 = nm_kana_out(e, role, &o.m3, &o.ysp, &o.o3) ? 2u : 0xFFu;
      } else if ((tabs.misc & NM_MISC_DEC3) && cp - NM_DEC3_BASE < NM_DEC3_SIZE) {
        // a character NFD splits in two three-byte ones: the first, the second, nothing
        const uint32_t e = nm_dec3_tab(tabs)[cp - NM_DEC3_BASE];
        if (e & ND_OK) o.len1 = nm_dec3_out(tabs, e, role, &o.m3, &o.ysp, &o.o3) ? 2u : 0xFFu;
      } else if ((tabs.misc & NM_MISC_HANGUL) && nm_hangul(cp)) {
        // a Hangul syllable: one of its jamo - or nothing
        o.len1 = nm_hangul_out(cp, role, &o.m3, &o.ysp, &o.o3) ? 2u : 0xFFu;
      }
    }
  }
  return o;
}

// stage the piece (+ margins) and classify every byte; continuation bytes inherit the class of their lead byte.
// returns the piece length; LDS index of document byte (pb + i) is PMARGIN + i.
// The staged range is a WINDOW of the wavefront on the raw text (tm_device.h: a buffer resource cut to the document): what lies outside the
// document reads as 0 - class O - without a compare or a branch per dword (the loop this replaces spent 18 scalar instructions per step on
// 64-bit range tests and exec masks, on a kernel that is bound by its scalar instructions).
template <typename LDS>
__device__ __forceinline__ int norm_load_piece(LDS& L, const uint8_t* __restrict__ raw, uint64_t rb, uint64_t re, uint64_t pb, int lane,
                                               const uint8_t* s_cls, const TabLds* ts, const NmTwo* __restrict__ two) {
  const uint64_t left = re - pb;
  const uint32_t before = pb != rb ? (uint32_t)PMARGIN : 0u;            // (pb - rb is a multiple of PIECE: the whole margin in front, or none of it)
  const uint32_t after = left > (uint64_t)(PIECE + PMARGIN) ? (uint32_t)(PIECE + PMARGIN) : (uint32_t)left;
  const uint32_t a = (uint32_t)PMARGIN - before, e = (uint32_t)PMARGIN + after;      // LDS indices of the first byte of the document in the range, and one past its last
  const TmWindow win = tm_window(raw + (pb - before), (before + after) & ~3u);        // whole dwords only: a dword that straddles the document's end is outside
  constexpr int STEPS = (PLDS / 4 + 63) / 64;
  uint32_t w4[STEPS], high = 0u;
#pragma unroll
  for (int s = 0; s < STEPS; s++) w4[s] = tm_window_u32(win, (uint32_t)(4 * (lane + 64 * s)) - a);       // (a is a multiple of 4: the dwords of the range are the dwords of the window)
#pragma unroll
  for (int s = 0; s < STEPS; s++) {
    const int i = lane + 64 * s;
    if (s < STEPS - 1 || i < PLDS / 4) {
      const uint32_t v = w4[s];
      reinterpret_cast<uint32_t*>(L.raw)[i] = v;
      // four table reads classify a dword of ASCII; a dword with anything else in it is done again below, byte by byte
      reinterpret_cast<uint32_t*>(L.f)[i] = (uint32_t)s_cls[v & 0x7Fu] | ((uint32_t)s_cls[(v >> 8) & 0x7Fu] << 8) | ((uint32_t)s_cls[(v >> 16) & 0x7Fu] << 16) | ((uint32_t)s_cls[(v >> 24) & 0x7Fu] << 24);
      high |= v;
    }
  }
  if (lane < 32) L.chg[lane] = 0;
  // the last dword of the document may be a partial one: its bytes one by one (behind the zero the loop above has put there)
  __builtin_amdgcn_wave_barrier();
  const uint32_t tail = e & 3u;
  if ((uint32_t)lane < tail) {
    const uint32_t x = (e & ~3u) + (uint32_t)lane, bt = raw[pb + x - (uint32_t)PMARGIN];
    L.raw[x] = (uint8_t)bt;
    L.f[x] = bt < 0x80u ? s_cls[bt] : (uint8_t)0;
    high |= bt;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  if (__ballot((high & 0x80808080u) != 0u) != 0ull) {
    // bytes beyond ASCII somewhere in the range (the exception): their dwords again.  A character is classified ONCE, at its first byte, by the
    // classifier that looks three bytes either way, and the lane that holds that byte writes the class bytes of the whole character (text in
    // another script is two- and three-byte characters from end to end: a decoding per byte was most of what the pass took there - 13 - 17 ms
    // per GiB against 3); every byte beyond ASCII is NF_BAD beforehand, which is what is left of a byte no well-formed character claims.
#pragma unroll 1
    for (int i = lane; i < PLDS / 4; i += 64) {
      const uint32_t v = reinterpret_cast<const uint32_t*>(L.raw)[i];
      if ((v & 0x80808080u) != 0u) {
        uint32_t f4 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int x = 4 * i + q;
          const uint32_t b = (v >> (8 * q)) & 0xFFu;
          const uint32_t fl = b < 0x80u ? (uint32_t)s_cls[b] : ((x >= 3 && x < PLDS - 3) ? (uint32_t)NF_BAD : 0u);      // (the three bytes at either end of the staged range are never looked at)
          f4 |= fl << (8 * q);
        }
        reinterpret_cast<uint32_t*>(L.f)[i] = f4;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll 1
    for (int i = lane; i < PLDS / 4; i += 64) {
      const uint32_t v = reinterpret_cast<const uint32_t*>(L.raw)[i];
      // the bytes >= 0xC0 of the dword, one after the other: a character begins at each (as many trips as the lane with the most of them has -
      // two in text of three-byte characters, where a loop over the four byte positions makes four calls, each for a third of the lanes)
      // (this shape and no other: as `for (lm = ...; lm != 0; lm &= lm - 1)` with a `continue` behind the call and a loop over the character's bytes,
      // k_norm_emit<3> - and only it - came out with wrong class bytes on the device; profiles/r06_scripts_on_device.txt)
      uint32_t lm = v & (v << 1) & 0x80808080u;
#pragma unroll 1
      while (lm != 0u) {
        const int x = 4 * i + (__builtin_ctz(lm) >> 3);
        lm &= lm - 1u;
        const uint32_t r = classify_high_char(L.raw + x, x, PLDS, ts, two);
        const uint32_t c0 = r & 0xFFu, c1 = (r >> 8) & 0xFFu, nb = c0 == NF_BAD ? 0u : (r >> 16) & 0xFFu;
        uint8_t* f = L.f + x;
        if (nb >= 1u && x >= 3 && x < PLDS - 3) f[0] = (uint8_t)c0;
        if (nb >= 2u && x + 1 >= 3 && x + 1 < PLDS - 3) f[1] = (uint8_t)c1;
        if (nb >= 3u && x + 2 >= 3 && x + 2 < PLDS - 3) f[2] = (uint8_t)c1;
        if (nb >= 4u && x + 3 >= 3 && x + 3 < PLDS - 3) f[3] = (uint8_t)c1;
        if ((r >> 24) != 0u && nb != 0u) {                     // a character the pass changes: its chunk(s) of the piece (every writer writes 1)
          const uint32_t i0 = (uint32_t)(x - PMARGIN), i1 = i0 + nb - 1u;
          if (i0 < (uint32_t)PIECE) L.chg[i0 >> 6] = 1;
          if (i1 < (uint32_t)PIECE) L.chg[i1 >> 6] = 1;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
  }
  return left > (uint64_t)PIECE ? PIECE : (int)left;
}

__global__ __launch_bounds__(256) void k_norm_summary(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rbegin,
                                                      const uint64_t* __restrict__ rend, const uint32_t* __restrict__ piece_doc,
                                                      const uint64_t* __restrict__ doc_piece_start, uint64_t npieces, uint32_t lower_all,
                                                      const NmTwo* __restrict__ two, uint32_t* __restrict__ piece_sum) {
  __shared__ PieceLds s_l[4];
  __shared__ uint8_t s_cls[128];
  __shared__ TabLds s_tab;
  if (threadIdx.x < 128) s_cls[threadIdx.x] = (uint8_t)ncls_ascii(threadIdx.x, lower_all != 0);
  const NmTabs tabs = stage_tabs(s_tab, two);
  __syncthreads();
  // (wave-uniform wavefront index: the run bookkeeping below is arithmetic on ballots and then runs on the scalar unit)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t k = (uint64_t)blockIdx.x * 4 + wv;
  if (k >= npieces) return;
  PieceLds& L = s_l[wv];
  const uint32_t d = piece_doc[k];
  const uint64_t rb = rbegin[d], re = rend[d], pb = rb + (k - doc_piece_start[d]) * PIECE;
  const int m = __builtin_amdgcn_readfirstlane(norm_load_piece(L, raw, rb, re, pb, lane, s_cls, &s_tab, two));
  bool lead_open = true, lead_tl = false, bad = false;
  uint32_t lead_u = 0, trail_u = 0;
  for (int c = 0; c * 64 < m; c++) {
    const int i = c * 64 + lane;
    const bool in = i < m;
    const uint32_t fl = in ? L.f[PMARGIN + i] : 0u, cls = fl & NF_CLASS;
    bad |= in && fl == NF_BAD;
    const unsigned long long V = __ballot(in), mB = __ballot(in && fl != NF_BAD && nblock(cls)), mU = __ballot(in && fl != NF_BAD && cls == NC_U),
                             mL = __ballot(in && fl != NF_BAD && cls == NC_L);
    const unsigned long long nonblock = V & ~mB;
    if (lead_open) {
      if (nonblock != 0) {
        const int e = __ffsll((long long)nonblock) - 1;
        lead_u = min(lead_u + (uint32_t)__popcll(mU & ((1ull << e) - 1ull)), 2u);
        lead_tl = (mL >> e) & 1ull;
        lead_open = false;
      } else lead_u = min(lead_u + (uint32_t)__popcll(mU), 2u);
    }
    if (nonblock != 0) {
      const int hi = 63 - __clzll((long long)nonblock);
      trail_u = min((uint32_t)__popcll(hi == 63 ? 0ull : (mU & (~0ull << (hi + 1)))), 2u);
    } else trail_u = min(trail_u + (uint32_t)__popcll(mU), 2u);
  }
  bad = __any(bad);
  if (lane == 0) {
    const uint32_t f0 = L.f[PMARGIN], c0 = f0 & NF_CLASS;
    uint32_t s = (lead_open ? PS_WHOLE : 0u) | (lead_u << PS_LEADU_SHIFT) | (lead_tl ? PS_LEADTL : 0u) | (trail_u << PS_TRAILU_SHIFT);
    if (m > 0 && f0 != NF_BAD && nblock(c0)) s |= PS_FIRSTBLOCK;
    if (m > 0 && f0 != NF_BAD && c0 == NC_L) s |= PS_FIRSTL;
    if (bad) s |= PS_BAD;
    piece_sum[k] = s;
  }
}

// carry byte of a piece: ub[0..1] | ua[2..3] | tL[4]
__global__ void k_norm_carry(const uint32_t* __restrict__ piece_sum, const uint64_t* __restrict__ doc_piece_start, uint32_t ndocs,
                             uint8_t* __restrict__ piece_carry, uint8_t* __restrict__ need_host, unsigned long long* __restrict__ ninfo,
                             uint32_t* __restrict__ fb_ids, uint32_t all_host) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndocs) return;
  const uint64_t ps = doc_piece_start[d], pe = doc_piece_start[d + 1];
  uint32_t ub = 0;
  bool bad = all_host != 0;        // normalizer flags the device pass does not implement: every document takes the host path
  for (uint64_t k = ps; k < pe; k++) {
    const uint32_t s = piece_sum[k];
    piece_carry[k] = (uint8_t)ub;
    bad |= (s & PS_BAD) != 0;
    if (s & PS_WHOLE) ub = min(ub + ((s >> PS_LEADU_SHIFT) & 3u), 2u);
    else ub = (s >> PS_TRAILU_SHIFT) & 3u;
  }
  uint32_t ua = 0, tl = 0;
  for (uint64_t k = pe; k > ps; k--) {
    const uint32_t s = piece_sum[k - 1];
    piece_carry[k - 1] = (uint8_t)(piece_carry[k - 1] | (ua << 2) | (tl << 4));
    if (s & PS_WHOLE) ua = min(ua + ((s >> PS_LEADU_SHIFT) & 3u), 2u);
    else if (s & PS_FIRSTBLOCK) { ua = (s >> PS_LEADU_SHIFT) & 3u; tl = (s & PS_LEADTL) ? 1u : 0u; }
    else { ua = 0; tl = (s & PS_FIRSTL) ? 1u : 0u; }
  }
  need_host[d] = bad ? 1 : 0;
  if (bad) fb_ids[atomicAdd(&ninfo[0], 1ull)] = d;           // the documents the host has to normalize, in no particular order
}

// MODE 0: normalized length of every piece.  MODE 1: the bytes, packed at piece_off.  MODE 2: the bytes into the piece's
// private slab (SLAB bytes apart; a piece that would not fit raises the overflow flag) and the length.  MODE 3: a vocabulary WITHOUT
// capcode in one pass, as k_norm_emit2<false> is for capcode 2: nothing is known about the documents beforehand (no k_norm_summary /
// k_norm_carry: without capcode a byte's output depends on nothing but its character), the bytes go into the slabs, and a piece that finds
// a byte it cannot normalize marks its document in need_host.
constexpr int SLAB = 2 * PIECE;
static_assert(SLAB == SLAB_BYTES, "k_match_branch stages the text from these slabs");
template <int MODE>
__global__ __launch_bounds__(256) void k_norm_emit(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rbegin,
                                                   const uint64_t* __restrict__ rend, const uint32_t* __restrict__ piece_doc,
                                                   const uint64_t* __restrict__ doc_piece_start, uint64_t npieces, uint32_t capcode,
                                                   uint32_t lower_all, const uint8_t* __restrict__ piece_carry,
                                                   uint8_t* __restrict__ need_host, uint32_t* __restrict__ piece_len,
                                                   const uint64_t* __restrict__ piece_off, uint8_t* __restrict__ out,
                                                   unsigned long long* __restrict__ overflow, const NmTwo* __restrict__ two,
                                                   const uint64_t* __restrict__ np_dev) {
  constexpr bool WRITE = MODE != 0;
  if (np_dev) npieces = min(npieces, np_dev[0]);       // (the ring behind a filter pass: launched over a bound, the count is the device's)
  __shared__ PieceLds s_l[4];
  __shared__ uint8_t s_cls[128];
  __shared__ TabLds s_tab;
  if (threadIdx.x < 128) s_cls[threadIdx.x] = (uint8_t)ncls_ascii(threadIdx.x, lower_all != 0);
  const NmTabs tabs = stage_tabs(s_tab, two);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t k = (uint64_t)blockIdx.x * 4 + wv;
  if (k >= npieces) return;
  PieceLds& L = s_l[wv];
  const uint32_t d = piece_doc[k];
  if (MODE != 3 && need_host[d]) { if (MODE != 1 && lane == 0) piece_len[k] = 0; return; }
  const uint64_t rb = rbegin[d], re = rend[d], pb = rb + (k - doc_piece_start[d]) * PIECE;
  const int m = norm_load_piece(L, raw, rb, re, pb, lane, s_cls, &s_tab, two);
  uint8_t* dst = MODE == 1 ? out + piece_off[k] : (MODE >= 2 ? out + k * (uint64_t)SLAB : nullptr);
  if (capcode != 2 || MODE == 3) {                          // no capcode: same length, only the lower-case flag applies
    if (MODE != 1 && lane == 0) piece_len[k] = (uint32_t)m;
    bool bad = false;
    if (WRITE) for (int i = lane; i < m; i += 64) {
      const int x = PMARGIN + i;
      if (MODE == 3) bad |= L.f[x] == NF_BAD;
      uint32_t b = L.raw[x], y = 0, m3 = 0;
      if (lower_all && b - 'A' < 26u) b |= 0x20u;
      const uint32_t bm1 = L.raw[x - 1];
      // (a two-byte character that stays one: its bytes under the vocabulary's flags; those that decompose are not in this table)
      if (nm_two_lead(b)) (void)nm_two_out(nm_two_get(tabs, nm_two_index(b, L.raw[x + 1])), false, false, false, &b, &y, &m3);
      else if (nm_cont_byte(b) && nm_two_lead(bm1)) (void)nm_two_out(nm_two_get(tabs, nm_two_index(bm1, b)), true, false, false, &b, &y, &m3);
      dst[i] = (uint8_t)b;
    }
    if (MODE == 3 && __any(bad) && lane == 0) need_host[d] = 1;      // (every writer writes 1; the piece lengths of such a document are zeroed by k_norm_bad)
    return;
  }
  const uint32_t carry = piece_carry[k];
  const unsigned long long below = (1ull << lane) - 1ull;
  // ---- backward sweep: ua and tL of every block byte --------------------------------------------------------------
  {
    uint32_t carry_ua = (carry >> 2) & 3u;
    bool carry_tl = (carry >> 4) & 1u;
    for (int c = (m - 1) / 64; c >= 0; c--) {
      const int i = c * 64 + lane;
      const bool in = i < m;
      const uint32_t fl = in ? L.f[PMARGIN + i] : 0u, cls = fl & NF_CLASS;
      const unsigned long long V = __ballot(in), mB = __ballot(in && nblock(cls)), mU = __ballot(in && cls == NC_U), mL = __ballot(in && cls == NC_L);
      const unsigned long long above = lane == 63 ? 0ull : (~0ull << (lane + 1));
      const unsigned long long stop = V & ~mB & above;         // first non-block byte to my right inside the chunk
      uint32_t ua;
      bool tl;
      if (stop != 0) {
        const int e = __ffsll((long long)stop) - 1;
        ua = (uint32_t)__popcll(mU & above & ((1ull << e) - 1ull));
        tl = (mL >> e) & 1ull;
      } else {
        ua = (uint32_t)__popcll(mU & above) + carry_ua;
        tl = carry_tl;
      }
      ua = min(ua, 2u);
      if (in && nblock(cls)) L.f[PMARGIN + i] = (uint8_t)(fl | (tl ? NF_TERML : 0u) | (ua << NF_UA_SHIFT));
      const uint32_t ua0 = __shfl(ua, 0), cls0 = __shfl(cls, 0);
      const bool tl0 = __shfl((int)tl, 0) != 0;
      if (nblock(cls0)) { carry_ua = min(ua0 + (cls0 == NC_U ? 1u : 0u), 2u); carry_tl = tl0; }
      else { carry_ua = 0; carry_tl = cls0 == NC_L; }
    }
  }
  // tL of the first byte of the NEXT piece (a space at the very end of this piece may turn into its marker)
  const bool next_tl = (carry >> 4) & 1u;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  // ---- forward sweep: ub, then every character's output ------------------------------------------------------------
  uint32_t carry_ub = carry & 3u;
  uint32_t pos = 0;
  for (int c = 0; c * 64 < m; c++) {
    const int i = c * 64 + lane, x = PMARGIN + i;
    const bool in = i < m;
    const uint32_t fl = in ? L.f[x] : 0u, cls = fl & NF_CLASS;
    const unsigned long long mB = __ballot(in && nblock(cls)), mU = __ballot(in && cls == NC_U);
    const unsigned long long gap = ~mB & below;                // non-block bytes to my left inside the chunk
    uint32_t ub;
    if (gap != 0) {
      const int s = 64 - __clzll((long long)gap);
      ub = (uint32_t)__popcll(mU & below & ~((1ull << s) - 1ull));
    } else ub = (uint32_t)__popcll(mU & below) + carry_ub;
    ub = min(ub, 2u);
    {
      const uint32_t ub63 = __shfl(ub, 63), cls63 = __shfl(cls, 63);
      carry_ub = nblock(cls63) ? min(ub63 + (cls63 == NC_U ? 1u : 0u), 2u) : 0u;
    }
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0, len = 0;
    if (in) {
      const uint32_t b = L.raw[x];
      const uint32_t fm1 = L.f[x - 1], P = fm1 & NF_CLASS;
      uint32_t P2 = NC_O;
      if (P == NC_AP) P2 = L.f[(fm1 & NF_CONT) ? x - 4 : x - 2] & NF_CLASS;
      const bool inword = ub >= 1;                             // javascript/tokenmonster.js `inWord` before this character
      const bool tl = (fl & NF_TERML) != 0;
      // the rules themselves are the ones k_norm_emit2 reads from its table (tm_norm_masks.h: nm_lut_entry); what this kernel does
      // differently is how it finds inWord and the 'C' look-ahead (per-lane counting instead of flood fills on the ballots)
      const uint32_t code = nm_lut_entry(nm_lut_index((fl & NF_CONT) ? (uint32_t)NC_O : cls, P, P2, inword ? 1u : 0u, tl ? 1u : 0u), lower_all != 0);
      len = (code & 3u) + 1u;
      o0 = 'D'; o1 = code >> 8; o2 = ' ';
      o3 = b | ((code & 4u) << 3);
      if (cls == NC_SP) {
        const bool last = i + 1 == m;                          // the next byte lives in the next piece (or nowhere)
        const uint32_t fp1 = L.f[x + 1];
        if ((fp1 & NF_CLASS) == NC_U && fp1 != NF_BAD && !(fp1 & NF_CONT)) o3 = (last ? next_tl : (fp1 & NF_TERML) != 0) ? 'C' : 'W';   // :976-979
      }
      // one byte of a two-byte character: its bytes come from the table (a decomposed character's second half emits the two bytes of the mark)
      const uint32_t bm1 = L.raw[x - 1];
      const bool lead2 = nm_two_lead(b), cont2 = nm_cont_byte(b) && nm_two_lead(bm1);
      if (lead2 || cont2) {
        const NmTwo e = nm_two_get(tabs, lead2 ? nm_two_index(b, L.raw[x + 1]) : nm_two_index(bm1, b));
        uint32_t y = 0, m3 = 0;
        const uint32_t extra = nm_two_out(e, cont2, (code & 4u) != 0, true, &o3, &y, &m3);
        if (extra >= 1u) { len = 1u + extra; o2 = y; o1 = m3; }
      }
      // one byte of a Hangul syllable (NFD): its lane emits one of the syllable's jamo - or nothing (tm_norm_masks.h)
      uint32_t hrole, hcp;
      if ((tabs.misc & NM_MISC_HANGUL) && fl != NF_BAD && nm_hangul_role(b, bm1, L.raw[x - 2], L.raw[x + 1], L.raw[x + 2], &hrole, &hcp)) len = nm_hangul_out(hcp, hrole, &o1, &o2, &o3);
      // ... or of a letter of Latin Extended Additional (NFD): the letter, its first mark, its second mark or nothing
      uint32_t lrole, lidx;
      if ((tabs.misc & NM_MISC_LEA) && fl != NF_BAD && nm_lea_role(b, bm1, L.raw[x - 2], L.raw[x + 1], L.raw[x + 2], &lrole, &lidx) && (tabs.lea[lidx].a & NT_OK)) {
        const NmLea le = tabs.lea[lidx];
        if (lrole == 0u) o3 = (code & 4u) ? ((le.a >> 16) & 0xFFu) : ((le.a >> 8) & 0xFFu);
        else if (lrole == 1u) { len = 2u; o2 = le.b & 0xFFu; o3 = (le.b >> 8) & 0xFFu; }
        else if (((le.a >> 24) & 3u) == 2u) { len = 2u; o2 = (le.b >> 16) & 0xFFu; o3 = le.b >> 24; }
        else len = 0u;
      }
      // ... or of a voiced kana (NFD): the base kana, the mark, nothing
      uint32_t krole, kidx;
      if ((tabs.misc & NM_MISC_KANA) && fl != NF_BAD && nm_kana_role(b, bm1, L.raw[x - 2], L.raw[x + 1], L.raw[x + 2], &krole, &kidx) && (nm_kana_tab(tabs)[kidx] & NK_OK))
        len = nm_kana_out(nm_kana_tab(tabs)[kidx], krole, &o1, &o2, &o3);
      // ... or of a character NFD splits in two three-byte ones
      uint32_t drole, dcp;
      if ((tabs.misc & NM_MISC_DEC3) && fl != NF_BAD && nm_three_role(b, bm1, L.raw[x - 2], L.raw[x + 1], L.raw[x + 2], &drole, &dcp) && (nm_dec3(tabs, dcp) & ND_OK))
        len = nm_dec3_out(tabs, nm_dec3(tabs, dcp), drole, &o1, &o2, &o3);
    }
    // inclusive prefix sum of len (0..4) over the wavefront: bytes-with-len>=k ballots, counted below the lane
    uint32_t incl = len;
#pragma unroll
    for (uint32_t q = 1; q <= 4; q++) {
      const unsigned long long bq = __ballot(len >= q);
      incl = __builtin_amdgcn_mbcnt_hi((uint32_t)(bq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bq, incl));
    }
    if (WRITE && in && (MODE == 1 || pos + incl <= (uint32_t)SLAB)) {
      uint8_t* w = dst + pos + (incl - len);
      if (len == 4) { w[0] = (uint8_t)o0; w[1] = (uint8_t)o1; w[2] = (uint8_t)o2; w[3] = (uint8_t)o3; }
      else if (len == 3) { w[0] = (uint8_t)o1; w[1] = (uint8_t)o2; w[2] = (uint8_t)o3; }
      else if (len == 2) { w[0] = (uint8_t)o2; w[1] = (uint8_t)o3; }
      else if (len == 1) w[0] = (uint8_t)o3;
    }
    pos += __shfl(incl, 63);
  }
  if (MODE != 1 && lane == 0) {
    piece_len[k] = pos;
    if (MODE == 2 && pos > (uint32_t)SLAB) atomicAdd(overflow, 1ull);
  }
}

// ---- capcode level 2, one pass into the slabs ------------------------------------------------------------------------------
// Same rules as k_norm_emit (which stays for the exact two-pass path and for capcode 0) at ~40 % of its vector instructions
// (the normalizer is bound by vector-instruction issue like the match kernel): the two run-dependent facts of a byte — inWord
// and the 'C'/'W' lookahead — are flood fills on the chunk's class ballots, done by the scalar unit; the rule chain is a
// 2048-entry table in LDS (tm_norm_masks.h; both checked on the CPU by tools/norm_masks_check.cpp); the output is assembled in
// LDS and leaves for the slab in 16-byte stores.
struct PieceLds2 { uint8_t raw[PLDS]; uint8_t f[PLDS]; uint8_t chg[32]; alignas(16) uint8_t out[2 * PIECE + 64]; };   // out: slab image + one dump byte per lane

__device__ const NmLut g_norm_lut = nm_make_lut();

// CARRY = true: the carries of the piece (inWord at its first byte, how the block at its end ends) and the documents that need the host
// normalizer are known (k_norm_summary + k_norm_carry have run: the exact path).  CARRY = false (the usual path): no pass before this
// one — the carries come from the 64 bytes either side of the piece (nm_margin_carries), a piece whose margins cannot tell raises
// overflow[1] and the exact path runs after all; a piece that finds a byte it cannot normalize marks its document in need_host.
#ifndef TM_NORM_UNROLL
#define TM_NORM_UNROLL 2
#endif
template <bool CARRY>
__global__ __launch_bounds__(256) void k_norm_emit2(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ rbegin,
                                                    const uint64_t* __restrict__ rend, const uint32_t* __restrict__ piece_doc,
                                                    const uint64_t* __restrict__ doc_piece_start, uint64_t npieces, uint32_t lower_all,
                                                    const uint8_t* __restrict__ piece_carry, uint8_t* __restrict__ need_host,
                                                    uint32_t* __restrict__ piece_len, uint8_t* __restrict__ slab,
                                                    unsigned long long* __restrict__ overflow, const NmTwo* __restrict__ two,
                                                    const uint64_t* __restrict__ np_dev) {
  constexpr int SLAB2 = 2 * PIECE, NCH = PIECE / 64;
  if (np_dev) npieces = min(npieces, np_dev[0]);       // (the ring behind a filter pass: launched over a bound, the count is the device's)
  __shared__ PieceLds2 s_l[4];
  __shared__ uint8_t s_cls[128];
  __shared__ TabLds s_tab;
  const NmTabs tabs = stage_tabs(s_tab, two);
  alignas(16) __shared__ uint16_t s_lut[NM_LUT_SIZE];
  static_assert(NM_LUT_SIZE * sizeof(uint16_t) == 256 * sizeof(uint4), "one 16-byte load per thread stages the rule table");
  if (threadIdx.x < 128) s_cls[threadIdx.x] = (uint8_t)ncls_ascii(threadIdx.x, lower_all != 0);
  reinterpret_cast<uint4*>(s_lut)[threadIdx.x] = reinterpret_cast<const uint4*>(g_norm_lut.e[lower_all ? 1 : 0])[threadIdx.x];
  __syncthreads();
  // the wavefront index is made visibly wave-uniform: everything derived from it (piece, length, carries) then lives in scalar
  // registers, and so does the mask algebra below
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  PieceLds2& L = s_l[wv];
  // A wavefront takes piece after piece (the grid: norm_grid()): the tables above are staged once per workgroup, not once per four pieces.
#pragma unroll 1
  for (uint64_t k = (uint64_t)blockIdx.x * 4 + wv; k < npieces; k += (uint64_t)gridDim.x * 4) {
  const uint32_t d = piece_doc[k];
  if (CARRY && need_host[d]) { if (lane == 0) piece_len[k] = 0; continue; }
  const uint64_t rb = rbegin[d], re = rend[d], pb = rb + (k - doc_piece_start[d]) * PIECE;
  const int m = __builtin_amdgcn_readfirstlane(norm_load_piece(L, raw, rb, re, pb, lane, s_cls, &s_tab, two));
  const uint32_t carry = CARRY ? __builtin_amdgcn_readfirstlane((uint32_t)piece_carry[k]) : 0u;
  const int nch = (m + 63) >> 6;
  const uint8_t* fl0 = L.f + PMARGIN + lane;       // class byte of byte (64 c + lane) of the piece = fl0[64 c]; chunk -1 and chunk NCH are the margins
  // (norm_load_piece has classified the LDS bytes from 66 before the piece to 66 after its 1024; behind the end of the document they are class O)
  unsigned long long w_seed = (carry & 3u) ? 1ull : 0ull;
  const unsigned long long t0_beyond = CARRY ? (unsigned long long)((carry >> 4) & 1u) : 0ull;       // T0 (tm_norm_masks.h) of what lies behind the margin: looked at only when all of the margin is one block
  if (!CARRY) {
    const uint32_t fb = fl0[-64], fa = m == PIECE ? (uint32_t)fl0[64 * NCH] : 0u;       // (a shorter piece is the last of its document: nothing follows)
    uint64_t w_in;
    const bool known = nm_margin_carries(__ballot((fb & NF_BLOCK) != 0), __ballot((fb & NF_CLASS) == NC_U), __ballot((fa & NF_BLOCK) != 0), &w_in);
    if (!known && lane == 0) atomicAdd(overflow + 1, 1ull);
    w_seed = w_in;
  }
  // ---- ONE sweep, forwards: W carried from chunk to chunk, T from the ballots of the chunk and of the one behind it (nm_t0 / nm_tx) -----------
  // No masks for the end of the document: the bytes behind it are class O, which take no part in anything and emit themselves - one zero
  // byte per lane, behind everything else in the output image, and subtracted from the length below.
  typedef TM_LDS_SPACE uint8_t lds_u8;
  const uint32_t out0 = TM_LDS_ADDR(L.out);
  const uint32_t dump = out0 + (uint32_t)SLAB2 + (uint32_t)lane;     // where a lane's "not this byte" stores go
  const uint32_t lim = out0 + (uint32_t)SLAB2;
  unsigned long long w = w_seed, badm = 0ull;
  uint32_t posabs = out0;                                             // LDS address of the next output byte (wave-uniform)
  bool over = false;
  uint32_t chC = 'C', chW = 'W', chSP = ' ', chD = 'D';
  TM_KEEP_IN_VGPRS4(chC, chW, chSP, chD);      // four registers for the whole sweep, not four moves per chunk
  const uint8_t* fc = fl0;                      // this lane's class byte in chunk c
  const uint8_t* rc = L.raw + PMARGIN + lane;   // and its byte
  uint32_t fl = fc[0];
  unsigned long long Bcur = __ballot((fl & NF_BLOCK) != 0), Lcur = __ballot((fl & NF_CLASS) == NC_L), Ucur = __ballot((fl & NF_CLASS) == NC_U);
  int c = 0;
  auto chunk = [&]() __attribute__((always_inline)) {
    // the chunk behind this one (chunk NCH is the margin): its capitals (its first byte may be the capital a trailing space announces), and what
    // a block that reaches the end of this chunk ends in
    const uint32_t fnext = fc[64];
    const unsigned long long Bn = __ballot((fnext & NF_BLOCK) != 0), Ln = __ballot((fnext & NF_CLASS) == NC_L), Un = __ballot((fnext & NF_CLASS) == NC_U);
    // (bit 63 set in what ctz looks at: defined when every byte is in the block, and no compare-and-select when - as good as always - not)
    unsigned long long t0n = Ln >> __builtin_ctzll(~Bn | (1ull << 63));
    if (__builtin_expect(Bn == ~0ull, 0)) {       // 64 bytes of one block behind this chunk: the first chunk behind that with anything else in it decides
      t0n = t0_beyond;
#pragma unroll 1
      for (int j = c + 2; j <= NCH; j++) {
        const uint32_t fj = fl0[64 * j];
        const unsigned long long Bj = __ballot((fj & NF_BLOCK) != 0), Lj = __ballot((fj & NF_CLASS) == NC_L);
        if (Bj != ~0ull) { t0n = nm_t0(Bj, Lj, 0ull); break; }
      }
    }
    const unsigned long long TXc = nm_tx(Bcur, Lcur, t0n);
    const uint32_t fp = fc[-1], f2 = fc[-2], f4 = fc[-4];
    const uint32_t b = rc[0];
    uint64_t w_out, spC, spW;
    if (!CARRY) badm |= __ballot(fl == NF_BAD);
    const unsigned long long W = nm_inword(Bcur, Ucur, ~0ull, w, &w_out);
    nm_space_markers(__ballot((fl & NF_CLASS) == NC_SP), Ucur, Un, ~0ull, TXc, t0n, &spC, &spW);
    w = w_out;
    // the rule table: index = class | previous class << 3 | class before the apostrophe << 6 | W << 9 | T << 10
    // (a continuation byte takes no part in the rules as a character of its own: class O here; as the PREVIOUS byte it stands for its character)
    const uint32_t p2 = (fp & NF_CONT) ? f4 : f2;
    uint32_t idx = ((fl & NF_CONT) ? (uint32_t)NC_O : (fl & 7u)) | ((fp & 7u) << 3) | ((p2 & 7u) << 6);
    idx = sel_mask(W, idx | 512u, idx);
    idx = sel_mask(TXc, idx | 1024u, idx);
    const uint32_t code = s_lut[idx];
    uint32_t len1 = code & 3u;                                         // bytes emitted - 1
    uint32_t o3 = b | ((code & 4u) << 3);
    o3 = sel_mask(spC, chC, o3);
    o3 = sel_mask(spW, chW, o3);
    uint32_t ysp = chSP, m3 = code >> 8;
    // Everything a character beyond ASCII needs sits behind ONE test of the chunk and out of line (emit_high_byte): chunks of plain ASCII are
    // the rule.  E = the lanes that emit at least one byte: all, but for the third lane of a Hangul syllable without a final consonant.
    unsigned long long E = ~0ull;
    uint32_t below = posabs + (uint32_t)lane, n = len1 + 1u;           // first output byte of the lane before the extra bytes of the lanes below it; bytes it emits
    if (__ballot(b >= 0x80u) != 0ull && __builtin_amdgcn_readfirstlane((int)L.chg[c]) != 0) {      // (a chunk of nothing but inert characters - punctuation, ideographs, Thai ... - emits its bytes as they are)
      if (b >= 0x80u) {
        const HighOut h = emit_high_byte(rc, fl, code, HighOut{o3, ysp, m3, len1}, &s_tab, two);
        o3 = h.o3; ysp = h.ysp; m3 = h.m3; len1 = h.len1;
      }
      E = ~__ballot(len1 == 0xFFu);
      len1 = len1 == 0xFFu ? 0u : len1;
      n = sel_mask(E, len1 + 1u, 0u);
      below = mbcnt64(E, posabs);
    }
    const unsigned long long ge2 = __ballot(len1 >= 1u), ge3 = __ballot(len1 >= 2u), ge4 = __ballot(len1 >= 3u);
    // first output byte of the lane = the bytes of the lanes below behind posabs; its last byte is len1 further; the chunk ends where lane 63 does
    const uint32_t first = mbcnt64(ge4, mbcnt64(ge3, mbcnt64(ge2, below)));
    const uint32_t last = first + len1;
    const uint32_t end = read_lane(first + n, 63);
    if (end <= lim) {
      *TM_LDS_PTR(lds_u8, sel_mask(E, last, dump)) = (uint8_t)o3;
      *TM_LDS_PTR(lds_u8, sel_mask(ge2, last - 1u, dump)) = (uint8_t)ysp;
      *TM_LDS_PTR(lds_u8, sel_mask(ge3, last - 2u, dump)) = (uint8_t)m3;
      *TM_LDS_PTR(lds_u8, sel_mask(ge4, first, dump)) = (uint8_t)chD;
    } else over = true;                                   // wave-uniform: the piece does not fit its slab (exact two-pass path)
    posabs = end;
    fl = fnext; Bcur = Bn; Lcur = Ln; Ucur = Un;
    c++; fc += 64; rc += 64;
  };
  // (two chunks per trip: what the chunk behind hands on stays where it is instead of being copied back - five scalar moves and a branch per chunk)
#pragma unroll 1
  while (c + TM_NORM_UNROLL <= nch) {
#pragma unroll
    for (int u = 0; u < TM_NORM_UNROLL; u++) chunk();
  }
#pragma unroll 1
  while (c < nch) chunk();
  const uint32_t pos = posabs - out0 - (uint32_t)(64 * nch - m);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  if (!over) {
    uint8_t* dst = slab + k * (uint64_t)SLAB2;
    for (uint32_t j = (uint32_t)lane * 16u; j < pos; j += 64u * 16u)
      *reinterpret_cast<uint4*>(dst + j) = *reinterpret_cast<const uint4*>(L.out + j);    // the slab is 16-byte aligned and 2 KiB long
  }
  if (lane == 0) {
    piece_len[k] = pos;
    if (over) atomicAdd(overflow, 1ull);             // (with the zero bytes of a last chunk counted in: a piece within 63 bytes of its slab's end may take the exact path for nothing)
    if (!CARRY && badm != 0ull) need_host[d] = 1;      // (every writer writes 1; the piece lengths of such a document are zeroed by k_norm_bad)
  }
  __builtin_amdgcn_wave_barrier();                     // (the output image is read above and written again by the next piece)
  }
}


// pieces per document; clears what the pass writes to on the side (need_host, the info words): no memset commands of their own
__global__ void k_norm_begin(const uint64_t* __restrict__ rbegin, const uint64_t* __restrict__ rend, uint32_t ndocs, uint32_t* __restrict__ doc_npiece, uint8_t* __restrict__ need_host,
                             unsigned long long* __restrict__ ninfo) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < 8) ninfo[d] = 0ull;
  if (d < ndocs) {
    doc_npiece[d] = (uint32_t)((rend[d] - rbegin[d] + PIECE - 1) / PIECE);
    need_host[d] = 0;
  }
}

// behind k_norm_emit2<false>: the pieces of the documents that turned out to need the host normalizer count for nothing, and those documents
// are listed for the host (in no particular order) - thread k looks at piece k and at document k
// ... and ninfo[6] counts the pieces that are not the last of their document yet shorter than what a segment looks at (TEXT_LEN): with one
// of those the text cannot be staged from the slabs by k_match_branch (which looks at two pieces at most) and is packed after all
__global__ void k_norm_bad(const uint32_t* __restrict__ piece_doc, const uint8_t* __restrict__ need_host, const uint64_t* __restrict__ doc_piece_start, uint64_t npieces,
                           uint32_t ndocs, uint32_t* __restrict__ piece_len, unsigned long long* __restrict__ ninfo, uint32_t* __restrict__ fb_ids,
                           const uint64_t* __restrict__ np_dev) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (np_dev && k >= np_dev[0]) { if (k < npieces) piece_len[k] = 0; }       // (launched over a bound: what lies behind the device's count is empty)
  else if (k < npieces) {
    const uint32_t d = piece_doc[k];
    if (need_host[d]) piece_len[k] = 0;
    else if (piece_len[k] < (uint32_t)TEXT_LEN && k + 1 != doc_piece_start[d + 1]) atomicAdd(&ninfo[6], 1ull);
  }
  if (k < ndocs && need_host[k]) fb_ids[atomicAdd(&ninfo[0], 1ull)] = (uint32_t)k;
}

// the exact path's share of k_norm_bad: ninfo[6] counts the pieces that are too short for the text to stay in the slabs
__global__ void k_norm_short(const uint32_t* __restrict__ piece_doc, const uint8_t* __restrict__ need_host, const uint64_t* __restrict__ doc_piece_start, uint64_t npieces,
                             const uint32_t* __restrict__ piece_len, unsigned long long* __restrict__ ninfo) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npieces) return;
  const uint32_t d = piece_doc[k];
  if (!need_host[d] && piece_len[k] < (uint32_t)TEXT_LEN && k + 1 != doc_piece_start[d + 1]) atomicAdd(&ninfo[6], 1ull);
}

// pack the slabs: piece k's bytes go to out[piece_off[k] ..); a piece that would end behind `cap` bytes is left out (the caller - which
// has not seen the total yet when it launches this - then reports TM_E_LIMIT)
__global__ __launch_bounds__(256) void k_norm_compact(const uint8_t* __restrict__ slab, const uint32_t* __restrict__ piece_len,
                                                      const uint64_t* __restrict__ piece_off, uint64_t npieces, uint8_t* __restrict__ out, uint64_t cap) {
  const int lane = threadIdx.x & 63;
  const uint64_t k = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= npieces) return;
  const uint32_t len = piece_len[k];
  if (piece_off[k] + len > cap) return;
  const uint8_t* src = slab + k * (uint64_t)SLAB;
  uint8_t* dst = out + piece_off[k];
  for (uint32_t i = (uint32_t)lane * 16u; i < len; i += 64u * 16u) {
    if (i + 16u <= len) { uint4 q; __builtin_memcpy(&q, src + i, 16); __builtin_memcpy(dst + i, &q, 16); }
    else for (uint32_t j = i; j < len; j++) dst[j] = src[j];
  }
}

// normalized range of every device-normalized document; segment / long-document counts
__global__ void k_norm_ranges(const uint64_t* __restrict__ piece_off, const uint64_t* __restrict__ doc_piece_start, const uint8_t* __restrict__ need_host,
                              uint32_t ndocs, uint64_t* __restrict__ nbegin, uint64_t* __restrict__ nend) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndocs) return;
  if (need_host[d]) return;                      // (k_place_fallback writes the range of a host-normalized document, possibly at the same time)
  nbegin[d] = piece_off[doc_piece_start[d]];
  nend[d] = piece_off[doc_piece_start[d + 1]];
}
// k_norm_ranges and k_norm_info in one launch, for a batch without host-normalized documents (a document that needs the host counts for
// nothing here: the caller runs k_norm_info again once those are placed)
__global__ void k_norm_ranges_info(const uint64_t* __restrict__ piece_off, const uint64_t* __restrict__ doc_piece_start, const uint8_t* __restrict__ need_host,
                                   uint32_t ndocs, uint64_t* __restrict__ nbegin, uint64_t* __restrict__ nend, unsigned long long* __restrict__ ninfo, uint32_t long_segs) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t nseg = 0;
  if (d < ndocs && !need_host[d]) {
    const uint64_t a = piece_off[doc_piece_start[d]], b = piece_off[doc_piece_start[d + 1]];
    nbegin[d] = a; nend[d] = b;
    nseg = (b - a + SEG - 1) / SEG;
    if (nseg > long_segs) atomicAdd(&ninfo[1], 1ull);
  }
  for (int o = 32; o > 0; o >>= 1) nseg += __shfl_xor(nseg, o);
  if ((threadIdx.x & 63) == 0 && nseg) atomicAdd(&ninfo[2], (unsigned long long)nseg);
}
__global__ void k_norm_info(const uint64_t* __restrict__ nbegin, const uint64_t* __restrict__ nend, uint32_t ndocs, unsigned long long* __restrict__ ninfo,
                            uint32_t long_segs) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t nseg = 0;
  if (d < ndocs) { nseg = (nend[d] - nbegin[d] + SEG - 1) / SEG; if (nseg > long_segs) atomicAdd(&ninfo[1], 1ull); }
  for (int o = 32; o > 0; o >>= 1) nseg += __shfl_xor(nseg, o);
  if ((threadIdx.x & 63) == 0 && nseg) atomicAdd(&ninfo[2], (unsigned long long)nseg);
}
// ---- the filter pass in front of the normalizer pass (round 6): the flags that drop and replace BYTES ------------------------------------------
// quotemarks 8, collapse 16, trim 32, leadingspace 64, unixlines 128 (training/README.md:110-123; tokenmonster.cpp:245-425, 428-462) and what
// accents 4 (:231-243) does to the two-byte characters, on the device: raw text in, filtered text + the documents' new ranges out, and the
// normalizer pass (NFD / lowercase / capcode) runs on that.  The host form is tm_normalize.cpp (squeeze, unix_lines, trim_and_lead,
// remove_marks), fuzzed against the reference runtime for all 256 flag values; this is the same function of the text, stated per byte:
//   * a space goes when the byte before it in the INPUT is a space; a '\r' goes when a '\n' follows; E2 80 {98,99 | 9C,9D} becomes ' or " -
//     all local - EXCEPT that the reference compacts in place and looks for the E2 80 in the buffer it is compacting: after exactly ONE
//     byte has been dropped (and until the next one is) the look-behind finds a byte that was already moved, and the quote stays.  The
//     offset by which the text has shrunk only grows - one for a dropped byte, two for a replaced quote - so it equals one exactly between the
//     first single drop of a document and the second, provided no quote was replaced before the first: three numbers per document
//     (first and second single drop, first quote candidate).
//     (unixlines WITHOUT collapse is a pass of its own in the reference, into a fresh buffer: no single drops in the loop that follows.)
//   * trim keeps what lies between the first and the last byte above 32, leadingspace puts a space in front of a first byte that is not
//     one; together, on a text WITHOUT leading blanks, the reference also cuts the last non-blank byte (:274-277) - kept.
//   * accents: a two-byte character becomes what NFD-and-drop-Mn leaves of it (nothing, one byte, two bytes: a table from the host's own
//     function, build_accent_table); whatever else would decompose or is a mark reaches the normalizer pass, whose tables in this mode
//     refuse it (norm_tables): that document is normalized by the host from its ORIGINAL bytes.
// Second form (the first - a piece per wavefront, a byte per lane, three launches that each read the text - took 9 ms per GiB): ONE launch, a
// wavefront per document, a dword per lane.  The wavefront first looks for the document's numbers - as a rule in its first and last 256
// bytes: the search for the single drops ends at the second one or at a quote in front of the first, the blanks at either end are short -,
// then sweeps the text once, 256 bytes per step: the tests are byte-parallel arithmetic on the lane's dword and on the same dword shifted
// against its neighbours' (DPP moves; the dword behind the step's last is the first of the next step's, which is already under way), a quote is
// found ONCE, at its first byte, and its other two bytes take their verdict from there, the lane packs the bytes it keeps and writes them
// with one store (two where it dropped something).  The filtered documents are not packed: document d begins on a 16-byte boundary of its
// own, PF_GAP bytes further from where it began for every document in front of it (the normalizer pass takes begin and end of a document
// from two arrays).  A document of more than PF_WHOLE bytes (rare) is filtered by a wavefront per span of PF_SPAN in three launches - numbers per span,
// sizes per span, bytes - of which the last two skip every other document.
constexpr uint32_t PF_NONE = 0xFFFFFFFFu;
#ifndef TM_PF_SPAN
#define TM_PF_SPAN 4096u        // (the emulated fuzz also runs with 1024 / 1024: documents of many spans)
#endif
#ifndef TM_PF_WHOLE
#define TM_PF_WHOLE 16384u
#endif
// a document of up to PF_WHOLE bytes is one wavefront's; a longer one is cut into spans of PF_SPAN (a span is a wavefront's work from end to end - 16 steps, three times)
constexpr uint32_t PF_SPAN = TM_PF_SPAN, PF_WHOLE = TM_PF_WHOLE, PF_GAP = 32u;
static_assert(PF_SPAN % 256u == 0u && PF_SPAN >= (uint32_t)PIECE && PF_WHOLE >= PF_SPAN, "a span is whole steps of the sweep, and the spans' table lives in the pieces' arrays");
__host__ __device__ __forceinline__ uint32_t pf_spans(uint64_t n) { return n <= PF_WHOLE ? (n ? 1u : 0u) : (uint32_t)((n + PF_SPAN - 1) / PF_SPAN); }
struct PfPiece { uint32_t s1, s2, fq, a, z; };      // of a span; positions in the document: first / second single drop, first quote candidate (its third byte), first / last byte above 32
struct PfDoc { uint32_t s1, s2, qb, a, z, zlo; };   // qb: a quote candidate lies before s1; zlo: where the last non-blank OUTPUT byte begins (z, or z - 2 for a replaced quote)
__host__ __device__ __forceinline__ uint64_t pf_out_begin(uint64_t raw_begin, uint32_t d) { return ((raw_begin + 15ull) & ~15ull) + (uint64_t)PF_GAP * d; }
__device__ __forceinline__ bool pf_isq(uint32_t y) { return y == 0x98u || y == 0x99u || y == 0x9Cu || y == 0x9Du; }
// four bytes at once: bit 7 of every byte of v that equals the byte c4 holds four times / that is one of 98 99 9C 9D / that is above 32
__device__ __forceinline__ uint32_t pf_eq(uint32_t v, uint32_t c4) { const uint32_t t = v ^ c4; return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u; }
__device__ __forceinline__ uint32_t pf_isq4(uint32_t v) { return pf_eq(v & 0xFAFAFAFAu, 0x98989898u); }
__device__ __forceinline__ uint32_t pf_nb4(uint32_t v) { return (((v & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | v) & 0x80808080u; }
// the dword at `off` of a document of n bytes behind `win` (a window of (n + 3) & ~3 bytes: the raw text has 256 bytes of room behind its last document): 0 for what lies behind the document
__device__ __forceinline__ uint32_t pf_load(const TmWindow& win, uint32_t off, uint32_t n) {
  uint32_t v = tm_window_u32(win, off);
  const uint32_t rem = n - off;
  if (rem < 4u) v &= (1u << (8u * rem)) - 1u;
  return v;
}
// position of the first (last) marked byte of the wavefront's 256: `any` = the ballot of m != 0
__device__ __forceinline__ uint32_t pf_first(unsigned long long any, uint32_t m, uint32_t base) {
  const int l = __builtin_ctzll(any);
  return base + 4u * (uint32_t)l + ((uint32_t)__builtin_ctz(read_lane(m, l)) >> 3);
}
__device__ __forceinline__ uint32_t pf_last(unsigned long long any, uint32_t m, uint32_t base) {
  const int l = 63 - __builtin_clzll(any);
  return base + 4u * (uint32_t)l + ((31u - (uint32_t)__builtin_clz(read_lane(m, l))) >> 3);
}
// is the quote whose third byte lies at t left alone (the in-place quirk above)?
__device__ __forceinline__ bool pf_suppressed(uint32_t flags, const PfDoc& d, uint32_t t) {
  return (flags & 16u) && d.s1 < t && t < d.s2 && d.qb == 0u;      // (d.s2 == PF_NONE: no second drop)
}
// the numbers of the span [rb, re) of a document.  `whole`: the span is the document, and the search for the drops may end at a quote
// candidate in front of the first of them as well (qb is then 1 whatever follows; spans are combined by positions and need the drops).
__device__ __forceinline__ PfPiece pf_scan_span(const TmWindow& win, uint32_t n, uint32_t rb, uint32_t re, uint32_t flags, bool whole, uint32_t lane) {
  PfPiece r{PF_NONE, PF_NONE, PF_NONE, PF_NONE, PF_NONE};
  const bool fused = (flags & 16u) && (flags & 128u), trim = flags & 32u;
  bool want_s = (flags & 8u) && (flags & 16u), want_a = trim;
  uint32_t carry = rb >= 4u ? pf_load(win, rb - 4u, n) : 0u;
#pragma unroll 1
  for (uint32_t base = rb; base < re && (want_s || want_a); base += 256u) {
    const uint32_t x = pf_load(win, base + 4u * lane, n);
    const uint32_t pd = TM_DPP(carry, x, 0x138, 0xF);           // wave_shr:1 - the dword of the lane below (lane 0: the last of the step before)
    carry = read_lane(x, 63);
    if (want_s) {
      const uint32_t p1 = (x << 8) | (pd >> 24), p2 = (x << 16) | (pd >> 16);      // the bytes one / two places in front of the lane's four
      uint32_t S = pf_eq(x, 0x20202020u) & pf_eq(p1, 0x20202020u);
      if (fused) S |= pf_eq(x, 0x0A0A0A0Au) & pf_eq(p1, 0x0D0D0D0Du);
      const uint32_t Q = pf_isq4(x) & pf_eq(p1, 0x80808080u) & pf_eq(p2, 0xE2E2E2E2u);
      unsigned long long mS = __ballot(S != 0u);
      while (mS != 0ull && r.s2 == PF_NONE) {
        const int l = __builtin_ctzll(mS);
        uint32_t m = read_lane(S, l);
        const uint32_t pos = base + 4u * (uint32_t)l + ((uint32_t)__builtin_ctz(m) >> 3);
        if (r.s1 == PF_NONE) r.s1 = pos; else r.s2 = pos;
        m &= m - 1u;
        if ((int)lane == l) S = m;
        if (m == 0u) mS &= mS - 1ull;
      }
      const unsigned long long mQ = __ballot(Q != 0u);
      if (mQ != 0ull && r.fq == PF_NONE) r.fq = pf_first(mQ, Q, base);
      if (r.s2 != PF_NONE || (whole && r.fq != PF_NONE && (r.s1 == PF_NONE || r.fq < r.s1))) want_s = false;
    }
    if (want_a) {
      const uint32_t N = pf_nb4(x);
      const unsigned long long mN = __ballot(N != 0u);
      if (mN != 0ull) { r.a = pf_first(mN, N, base); want_a = false; }
    }
  }
  if (trim && r.a != PF_NONE) {
#pragma unroll 1
    for (uint32_t base = rb + ((re - 1u - rb) & ~255u);; base -= 256u) {
      const uint32_t N = pf_nb4(pf_load(win, base + 4u * lane, n));
      const unsigned long long mN = __ballot(N != 0u);
      if (mN != 0ull) { r.z = pf_last(mN, N, base); break; }
      if (base == rb) break;
    }
  }
  return r;
}
// the document's numbers from those of its spans, in order (the numbers of one span: the document's own)
__device__ __forceinline__ void pf_add_span(PfDoc& r, uint32_t& fq, const PfPiece& p) {
  if (r.s1 == PF_NONE) { r.s1 = p.s1; r.s2 = p.s2; } else if (r.s2 == PF_NONE) r.s2 = p.s1;
  if (fq == PF_NONE) fq = p.fq;
  if (r.a == PF_NONE) r.a = p.a;
  if (p.z != PF_NONE) r.z = p.z;
}
__device__ __forceinline__ void pf_finish(PfDoc& r, uint32_t fq, const uint8_t* __restrict__ doc, uint32_t flags) {
  r.qb = fq < r.s1 ? 1u : 0u;
  r.zlo = r.z;
  // the last non-blank byte is the end of a quote that is replaced: the output byte begins two bytes earlier
  if (r.z != PF_NONE && (flags & 8u) && r.z >= 2u && pf_isq(doc[r.z]) && doc[r.z - 1u] == 0x80u && doc[r.z - 2u] == 0xE2u && !pf_suppressed(flags, r, r.z)) r.zlo = r.z - 2u;
}
// what the span [rb, re) of the document leaves: the number of bytes, and (EMIT) the bytes themselves at `out`
template <bool EMIT>
__device__ __forceinline__ uint32_t pf_emit_span(const uint8_t* __restrict__ doc, const TmWindow& win, uint32_t n, uint32_t rb, uint32_t re, uint32_t flags, const PfDoc& dd,
                                                 const uint32_t* __restrict__ acc, uint8_t* __restrict__ out, uint32_t lane) {
  const bool q = flags & 8u, c = flags & 16u, u = flags & 128u, trim = flags & 32u, lead = flags & 64u, accents = flags & 4u;
  // trim / leadingspace: [lo, hi) stays, and a space may go in front of it
  uint32_t lo = 0u, hi = n;
  bool pre = false;
  if (trim) {
    if (dd.a == PF_NONE) return 0u;                          // blank from end to end: nothing is left
    if (lead && dd.a == 0u) { if (dd.zlo == 0u) return 0u; hi = dd.zlo; pre = true; }      // (:274-277: the last non-blank byte goes as well)
    else { lo = dd.a; hi = dd.z + 1u; pre = lead; }
  } else if (lead) pre = doc[0] != ' ';                      // (n > 0: an empty document has no span)
  const uint32_t elo = max(lo, rb), ehi = min(hi, re);
  if (elo >= ehi) return 0u;
  uint32_t o = 0u;                                           // bytes of the span so far (wave-uniform)
  if (pre && lo >= rb) { if (EMIT && lane == 0u) out[0] = ' '; o = 1u; }
  const bool sup_on = q && c && dd.qb == 0u && dd.s1 != PF_NONE;
  // the sweep begins a dword in front of the first byte that stays: a quote that began there is found where it begins, and the first lane that
  // matters has its neighbour (nothing of what the sweep looks at lies further back than two bytes)
  uint32_t base = elo & ~3u;
  if (base >= 4u) base -= 4u;
  uint32_t xn = pf_load(win, base + 4u * lane, n), carry = 0u, carry_s = 0u;
#pragma unroll 1
  for (; base < ehi; base += 256u) {
    const uint32_t x = xn, off = base + 4u * lane;
    xn = pf_load(win, off + 256u, n);
    const uint32_t pd = TM_DPP(carry, x, 0x138, 0xF);                   // the dword of the lane below (lane 0: the last of the step before)
    const uint32_t nx = TM_DPP(read_lane(xn, 0), x, 0x130, 0xF);        // the dword of the lane above (lane 63: the first of the next step)
    carry = read_lane(x, 63);
    uint32_t D = 0u, y = x;                                              // bit 7 of a byte: it goes; the bytes as they leave
    if (c) D |= pf_eq(x, 0x20202020u) & pf_eq((x << 8) | (pd >> 24), 0x20202020u);
    if (u) D |= pf_eq(x, 0x0D0D0D0Du) & pf_eq((x >> 8) | (nx << 24), 0x0A0A0A0Au);
    if (q) {
      const uint32_t n1 = (x >> 8) | (nx << 24), n2 = (x >> 16) | (nx << 16);
      uint32_t S = pf_eq(x, 0xE2E2E2E2u) & pf_eq(n1, 0x80808080u) & pf_isq4(n2);      // a quote begins here
      if (sup_on && S != 0u) {
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) { const uint32_t t = off + j + 2u; if (dd.s1 < t && t < dd.s2) S &= ~(0x80u << (8u * j)); }
      }
      const uint32_t sp = TM_DPP(carry_s, S, 0x138, 0xF);
      carry_s = read_lane(S, 63);
      D |= ((S << 8) | (sp >> 24)) | ((S << 16) | (sp >> 16));           // its second and third byte go
      const uint32_t full = (S >> 7) * 0xFFu;
      y = (x & ~full) | ((0x27272727u ^ (((n2 >> 2) & 0x01010101u) * 5u)) & full);      // 98 99 -> ', 9C 9D -> "
    }
    if (accents && (x & 0x80808080u) != 0u) {
      const unsigned long long xx = (unsigned long long)x | ((unsigned long long)nx << 32);
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) {
        const uint32_t b = (x >> (8u * j)) & 0xFFu, i = off + j;
        if (b < 0x80u) continue;
        const uint32_t bn = (uint32_t)(xx >> (8u * j + 8u)) & 0xFFu, bp = j ? (x >> (8u * j - 8u)) & 0xFFu : pd >> 24;
        // (both halves of the character must have stayed: the cut of trim + leadingspace may have taken the second)
        if (nm_two_lead(b) && i + 1u < hi && nm_cont_byte(bn)) {
          const uint32_t e = acc[nm_two_index(b, bn)];
          if ((e & 3u) == 1u) D |= 0x80u << (8u * j);
          else if ((e & 3u) >= 2u) y = (y & ~(0xFFu << (8u * j))) | (((e >> 8) & 0xFFu) << (8u * j));
        } else if (nm_cont_byte(b) && i >= lo + 1u && nm_two_lead(bp)) {
          const uint32_t e = acc[nm_two_index(bp, b)];
          if ((e & 3u) == 1u || (e & 3u) == 2u) D |= 0x80u << (8u * j);
          else if ((e & 3u) == 3u) y = (y & ~(0xFFu << (8u * j))) | (((e >> 16) & 0xFFu) << (8u * j));
        }
      }
    }
    if (base < elo || base + 256u > ehi) {                               // the first and the last step: what lies outside [elo, ehi) goes
#pragma unroll
      for (uint32_t j = 0; j < 4u; j++) if (off + j < elo || off + j >= ehi) D |= 0x80u << (8u * j);
    }
    // the bytes that stay, packed: w, cnt of them
    uint32_t w = 0u, s = 0u;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
      const uint32_t gone = (D >> (8u * j + 7u)) & 1u;
      w |= (gone ? 0u : (y >> (8u * j)) & 0xFFu) << s;
      s += gone ? 0u : 8u;
    }
    const uint32_t cnt = s >> 3;
    const unsigned long long b0 = __ballot((cnt & 1u) != 0u), b1 = __ballot((cnt & 2u) != 0u), b2 = __ballot((cnt & 4u) != 0u);
    if (EMIT) {
      uint8_t* p = out + (o + mbcnt64(b0, 0u) + 2u * mbcnt64(b1, 0u) + 4u * mbcnt64(b2, 0u));
      if (cnt == 4u) __builtin_memcpy(p, &w, 4);
      else {
        if (cnt & 2u) { const uint16_t h = (uint16_t)w; __builtin_memcpy(p, &h, 2); p += 2; w >>= 16; }
        if (cnt & 1u) *p = (uint8_t)w;
      }
    }
    o += (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
  }
  return o;
}
// A wavefront per span (span_doc null: per document, every document one span or none).  A document that is one span is filtered here from end
// to end; of a longer one the span's numbers are left for the two kernels below.  rbegin / rend: the filtered document's range in `out`.
__global__ __launch_bounds__(256) void k_pf_filter(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ raw_off, const uint32_t* __restrict__ span_doc,
                                                   const uint64_t* __restrict__ doc_span_start, uint64_t nspans, uint32_t flags, const uint32_t* __restrict__ acc,
                                                   PfPiece* __restrict__ span_sum, uint8_t* __restrict__ out, uint64_t* __restrict__ rbegin, uint64_t* __restrict__ rend) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t k = (uint64_t)blockIdx.x * 4u + (uint64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (k >= nspans) return;
  const uint32_t d = span_doc ? span_doc[k] : (uint32_t)k;
  const uint64_t db = raw_off[d], ob = pf_out_begin(db, d);
  const uint32_t n = (uint32_t)(raw_off[d + 1] - db);
  if (n == 0u) { if (lane == 0u) { rbegin[d] = ob; rend[d] = ob; } return; }
  const bool whole = n <= PF_WHOLE;
  const uint32_t rb = span_doc ? (uint32_t)(k - doc_span_start[d]) * PF_SPAN : 0u, re = whole ? n : min(n, rb + PF_SPAN);
  const uint8_t* doc = raw + db;
  const TmWindow win = tm_window(doc, (n + 3u) & ~3u);
  const PfPiece p = pf_scan_span(win, n, rb, re, flags, whole, lane);
  if (!whole) { if (lane == 0u) span_sum[k] = p; return; }
  PfDoc dd{PF_NONE, PF_NONE, 0u, PF_NONE, PF_NONE, PF_NONE};
  uint32_t fq = PF_NONE;
  pf_add_span(dd, fq, p);
  pf_finish(dd, fq, doc, flags);
  const uint32_t len = pf_emit_span<true>(doc, win, n, rb, re, flags, dd, acc, out + ob, lane);
  if (lane == 0u) { rbegin[d] = ob; rend[d] = ob + len; }
}
// the spans of the documents of more than one: EMIT == false - the document's numbers (left in `docs` by its first span) and the bytes the span
// leaves (span_len); EMIT == true - the bytes themselves, behind those of the document's spans in front
template <bool EMIT>
__global__ __launch_bounds__(256) void k_pf_long(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ raw_off, const uint32_t* __restrict__ span_doc,
                                                 const uint64_t* __restrict__ doc_span_start, uint64_t nspans, uint32_t flags, const uint32_t* __restrict__ acc,
                                                 const PfPiece* __restrict__ span_sum, PfDoc* __restrict__ docs, uint32_t* __restrict__ span_len,
                                                 uint8_t* __restrict__ out, uint64_t* __restrict__ rbegin, uint64_t* __restrict__ rend) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t k = (uint64_t)blockIdx.x * 4u + (uint64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (k >= nspans) return;
  const uint32_t d = span_doc[k];
  const uint64_t db = raw_off[d], ob = pf_out_begin(db, d), k0 = doc_span_start[d];
  const uint32_t n = (uint32_t)(raw_off[d + 1] - db);
  if (n <= PF_WHOLE) return;
  const uint32_t rb = (uint32_t)(k - k0) * PF_SPAN, re = min(n, rb + PF_SPAN);
  const uint8_t* doc = raw + db;
  const TmWindow win = tm_window(doc, (n + 3u) & ~3u);
  if (!EMIT) {
    PfDoc dd{PF_NONE, PF_NONE, 0u, PF_NONE, PF_NONE, PF_NONE};
    uint32_t fq = PF_NONE;
    for (uint64_t j = k0; j < doc_span_start[d + 1]; j++) pf_add_span(dd, fq, span_sum[j]);
    pf_finish(dd, fq, doc, flags);
    const uint32_t len = pf_emit_span<false>(doc, win, n, rb, re, flags, dd, acc, nullptr, lane);
    if (lane == 0u) { span_len[k] = len; if (k == k0) docs[d] = dd; }
  } else {
    const PfDoc dd = docs[d];
    uint64_t at = 0;
    for (uint64_t j = k0; j < k; j++) at += span_len[j];
    const uint32_t len = pf_emit_span<true>(doc, win, n, rb, re, flags, dd, acc, out + ob + at, lane);
    if (lane == 0u && k + 1 == doc_span_start[d + 1]) { rbegin[d] = ob; rend[d] = ob + at + len; }
  }
}
// spans per document, for a batch with a document of more than one (an empty document has none: its range is written here)
__global__ void k_pf_begin(const uint64_t* __restrict__ raw_off, uint32_t ndocs, uint32_t* __restrict__ doc_nspan, uint64_t* __restrict__ rbegin, uint64_t* __restrict__ rend) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndocs) return;
  const uint64_t db = raw_off[d], n = raw_off[d + 1] - db;
  doc_nspan[d] = pf_spans(n);
  if (n == 0) { rbegin[d] = pf_out_begin(db, d); rend[d] = rbegin[d]; }
}

// One side of these two copies is pinned HOST memory, reached over PCIe: that side is accessed in aligned 16-byte units (a byte
// per lane made a 64-byte request per wavefront), the device side at whatever alignment is left.
template <bool DST_IS_HOST>
__device__ __forceinline__ void copy_doc(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t len) {
  const uintptr_t host_side = DST_IS_HOST ? (uintptr_t)dst : (uintptr_t)src;
  uint64_t head = (16u - (uint32_t)(host_side & 15u)) & 15u;
  if (head > len) head = len;
  for (uint64_t j = threadIdx.x; j < head; j += blockDim.x) dst[j] = src[j];
  const uint64_t nvec = (len - head) >> 4;
  for (uint64_t v = threadIdx.x; v < nvec; v += blockDim.x) {
    uint4 t;
    __builtin_memcpy(&t, src + head + 16 * v, 16);
    __builtin_memcpy(dst + head + 16 * v, &t, 16);
  }
  for (uint64_t j = head + 16 * nvec + threadIdx.x; j < len; j += blockDim.x) dst[j] = src[j];
}
__global__ void k_gather_docs(const uint8_t* __restrict__ raw, const uint64_t* __restrict__ raw_off, const uint32_t* __restrict__ doc_ids,
                              const uint64_t* __restrict__ dst_off, uint32_t n, uint8_t* __restrict__ staging) {
  const uint32_t k = blockIdx.x;
  if (k >= n) return;
  const uint64_t s = raw_off[doc_ids[k]], len = raw_off[doc_ids[k] + 1] - s, o = dst_off[k];
  copy_doc<true>(staging + o, raw + s, len);
}
// place the host-normalized documents after the device-normalized text and record their ranges
__global__ void k_place_fallback(const uint8_t* __restrict__ staging, const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ doc_ids,
                                 uint32_t n, uint64_t base, uint8_t* __restrict__ out, uint64_t* __restrict__ nbegin, uint64_t* __restrict__ nend) {
  const uint32_t k = blockIdx.x;
  if (k >= n) return;
  const uint64_t s = src_off[k], len = src_off[k + 1] - s;
  if (threadIdx.x == 0) { nbegin[doc_ids[k]] = base + s; nend[doc_ids[k]] = base + s + len; }
  copy_doc<false>(out + base + s, staging + s, len);
}

}  // namespace tmh

namespace {
// the normalizer's tables for {NFD, lowercase} x {capcode 2 or not}, built once per process from the host normalizer's functions (a few
// milliseconds of ICU calls: every lane's workspace uploads its own copy)
// flags: NFD 1, lowercase 2, accents 4.  `accents` (which implies NFD) strips the marks in the filter pass in front of the normalizer pass;
// what that pass does not handle - a two-byte character that would still decompose or is a non-spacing mark cannot reach this one, a
// three-byte mark can - has no entry here, so that its document goes to the host normalizer (with its original bytes).
const std::vector<uint8_t>& norm_tables(uint32_t flags, bool capcode2) {
  static std::mutex mu;
  static std::vector<uint8_t> cache[16];
  std::lock_guard<std::mutex> g(mu);
  const bool accents = (flags & 4u) != 0;
  if (accents) flags |= 1u;
  std::vector<uint8_t>& t = cache[(flags & 7u) | (capcode2 ? 8u : 0u)];
  if (t.empty()) {
    t.resize(NM_TABLE_BYTES);
    NmTwo* two = reinterpret_cast<NmTwo*>(t.data());
    uint32_t* blk = reinterpret_cast<uint32_t*>(two + NM_TWO_SIZE);
    build_two_table(flags & 3u, two);
    if (!capcode2) for (int k = 0; k < NM_TWO_SIZE; k++) if (two[k].a & (NT_DECOMP | NT_DECOMP2)) two[k].a = 0;      // (without capcode the pass keeps lengths: decomposing characters take the host path)
    build_three_tables(flags & 3u, blk, blk + NM_BLK_WORDS);
    build_four_table(flags & 3u, blk + NM_BLK_WORDS + NM_CP_WORDS);
    blk[NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS] = (capcode2 && (flags & 1u)) ? NM_MISC_HANGUL : 0u;      // NFD of the Hangul syllables: by arithmetic, where the pass may change lengths
    NmLea* lea = reinterpret_cast<NmLea*>(blk + NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS + 4);
    for (int k = 0; k < NM_LEA_SIZE; k++) lea[k] = NmLea{0, 0};
    if (capcode2 && (flags & 1u) && !accents) {      // ... and of Latin Extended Additional: a letter and one or two marks (under `accents` the marks would have to go: the host)
      build_lea_table(flags & 3u, lea);
      blk[NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS] |= NM_MISC_LEA;
    }
    uint16_t* kana = reinterpret_cast<uint16_t*>(lea + NM_LEA_SIZE);
    for (int k = 0; k < NM_KANA_SIZE; k++) kana[k] = 0;
    if (capcode2 && (flags & 1u) && !accents) {      // ... and of the voiced kana: a kana and its mark
      build_kana_table(kana);
      blk[NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS] |= NM_MISC_KANA;
    }
    uint8_t* ccc = reinterpret_cast<uint8_t*>(kana + NM_KANA_SIZE);
    for (uint32_t k = 0; k < NM_CCC_SIZE; k++) ccc[k] = 0;
    // ... the three-byte marks of canonical class > 0 (Indic, Thai ...): in place where they stand in order (under NFD without `accents`; without
    // NFD they are inert, build_three_tables), and the three-byte digits
    build_ccc_table(flags & 3u, (flags & 1u) && !accents, ccc);
    blk[NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS] |= NM_MISC_CCC;
    uint32_t* dec3 = reinterpret_cast<uint32_t*>(ccc + NM_CCC_SIZE);
    for (uint32_t k = 0; k < NM_DEC3_SIZE + NM_DEC3_THIRDS; k++) dec3[k] = 0;
    if (capcode2 && (flags & 1u) && !accents) {      // ... and the three-byte characters NFD splits in two (the two-part vowel signs of Bengali, Tamil ...; the nukta letters)
      build_dec3_table(dec3);
      blk[NM_BLK_WORDS + NM_CP_WORDS + NM_BLK4_WORDS] |= NM_MISC_DEC3;
    }
    if (accents) {
      std::vector<uint32_t> acc(NM_TWO_SIZE);
      build_accent_table(acc.data());
      for (int k = 0; k < NM_TWO_SIZE; k++) if ((acc[k] & 3u) != 0u || (two[k].a & (NT_DECOMP | NT_DECOMP2))) two[k].a = 0;      // (the filter pass has dealt with the first kind; a decomposition whose mark stays is rare enough for the host)
      uint32_t* cpt = blk + NM_BLK_WORDS;
      for (int w = 0; w < NM_CP_WORDS; w++) for (int j = 0; j < 16; j++) if (((cpt[w] >> (2 * j)) & 3u) == 3u) cpt[w] &= ~(3u << (2 * j));      // three-byte marks (a variation selector is Mn): the host
    }
  }
  return t;
}
template <typename T>
hipError_t grow(T** p, uint64_t* cap, uint64_t need, uint64_t slack = 0) {
  if (*p && *cap >= need) return hipSuccess;
  if (*p) trace_grow("normalizer buffer", need * sizeof(T));
  (void)hipFree(*p);
  *p = nullptr;
  *cap = need + need / 4 + slack;
  return hipMalloc((void**)p, *cap * sizeof(T));
}
}  // namespace

extern "C" {

int tm_batch_upload_raw(tm_batch* b, const uint8_t* raw, const uint64_t* raw_offsets, uint32_t ndocs) {
  int rc = tmh::batch_upload_raw_on(b, raw, raw_offsets, ndocs, nullptr);
  if (rc == TM_OK) { hipError_t e = hipStreamSynchronize(nullptr); if (e != hipSuccess) rc = hip_fail(e, "H2D raw text"); }
  return rc;
}

}  // extern "C"
namespace tmh {
int batch_upload_raw_on(tm_batch* b, const uint8_t* raw, const uint64_t* raw_offsets, uint32_t ndocs, hipStream_t st) {
  if (!b || (ndocs && !raw_offsets)) return set_error(TM_E_INVALID, "null argument");
  { int rc = enter_device(b->vocab); if (rc != TM_OK) return rc; }
  if (ndocs > b->max_docs) return set_error(TM_E_LIMIT, "batch has %u documents, workspace sized for %u", ndocs, b->max_docs);
  const uint64_t nbytes = ndocs ? raw_offsets[ndocs] : 0;
  if (ndocs && raw_offsets[0] != 0) return set_error(TM_E_INVALID, "offsets[0] must be 0");
  uint64_t npieces = 0;
  const uint64_t grows = (b->vocab->host.norm_flag & 64u) ? 1u : 0u;       // leadingspace: the filter pass may put a byte in front of a document
  for (uint32_t d = 0; d < ndocs; d++) {
    if (raw_offsets[d + 1] < raw_offsets[d]) return set_error(TM_E_INVALID, "offsets not monotone at document %u", d);
    npieces += (raw_offsets[d + 1] - raw_offsets[d] + grows + PIECE - 1) / PIECE;
  }
  { int rc = raw_prepare(b, nbytes, ndocs, npieces, st); if (rc != TM_OK) return rc; }
  hipError_t e;
  if (nbytes && (e = hipMemcpyAsync(b->d_raw, raw, nbytes, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(e, "H2D raw text");
  b->h_raw_off.assign(raw_offsets, raw_offsets + (ndocs ? ndocs + 1 : 0));      // (the batch's own copy: it outlives the caller's array)
  if (ndocs) { int rc = small_h2d(b, b->d_raw_off, b->h_raw_off.data(), ((uint64_t)ndocs + 1) * 8, st); if (rc != TM_OK) return rc; }
  return TM_OK;
}

// the device buffers of a raw batch of this size (grow-only), the normalizer's tables on first use, and the batch's counts
int raw_prepare(tm_batch* b, uint64_t nbytes, uint32_t ndocs, uint64_t npieces, hipStream_t st) {
  // the normalized text cannot be shorter than what capcode leaves of the raw text, and the scans of tm_batch_normalize run over
  // one entry per piece with block sums sized (make_workspace) for max_bytes / 256 + max_docs entries
  if (ndocs > b->max_docs) return set_error(TM_E_LIMIT, "batch has %u documents, workspace sized for %u", ndocs, b->max_docs);
  if (nbytes > b->max_bytes) return set_error(TM_E_LIMIT, "raw batch has %llu bytes, workspace sized for %llu", (unsigned long long)nbytes, (unsigned long long)b->max_bytes);
  if (npieces > b->max_bytes / 256 + (uint64_t)b->max_docs) return set_error(TM_E_LIMIT, "raw batch has %llu pieces, workspace scans hold %llu", (unsigned long long)npieces, (unsigned long long)(b->max_bytes / 256 + b->max_docs));
  hipError_t e;
  uint64_t docs_cap = b->raw_docs_cap;
  if ((e = grow(&b->d_raw, &b->raw_cap, nbytes + 256)) != hipSuccess) return hip_fail(e, "hipMalloc (raw text)");
  if (!b->d_raw_off || ndocs + 2 > docs_cap) {
    void** ps[] = {(void**)&b->d_raw_off, (void**)&b->d_doc_npiece, (void**)&b->d_doc_piece_start, (void**)&b->d_need_host, (void**)&b->d_nbegin, (void**)&b->d_nend,
                   (void**)&b->d_fb_ids, (void**)&b->d_fb_roff, (void**)&b->d_fb_noff};
    if (b->d_raw_off) trace_grow("normalizer per-document arrays", (uint64_t)ndocs * 60);
    for (void** q : ps) { (void)hipFree(*q); *q = nullptr; }
    docs_cap = (uint64_t)ndocs + ndocs / 4 + 16;
    if ((e = hipMalloc((void**)&b->d_raw_off, docs_cap * 8)) != hipSuccess || (e = hipMalloc((void**)&b->d_doc_npiece, docs_cap * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_doc_piece_start, (docs_cap + 1) * 8)) != hipSuccess || (e = hipMalloc((void**)&b->d_need_host, docs_cap)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_nbegin, docs_cap * 8)) != hipSuccess || (e = hipMalloc((void**)&b->d_nend, docs_cap * 8)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_fb_ids, docs_cap * 4)) != hipSuccess || (e = hipMalloc((void**)&b->d_fb_roff, (docs_cap + 1) * 8)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_fb_noff, (docs_cap + 1) * 8)) != hipSuccess)
      return hip_fail(e, "hipMalloc (raw documents)");
    b->raw_docs_cap = (uint32_t)std::min<uint64_t>(docs_cap - 2, 0xFFFFFFFFull);
  }
  if (!b->d_ninfo && (e = hipMalloc((void**)&b->d_ninfo, 64)) != hipSuccess) return hip_fail(e, "hipMalloc");
  if (!b->d_two) {
    // what the vocabulary's flags {NFD, lowercase} do to the two-byte characters U+0080..U+017F, from the host normalizer's own building blocks
    if ((e = hipMalloc((void**)&b->d_two, NM_TABLE_BYTES)) != hipSuccess) return hip_fail(e, "hipMalloc");
    const std::vector<uint8_t>& tab = norm_tables(b->vocab->host.norm_flag & 7u, b->vocab->host.capcode == 2);
    int rc = small_h2d(b, b->d_two, tab.data(), NM_TABLE_BYTES, st);
    if (rc != TM_OK) return rc;
  }
  if (!b->d_piece_doc || npieces + 2 > b->piece_cap) {
    void** ps[] = {(void**)&b->d_piece_doc, (void**)&b->d_piece_sum, (void**)&b->d_piece_carry, (void**)&b->d_piece_len, (void**)&b->d_piece_off};
    if (b->d_piece_doc) trace_grow("normalizer per-piece arrays", npieces * 21);
    for (void** q : ps) { (void)hipFree(*q); *q = nullptr; }
    b->piece_cap = npieces + npieces / 4 + 16;
    if ((e = hipMalloc((void**)&b->d_piece_doc, b->piece_cap * 4)) != hipSuccess || (e = hipMalloc((void**)&b->d_piece_sum, b->piece_cap * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_piece_carry, b->piece_cap)) != hipSuccess || (e = hipMalloc((void**)&b->d_piece_len, b->piece_cap * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_piece_off, (b->piece_cap + 1) * 8)) != hipSuccess)
      return hip_fail(e, "hipMalloc (pieces)");
  }
  if ((e = grow(&b->d_slab, &b->slab_cap, (npieces + 1) * (uint64_t)SLAB)) != hipSuccess) return hip_fail(e, "hipMalloc (normalizer slabs)");
  b->raw_bytes = nbytes;
  b->raw_docs = ndocs;
  b->raw_pieces = npieces;
  return TM_OK;
}

// Would batch_upload_raw_on(these documents) replace a buffer that the batch in flight still reads?  After its normalizer pass that is the
// per-document ranges (always) and - when the text stays in the slabs for k_match_branch - the slabs and the pieces' offsets.  The
// host-to-host pipeline asks before it uploads a lane's next chunk beside the tokenizer kernels of the current one.
bool raw_upload_replaces_buffers(const tm_batch* b, const uint64_t* raw_offsets, uint32_t ndocs) {
  if (!b || !b->d_raw_off || !b->d_piece_doc) return true;
  uint64_t npieces = 0;
  const uint64_t grows = (b->vocab->host.norm_flag & 64u) ? 1u : 0u;       // (as batch_upload_raw_on counts them)
  for (uint32_t d = 0; d < ndocs; d++) npieces += (raw_offsets[d + 1] - raw_offsets[d] + grows + PIECE - 1) / PIECE;
  return (uint64_t)ndocs + 2 > b->raw_docs_cap || npieces + 2 > b->piece_cap || (npieces + 1) * (uint64_t)SLAB > b->slab_cap;
}

// the slabs of the current batch packed into d_text (what tm_batch_normalize leaves out in the usual case); piece lengths and offsets are
// those of the last normalize call
void pack_text(tm_batch* b, hipStream_t st) {
  if (!b->text_in_slabs) return;
  b->text_in_slabs = false;
  const uint64_t np = b->slab_pieces;
  if (np > 0) TM_LAUNCH(k_norm_compact, (uint32_t)((np + 3) / 4), 256, 0, st, b->d_slab, b->d_piece_len, b->d_piece_off, np, b->d_text, b->max_bytes);
}
}  // namespace tmh
extern "C" {


// The grid of k_norm_emit2: its wavefronts take piece after piece, so that the tables are staged once per workgroup instead of once per four
// pieces - but NOT one workgroup per slot of the device (6 per compute unit: 25 KB of LDS each): the dispatcher does not spread those evenly, and
// a compute unit that got a seventh finishes late.  Measured per 512 MiB (MI355X, 256 compute units): one piece per wavefront 1.54 ms; 6 workgroups
// per compute unit 1.58; 12: 1.49; 24: 1.43; 48 - 96: 1.365; 256: 1.43.  (TM_NORM_WG_PER_CU: the sweep.)
static uint32_t norm_grid() {
  // (once per device and process: this runs per chunk and lane of the host-to-host pipeline)
  static std::mutex mu;
  static uint32_t cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  std::lock_guard<std::mutex> g(mu);
  uint32_t& slot = cached[dev >= 0 && dev < 64 ? dev : 0];
  if (slot == 0) {
    const char* e = getenv("TM_NORM_WG_PER_CU");
    const int v = e ? atoi(e) : 0;
    const uint32_t wg_per_cu = v > 0 ? (uint32_t)v : 64u;
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) { (void)hipGetLastError(); cu = 256; }
    slot = (uint32_t)cu * wg_per_cu;
  }
  return slot;
}

// The filter pass (k_pf_*, above) on the uploaded text, enqueued: *R = the filtered text, *RB / *RE = where its documents begin and end (device); the
// table of its pieces (k_norm_begin + scan) is made here as well, their count is left in d_totals[3].
static int prefilter_enqueue(tm_batch* b, hipStream_t st, uint32_t norm_flag, const uint64_t* h_raw_off, const uint8_t** R, const uint64_t** RB, const uint64_t** RE) {
  const uint32_t nd = b->raw_docs;
  hipError_t e;
  uint64_t nspans = 0;
  bool any_long = false;
  for (uint32_t d = 0; d < nd; d++) {
    const uint64_t len = h_raw_off[d + 1] - h_raw_off[d];
    if (len >= 0xFFFF0000ull) return set_error(TM_E_LIMIT, "document %u has %llu bytes: beyond what the filter pass addresses", d, (unsigned long long)len);
    nspans += pf_spans(len);
    any_long = any_long || len > PF_WHOLE;
  }
  const uint64_t out_cap = b->raw_bytes + (uint64_t)PF_GAP * nd + 256;
  if ((e = grow(&b->d_rawf, &b->rawf_cap, out_cap)) != hipSuccess) return hip_fail(e, "hipMalloc (filtered text)");
  if (!b->d_rawf_off || b->rawf_docs_cap < (uint64_t)nd + 2) {
    (void)hipFree(b->d_rawf_off); (void)hipFree(b->d_pf_doc);
    b->d_rawf_off = nullptr; b->d_pf_doc = nullptr;
    b->rawf_docs_cap = (uint64_t)nd + nd / 4 + 16;
    if ((e = hipMalloc((void**)&b->d_rawf_off, b->rawf_docs_cap * 16)) != hipSuccess || (e = hipMalloc((void**)&b->d_pf_doc, b->rawf_docs_cap * sizeof(PfDoc))) != hipSuccess)
      return hip_fail(e, "hipMalloc (filtered documents)");
  }
  if (any_long) {
    uint8_t* pp = (uint8_t*)b->d_pf_piece; uint64_t cap = b->pf_piece_cap;
    if ((e = grow(&pp, &cap, (nspans + 1) * sizeof(PfPiece))) != hipSuccess) return hip_fail(e, "hipMalloc (filter pass)");
    b->d_pf_piece = pp; b->pf_piece_cap = cap;
    if (nspans + 2 > b->piece_cap) return set_error(TM_E_INTERNAL, "the filter pass has %llu spans, the workspace holds %llu pieces", (unsigned long long)nspans, (unsigned long long)b->piece_cap);
  }
  if ((norm_flag & 4u) && !b->d_acc) {
    std::vector<uint32_t> acc(NM_TWO_SIZE);
    build_accent_table(acc.data());
    if ((e = hipMalloc((void**)&b->d_acc, acc.size() * 4)) != hipSuccess) return hip_fail(e, "hipMalloc");
    int rc = small_h2d(b, b->d_acc, acc.data(), acc.size() * 4, st);
    if (rc != TM_OK) return rc;
  }
  uint64_t* rb = b->d_rawf_off;
  uint64_t* re = b->d_rawf_off + b->rawf_docs_cap;
  if (!any_long)
    TM_LAUNCH(k_pf_filter, (nd + 3) / 4, 256, 0, st, b->d_raw, b->d_raw_off, nullptr, nullptr, (uint64_t)nd, norm_flag, b->d_acc, nullptr, b->d_rawf, rb, re);
  else {
    // a document of more than one span: the spans' table (in the arrays the pieces' table takes afterwards), and three launches over it
    PfPiece* sums = (PfPiece*)b->d_pf_piece;
    PfDoc* docs = (PfDoc*)b->d_pf_doc;
    const uint32_t sgrid = (uint32_t)((nspans + 3) / 4);
    TM_LAUNCH(k_pf_begin, (nd + 255) / 256, 256, 0, st, b->d_raw_off, nd, b->d_doc_npiece, rb, re);
    scan_u32(b->d_doc_npiece, nd, b->d_scan_tmp, b->d_totals + 3, b->d_doc_piece_start, st);
    launch_unit_owner(b->d_doc_piece_start, nd, nspans, b->d_piece_doc, st);
    TM_LAUNCH(k_pf_filter, sgrid, 256, 0, st, b->d_raw, b->d_raw_off, b->d_piece_doc, b->d_doc_piece_start, nspans, norm_flag, b->d_acc, sums, b->d_rawf, rb, re);
    TM_LAUNCH(k_pf_long<false>, sgrid, 256, 0, st, b->d_raw, b->d_raw_off, b->d_piece_doc, b->d_doc_piece_start, nspans, norm_flag, b->d_acc, sums, docs, b->d_piece_len, nullptr, rb, re);
    TM_LAUNCH(k_pf_long<true>, sgrid, 256, 0, st, b->d_raw, b->d_raw_off, b->d_piece_doc, b->d_doc_piece_start, nspans, norm_flag, b->d_acc, sums, docs, b->d_piece_len, b->d_rawf, rb, re);
  }
  // the pieces of the filtered documents
  TM_LAUNCH(k_norm_begin, (std::max(nd, 8u) + 255) / 256, 256, 0, st, rb, re, nd, b->d_doc_npiece, b->d_need_host, (unsigned long long*)b->d_ninfo);
  scan_u32(b->d_doc_npiece, nd, b->d_scan_tmp, b->d_totals + 3, b->d_doc_piece_start, st);
  *R = b->d_rawf; *RB = rb; *RE = re;
  return TM_OK;
}
// ... for tm_batch_normalize: with one trip to the host of its own (*np = the count of the filtered documents' pieces: the normalizer pass is launched over it)
static int prefilter(tm_batch* b, hipStream_t st, uint32_t norm_flag, const uint8_t** R, const uint64_t** RB, const uint64_t** RE, uint64_t* np_out) {
  int rc = prefilter_enqueue(b, st, norm_flag, b->h_raw_off.data(), R, RB, RE);
  if (rc != TM_OK) return rc;
  hipError_t e;
  uint64_t np = 0;
  if ((e = hipMemcpyAsync(&np, b->d_totals + 3, 8, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "D2H filtered piece count");
  if (np + 2 > b->piece_cap || (np + 1) * (uint64_t)SLAB > b->slab_cap * sizeof(b->d_slab[0]))
    return set_error(TM_E_INTERNAL, "the filter pass left %llu pieces, the workspace holds %llu", (unsigned long long)np, (unsigned long long)b->piece_cap);
  *np_out = np;
  return TM_OK;
}

int tm_batch_normalize(tm_batch* b, void* stream) {
  if (!b) return set_error(TM_E_INVALID, "null argument");
  const tm_vocab* v = b->vocab;
  { int rc = enter_device(v); if (rc != TM_OK) return rc; }
  const uint32_t capcode = v->host.capcode, norm_flag = v->host.norm_flag;
  if (!normalize_supported(capcode, norm_flag))
    return set_error(TM_E_INVALID, "normalization flags %u / capcode %u not supported by the normalizer", norm_flag, capcode);
  hipStream_t st = (hipStream_t)stream;
  const uint32_t nd = b->raw_docs;
  uint64_t np = b->raw_pieces;
  hipError_t e;
  const uint8_t* R = b->d_raw;                 // the text the normalizer pass reads, and its documents: the upload's, or what the filter pass makes of it
  const uint64_t* RB = b->d_raw_off;
  const uint64_t* RE = b->d_raw_off + 1;
  const bool filtered = (norm_flag & ~3u) && nd > 0 && (capcode == 0 || capcode == 2);
  if (filtered) {
    int rc = prefilter(b, st, norm_flag, &R, &RB, &RE, &np);
    if (rc != TM_OK) return rc;
  }
  b->host_fallback_docs = 0;
  b->d_doc_begin = b->d_nbegin;
  b->d_doc_end = b->d_nend;
  b->ndocs = nd; b->nbytes = 0; b->nseg = 0; b->ngroups = 0; b->nlong = 0;
  b->text_in_slabs = false;
  b->slab_pieces = np;
  if (nd == 0) return TM_OK;
  static const bool trace = getenv("TM_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  const uint32_t lower_all = (norm_flag & 2u) ? 1u : 0u;
  unsigned long long* ninfo = (unsigned long long*)b->d_ninfo;
  // piece table
  if (!filtered) {                             // (the filter pass has made the table of ITS documents' pieces)
    TM_LAUNCH(k_norm_begin, (std::max(nd, 8u) + 255) / 256, 256, 0, st, RB, RE, nd, b->d_doc_npiece, b->d_need_host, ninfo);
    scan_u32(b->d_doc_npiece, nd, b->d_scan_tmp, b->d_totals + 3, b->d_doc_piece_start, st);
  }
  const uint32_t pgrid = (uint32_t)((np + 3) / 4);
  const uint32_t egrid = std::min(pgrid, norm_grid());       // k_norm_emit2: its wavefronts take piece after piece
  if (np > 0) launch_unit_owner(b->d_doc_piece_start, nd, np, b->d_piece_doc, st);
  unsigned long long h_info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // The usual path is ONE pass over the raw text: k_norm_emit2<false> takes the carries of a piece from the 64 bytes either side of it and
  // finds the documents that need the host normalizer as it goes.  A piece whose margins cannot tell (a run of 64 digits / apostrophes /
  // capitals across a piece boundary) or whose output outgrows its slab sends the batch through the exact path after all: summaries of the
  // pieces, carries per document, then the same emit kernel with the carries given.
  bool fast = np > 0 && (capcode == 2 || capcode == 0) && normalize_on_device(capcode, norm_flag) && !(tm_debug_flags(-1) & 256);
  // ... and ONE trip to the host: the whole device part — the pass, the scan of the piece lengths, the compaction, the documents' ranges
  // and segment counts — is enqueued before anything comes back, and what comes back is everything at once: how many documents need the
  // host normalizer, whether a piece was undecided or outgrew its slab, the normalized size, the segment and long-document counts.  (A
  // chunk of the host-to-host pipeline used to wait for the device three times here; with four lanes' launches queueing on the runtime's
  // lock every wait was also a gap in its stream.)  `pre`: the device part is done and valid.
  bool pre = false;
  uint64_t pre_bytes = 0;
  if (fast) {
    if (capcode == 2)
      TM_LAUNCH(k_norm_emit2<false>, egrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, lower_all, nullptr,
                                                 b->d_need_host, b->d_piece_len, b->d_slab, ninfo + 3, b->d_two, nullptr);
    else
      TM_LAUNCH(k_norm_emit<3>, pgrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, capcode, lower_all,
                                            nullptr, b->d_need_host, b->d_piece_len, nullptr, b->d_slab, ninfo + 3, b->d_two, nullptr);
    TM_LAUNCH(k_norm_bad, (uint32_t)((std::max<uint64_t>(np, nd) + 255) / 256), 256, 0, st, b->d_piece_doc, b->d_need_host, b->d_doc_piece_start, np, nd, b->d_piece_len, ninfo, b->d_fb_ids, nullptr);
    scan_u32(b->d_piece_len, np, b->d_scan_tmp, reinterpret_cast<uint64_t*>(ninfo + 5), b->d_piece_off, st);      // (the total lands beside the info words: one copy brings everything)
    // NO compaction pass here: in the usual case the text stays in the slabs and k_match_branch stages its segments from there (k_seg_src) —
    // packing it was 2.2 GB of traffic and 0.8 ms per GiB; pack_text() below is for the cases that need the packed text after all
    TM_LAUNCH(k_norm_ranges_info, (nd + 255) / 256, 256, 0, st, b->d_piece_off, b->d_doc_piece_start, b->d_need_host, nd, b->d_nbegin, b->d_nend, ninfo, long_segs());
    { int rc = small_d2h(b, h_info, ninfo, 56, st); if (rc == TM_OK) rc = small_sync(b, st);
      if (rc != TM_OK) return rc; }
    pre_bytes = h_info[5];
    if (h_info[3] != 0 || h_info[4] != 0) {        // a piece whose margins could not tell, or one that outgrew its slab: the exact path, from the start
      fast = false;
      (void)hipMemsetAsync(ninfo, 0, 64, st);
    } else if (h_info[0] == 0) {                    // the usual case: every document was normalized on the device
      if (pre_bytes > b->max_bytes)
        return set_error(TM_E_LIMIT, "normalized text needs %llu bytes, workspace sized for %llu", (unsigned long long)pre_bytes, (unsigned long long)b->max_bytes);
      b->nbytes = pre_bytes;
      b->nseg = h_info[2];
      b->text_in_slabs = true;
      if (h_info[6] != 0 || (tm_debug_flags(-1) & 2048)) pack_text(b, st);        // a short piece inside a document (or the test hook): the packed text after all
      int rc = TM_OK;
      if (h_info[1] > 0) {
        std::vector<uint64_t> hb(nd), he(nd);
        if ((e = hipMemcpyAsync(hb.data(), b->d_nbegin, (size_t)nd * 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
            (e = hipMemcpyAsync(he.data(), b->d_nend, (size_t)nd * 8, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "D2H ranges");
        rc = build_groups(b, hb.data(), he.data(), nd, st);
      }
      if (trace) fprintf(stderr, "[tm_batch_normalize] one trip: %.2f ms\n", now() - t0);
      return rc;
    } else {                                        // some documents need the host normalizer: fetch, normalize, place them behind the device part below
      pre = true;
      b->text_in_slabs = true;
      pack_text(b, st);
    }
  }
  if (!fast) {
    if (np > 0) TM_LAUNCH(k_norm_summary, pgrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, lower_all, b->d_two, b->d_piece_sum);
    TM_LAUNCH(k_norm_carry, (nd + 255) / 256, 256, 0, st, b->d_piece_sum, b->d_doc_piece_start, nd, b->d_piece_carry, b->d_need_host, ninfo, b->d_fb_ids,
                                                    0u);
    int rc = small_d2h(b, h_info, ninfo, 8, st); if (rc == TM_OK) rc = small_sync(b, st); if (rc != TM_OK) return rc;
  }
  const double t1 = now();
  const uint32_t nf = (uint32_t)h_info[0];
  std::vector<uint32_t> ids;
  std::vector<uint64_t> roff;
  double f1 = now(), f2 = 0, f3 = 0, f4 = 0;
  // the exact path: the device normalizes its documents with the carries given (one pass into per-piece slabs, lengths on the side) ...
  if (!fast && np > 0 && capcode == 2 && !(tm_debug_flags(-1) & 256))
    TM_LAUNCH(k_norm_emit2<true>, egrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, lower_all, b->d_piece_carry,
                                              b->d_need_host, b->d_piece_len, b->d_slab, ninfo + 3, b->d_two, nullptr);
  else if (!fast && np > 0)           // capcode 0, or debug bit 8: the per-lane version of the rules
    TM_LAUNCH(k_norm_emit<2>, pgrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, capcode, lower_all,
                                          b->d_piece_carry, b->d_need_host, b->d_piece_len, nullptr, b->d_slab, ninfo + 3, b->d_two, nullptr);
  if (!pre) scan_u32(b->d_piece_len, np, b->d_scan_tmp, b->d_totals + 2, b->d_piece_off, st);
  if (!pre && np > 0 && nf == 0) TM_LAUNCH(k_norm_short, (uint32_t)((np + 255) / 256), 256, 0, st, b->d_piece_doc, b->d_need_host, b->d_doc_piece_start, np, b->d_piece_len, ninfo);
  if (nf > 0) {
    // ... while the documents it cannot normalize (other non-ASCII content: NFD / Unicode case need ICU) are fetched on a
    // second stream, so that the fetch does not hold up the pass above
    if (!b->aux_stream && (e = hipStreamCreateWithFlags(&b->aux_stream, hipStreamNonBlocking)) != hipSuccess) return hip_fail(e, "hipStreamCreate");
    hipStream_t sx = b->aux_stream;
    // the (unordered) list k_norm_carry / k_norm_bad has left on the device, put into document order
    ids.resize(nf);
    { int rc = small_d2h(b, ids.data(), b->d_fb_ids, (uint64_t)nf * 4, sx); if (rc == TM_OK) rc = small_sync(b, sx); if (rc != TM_OK) return rc; }
    std::sort(ids.begin(), ids.end());
    roff.assign(ids.size() + 1, 0);
    for (size_t k = 0; k < ids.size(); k++) roff[k + 1] = roff[k] + (b->h_raw_off[ids[k] + 1] - b->h_raw_off[ids[k]]);
    if (!b->h_fb_raw || b->h_fb_raw_cap < roff.back() + 16) {
      (void)hipHostFree(b->h_fb_raw);
      b->h_fb_raw = nullptr;
      b->h_fb_raw_cap = roff.back() + roff.back() / 4 + 4096;
      if ((e = hipHostMalloc((void**)&b->h_fb_raw, b->h_fb_raw_cap, hipHostMallocDefault)) != hipSuccess) return hip_fail(e, "hipHostMalloc (fallback staging)");
    }
    { int rc = small_h2d(b, b->d_fb_ids, ids.data(), ids.size() * 4, sx); if (rc == TM_OK) rc = small_h2d(b, b->d_fb_roff, roff.data(), roff.size() * 8, sx); if (rc != TM_OK) return rc; }
    // the gather writes straight into the pinned host buffer (no staging copy, nothing on the copy engines)
    TM_LAUNCH(k_gather_docs, (uint32_t)ids.size(), 256, 0, sx, b->d_raw, b->d_raw_off, b->d_fb_ids, b->d_fb_roff, (uint32_t)ids.size(), b->h_fb_raw);
    if ((e = hipStreamSynchronize(sx)) != hipSuccess) return hip_fail(e, "fallback documents to the host");
    f2 = now();
  }
  uint64_t gpu_bytes = 0;
  // the host normalizes those; they are appended after the device part
  uint8_t* hnorm = nullptr;
  std::vector<uint64_t> noff(ids.size() + 1, 0);
  if (nf > 0) {
    // pooled workers: up to 128, but only this process's share of the host when there is one process per GPU of the node
    static const uint32_t host_share = std::max(8u, std::max(1u, std::thread::hardware_concurrency()) / (uint32_t)std::max(1, tm_device_count()));
    // ... and only as many as the bytes are worth: the pool runs one job at a time, so a chunk of the host-to-host pipeline (a few
    // dozen fallback documents) normalizes them on its own lane thread and leaves the pool to batches that need it
    const uint32_t threads = (uint32_t)std::min<size_t>(std::min<size_t>(std::min<uint32_t>(128u, host_share), ids.size() / 8 + 1), (size_t)(roff.back() >> 18) + 1);
    hipError_t he = hipSuccess;
    int rc = normalize_batch_into(b->h_fb_raw, roff.data(), (uint32_t)ids.size(), capcode, norm_flag, threads, noff.data(), [&](uint64_t total) -> uint8_t* {
      if (!b->h_fb_norm || b->h_fb_norm_cap < total + 16) {                         // pinned: the placement kernel reads it where it lies
        (void)hipHostFree(b->h_fb_norm);
        b->h_fb_norm = nullptr;
        b->h_fb_norm_cap = total + total / 4 + 4096;
        if ((he = hipHostMalloc((void**)&b->h_fb_norm, b->h_fb_norm_cap, hipHostMallocDefault)) != hipSuccess) return nullptr;
      }
      return b->h_fb_norm;
    });
    if (he != hipSuccess) return hip_fail(he, "hipHostMalloc (fallback output)");
    if (rc != TM_OK) return rc;
    hnorm = b->h_fb_norm;
    f3 = now();
  }
  { int rc = small_d2h(b, h_info, ninfo, 56, st); if (rc == TM_OK && !pre) rc = small_d2h(b, &gpu_bytes, b->d_totals + 2, 8, st); if (rc == TM_OK) rc = small_sync(b, st);
    if (rc != TM_OK) return rc; }
  if (pre) gpu_bytes = pre_bytes;
  if (gpu_bytes + noff.back() > b->max_bytes) {
    return set_error(TM_E_LIMIT, "normalized text needs %llu bytes, workspace sized for %llu", (unsigned long long)(gpu_bytes + noff.back()), (unsigned long long)b->max_bytes);
  }
  // the host-normalized documents go behind the device part; placing them (reads over PCIe) runs on the second stream beside the
  // compaction of the device part — different bytes of d_text, different documents' ranges
  if (nf > 0) {
    hipStream_t sx = b->aux_stream;
    { int rc = small_h2d(b, b->d_fb_noff, noff.data(), noff.size() * 8, sx); if (rc != TM_OK) return rc; }
    TM_LAUNCH(k_place_fallback, (uint32_t)ids.size(), 256, 0, sx, hnorm, b->d_fb_noff, b->d_fb_ids, (uint32_t)ids.size(), gpu_bytes, b->d_text, b->d_nbegin, b->d_nend);
  }
  if (np > 0 && !pre) {
    if (h_info[3] == 0) {
      // every document normalized on the device and no short piece inside a document: the text stays in the slabs here too (k_match_branch
      // stages it from there); otherwise it is packed - the host-normalized documents are placed behind the packed device part
      b->text_in_slabs = true;
      if (nf > 0 || h_info[6] != 0 || (tm_debug_flags(-1) & 2048)) pack_text(b, st);
    } else {
      // some piece expands beyond its slab (long runs of capitals): exact two-pass path
      TM_LAUNCH(k_norm_emit<1>, pgrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, capcode, lower_all,
                                            b->d_piece_carry, b->d_need_host, b->d_piece_len, b->d_piece_off, b->d_text, nullptr, b->d_two, nullptr);
    }
  }
  TM_LAUNCH(k_norm_ranges, (nd + 255) / 256, 256, 0, st, b->d_piece_off, b->d_doc_piece_start, b->d_need_host, nd, b->d_nbegin, b->d_nend);
  uint64_t total = gpu_bytes;
  if (nf > 0) {
    { int rc = small_sync(b, b->aux_stream); if (rc != TM_OK) return rc; }       // the placement: the text and ranges it wrote are read on `st` next
    total += noff.back();
    b->host_fallback_docs = (uint32_t)ids.size();
    f4 = now();
    if (trace) fprintf(stderr, "[fallback] flags+gather+D2H %.2f ms, host normalize (overlaps the device pass) %.2f ms, wait + H2D + place %.2f ms\n", f2 - f1, f3 - f2, f4 - f3);
  }
  const double t2 = now();
  // ---- what the tokenize pipeline needs to know on the host: #segments, and the long documents if any ------------
  if (pre) (void)hipMemsetAsync(ninfo + 1, 0, 16, st);          // (k_norm_ranges_info has counted the device documents already: count them all again)
  TM_LAUNCH(k_norm_info, (nd + 255) / 256, 256, 0, st, b->d_nbegin, b->d_nend, nd, ninfo, long_segs());
  { int rc = small_d2h(b, h_info, ninfo, 24, st); if (rc == TM_OK) rc = small_sync(b, st); if (rc != TM_OK) return rc; }
  b->nbytes = total;
  b->nseg = h_info[2];
  int rc = TM_OK;
  if (h_info[1] > 0) {
    std::vector<uint64_t> hb(nd), he(nd);
    if ((e = hipMemcpy(hb.data(), b->d_nbegin, (size_t)nd * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
        (e = hipMemcpy(he.data(), b->d_nend, (size_t)nd * 8, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H ranges");
    rc = build_groups(b, hb.data(), he.data(), nd, st);
  }
  if (trace) fprintf(stderr, "[tm_batch_normalize] summaries %.2f ms, device pass + host fallback (%u docs) %.2f ms, info %.2f ms\n", t1 - t0, nf, t2 - t1, now() - t2);
  return rc;
}

}  // extern "C"
namespace tmh {
// The host-to-host ring takes a vocabulary whose normalizer pass is the one-pass form (capcode 0 or 2 with flags the device implements)
bool ring_supported(const tm_vocab* v) {
  const uint32_t capcode = v->host.capcode, norm_flag = v->host.norm_flag;
  // (a vocabulary with byte-level flags: the filter pass in front is enqueued like everything else, and the normalizer pass behind it is launched
  // over the pieces of the RAW documents - a bound: a filtered document is no longer than the raw one, but for the space leadingspace may put in
  // front, which raw_prepare has counted - and takes the count from the device)
  return normalize_supported(capcode, norm_flag) && (capcode == 2 || capcode == 0) && !(tm_debug_flags(-1) & (256 | 2048));
}
// tm_batch_normalize's usual path - ONE pass over the raw text that raw_prepare + the upload have put into the workspace - enqueued on `st`
// and NOT waited for: what the host would read back stays in d_ninfo, k_chunk_ctl turns it into the control words the kernels behind it
// look at (the segments are launched over `seg_bound`), and a chunk the pass cannot finish by itself (documents for the host normalizer, a
// piece whose margins could not tell, a long document ...) is marked there and run through tm_batch_normalize by the caller afterwards.
int ring_enqueue_normalize(tm_batch* b, hipStream_t st, uint64_t seg_bound, const uint64_t* h_raw_off) {
  const tm_vocab* v = b->vocab;
  const uint32_t capcode = v->host.capcode, norm_flag = v->host.norm_flag;
  const uint32_t nd = b->raw_docs;
  const uint64_t np = b->raw_pieces;
  if (nd == 0 || np == 0) return set_error(TM_E_INTERNAL, "ring_enqueue_normalize: empty chunk");
  hipError_t e;
  if (!b->d_ctl_store && (e = hipMalloc((void**)&b->d_ctl_store, 64)) != hipSuccess) return hip_fail(e, "hipMalloc");
  b->d_ctl = b->d_ctl_store;
  b->host_fallback_docs = 0;
  b->d_doc_begin = b->d_nbegin;
  b->d_doc_end = b->d_nend;
  b->ndocs = nd; b->nbytes = 0; b->ngroups = 0; b->nlong = 0;
  b->nseg = std::min<uint64_t>(seg_bound, b->max_segs);
  b->text_in_slabs = true;
  b->slab_pieces = np;
  (void)hipGetLastError();
  const uint32_t lower_all = (norm_flag & 2u) ? 1u : 0u;
  unsigned long long* ninfo = (unsigned long long*)b->d_ninfo;
  const uint8_t* R = b->d_raw;
  const uint64_t* RB = b->d_raw_off;
  const uint64_t* RE = b->d_raw_off + 1;
  const uint64_t* np_dev = nullptr;             // behind a filter pass np is a bound, and this the count
  if (norm_flag & ~3u) {
    int rc = prefilter_enqueue(b, st, norm_flag, h_raw_off, &R, &RB, &RE);
    if (rc != TM_OK) return rc;
    np_dev = b->d_totals + 3;
  } else {
    TM_LAUNCH(k_norm_begin, (std::max(nd, 8u) + 255) / 256, 256, 0, st, RB, RE, nd, b->d_doc_npiece, b->d_need_host, ninfo);
    scan_u32(b->d_doc_npiece, nd, b->d_scan_tmp, b->d_totals + 3, b->d_doc_piece_start, st);
  }
  const uint32_t pgrid = (uint32_t)((np + 3) / 4);
  const uint32_t egrid = std::min(pgrid, norm_grid());
  launch_unit_owner(b->d_doc_piece_start, nd, np, b->d_piece_doc, st, np_dev);
  if (capcode == 2)
    TM_LAUNCH(k_norm_emit2<false>, egrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, lower_all, nullptr,
                                               b->d_need_host, b->d_piece_len, b->d_slab, ninfo + 3, b->d_two, np_dev);
  else
    TM_LAUNCH(k_norm_emit<3>, pgrid, 256, 0, st, R, RB, RE, b->d_piece_doc, b->d_doc_piece_start, np, capcode, lower_all,
                                          nullptr, b->d_need_host, b->d_piece_len, nullptr, b->d_slab, ninfo + 3, b->d_two, np_dev);
  TM_LAUNCH(k_norm_bad, (uint32_t)((std::max<uint64_t>(np, nd) + 255) / 256), 256, 0, st, b->d_piece_doc, b->d_need_host, b->d_doc_piece_start, np, nd, b->d_piece_len, ninfo, b->d_fb_ids, np_dev);
  scan_u32(b->d_piece_len, np, b->d_scan_tmp, reinterpret_cast<uint64_t*>(ninfo + 5), b->d_piece_off, st);
  TM_LAUNCH(k_norm_ranges_info, (nd + 255) / 256, 256, 0, st, b->d_piece_off, b->d_doc_piece_start, b->d_need_host, nd, b->d_nbegin, b->d_nend, ninfo, long_segs());
  launch_chunk_ctl(b, b->nseg, st);
  e = hipGetLastError();
  return e == hipSuccess ? TM_OK : hip_fail(e, "kernel launch");
}
}  // namespace tmh
extern "C" {

uint64_t tm_batch_normalized_bytes(const tm_batch* b) { return b->nbytes; }
uint32_t tm_batch_host_fallback_docs(const tm_batch* b) { return b->host_fallback_docs; }

// D2H of the normalized text of the current batch in DOCUMENT ORDER (for tests): text_out[nbytes], offsets_out[ndocs+1]
int tm_batch_download_text(tm_batch* b, uint8_t* text_out, uint64_t text_cap, uint64_t* offsets_out) {
  if (!b) return set_error(TM_E_INVALID, "null argument");
  hipError_t e;
  if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "sync");
  pack_text(b, nullptr);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "sync");
  if (b->nbytes > text_cap) return set_error(TM_E_NOSPACE, "text_cap too small");
  const uint32_t nd = b->ndocs;
  std::vector<uint64_t> hb(nd), he(nd);
  std::vector<uint8_t> all(b->nbytes);
  if (nd && ((e = hipMemcpy(hb.data(), b->d_doc_begin, (size_t)nd * 8, hipMemcpyDeviceToHost)) != hipSuccess ||
             (e = hipMemcpy(he.data(), b->d_doc_end, (size_t)nd * 8, hipMemcpyDeviceToHost)) != hipSuccess)) return hip_fail(e, "D2H ranges");
  if (b->nbytes && (e = hipMemcpy(all.data(), b->d_text, b->nbytes, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H text");
  uint64_t o = 0;
  for (uint32_t d = 0; d < nd; d++) {
    if (offsets_out) offsets_out[d] = o;
    if (he[d] > hb[d]) std::memcpy(text_out + o, all.data() + hb[d], he[d] - hb[d]);
    o += he[d] - hb[d];
  }
  if (offsets_out) offsets_out[nd] = o;
  return TM_OK;
}

}  // extern "C"
