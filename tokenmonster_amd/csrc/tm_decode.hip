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
// The gather works on TILES of DEC_TILE ids, one workgroup each, in two launches with a scan of the tiles' byte counts between them (the
// byte offset of every id never exists in memory: 12 bytes of traffic per id in the first form of this path, and a byte-by-byte copy by one
// work-item per id).  k_dec_tile_len: bytes of a tile.  k_dec_doc_tile: the first document whose first id lies in a tile.  k_dec_gather: the
// lengths once more (one word of the reverse table per id - place and length packed -, L2 hits), their prefix sums inside the tile in LDS, the keys copied into
// an LDS window of DEC_WIN bytes - a work-item owns eight consecutive ids, so it writes one run of the window, a dword at a time - and the
// window written out in 16-byte stores aligned in global memory (partial first / last chunk bytewise: the neighbours' bytes are theirs);
// the documents that begin in the tile get their byte offset from the same prefix sums.  A tile of more than DEC_WIN bytes (ids of up to
// 40 bytes: 80 KB at most) takes several windows.
constexpr uint32_t DEC_TILE = DEC_TILE_IDS, DEC_PER = DEC_TILE / 256, DEC_WIN = 16384, DEC_NO_DOC = 0xFFFFFFFFu;
typedef uint32_t __attribute__((aligned(1))) dec_u32u;
// ids [k0, k0 + DEC_PER) of the tile that begins at t0: id, place of the key in the reverse table, its length (0: no such id, or beyond the end)
__device__ __forceinline__ uint32_t dec_load_ids(const uint32_t* __restrict__ tokens, uint64_t t0, uint32_t k0, uint32_t ntok, const uint32_t* __restrict__ rev_off,
                                                 uint32_t n_ids, uint32_t* __restrict__ src, uint32_t* __restrict__ len) {
  uint32_t id[DEC_PER];
  if (k0 + DEC_PER <= ntok) {
    const uint4* q = reinterpret_cast<const uint4*>(tokens + t0 + k0);            // (t0 and k0 are multiples of 8 ids, the buffer 256-byte aligned)
#pragma unroll
    for (uint32_t j = 0; j < DEC_PER / 4; j++) { const uint4 v = q[j]; id[4 * j] = v.x; id[4 * j + 1] = v.y; id[4 * j + 2] = v.z; id[4 * j + 3] = v.w; }
  } else {
#pragma unroll
    for (uint32_t j = 0; j < DEC_PER; j++) id[j] = k0 + j < ntok ? tokens[t0 + k0 + j] : 0xFFFFFFFFu;
  }
  uint32_t sum = 0;
#pragma unroll
  for (uint32_t j = 0; j < DEC_PER; j++) {
    const bool ok = id[j] < n_ids;                                                // ids >= n_ids are skipped like the reference does
    const uint32_t pk = ok ? rev_off[id[j]] : 0u;                                 // place | length << 26 (tm_tables.h: kRevPlaceBits)
    src[j] = pk & ((1u << kRevPlaceBits) - 1u); len[j] = pk >> kRevPlaceBits; sum += pk >> kRevPlaceBits;
  }
  return sum;
}
__global__ __launch_bounds__(256) void k_dec_tile_len(const uint32_t* __restrict__ tokens, uint64_t n, const uint32_t* __restrict__ rev_off, uint32_t n_ids,
                                                     uint32_t* __restrict__ tile_len, uint32_t* __restrict__ tile_first) {
  __shared__ uint32_t s_w[4];
  const uint64_t t0 = (uint64_t)blockIdx.x * DEC_TILE;
  const uint32_t ntok = (uint32_t)(n - t0 < DEC_TILE ? n - t0 : DEC_TILE);
  uint32_t src[DEC_PER], len[DEC_PER];
  uint32_t sum = dec_load_ids(tokens, t0, threadIdx.x * DEC_PER, ntok, rev_off, n_ids, src, len);
  for (int m = 32; m > 0; m >>= 1) sum += __shfl_xor(sum, m);
  if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { tile_len[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3]; tile_first[blockIdx.x] = DEC_NO_DOC; }
}
// tile of the id at offset t (the one-past-the-end offset belongs to the last tile)
__device__ __forceinline__ uint64_t dec_tile_of(uint64_t t, uint64_t ntiles) { const uint64_t k = t / DEC_TILE; return k < ntiles ? k : ntiles - 1; }
__global__ void k_dec_doc_tile(const uint64_t* __restrict__ tok_offsets, uint32_t ndocs, uint64_t ntiles, uint32_t* __restrict__ tile_first) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d > ndocs) return;
  const uint64_t tl = dec_tile_of(tok_offsets[d], ntiles);
  if (d == 0 || dec_tile_of(tok_offsets[d - 1], ntiles) != tl) tile_first[tl] = d;
}
__global__ __launch_bounds__(256) void k_dec_gather(const uint32_t* __restrict__ tokens, uint64_t n, const uint32_t* __restrict__ rev_off, const uint8_t* __restrict__ rev_bytes,
                                                   uint32_t n_ids, const uint64_t* __restrict__ tile_base, const uint32_t* __restrict__ tile_first,
                                                   const uint64_t* __restrict__ tok_offsets, uint32_t ndocs, uint64_t ntiles, uint8_t* __restrict__ out,
                                                   uint64_t* __restrict__ doc_off) {
  __shared__ uint32_t s_loff[DEC_TILE + 1];
  __shared__ uint32_t s_w[4];
  alignas(16) __shared__ uint8_t s_stage[DEC_WIN + 16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint64_t tile = blockIdx.x, t0 = tile * DEC_TILE;
  const uint32_t ntok = (uint32_t)(n - t0 < DEC_TILE ? n - t0 : DEC_TILE);
  const uint32_t k0 = tid * DEC_PER;
  uint32_t src[DEC_PER], len[DEC_PER];
  const uint32_t mine = dec_load_ids(tokens, t0, k0, ntok, rev_off, n_ids, src, len);
  // where the work-item's run of keys begins in the tile: prefix sum over the wavefront, then over the four wavefronts
  uint32_t incl = mine;
  for (int dlt = 1; dlt < 64; dlt <<= 1) { const uint32_t t = __shfl(incl, (int)lane - dlt); if ((int)lane >= dlt) incl += t; }
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  uint32_t tb = incl - mine;
  for (uint32_t q = 0; q < wv; q++) tb += s_w[q];
  const uint32_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  {
    uint32_t o = tb;
#pragma unroll
    for (uint32_t j = 0; j < DEC_PER; j++) { if (k0 + j <= ntok) s_loff[k0 + j] = o; o += len[j]; }      // (index ntok: the bytes of the tile - lengths beyond the end are 0)
    if (k0 + DEC_PER == ntok) s_loff[ntok] = o;
  }
  __syncthreads();
  const uint64_t base = tile_base[tile];
  {
    const uint32_t first = tile_first[tile];
    if (first != DEC_NO_DOC) {
      const uint64_t span = tile + 1 == ntiles ? (uint64_t)ntok + 1 : (uint64_t)DEC_TILE;
      for (uint64_t d = (uint64_t)first + tid; d <= ndocs; d += 256) {
        const uint64_t t = tok_offsets[d] - t0;
        if (t >= span) break;
        doc_off[d] = base + s_loff[t];
      }
    }
  }
  for (uint32_t w0 = 0; w0 < total; w0 += DEC_WIN) {
    const uint32_t pad = (uint32_t)((base + w0) & 15u);                            // the window's first byte lies `pad` bytes into its 16-byte chunk of global memory
    const uint32_t wbytes = total - w0 < DEC_WIN ? total - w0 : DEC_WIN;
    uint32_t o = tb;
#pragma unroll
    for (uint32_t j = 0; j < DEC_PER; j++) {
      const uint32_t lo = o > w0 ? o : w0, hi = o + len[j] < w0 + DEC_WIN ? o + len[j] : w0 + DEC_WIN;
      if (lo < hi) {
        const uint8_t* sp = rev_bytes + src[j] + (lo - o);
        const uint32_t dp = pad + (lo - w0), c = hi - lo;
        uint32_t q = 0;
        for (; q + 4 <= c; q += 4) *reinterpret_cast<dec_u32u*>(&s_stage[dp + q]) = *reinterpret_cast<const dec_u32u*>(sp + q);
        if (q < c) {
          const uint32_t wd = *reinterpret_cast<const dec_u32u*>(sp + q);       // (up to three bytes beyond the key: the table's own bytes, or the padding behind it)
          s_stage[dp + q] = (uint8_t)wd;
          if (q + 1 < c) s_stage[dp + q + 1] = (uint8_t)(wd >> 8);
          if (q + 2 < c) s_stage[dp + q + 2] = (uint8_t)(wd >> 16);
        }
      }
      o += len[j];
    }
    __syncthreads();
    uint8_t* g0 = out + (base + w0 - pad);                                          // 16-byte aligned (out is)
    const uint32_t nch = (pad + wbytes + 15u) / 16u;
    for (uint32_t ch = tid; ch < nch; ch += 256) {
      const uint32_t lo = 16u * ch, hi = lo + 16u;
      if (lo >= pad && hi <= pad + wbytes) *reinterpret_cast<uint4*>(g0 + lo) = *reinterpret_cast<const uint4*>(&s_stage[lo]);
      else for (uint32_t q = lo > pad ? lo : pad; q < hi && q < pad + wbytes; q++) g0[q] = s_stage[q];
    }
    __syncthreads();
  }
}

// Capcode level 2 decoding (javascript/tokenmonster.js:1007-1065; the host form is capcode_decode_stream, tm_normalize.cpp) of documents
// made of ASCII, the two-byte characters U+0080..U+07FF (Latin, the combining marks, Greek, Cyrillic, Armenian, Hebrew, Arabic ...) and the
// three-byte characters without case (punctuation, CJK, kana, symbols ...) — there the decoder is a four-bit state machine over CHARACTERS: 'D' deletes the
// next character, 'C' capitalises the next one that is not a (kept) space, 'W' capitalises letters until the word ends, and a space
// straight after 'W' does not end it.  One wavefront per document walks it 64 bytes at a time.  Each of the four flags is a "set here / passes
// here" pair per byte position, and what a position sees is the state behind the position before it: an inclusive scan of the pairs over
// the wavefront under (s2, p2) o (s1, p1) = (s2 | p2 & s1, p2 & p1), two flags per scan (delete and ignore-space first; what they decide - which
// characters are kept - is what capitalise-next and capitalise-word pass through), six steps of DPP moves each, then one more move for the
// state of the lane below and a v_readlane for the next chunk's entry state.  (The first form of this kernel did the four fills on the ballots
// of the byte classes with the carry chain of a 64-bit scalar add: 337 scalar and 208 vector instructions per 64 bytes, and its time WAS the
// scalar unit's - 335 clocks of a CU per chunk, 9.8 ms per GiB; a CU has one scalar unit and four vector units.)  The class of an ASCII byte
// comes from a 256-entry table in LDS.  A character is decided at its first byte; its other bytes are transparent to every flag and take the
// decision (kept / capitalised) of the first, also across the end of a chunk - that part, only run for a chunk with a byte >= 0x80, stays on
// ballots.  Capitalising a two-byte letter replaces its two bytes by those of
// its upper-case form (`tab`, built by the host from the host decoder's own functions: tm_normalize.cpp build_dec_tables; р D1 80 -> Р D0 A0
// changes the lead byte too); four-byte characters whose block of 64 code points is caseless throughout (emoji, symbols, the ideographs of
// plane 2 ...) are passed on like the three-byte ones; a character whose upper-case form has another length, a three- or four-byte letter
// with case, and any byte sequence that is not well-formed UTF-8 leave the document to the host decoder (dec_len = DEC_HOST), as does a
// document of 4 GiB or more.
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
// class bits of a byte position (k_dec_capcode)
constexpr uint32_t DC_M = 1u, DC_D = 2u, DC_W = 4u, DC_C = 8u, DC_SP = 16u, DC_LET = 32u, DC_WC = 64u, DC_LOW = 128u, DC_START = 256u;
__device__ __forceinline__ uint32_t dec_ascii_class(uint32_t c) {
  if (c >= 0x80u) return 0u;
  uint32_t f = DC_START;
  if (c == 'D') f |= DC_M | DC_D;
  if (c == 'W') f |= DC_M | DC_W;
  if (c == 'C') f |= DC_M | DC_C;
  if (c == ' ') f |= DC_SP;
  if (c - 'a' < 26u) f |= DC_LET | DC_LOW;
  if (c - 'A' < 26u) f |= DC_LET;
  if (c - '0' < 10u || c == '\'') f |= DC_WC;      // what keeps a capitalised word going besides letters: digits, the apostrophe (and U+2019, marks: below)
  return f;
}
// inclusive scan of (set, stop) pairs over the wavefront under (s2, q2) o (s1, q1) = (s2 | s1 & ~q2, q2 | q1): afterwards a lane's pair summarises the
// positions 0 .. lane of the chunk (stop = the flag does not pass this position; as "stop" and not "pass" so that 0 is the identity of both
// words and a lane without a source needs no value set up)
__device__ __forceinline__ void dec_scan(uint32_t& s, uint32_t& q) {
#define TM_DEC_ROW(ctrl) { const uint32_t ts = TM_DPP0(s, ctrl), tq = TM_DPP0(q, ctrl); s |= ts & ~q; q |= tq; }
#define TM_DEC_BC(ctrl, rows) { const uint32_t ts = TM_DPP(0u, s, ctrl, rows), tq = TM_DPP(0u, q, ctrl, rows); s |= ts & ~q; q |= tq; }
  TM_DEC_ROW(0x111) TM_DEC_ROW(0x112) TM_DEC_ROW(0x114) TM_DEC_ROW(0x118)      // row_shr:1, 2, 4, 8
  TM_DEC_BC(0x142, 0xA) TM_DEC_BC(0x143, 0xC)                                  // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
#undef TM_DEC_ROW
#undef TM_DEC_BC
}
// (The text comes through LDS: a wavefront fetches the document in blocks of DEC_BLK bytes aligned in the text buffer - sixteen bytes a lane through a
// buffer-resource window, and sixteen more on either side for the characters that straddle a block -, the next block under way while the
// sixteen 64-byte chunks of this one are decoded, so that the wait for memory comes once per block and not once per chunk: inside the chunk
// loop nothing is loaded from global memory but the table entries of characters beyond ASCII, and a lane reads its byte, its neighbours and
// its class from LDS.  The chunks are aligned like the blocks: the first one of a document may begin with positions that are not the
// document's, which behave like the ones behind its end.)
constexpr uint32_t DEC_BLK = 1024, DEC_HALF = 16 + DEC_BLK + 16, DEC_ILP = 4, DEC_CLASSES = 3;
constexpr uint64_t DEC_LONG = 16384, DEC_MID = 4096;
#ifdef TM_EMU
#define TM_SETPRIO_BY_CLASS(c) ((void)(c))
#else
#define TM_SETPRIO_BY_CLASS(c) do { if ((c) == 0u) __builtin_amdgcn_s_setprio(3); else if ((c) == 1u) __builtin_amdgcn_s_setprio(1); } while (0)
#endif
__global__ __launch_bounds__(256) void k_dec_capcode(const uint8_t* __restrict__ in, const uint64_t* __restrict__ doc_off, uint32_t ndocs,
                                                      uint8_t* __restrict__ out, uint64_t* __restrict__ dec_len, const uint32_t* __restrict__ tab) {
  __shared__ uint16_t s_cls[256];
  alignas(16) __shared__ uint8_t s_ring[4][2 * DEC_HALF];
  s_cls[threadIdx.x] = (uint16_t)dec_ascii_class(threadIdx.x);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // One wavefront per document, and a document is a serial walk: the launch's time is what its longest documents take, so they go first and
  // run ahead.  The grid is the documents DEC_CLASSES times over: the first third of the workgroups (the ones dispatched first) take only the
  // documents of DEC_LONG bytes and more - the wavefronts of the others end at once -, the second third those of DEC_MID and more, the last
  // third the rest; and the wavefront of a long document asks for a higher issue priority than its neighbours on the SIMD (s_setprio), which
  // keeps it near the speed of a wavefront that is alone (0.5 us per 64-byte chunk against 2.3 when eight share a SIMD evenly).
  const uint32_t per_class = (ndocs + 3u) / 4u, cls = blockIdx.x / per_class;
  const uint32_t d = (blockIdx.x - cls * per_class) * 4u + wv;
  if (d >= ndocs) return;
  const uint64_t b = doc_off[d], e = doc_off[d + 1];
  const uint32_t mine = e - b >= DEC_LONG ? 0u : (e - b >= DEC_MID ? 1u : 2u);
  if (mine != cls) return;
  if (e - b >= 0xFFFFF000ull) { if (lane == 0) dec_len[d] = DEC_HOST; return; }
  TM_SETPRIO_BY_CLASS(mine);
  const uint64_t wb = b & ~(uint64_t)(DEC_BLK - 1);      // the blocks are aligned in the buffer; positions below are relative to wb
  const uint32_t rb = (uint32_t)(b - wb), re = (uint32_t)(e - wb);
  const TmWindow win = tm_window(in + wb, (re + 3u) & ~3u);
  uint8_t* __restrict__ dst = out + b;                   // decoded bytes of the document go to out[b ..): never more than it had
  uint8_t* ring = s_ring[wv];
  uint32_t o = 0;
  uint32_t cin1 = 0, cin2 = 0;                           // the decoder's state (tm_internal.h CapcodeState) between chunks: delete | ignore << 1, capitalise-next | capitalise-word << 1
  unsigned long long kept_in = 0, cap_in = 0;            // bytes at the start of this chunk that continue a character of the chunk before: kept / capitalised
  bool host = false;
  const uint32_t nblk = (re + DEC_BLK - 1) / DEC_BLK;
  // a block and its margins: lane l the bytes [16 l, 16 l + 16) of the block; lane 0 also the sixteen before it, lane 1 the sixteen behind it
  // (before the window / behind its end: zeros - the range check is the hardware's)
  const uint32_t xoff = lane == 0 ? 0u - 16u : DEC_BLK, xat = lane == 0 ? 0u : 16u + DEC_BLK;
  uint4 nxt = tm_window_u128(win, 16u * lane), nxt2 = make_uint4(0u, 0u, 0u, 0u);
  if (lane < 2) nxt2 = tm_window_u128(win, xoff);
  for (uint32_t k = 0; k < nblk && !host; k++) {
    const uint32_t hb = (k & 1u) * DEC_HALF;
    *reinterpret_cast<uint4*>(&ring[hb + 16u + 16u * lane]) = nxt;
    if (lane < 2) *reinterpret_cast<uint4*>(&ring[hb + xat]) = nxt2;
    if (k + 1 < nblk) {
      nxt = tm_window_u128(win, (k + 1) * DEC_BLK + 16u * lane);
      if (lane < 2) nxt2 = tm_window_u128(win, (k + 1) * DEC_BLK + xoff);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t j0 = k == 0 ? rb / 64u : 0u, left = re - k * DEC_BLK, j1 = left >= DEC_BLK ? DEC_BLK / 64u : (left + 63u) / 64u;
    for (uint32_t jg = j0 & ~(DEC_ILP - 1u); jg < j1 && !host; jg += DEC_ILP) {
    // DEC_ILP chunks at once when all of them are ASCII: the scans of a chunk do not depend on the chunk before it - only the two carried state
    // words do, and those are chained through scalar registers between the scans -, so a wavefront has DEC_ILP independent instruction streams
    // to issue from instead of one chain of ~130 dependent instructions per chunk (what bounded this kernel: profiles/r06_decode_bound.txt)
    {
      uint32_t cc[DEC_ILP], ff[DEC_ILP], any = 0;
#pragma unroll
      for (uint32_t u = 0; u < DEC_ILP; u++) {
        const uint32_t at = k * DEC_BLK + 64u * (jg + u) + lane;
        const bool valid = at >= rb && at < re;
        uint32_t c = ring[hb + 16u + 64u * (jg + u) + lane];
        c = valid ? c : 0u;
        cc[u] = c; any |= c;
      }
      if (!__any(any >= 0x80u)) {
        // the (set, stop) pairs of the DEC_ILP chunks side by side in ONE register, a byte per chunk: the scan is bitwise, so its six steps serve all
        // the chunks at once; what a chunk hands to the next - the state behind its lane 63 - is chained through scalar registers behind the scan
        static_assert(DEC_ILP <= 4, "a byte per chunk");
        auto chain = [&](uint32_t sp, uint32_t qp, uint32_t& cin) -> uint32_t {
          const uint32_t S = read_lane(sp, 63), Q = read_lane(qp, 63);
          uint32_t cinp = 0u, c = cin;
#pragma unroll
          for (uint32_t u = 0; u < DEC_ILP; u++) { cinp |= c << (8u * u); c = (((S >> (8u * u)) | (c & ~(Q >> (8u * u)))) & 3u); }
          cin = c;
          return TM_DPP(cinp, sp | (cinp & ~qp), 0x138, 0xF);        // the state behind the position before, every chunk in its byte
        };
        uint32_t sp = 0u, qp = 0u;
#pragma unroll
        for (uint32_t u = 0; u < DEC_ILP; u++) {
          const uint32_t at = k * DEC_BLK + 64u * (jg + u) + lane;
          uint32_t f = s_cls[cc[u]];
          ff[u] = (at >= rb && at < re) ? f : 0u;
          sp |= ((ff[u] >> 1) & 3u) << (8u * u); qp |= (((ff[u] & DC_M) - 1u) & 3u) << (8u * u);
        }
        dec_scan(sp, qp);
        const uint32_t pv1 = chain(sp, qp, cin1);
        uint32_t kept[DEC_ILP], Kk[DEC_ILP];
        sp = 0u; qp = 0u;
#pragma unroll
        for (uint32_t u = 0; u < DEC_ILP; u++) {
          const uint32_t f = ff[u], nm = (f >> 8) & ~f & 1u, pv = pv1 >> (8u * u);
          kept[u] = nm & ~pv;
          Kk[u] = kept[u] & ~(f >> 4);
          const uint32_t ends_word = ((f >> 3) | ((f >> 4) & kept[u] & ~(pv >> 1)) | (Kk[u] & ~(f >> 5) & ~(f >> 6))) & 1u;
          sp |= (((f >> 3) & 1u) | ((f >> 1) & 2u)) << (8u * u); qp |= ((((f >> 2) | Kk[u]) & 1u) | (ends_word << 1)) << (8u * u);
        }
        dec_scan(sp, qp);
        const uint32_t pv2 = chain(sp, qp, cin2);
#pragma unroll
        for (uint32_t u = 0; u < DEC_ILP; u++) {
          const uint32_t f = ff[u], pv = pv2 >> (8u * u);
          const uint32_t capS = ((Kk[u] & pv) | (Kk[u] & (f >> 5) & (pv >> 1))) & 1u;
          const uint32_t oc = cc[u] - ((capS & (f >> 7)) << 5);
          const unsigned long long km = __ballot(kept[u] != 0u);
          if (kept[u]) dst[o + mbcnt64(km, 0u)] = (uint8_t)oc;
          o += (uint32_t)__popcll(km);
        }
        continue;
      }
    }
    const uint32_t ja = jg > j0 ? jg : j0, jb = jg + DEC_ILP < j1 ? jg + DEC_ILP : j1;
    for (uint32_t j = ja; j < jb; j++) {
    const uint32_t pos = k * DEC_BLK + 64u * j;
    const uint32_t at = pos + lane;
    const bool valid = at >= rb && at < re;
    const uint32_t ri = hb + 16u + 64u * j + lane;      // where the lane's byte lies in the ring
    uint32_t c = ring[ri];
    c = valid ? c : 0u;
    uint32_t f = s_cls[c];
    f = valid ? f : 0u;
    const bool hi = __any(c >= 0x80u);
    // (what a chunk with bytes beyond ASCII carries from here to its output - deliberately without a value on the ASCII path: a zero written
    // here every chunk would have to wait for a table load of the chunk before that may still be under way, and with it for the next block's prefetch)
    uint32_t te;                                         // table entry of the two-byte character this lane starts or ends
    bool lead2, tail2;
    unsigned long long L2, L3, L4;
    if (hi) {
      te = 0;
      // the bytes around it (inside the document)
      const uint32_t cn = at + 1 < re ? ring[ri + 1] : 0u, cnn = at + 2 < re ? ring[ri + 2] : 0u, cn3 = at + 3 < re ? ring[ri + 3] : 0u;
      const uint32_t cp = valid && at >= rb + 1 ? ring[ri - 1] : 0u, cpp = valid && at >= rb + 2 ? ring[ri - 2] : 0u, cp3 = valid && at >= rb + 3 ? ring[ri - 3] : 0u;
      auto is_lead2 = [](uint32_t x) { return x - 0xC2u < 30u; };
      auto is_lead3 = [](uint32_t x) { return (x & 0xF0u) == 0xE0u; };
      auto is_cont = [](uint32_t x) { return (x & 0xC0u) == 0x80u; };
      auto is_lead4 = [](uint32_t x) { return x - 0xF0u < 5u; };
      const bool ascii = c < 0x80u;
      lead2 = valid && is_lead2(c);
      const bool lead3 = valid && is_lead3(c), lead4 = valid && is_lead4(c);
      uint32_t t3 = 0;                                   // code of the three- or four-byte character it starts
      if (lead2 && is_cont(cn)) te = tab[dec_tab_index(c, cn)];
      tail2 = valid && is_cont(c) && is_lead2(cp);
      if (tail2) te = tab[dec_tab_index(cp, c)];
      if (lead3 && is_cont(cn) && is_cont(cnn)) t3 = dec_three_code(tab, c, cn, cnn);
      if (lead4 && is_cont(cn) && is_cont(cnn) && is_cont(cn3)) t3 = dec_four_code(tab, c, cn, cnn, cn3);
      // (the other bytes of a three- or four-byte character: the lane of its first byte vouches for it)
      const bool tail3a = valid && is_cont(c) && is_lead3(cp), tail3b = valid && is_cont(c) && is_cont(cp) && is_lead3(cpp);
      const bool tail4 = valid && is_cont(c) && (is_lead4(cp) || (is_cont(cp) && (is_lead4(cpp) || (is_cont(cpp) && is_lead4(cp3)))));
      const bool ok = !valid || ascii || (lead2 && (te & 1u)) || ((lead3 || lead4) && t3 != 0u) || (tail2 && (te & 1u)) || tail3a || tail3b || tail4;
      if (__any(!ok)) { host = true; break; }
      if (lead2) f = DC_START | ((te & 2u) ? DC_LET : 0u) | ((te & 4u) ? DC_WC : 0u);
      if (lead3 || lead4) f = DC_START | ((t3 == 2u || (c == 0xE2u && cn == 0x80u && cnn == 0x99u)) ? DC_WC : 0u);
      L2 = __ballot(lead2); L4 = __ballot(lead4); L3 = __ballot(lead3) | L4;                            // (L3: characters of three bytes or more)
    }
    // delete and ignore-space: set by 'D' / 'W', alive over the markers behind them, seen by the first character that is not a marker
    uint32_t s1 = (f >> 1) & 3u, q1 = (f & DC_M) - 1u;
    dec_scan(s1, q1);
    const uint32_t st1 = s1 | (cin1 & ~q1);
    const uint32_t pv1 = TM_DPP(cin1, st1, 0x138, 0xF);                            // wave_shr:1 - the state behind the position before (lane 0: behind the chunk before)
    cin1 = read_lane(st1, 63);
    const uint32_t nm = (f >> 8) & ~f & 1u;                                         // a character that is not a marker
    const uint32_t kept = nm & ~pv1;                                                // ... and not deleted
    const uint32_t K = kept & ~(f >> 4);                                            // the characters that use up a pending 'C': kept, not a space
    const uint32_t letK = K & (f >> 5);
    const uint32_t ends_word = ((f >> 3) | ((f >> 4) & kept & ~(pv1 >> 1)) | (K & ~(f >> 5) & ~(f >> 6))) & 1u;
    // capitalise-next: set by 'C', passes everything but 'W' and the character it capitalises; capitalise-word: set by 'W', until the word ends
    uint32_t s2 = ((f >> 3) & 1u) | ((f >> 1) & 2u), q2 = (((f >> 2) | K) & 1u) | (ends_word << 1);
    dec_scan(s2, q2);
    const uint32_t st2 = s2 | (cin2 & ~q2);
    const uint32_t pv2 = TM_DPP(cin2, st2, 0x138, 0xF);
    cin2 = read_lane(st2, 63);
    const uint32_t capS = ((K & pv2) | (letK & (pv2 >> 1))) & 1u;                   // characters that come out in upper case
    uint32_t oc = c - ((capS & (f >> 7)) << 5);                                     // (an ASCII lower-case letter)
    unsigned long long kept_all = __ballot(kept != 0u);
    if (hi) {
      // the other bytes of a character: kept / capitalised like its first byte
      const unsigned long long cap2 = __ballot(capS != 0u) & L2;
      const unsigned long long k23 = kept_all & (L2 | L3), k3 = kept_all & L3, k4 = kept_all & L4;
      kept_all = kept_all | (k23 << 1) | (k3 << 2) | (k4 << 3) | kept_in;
      const unsigned long long cap_tail = (cap2 << 1) | cap_in;
      kept_in = (k23 >> 63) | (k3 >> 62) | (k4 >> 61);          // what the shifts above push beyond bit 63: the first bytes of the next chunk
      cap_in = cap2 >> 63;
      if (lead2 && capS) oc = (te >> 8) & 0xFFu;
      if (tail2 && ((cap_tail >> lane) & 1ull)) oc = (te >> 16) & 0xFFu;
    }
    if ((kept_all >> lane) & 1ull) dst[o + mbcnt64(kept_all, 0u)] = (uint8_t)oc;
    o += (uint32_t)__popcll(kept_all);
    }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) dec_len[d] = host ? DEC_HOST : (uint64_t)o;
}
// what the batch adds up to: sums[0] bytes of the documents the device decoded, sums[1] documents it left to the host decoder.  (One workgroup per 2 048
// documents and two atomic adds per workgroup: an atomic add per DOCUMENT from the decoder itself cost 12 ns each - they all go to one address, and the
// L2 takes them one after the other -, 3.6 of that kernel's 4.9 ms per GiB: profiles/r06_decode_bound.txt.)
__global__ __launch_bounds__(256) void k_dec_sum(const uint64_t* __restrict__ dec_len, uint32_t ndocs, unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long s_b[4], s_h[4];
  unsigned long long bytes = 0, host = 0;
  for (uint32_t k = 0; k < 8; k++) {
    const uint32_t d = blockIdx.x * 2048u + k * 256u + threadIdx.x;
    if (d < ndocs) { const uint64_t l = dec_len[d]; if (l == DEC_HOST) host++; else bytes += l; }
  }
  for (int m = 32; m > 0; m >>= 1) { bytes += __shfl_xor(bytes, m); host += __shfl_xor(host, m); }
  if ((threadIdx.x & 63u) == 0) { s_b[threadIdx.x >> 6] = bytes; s_h[threadIdx.x >> 6] = host; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long bsum = s_b[0] + s_b[1] + s_b[2] + s_b[3], hsum = s_h[0] + s_h[1] + s_h[2] + s_h[3];
    if (bsum) atomicAdd(&sums[0], bsum);
    if (hsum) atomicAdd(&sums[1], hsum);
  }
}
}  // namespace tmh

namespace tmh {
// the three stages of a decode on a stream, in buffers of the caller (tm_host.hip: a lane's grow-only arenas)
void launch_decode_lengths(const tm_vocab* v, const uint32_t* d_tok, uint64_t n, const uint64_t* d_toff, uint32_t ndocs, const DecArena& a, uint8_t* A, hipStream_t st) {
  uint32_t* tile_len = (uint32_t*)(A + a.o_len); uint32_t* tile_first = (uint32_t*)(A + a.o_first);
  uint64_t* tile_base = (uint64_t*)(A + a.o_off); uint64_t* d_total = (uint64_t*)(A + a.o_total);
  if (a.ntiles == 0) {                                                             // no ids: every document is empty
    (void)hipMemsetAsync(A + a.o_doff, 0, ((uint64_t)ndocs + 1) * 8, st);
    (void)hipMemsetAsync(d_total, 0, 8, st);
    return;
  }
  TM_LAUNCH(k_dec_tile_len, (uint32_t)a.ntiles, 256, 0, st, d_tok, n, v->d_rev_off, v->host.n_ids, tile_len, tile_first);
  note_table_use(v, st);
  TM_LAUNCH(k_dec_doc_tile, (ndocs + 256) / 256, 256, 0, st, d_toff, ndocs, a.ntiles, tile_first);
  scan_u32(tile_len, a.ntiles, (uint64_t*)(A + a.o_sums), d_total, tile_base, st);
}
void launch_decode_copy(const tm_vocab* v, const uint32_t* d_tok, uint64_t n, const uint64_t* d_toff, uint32_t ndocs, const DecArena& a, uint8_t* A, uint8_t* d_out, hipStream_t st) {
  if (a.ntiles == 0) return;
  TM_LAUNCH(k_dec_gather, (uint32_t)a.ntiles, 256, 0, st, d_tok, n, v->d_rev_off, v->d_rev_bytes, v->host.n_ids, (const uint64_t*)(A + a.o_off), (const uint32_t*)(A + a.o_first),
            d_toff, ndocs, a.ntiles, d_out, (uint64_t*)(A + a.o_doff));
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
int launch_decode_capcode(const tm_vocab* v, const uint8_t* d_out, const uint64_t* d_doff, uint32_t ndocs, uint8_t* d_dec, uint64_t* d_declen, uint64_t* d_sums, hipStream_t st) {
  const uint32_t* tab = dec_table(v->device);
  if (!tab) return set_error(TM_E_HIP, "the decoder's character table could not be placed on device %d", v->device);
  const hipError_t e = hipMemsetAsync(d_sums, 0, 16, st);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
  if (ndocs) {
    if ((uint64_t)DEC_CLASSES * ((ndocs + 3) / 4) > 0x7FFFFFFFull) return set_error(TM_E_LIMIT, "%u documents in one decode: more than a launch takes (split the batch)", ndocs);
    TM_LAUNCH(k_dec_capcode, DEC_CLASSES * ((ndocs + 3) / 4), 256, 0, st, d_out, d_doff, ndocs, d_dec, d_declen, tab);
    TM_LAUNCH(k_dec_sum, (ndocs + 2047) / 2048, 256, 0, st, d_declen, ndocs, (unsigned long long*)d_sums);
  }
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
  const DecArena a = dec_arena(0, n, nd);
  if ((e = grow(&b->d_dec_a, &b->dec_a_cap, a.bytes)) != hipSuccess) { b->dec_a_cap = 0; return hip_fail(e, "hipMalloc (decode)"); }
  uint8_t* A = b->d_dec_a;
  uint64_t* d_total = (uint64_t*)(A + a.o_total); uint64_t* d_doff = (uint64_t*)(A + a.o_doff); uint64_t* d_declen = (uint64_t*)(A + a.o_declen);
  mark(0);
  launch_decode_lengths(v, b->d_out, n, b->d_tok_offsets, nd, a, A, st);
  mark(1);
  uint64_t total = 0;
  { int rc = small_d2h(b, &total, d_total, 8, st); if (rc == TM_OK) rc = small_sync(b, st); if (rc != TM_OK) return rc; }
  const uint64_t o_dec = up(total + 16);
  if ((e = grow(&b->d_dec_b, &b->dec_b_cap, o_dec + up(total + 16))) != hipSuccess) { b->dec_b_cap = 0; return hip_fail(e, "hipMalloc (decode output)"); }
  mark(2);
  launch_decode_copy(v, b->d_out, n, b->d_tok_offsets, nd, a, A, b->d_dec_b, st);
  mark(3);
  const bool dev_capcode = !raw && v->host.capcode == 2 && v->host.charset == 1;
  uint64_t sum[2] = {total, 0};
  if (dev_capcode) {
    int rc = launch_decode_capcode(v, b->d_dec_b, d_doff, nd, b->d_dec_b + o_dec, d_declen, d_total + 8, st);      // (+ 8 words: bytes the device decoded, documents it left to the host)
    if (rc != TM_OK) return rc;
    mark(4);
    if ((rc = small_d2h(b, sum, d_total + 8, 16, st)) != TM_OK) return rc;
  }
  { int rc = small_sync(b, st); if (rc != TM_OK) return rc; }
  if (ms) {
    (void)hipEventElapsedTime(&ms[0], b->ev[0], b->ev[1]);
    (void)hipEventElapsedTime(&ms[1], b->ev[2], b->ev[3]);
    if (dev_capcode) (void)hipEventElapsedTime(&ms[2], b->ev[3], b->ev[4]);
  }
  b->dec_total = total; b->dec_o_doff = a.o_doff; b->dec_o_declen = a.o_declen; b->dec_o_dec = o_dec; b->dec_capcode = dev_capcode;
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
