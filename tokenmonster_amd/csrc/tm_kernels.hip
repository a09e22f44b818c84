// tm_kernels.hip — gfx950 kernels of the ungreedy tokenization path and the tm_batch pipeline.
//
// Reference path: go/tokenmonster.go:1017-1279 (Vocab.tokenize).  Restated as a specification in
// SURVEY.md Appendix B; the property everything here rests on (derived from go :1051-1267) is that
// the walk's whole carried state at a token boundary is (i, forwardDelete): `index/length` are always
// the longest match at i (or of ' '+data[i:] when forwardDelete == 1).  Hence
//     T(i, fd) -> (emitted id, advance, fd')
// is a pure function of the text around i and can be evaluated for EVERY byte position independently.
//
// Pipeline (one wavefront owns one <=512-byte document segment; LDS holds its text and the per-position
// second-token descriptors, i.e. the live cursors of all six branches of every position at once):
//   K0 segments      doc -> segment table (binary search per segment)
//   K1 match_branch  per position: longest match (trie walk) and forward-delete match -> descriptors in
//                    LDS; then the 6-branch score/select of go :1068-1262 per (position, fd) -> R[p] = {T(p,0), T(p,1)}
//   K2 link          per segment: follow T from each of the 80 possible entry states to the segment exit
//                    -> exit map (next entry state, #tokens, #forward-deletes, #missing)
//   K3 resolve       per document: chain the exit maps -> entry state + token base of every segment
//   K4 emit          per segment: follow T from the true entry state, write token ids densely
// All integer/byte work: no MFMA.  HBM-side traffic is streaming (text in, R out/in, ids out); the
// vocabulary tables (<= a few MB) live in L2 / Infinity Cache and are the gather-bound part.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "tm_device.h"

namespace tmh {

constexpr int SEG = 320;                 // bytes of one document segment (one wavefront); 320 -> 24 wavefronts per CU
constexpr int NPOS = SEG + 40;           // positions whose descriptors a segment needs (look-ahead <= 40)
constexpr int NPOS_PAD = (NPOS + 63) / 64 * 64;
constexpr int TEXT_LEN = SEG + 96;       // staged text: position i may read up to i + 40
constexpr int ENT = 80;                  // entry states of a segment: 40 offsets x fd{0,1}
constexpr int WAVES = 4;                 // wavefronts per workgroup in K1 (plain variant)
constexpr uint32_t R_INVALID = 0xFFFFFFFFu;
constexpr uint32_t J_EXIT = 1024;           // jump targets >= J_EXIT: left the segment; J_EXIT + next entry state
constexpr uint32_t J_INVALID = 4095;        // state is not reachable (no forward-delete match there)
constexpr uint32_t ID_NONE = 0xFFFFFFu;
constexpr int NOSCORE = -1000000;
constexpr uint32_t LONG_SEGS = 512;        // documents with more segments than this are resolved hierarchically

// R word: id[0..23] | advance[24..29] | fd'[30] | missing[31]

// ------------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------------
// Documents are given as doc_begin[d] .. doc_end[d] (for packed batches doc_end == doc_begin + 1 of the same
// offsets array; for the scoring pass they are arbitrary disjoint strips of the dataset).
__global__ void k_doc_nseg(const uint64_t* __restrict__ doc_begin, const uint64_t* __restrict__ doc_end, uint32_t ndocs,
                           uint32_t* __restrict__ doc_nseg) {
  uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < ndocs) {
    uint64_t len = doc_end[d] - doc_begin[d];
    doc_nseg[d] = (uint32_t)((len + SEG - 1) / SEG);
  }
}

// exclusive scan u32 -> u64, three phases, CH elements per block
constexpr int SCAN_T = 256, SCAN_PER = 16, SCAN_CH = SCAN_T * SCAN_PER;

__global__ void k_scan_partial(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ block_sums) {
  __shared__ uint64_t s[SCAN_T];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_CH + (uint64_t)threadIdx.x * SCAN_PER;
  uint64_t acc = 0;
  for (int k = 0; k < SCAN_PER; k++) if (base + k < n) acc += in[base + k];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = SCAN_T / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) block_sums[blockIdx.x] = s[0];
}

__global__ void k_scan_sums(uint64_t* __restrict__ block_sums, uint32_t nblocks, uint64_t* __restrict__ total) {
  // single workgroup; serial over chunks of SCAN_T
  __shared__ uint64_t s[SCAN_T];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblocks; base += SCAN_T) {
    uint32_t i = base + threadIdx.x;
    uint64_t v = i < nblocks ? block_sums[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int st = 1; st < SCAN_T; st <<= 1) {
      uint64_t t = (int)threadIdx.x >= st ? s[threadIdx.x - st] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_sums[i] = carry + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += s[SCAN_T - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void k_scan_final(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ block_sums,
                             uint64_t* __restrict__ out) {
  __shared__ uint64_t s[SCAN_T];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_CH + (uint64_t)threadIdx.x * SCAN_PER;
  uint32_t v[SCAN_PER];
  uint64_t acc = 0;
  for (int k = 0; k < SCAN_PER; k++) { v[k] = base + k < n ? in[base + k] : 0; acc += v[k]; }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 1; st < SCAN_T; st <<= 1) {
    uint64_t t = (int)threadIdx.x >= st ? s[threadIdx.x - st] : 0;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  uint64_t run = block_sums[blockIdx.x] + s[threadIdx.x] - acc;
  for (int k = 0; k < SCAN_PER; k++) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (base <= n && n < base + SCAN_PER) out[n] = run - 0;  // one-past-the-end = grand total (run == prefix up to n)
}

// K0: segment g belongs to the document d with doc_seg_start[d] <= g < doc_seg_start[d+1]
__global__ void k_segments(const uint64_t* __restrict__ doc_seg_start, uint32_t ndocs, uint64_t nseg,
                           uint32_t* __restrict__ seg_doc) {
  uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nseg) return;
  uint32_t lo = 0, hi = ndocs;   // invariant: start[lo] <= g < start[hi]
  while (hi - lo > 1) {
    uint32_t mid = lo + (hi - lo) / 2;
    if (doc_seg_start[mid] <= g) lo = mid; else hi = mid;
  }
  seg_doc[g] = lo;
}

// ------------------------------------------------------------------------------------------------
// K1: match + branch
// ------------------------------------------------------------------------------------------------
// descriptor word kept in LDS per position: len[0..5] | nWords[6..10] | flag5[11..15] | nextByteClass[16..19]
// (nWords|flag5 are bits 22..31 of the node value, so a descriptor is (v >> 22) << 6 | len)
__device__ __forceinline__ uint32_t make_desc(uint32_t len, uint32_t v, uint32_t nb) {
  return len | ((v >> 22) << 6) | (nb << 16);
}

// one in-flight trie walk of a lane: text byte number d of the string being matched is text[tbase + d]
struct Walk { int pos, tbase, depth, limit, bestlen; uint32_t haddr, key, bestv, h32; bool active; };

// consume one hash probe: follow the edge, remember the deepest accepting node, arm the next probe or stop
// (pansearch LongestSubstring semantics, tokenmonster.cpp:786-877: longest prefix that is a key).
// Written without branches: every lane executes the same ~25 instructions, idle lanes probe slot 0 and ignore it,
// so the NWALK loads of a round are issued back to back and the round has a single wait.
// Returns true when the walk finished in this round.
__device__ __forceinline__ bool walk_consume(const Tables& T, const uint8_t* text, Walk& k, const uint2 e) {
  const bool was = k.active;
  const bool hit = was && e.x == k.key;
  const bool again = was && !hit && e.x != kNone;                 // occupied by another key: linear probing
  const uint32_t cur = e.y;
  k.depth += hit ? 1 : 0;
  const bool acc = hit && node_id(cur) < T.n_info;
  k.bestv = acc ? cur : k.bestv;
  k.bestlen = acc ? k.depth : k.bestlen;
  const bool cont = hit && k.depth < k.limit && (cur & kHasChildren) != 0;
  const uint32_t c = text[k.tbase + k.depth];                     // always inside the staged text
  const uint32_t nkey = (node_id(cur) << 8) | c;
  k.key = cont ? nkey : k.key;
  const uint32_t nh32 = nkey * 0x9E3779B1u;
  const uint32_t lin = (k.haddr + 1) & T.edge_mask;
  k.h32 = cont ? nh32 : k.h32;
  k.haddr = cont ? (nh32 >> T.edge_shift) : (again ? lin : 0u);
  k.active = cont || again;
  return was && !k.active;
}

struct First { int flen; int nw; uint32_t f3; };   // candidate first token: length consumed, nWords - fd, flag bits {1, 8>>3, 128>>7}

// score of one branch, go/tokenmonster.go:1075-1084 (a), :1096-1105 (b); k > 0 adds :1132-1133
__device__ __forceinline__ int branch_score(const First& F, uint32_t dS, bool bvariant, bool alt, int len) {
  int l = (int)(dS & 63u);
  uint32_t f5 = (dS >> 11) & 31u;
  int snw = (int)((dS >> 6) & 31u);
  int nb = (int)((dS >> 16) & 15u);
  int BL = F.flen + l;
  int fend = (int)(F.f3 & 1u), fcap = (int)((F.f3 >> 1) & 1u), fall = (int)((F.f3 >> 2) & 1u);
  int send = (int)(f5 & 1u), sbegl = (int)((f5 >> 1) & 1u), sbegs = (int)((f5 >> 2) & 1u), sbegc = (int)((f5 >> 3) & 1u),
      sall = (int)((f5 >> 4) & 1u);
  int sc = BL + fall + sall + max(F.nw - 1, 0) + max(snw - 1, 0) + ((nb >> 2) & 1) + (F.nw + snw + (nb >> 3)) * 100;
  if (!bvariant) sc += sbegs;
  int pen = (fcap & sbegc) * 100 + (send & nb & 1) * 3;
  pen += bvariant ? fend * 103 + 1 : (fend & sbegl) * 103;
  if (alt) pen += (BL < len ? 100 : 0) + (BL == len ? 10000 : 0);
  return sc - pen;
}

struct WaveLds {
  alignas(16) uint8_t text[TEXT_LEN];
  uint32_t D[NPOS_PAD];    // longest match at p                      (second-token descriptor)
  uint32_t Db[NPOS_PAD];   // longest match of ' '+text[p:], if usable (forward-delete descriptor), 0 = none
  uint32_t X[SEG];         // node value of D's token (node id = record ordinal)
  uint32_t Xb[SEG];        // node value of Db's token
  uint8_t xchg[64];        // lane exchange scratch for handing out forward-delete tasks
};

// T(p, fd): go/tokenmonster.go:1051-1276
__device__ __forceinline__ uint32_t transition(const Tables& T, const WaveLds& w, const uint8_t* s_bb, int p, int dl,
                                               uint32_t d, const Row& O, int fd) {
  if (d == 0) return (T.unk_id != TM_NONE ? T.unk_id : ID_NONE) | (1u << 24) | (1u << 31);   // go :1269-1276
  const int len = (int)(d & 63u);
  const uint32_t id = O.x & ID_NONE, oflag = O.x >> 24;
  const int i1 = p + len;
  if (i1 < dl && ((oflag & 32u) == 0 || s_bb[w.text[i1]] != 12)) {                           // go :1057
    const int len1 = (int)(O.z >> 24), len2 = (int)(O.w & 63u);
    int s[6] = {NOSCORE, NOSCORE, NOSCORE, NOSCORE, NOSCORE, NOSCORE};   // s1 s2 s3 s1b s2b s3b
    int best = NOSCORE;
    First F[3];
    F[0] = {len, (int)(O.y >> 24) - fd, (oflag & 1u) | (((oflag >> 3) & 1u) << 1) | (((oflag >> 7) & 1u) << 2)};
    F[1] = {len1 - fd, (int)((O.w >> 6) & 31u) - fd, (O.w >> 16) & 7u};
    F[2] = {len2 - fd, (int)((O.w >> 11) & 31u) - fd, (O.w >> 19) & 7u};
    const int nk = len1 == 0 ? 1 : (len2 == 0 ? 2 : 3);                                      // go :1111, :1163
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k < nk) {
        const int ik = p + F[k].flen;
        const uint32_t dS = w.D[ik];
        if (dS != 0) {
          s[k] = branch_score(F[k], dS, false, k > 0, len);
          best = max(best, s[k]);
          const uint32_t dB = w.Db[ik];
          if (dB != 0) {
            s[3 + k] = branch_score(F[k], dB, true, k > 0, len);
            best = max(best, s[3 + k]);
          }
        }
      }
    }
    if (best != NOSCORE) {                                                                     // go :1217-1262
      if (best == s[0]) return id | ((uint32_t)len << 24);
      if (best == s[1]) return (O.y & ID_NONE) | ((uint32_t)F[1].flen << 24);
      if (best == s[2]) return (O.z & ID_NONE) | ((uint32_t)F[2].flen << 24);
      if (best == s[3]) return id | ((uint32_t)len << 24) | (1u << 30);
      if (best == s[4]) return (O.y & ID_NONE) | ((uint32_t)F[1].flen << 24) | (1u << 30);
      return (O.z & ID_NONE) | ((uint32_t)F[2].flen << 24) | (1u << 30);
    }
  }
  return id | ((uint32_t)len << 24);                                                           // go :1265-1267
}

constexpr int NWALK = 2;                 // independent trie walks in flight per lane
constexpr int REFILL_THR = 32;           // K1 refills idle lanes when fewer than this many (per walk slot) are still walking

__global__ __launch_bounds__(WAVES * 64, 6) void k_match_branch(Tables T, const uint8_t* __restrict__ text,
                                                                const uint64_t* __restrict__ doc_begin,
                                                                const uint64_t* __restrict__ doc_end,
                                                                const uint32_t* __restrict__ seg_doc,
                                                                const uint64_t* __restrict__ doc_seg_start, uint64_t nseg,
                                                                uint2* __restrict__ R, uint2* __restrict__ exitmap, int dbg) {
  __shared__ uint32_t s_root[256];
  __shared__ uint8_t s_bb[256];
  __shared__ WaveLds s_wave[WAVES];
  const int lane = threadIdx.x & 63, wvi = threadIdx.x >> 6;
  s_root[threadIdx.x] = T.root[threadIdx.x];
  s_bb[threadIdx.x] = T.begin_byte[threadIdx.x];
  __syncthreads();
  const uint64_t g = (uint64_t)blockIdx.x * WAVES + wvi;
  if (g >= nseg) return;
  WaveLds& w = s_wave[wvi];
  const int Lmax = (int)T.max_len;
  const unsigned long long lane_below = (1ull << lane) - 1ull;
  const uint2* __restrict__ hash_tab = T.tab + kL2Size;
  const uint32_t doc = seg_doc[g];
  const uint64_t begin = doc_begin[doc] + (g - doc_seg_start[doc]) * SEG;
  const uint64_t rem = doc_end[doc] - begin;
  const int dl = rem > (uint64_t)(1 << 20) ? (1 << 20) : (int)rem;   // bytes of the document from `begin` on (clamped)
  const int seglen = min(dl, SEG);

  // stage the text with (unaligned) dword loads; bytes at and after the end of the document read as 0: the pad
  // byte of go/tokenmonster.go:1038-1046 (quirk Q1: we define it as 0 like tokenmonster.cpp:1724-1726)
  for (int j = lane; j < TEXT_LEN / 4; j += 64) {
    uint32_t tw = 0;
    if (4 * j < dl) {
      __builtin_memcpy(&tw, text + begin + 4 * j, 4);       // the text buffer has >= 256 bytes of slack
      if (4 * j + 4 > dl) tw &= (1u << (8 * (dl - 4 * j))) - 1u;
    }
    reinterpret_cast<uint32_t*>(w.text)[j] = tw;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);

  // ---- step A: descriptors for every position the segment can look at -----------------------------
  // The pass is VALU-issue bound (profiles/r01_v1_pmc.txt), so the trie walks run as a TIGHT probe loop — one
  // 8-byte hash probe per lane per round, ~20 instructions — and everything rare (taking the next position, the
  // first two bytes through the direct map, storing a result) happens in a separate refill phase that runs only
  // when fewer than REFILL_THR lanes are still walking.  Lanes take positions from a wave-wide counter, so no lane
  // idles behind the longest walk of a block.  (pansearch LongestSubstring, call sites go/tokenmonster.go:1049..)
  for (int j = lane; j < NPOS_PAD; j += 64) { w.D[j] = 0; w.Db[j] = 0; }
  __builtin_amdgcn_wave_barrier();
  const int ntask = (dbg & 1) ? 0 : min(NPOS, dl);     // positions >= dl keep descriptor 0 (nothing there)
  {
    // ---- A1: longest match at every position -> D[p] = len | nWords | flag5 (next-byte class added in A2), X[p] = node value
    // The probe loop is latency bound (K1 time scales ~1/occupancy), so every lane keeps NWALK independent walks in
    // flight: NWALK probes are issued back to back before any result is consumed.
    int next_task = 0;                        // wave-uniform
    Walk k[NWALK];
#pragma unroll
    for (int s = 0; s < NWALK; s++) k[s] = Walk{0, 0, 0, 0, 0, 0u, 0u, 0u, 0u, false};
    for (;;) {
      // refill: idle slots take the next positions; the direct map answers the first two bytes
      for (int rep = 0; rep < 2 && next_task < ntask; rep++) {
#pragma unroll
        for (int s = 0; s < NWALK; s++) {
          if (next_task >= ntask) break;
          const unsigned long long wmask = __ballot(!k[s].active);
          if (wmask == 0) continue;
          const int p = next_task + __popcll(wmask & lane_below);
          next_task += __popcll(wmask);
          if (!k[s].active && p < ntask) {
            const int limit = min(dl - p, Lmax);
            uint2 e;
            if (limit >= 2) e = T.tab[((uint32_t)w.text[p] << 8) | w.text[p + 1]];
            else { const uint32_t r = s_root[w.text[p]]; e = make_uint2((r != kNone && node_id(r) < T.n_info) ? 1u : 0u, r); }
            const int bestlen = (int)(e.x & 3u);
            if ((e.x & 4u) && limit > 2 && !(dbg & 4)) {
              k[s].pos = p; k[s].tbase = p; k[s].depth = 2; k[s].limit = limit; k[s].active = true;
              k[s].bestlen = bestlen; k[s].bestv = e.y;
              k[s].key = ((e.x >> 3) << 8) | w.text[p + 2];
              k[s].h32 = k[s].key * 0x9E3779B1u;
              k[s].haddr = k[s].h32 >> T.edge_shift;
            } else if (bestlen != 0) {
              w.D[p] = (uint32_t)bestlen | ((e.y >> 22) << 6);
              if (p < SEG) w.X[p] = e.y;
            }
          }
        }
      }
      int nactive = 0;
#pragma unroll
      for (int s = 0; s < NWALK; s++) nactive += __popcll(__ballot(k[s].active));
      if (nactive == 0) { if (next_task >= ntask) break; continue; }
      // tight probe loop
      const int thr = next_task < ntask ? NWALK * REFILL_THR : 1;
      do {
        uint2 e[NWALK];
#pragma unroll
        for (int s = 0; s < NWALK; s++) e[s] = hash_tab[k[s].haddr];
        nactive = 0;
#pragma unroll
        for (int s = 0; s < NWALK; s++) {
          if (walk_consume(T, w.text, k[s], e[s]) && k[s].bestlen != 0) {
            w.D[k[s].pos] = (uint32_t)k[s].bestlen | ((k[s].bestv >> 22) << 6);
            if (k[s].pos < SEG) w.X[k[s].pos] = k[s].bestv;
          }
          nactive += __popcll(__ballot(k[s].active));
        }
      } while (nactive >= thr);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  // ---- A2: class of the byte after each match; which positions need the forward-delete probe (go :1088) ----------
  // second token begins with a letter, has no word boundary and the byte after it is letter-class
  unsigned long long elig[NPOS_PAD / 64];
  {
    const int off = (int)T.off;
    const bool can_b = T.has_delete && T.bstart != kNone && (T.bstart & kHasChildren) && !(dbg & 8);
#pragma unroll
    for (int it = 0; it < NPOS_PAD / 64; it++) {
      const int p = it * 64 + lane;
      uint32_t d = w.D[p];
      bool el = false;
      if (d != 0) {
        const uint32_t nb = s_bb[w.text[p + (d & 63u)]];
        d |= nb << 16;
        w.D[p] = d;
        el = can_b && ((d >> 12) & 1u) && nb == 1 && ((d >> 6) & 31u) == 0 && min(dl - p, Lmax - off) > 0;
      }
      elig[it] = __ballot(el);
    }
  }
  {
    // ---- A3: longest match of ' '+text[p:] at the eligible positions -> Db[p], Xb[p] (accepted only if longer, go :1092)
    const int off = (int)T.off;
    int blk = 0;                              // wave-uniform: block of 64 positions tasks are taken from
    unsigned long long avail = elig[0];
    Walk k[NWALK];
    int mainlen[NWALK];
#pragma unroll
    for (int s = 0; s < NWALK; s++) { k[s] = Walk{0, 0, 0, 0, 0, 0u, 0u, 0u, 0u, false}; mainlen[s] = 0; }
    for (;;) {
      // refill from the eligibility masks
      for (int rep = 0; rep < 2; rep++) {
#pragma unroll
        for (int s = 0; s < NWALK; s++) {
          while (avail == 0 && blk + 1 < NPOS_PAD / 64) {
            blk++;
#pragma unroll
            for (int q = 0; q < NPOS_PAD / 64; q++) if (q == blk) avail = elig[q];
          }
          if (avail == 0) break;
          const unsigned long long wmask = __ballot(!k[s].active);
          if (wmask == 0) continue;
          // the r-th idle lane takes the r-th available position of the block: exchanged through LDS
          const int n = min(__popcll(avail), __popcll(wmask));
          const int prank = __popcll(avail & lane_below);                  // rank of MY position bit, if set
          if (((avail >> lane) & 1ull) && prank < n) w.xchg[prank] = (uint8_t)lane;
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_s_waitcnt(0);
          const int wrank = __popcll(wmask & lane_below);
          const bool take = !k[s].active && wrank < n;
          int p = 0;
          if (take) p = blk * 64 + w.xchg[wrank];
          avail = __ballot(((avail >> lane) & 1ull) && prank >= n);       // the n lowest positions are handed out
          __builtin_amdgcn_wave_barrier();
          if (take) {
            k[s].pos = p; k[s].tbase = p - off; k[s].bestlen = 0; k[s].bestv = 0; k[s].depth = 2;
            mainlen[s] = (int)(w.D[p] & 63u);
            k[s].limit = min(dl - p, Lmax - off) + off;
            if (off == 1) {
              const uint2 e = T.tab[((uint32_t)' ' << 8) | w.text[p]];
              if ((e.x & 4u) && k[s].limit > 2) {
                k[s].active = true;
                k[s].key = ((e.x >> 3) << 8) | w.text[p + 1];
                k[s].h32 = k[s].key * 0x9E3779B1u;
                k[s].haddr = k[s].h32 >> T.edge_shift;
              }
            } else {
              k[s].active = true;
              k[s].key = (node_id(T.bstart) << 8) | w.text[p];
              k[s].h32 = k[s].key * 0x9E3779B1u;
              k[s].haddr = k[s].h32 >> T.edge_shift;
            }
          }
        }
      }
      int nactive = 0;
#pragma unroll
      for (int s = 0; s < NWALK; s++) nactive += __popcll(__ballot(k[s].active));
      bool more = avail != 0;
#pragma unroll
      for (int q = 0; q < NPOS_PAD / 64; q++) more |= (q > blk && elig[q] != 0);
      if (nactive == 0) { if (!more) break; continue; }
      const int thr = more ? NWALK * REFILL_THR : 1;
      do {
        uint2 e[NWALK];
#pragma unroll
        for (int s = 0; s < NWALK; s++) e[s] = hash_tab[k[s].haddr];
        nactive = 0;
#pragma unroll
        for (int s = 0; s < NWALK; s++) {
          if (walk_consume(T, w.text, k[s], e[s]) && k[s].bestlen > mainlen[s] + 1) {
            const int lb = k[s].bestlen - off;                              // go :1093
            w.Db[k[s].pos] = make_desc((uint32_t)lb, k[s].bestv, s_bb[w.text[k[s].pos + lb]]);
            if (k[s].pos < SEG) w.Xb[k[s].pos] = k[s].bestv;
          }
          nactive += __popcll(__ballot(k[s].active));
        }
      } while (nactive >= thr);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);

  // ---- step B: T(p,0) and T(p,1) for every position of the segment --------------------------------
  // all row gathers of the segment are issued first (independent), then the scoring runs out of registers + LDS
  uint32_t r0[SEG / 64], r1[SEG / 64];
  {
    Row row0[SEG / 64], row1[SEG / 64];
    uint32_t d0[SEG / 64], d1[SEG / 64];
#pragma unroll
    for (int it = 0; it < SEG / 64; it++) {
      const int p = it * 64 + lane;
      d0[it] = w.D[p];
      d1[it] = w.Db[p];
      if (p < seglen && d0[it] != 0) row0[it] = T.rows[node_id(w.X[p])];
      if (p < seglen && d1[it] != 0) row1[it] = T.rows[node_id(w.Xb[p])];
    }
#pragma unroll
    for (int it = 0; it < SEG / 64; it++) {
      const int p = it * 64 + lane;
      r0[it] = R_INVALID; r1[it] = R_INVALID;
      if (p < seglen) {
        r0[it] = transition(T, w, s_bb, p, dl, d0[it], row0[it], 0);
        if (d1[it] != 0) r1[it] = transition(T, w, s_bb, p, dl, d1[it], row1[it], 1);
        R[begin + p] = make_uint2(r0[it], r1[it]);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);

  // ---- step C: exit map of the segment by pointer doubling over all (p, fd) states -------------------
  // J[s] summarises the path from state s to where it currently points: {target, #id events, #forward deletes,
  // #missing}.  Composing J[s] with J[target] doubles the path; after <= log2(chain length) rounds every state
  // points at a segment exit.  Only the 80 possible entry states (offset < 40, fd) are written out, but their
  // chains run through arbitrary states, so all 2 x 512 states take part.  In-place updates are safe: an entry is
  // read and written as one 8-byte LDS access and always describes a valid prefix of its state's chain.
  {
    uint2* J = reinterpret_cast<uint2*>(w.D);            // overlays D, Db, X, Xb (dead after step B): 1024 x 8 B
    static_assert(sizeof(uint32_t) * (2 * NPOS_PAD + 2 * SEG) >= 2 * SEG * sizeof(uint2), "J overlay does not fit");
    const bool more_text = rem > (uint64_t)SEG;           // not the last segment of the document
    auto first_hop = [&](uint32_t r, int p) -> uint2 {
      if (p >= seglen) return make_uint2(J_EXIT, 0u);     // at/after the end of text: terminal, nothing emitted
      if (r == R_INVALID) return make_uint2(J_INVALID, 0u);
      const int pn = p + (int)((r >> 24) & 63u);
      const uint32_t fdn = (r >> 30) & 1u;
      const uint32_t ev = (r & ID_NONE) != ID_NONE ? 1u : 0u;
      const uint32_t tgt = pn >= seglen ? J_EXIT + (more_text ? (uint32_t)((pn - SEG) * 2) + fdn : 0u) : fdn * SEG + (uint32_t)pn;
      return make_uint2(tgt | (ev << 12), fdn | ((r >> 31) << 16));
    };
#pragma unroll
    for (int it = 0; it < SEG / 64; it++) {
      const int p = it * 64 + lane;
      J[p] = first_hop(r0[it], p);
      J[SEG + p] = first_hop(r1[it], p);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    // entry state e = offset*2 + fd  <->  state index fd*SEG + offset
    const int e0 = lane, e1 = 64 + lane;
    const int se0 = (e0 & 1) * SEG + (e0 >> 1), se1 = (e1 & 1) * SEG + (e1 >> 1);
    for (int round = 0; round < 12; round++) {
#pragma unroll
      for (int k = 0; k < 2 * SEG / 64; k++) {
        const int sidx = k * 64 + lane;
        uint2 a = J[sidx];
        const uint32_t t = a.x & 0xFFFu;
        if (t < J_EXIT) {
          const uint2 bnext = J[t];
          a.x = (bnext.x & 0xFFFu) | (((a.x >> 12) + (bnext.x >> 12)) << 12);
          a.y = a.y + bnext.y;                            // two 16-bit counters, neither can overflow (<= 512 each)
          J[sidx] = a;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0);
      bool pending = (J[se0].x & 0xFFFu) < J_EXIT;
      if (lane < ENT - 64) pending |= (J[se1].x & 0xFFFu) < J_EXIT;
      if (!__any(pending)) break;
    }
    for (int e = lane; e < ENT; e += 64) {
      const uint2 a = J[(e & 1) * SEG + (e >> 1)];
      const uint32_t t = a.x & 0xFFFu;
      uint2 o = make_uint2(R_INVALID, 0u);
      if (t >= J_EXIT && t != J_INVALID) o = make_uint2((t - J_EXIT) | ((a.x >> 12) << 8), a.y);
      exitmap[g * ENT + e] = o;
    }
  }
}

// exit map entry (uint2): x = next entry state [0..7] | #id events << 8 ; y = #forward-deletes | #missing << 16
// (#tokens emitted = events + forward-deletes; Count() = events, quirk Q2).  x == 0xFFFFFFFF: entry unreachable.

// ------------------------------------------------------------------------------------------------
// K3: resolve — per document, chain the exit maps
// ------------------------------------------------------------------------------------------------
__global__ void k_resolve(const uint2* __restrict__ exitmap, const uint64_t* __restrict__ doc_seg_start, uint32_t ndocs,
                          uint8_t* __restrict__ seg_entry, uint32_t* __restrict__ seg_tokbase,
                          uint32_t* __restrict__ doc_ntok, uint32_t* __restrict__ doc_events,
                          uint32_t* __restrict__ doc_missing, uint32_t* __restrict__ error_flag) {
  uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ndocs) return;
  uint64_t g0 = doc_seg_start[d], g1 = doc_seg_start[d + 1];
  if (g1 - g0 > LONG_SEGS) return;               // long documents: k_group_compose / k_long_top / k_group_expand
  uint32_t e = 0, ntok = 0, events = 0, nmiss = 0;
  for (uint64_t g = g0; g < g1; g++) {
    seg_entry[g] = (uint8_t)e;
    seg_tokbase[g] = ntok;
    uint2 x = exitmap[g * ENT + e];
    if (x.x == R_INVALID) { atomicOr(error_flag, 1u); break; }
    e = x.x & 0xFFu;
    uint32_t ev = x.x >> 8, nfd = x.y & 0xFFFFu;
    events += ev;
    ntok += ev + nfd;
    nmiss += x.y >> 16;
  }
  doc_ntok[d] = ntok;
  doc_events[d] = events;
  doc_missing[d] = nmiss;
}

// Long documents (more than LONG_SEGS segments: a multi-megabyte document, or a strip of the trainvocab dataset)
// would make k_resolve a serial chain of millions of dependent loads.  Their segments are cut into ~sqrt(S)
// groups (table built on the host at upload): k_group_compose composes the exit maps of a group for all 80 entry
// states at once (one lane per entry state), k_long_top chains the ~sqrt(S) group maps per document, and
// k_group_expand replays every group from its now known entry state.  Exit-map composition is associative, so the
// result is the same as the serial chain.
struct Group { uint32_t first_seg, nsegs, doc, pad; };
struct LongDoc { uint32_t doc, first_group, ngroups, pad; };

__global__ __launch_bounds__(128) void k_group_compose(const uint2* __restrict__ exitmap, const Group* __restrict__ groups,
                                                       uint4* __restrict__ gmap) {
  const Group gr = groups[blockIdx.x];
  const uint32_t e0 = threadIdx.x;
  if (e0 >= ENT) return;
  uint32_t e = e0, events = 0, nfd = 0, nmiss = 0;
  bool ok = true;
  for (uint32_t k = 0; k < gr.nsegs; k++) {
    const uint2 x = exitmap[(uint64_t)(gr.first_seg + k) * ENT + e];
    if (x.x == R_INVALID) { ok = false; break; }
    e = x.x & 0xFFu;
    events += x.x >> 8;
    nfd += x.y & 0xFFFFu;
    nmiss += x.y >> 16;
  }
  gmap[(uint64_t)blockIdx.x * ENT + e0] = ok ? make_uint4(e, events, nfd, nmiss) : make_uint4(R_INVALID, 0u, 0u, 0u);
}

__global__ void k_long_top(const uint4* __restrict__ gmap, const LongDoc* __restrict__ longs, uint32_t nlong,
                           uint8_t* __restrict__ group_entry, uint4* __restrict__ group_base,
                           uint32_t* __restrict__ doc_ntok, uint32_t* __restrict__ doc_events,
                           uint32_t* __restrict__ doc_missing, uint32_t* __restrict__ error_flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlong) return;
  const LongDoc ld = longs[i];
  uint32_t e = 0, ntok = 0, events = 0, nmiss = 0;
  for (uint32_t k = 0; k < ld.ngroups; k++) {
    const uint32_t gi = ld.first_group + k;
    group_entry[gi] = (uint8_t)e;
    group_base[gi] = make_uint4(ntok, events, nmiss, 0u);
    const uint4 x = gmap[(uint64_t)gi * ENT + e];
    if (x.x == R_INVALID) { atomicOr(error_flag, 1u); break; }
    e = x.x;
    events += x.y;
    ntok += x.y + x.z;
    nmiss += x.w;
  }
  doc_ntok[ld.doc] = ntok;
  doc_events[ld.doc] = events;
  doc_missing[ld.doc] = nmiss;
}

__global__ void k_group_expand(const uint2* __restrict__ exitmap, const Group* __restrict__ groups, uint32_t ngroups,
                               const uint8_t* __restrict__ group_entry, const uint4* __restrict__ group_base,
                               uint8_t* __restrict__ seg_entry, uint32_t* __restrict__ seg_tokbase,
                               uint32_t* __restrict__ error_flag) {
  const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= ngroups) return;
  const Group gr = groups[gi];
  uint32_t e = group_entry[gi], ntok = group_base[gi].x;
  for (uint32_t k = 0; k < gr.nsegs; k++) {
    const uint64_t g = (uint64_t)gr.first_seg + k;
    seg_entry[g] = (uint8_t)e;
    seg_tokbase[g] = ntok;
    const uint2 x = exitmap[g * ENT + e];
    if (x.x == R_INVALID) { atomicOr(error_flag, 1u); break; }
    e = x.x & 0xFFu;
    ntok += (x.x >> 8) + (x.y & 0xFFFFu);
  }
}

// ------------------------------------------------------------------------------------------------
// K4: emit ids (or, HIST, accumulate the trainvocab histogram, training/trainvocab.go:1105-1174)
// ------------------------------------------------------------------------------------------------
// The chain of a segment from its true entry state is ranked in parallel instead of being chased by one lane:
// J_k[s] = state 2^k steps after s (+ ids emitted on the way).  Round k marks, for every already-marked state s,
// the state J_k[s] with rank[s] + ids(s -> J_k[s]); then J is squared.  After ceil(log2(chain length)) rounds all
// chain states carry their output offset and write their ids independently.
// hist layout (all uint32, so that one RCCL all-reduce(sum) merges ranks): scores[n_ids] | tokens_in_text as
// four 16-bit limbs | missing[256] (per-byte counters, > 0 = that byte had no token)
// HIST keeps most of the histogram traffic in LDS: persistent workgroups (one per CU, WV wavefronts) own a
// direct-mapped table of HSLOTS {id, count} counters; an id that finds its slot free or already its own adds in
// LDS, anything else falls through to a global atomic; slots are flushed once when the workgroup retires.  Without
// this, the hot ids (" the", ",") serialise tens of millions of L2 atomics on a handful of addresses.
constexpr int HSLOTS = 8192;
template <bool HIST, int WV>
__global__ __launch_bounds__(WV * 64) void k_chain(const uint2* __restrict__ R, const uint8_t* __restrict__ text,
                                               const uint64_t* __restrict__ doc_begin,
                                               const uint64_t* __restrict__ doc_end,
                                               const uint32_t* __restrict__ seg_doc,
                                               const uint64_t* __restrict__ doc_seg_start, uint64_t nseg,
                                               const uint8_t* __restrict__ seg_entry,
                                               const uint32_t* __restrict__ seg_tokbase,
                                               const uint64_t* __restrict__ tok_offsets, uint32_t delete_id,
                                               uint64_t out_cap, uint32_t* __restrict__ out,
                                               uint32_t* __restrict__ scores, unsigned long long* __restrict__ tokens,
                                               uint32_t* __restrict__ missing_bits) {
  __shared__ uint32_t s_j[WV][2 * SEG];     // target[0..11] | ids on the path << 12
  __shared__ uint16_t s_rk[WV][2 * SEG];    // output rank of a chain state, 0xFFFF = not on the chain
  __shared__ uint32_t s_tag[HIST ? HSLOTS : 1], s_cnt[HIST ? HSLOTS : 1];
  __shared__ unsigned long long s_ntok;
  __shared__ uint32_t s_ndel;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (HIST) {
    for (int j = threadIdx.x; j < HSLOTS; j += WV * 64) { s_tag[j] = 0xFFFFFFFFu; s_cnt[j] = 0; }
    if (threadIdx.x == 0) { s_ntok = 0; s_ndel = 0; }
    __syncthreads();
  }
  uint32_t* J = s_j[wv];
  uint16_t* RK = s_rk[wv];
  for (uint64_t g = (uint64_t)blockIdx.x * WV + wv; g < nseg; g += (uint64_t)gridDim.x * WV) {
  const uint32_t doc = seg_doc[g];
  const uint64_t begin = doc_begin[doc] + (g - doc_seg_start[doc]) * SEG;
  const uint64_t rem = doc_end[doc] - begin;
  const int seglen = rem > SEG ? SEG : (int)rem;
  constexpr int NS = 2 * SEG / 64;          // states per lane: k < NS/2 -> (p = k*64+lane, fd 0), else fd 1
  uint32_t r[NS], jr[NS];
#pragma unroll
  for (int it = 0; it < SEG / 64; it++) {
    const int p = it * 64 + lane;
    uint2 v = make_uint2(R_INVALID, R_INVALID);
    if (p < seglen) v = R[begin + p];
    r[it] = v.x;
    r[SEG / 64 + it] = v.y;
  }
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const int p = (k % (SEG / 64)) * 64 + lane;
    uint32_t j = J_EXIT;                                  // p >= seglen or unreachable: absorbing, emits nothing
    if (r[k] != R_INVALID) {
      const int pn = p + (int)((r[k] >> 24) & 63u);
      const uint32_t fdn = (r[k] >> 30) & 1u;
      const uint32_t nt = ((r[k] & ID_NONE) != ID_NONE ? 1u : 0u) + fdn;
      j = (pn >= seglen ? J_EXIT : fdn * SEG + (uint32_t)pn) | (nt << 12);
    }
    jr[k] = j;
    J[k * 64 + lane] = j;
    RK[k * 64 + lane] = 0xFFFFu;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  const int e = seg_entry[g];
  const int se = (e & 1) * SEG + (e >> 1);
  if (lane == 0) RK[se] = 0;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  uint32_t rk[NS];
  for (int round = 0; round < 12; round++) {
    uint32_t bn[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
      rk[k] = RK[k * 64 + lane];
      const uint32_t t = jr[k] & 0xFFFu;
      bn[k] = t < J_EXIT ? J[t] : 0u;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const uint32_t t = jr[k] & 0xFFFu;
      if (t < J_EXIT) {
        if (rk[k] != 0xFFFFu) RK[t] = (uint16_t)(rk[k] + (jr[k] >> 12));
        jr[k] = (bn[k] & 0xFFFu) | (((jr[k] >> 12) + (bn[k] >> 12)) << 12);
        J[k * 64 + lane] = jr[k];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    if ((J[se] & 0xFFFu) >= J_EXIT) break;                 // wave-uniform (same address in every lane)
  }
  const uint64_t base = HIST ? 0 : tok_offsets[doc] + seg_tokbase[g];
  uint32_t ntok = 0, ndel = 0;
#pragma unroll
  for (int k = 0; k < NS; k++) {
    const uint32_t rank = RK[k * 64 + lane];
    if (rank != 0xFFFFu && r[k] != R_INVALID) {
      const uint32_t id = r[k] & ID_NONE, fdn = (r[k] >> 30) & 1u;
      if (!HIST) {
        uint64_t o = base + rank;
        if (id != ID_NONE) { if (o < out_cap) out[o] = id; o++; }
        if (fdn && o < out_cap) out[o] = delete_id;
      } else {
        const int p = (k % (SEG / 64)) * 64 + lane;
        if (r[k] >> 31) {                                  // trainvocab.go:1166-1173: no token for this byte
          const uint32_t byte = text[begin + p];
          atomicOr(&missing_bits[byte >> 5], 1u << (byte & 31));
        } else {
          const uint32_t adv = (r[k] >> 24) & 63u;         // scores[id] += bytes covered (:1109..1162)
          const uint32_t slot = id & (HSLOTS - 1);
          uint32_t owner = s_tag[slot];
          if (owner == 0xFFFFFFFFu) { owner = atomicCAS(&s_tag[slot], 0xFFFFFFFFu, id); if (owner == 0xFFFFFFFFu) owner = id; }
          if (owner == id) atomicAdd(&s_cnt[slot], adv);
          else atomicAdd(&scores[id], adv);
        }
        ntok += 1 + fdn;                                   // tokensInText++ (also for a missing byte, :1169) / += 2
        ndel += fdn;                                       // scores[deleteToken]++ (:1134,1143,1152)
      }
    }
  }
  if (HIST) {
    for (int o = 32; o > 0; o >>= 1) { ntok += __shfl_xor(ntok, o); ndel += __shfl_xor(ndel, o); }
    if (lane == 0) {
      if (ndel) atomicAdd(&s_ndel, ndel);
      if (ntok) atomicAdd(&s_ntok, (unsigned long long)ntok);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  }  // segments of this wavefront
  if (HIST) {
    __syncthreads();
    for (int j = threadIdx.x; j < HSLOTS; j += WV * 64)
      if (s_cnt[j] != 0) atomicAdd(&scores[s_tag[j]], s_cnt[j]);
    if (threadIdx.x == 0) {
      if (s_ndel) atomicAdd(&scores[delete_id], s_ndel);
      if (s_ntok) atomicAdd(tokens, s_ntok);
    }
  }
}

__global__ void k_hist_finish(const unsigned long long* __restrict__ tokens, const uint32_t* __restrict__ missing_bits,
                              uint32_t* __restrict__ tail) {
  uint32_t t = threadIdx.x;
  if (t < 4) tail[t] = (uint32_t)((*tokens >> (16 * t)) & 0xFFFFull);
  if (t < 256) tail[4 + t] = (missing_bits[t >> 5] >> (t & 31)) & 1u;
}

// pack u32 ids to 2/3/4 little-endian bytes (go/tokenmonster.go:1545, :1817, :2089)
__global__ void k_serialize(const uint32_t* __restrict__ ids, uint64_t n, uint32_t enc, uint8_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t v = ids[i];
  uint8_t* o = out + i * enc;
  o[0] = (uint8_t)v;
  o[1] = (uint8_t)(v >> 8);
  if (enc >= 3) o[2] = (uint8_t)(v >> 16);
  if (enc == 4) o[3] = 0;                     // go :2089 writes a zero high byte
}

}  // namespace tmh

using namespace tmh;

// ------------------------------------------------------------------------------------------------
// tm_batch
// ------------------------------------------------------------------------------------------------
struct tm_batch {
  const tm_vocab* vocab = nullptr;
  uint64_t max_bytes = 0;
  uint32_t max_docs = 0;
  uint64_t max_segs = 0;
  uint64_t nbytes = 0, nseg = 0;
  uint32_t ndocs = 0;
  uint64_t device_bytes = 0;
  hipStream_t last_stream = nullptr;
  // device buffers
  uint8_t* d_text = nullptr;
  bool text_borrowed = false;          // scoring pass: text belongs to a tm_dataset
  uint64_t* d_offsets = nullptr;       // packed batches: doc_begin = d_offsets, doc_end = d_offsets + 1
  const uint64_t* d_doc_begin = nullptr;
  const uint64_t* d_doc_end = nullptr;
  uint32_t* d_doc_nseg = nullptr;
  uint64_t* d_doc_seg_start = nullptr;
  uint32_t* d_seg_doc = nullptr;
  uint2* d_R = nullptr;
  uint2* d_exitmap = nullptr;
  uint8_t* d_seg_entry = nullptr;
  uint32_t* d_seg_tokbase = nullptr;
  uint32_t* d_doc_ntok = nullptr;
  uint32_t* d_doc_events = nullptr;
  uint32_t* d_doc_missing = nullptr;
  uint64_t* d_tok_offsets = nullptr;
  uint64_t* d_scan_tmp = nullptr;   // block sums
  uint64_t* d_totals = nullptr;     // [0] nseg total (device-computed), [1] token total, [2] missing total
  uint32_t* d_error = nullptr;
  // long documents (hierarchical resolve)
  uint32_t ngroups = 0, nlong = 0, cap_groups = 0, cap_long = 0;
  Group* d_groups = nullptr;
  LongDoc* d_longs = nullptr;
  uint4* d_gmap = nullptr;
  uint8_t* d_group_entry = nullptr;
  uint4* d_group_base = nullptr;
  uint32_t* d_out = nullptr;
  uint64_t out_cap = 0;
  hipEvent_t ev[TM_NUM_KERNELS + 1] = {};
  bool have_events = false;
};

namespace {

const char* kKernelNames[TM_NUM_KERNELS] = {"segments", "match_branch", "resolve", "scan", "emit"};

template <typename T>
hipError_t dalloc(tm_batch* b, T** p, uint64_t count) {
  uint64_t bytes = count * sizeof(T);
  hipError_t e = hipMalloc((void**)p, bytes ? bytes : 16);
  if (e == hipSuccess) b->device_bytes += bytes;
  return e;
}

void scan_u32(const uint32_t* in, uint64_t n, uint64_t* block_sums, uint64_t* total, uint64_t* out, hipStream_t st) {
  uint32_t nblocks = (uint32_t)((n + 1 + SCAN_CH - 1) / SCAN_CH);   // covers index n (the total slot)
  k_scan_partial<<<nblocks, SCAN_T, 0, st>>>(in, n, block_sums);
  k_scan_sums<<<1, SCAN_T, 0, st>>>(block_sums, nblocks, total);
  k_scan_final<<<nblocks, SCAN_T, 0, st>>>(in, n, block_sums, out);
}

// host: segment groups of the long documents.  doc d has lens[d] bytes; segments are numbered in document order.
int build_groups(tm_batch* b, const uint64_t* begin, const uint64_t* end, uint32_t ndocs) {
  std::vector<Group> groups;
  std::vector<LongDoc> longs;
  uint64_t seg = 0;
  for (uint32_t d = 0; d < ndocs; d++) {
    const uint64_t S = (end[d] - begin[d] + SEG - 1) / SEG;
    if (S > LONG_SEGS) {
      uint64_t G = 64;
      while (G * G < S) G++;                              // ~sqrt(S) segments per group -> ~sqrt(S) groups
      LongDoc ld{d, (uint32_t)groups.size(), 0u, 0u};
      for (uint64_t k = 0; k < S; k += G) groups.push_back(Group{(uint32_t)(seg + k), (uint32_t)std::min<uint64_t>(G, S - k), d, 0u});
      ld.ngroups = (uint32_t)groups.size() - ld.first_group;
      longs.push_back(ld);
    }
    seg += S;
  }
  if (seg >= (1ull << 32)) return set_error(TM_E_LIMIT, "batch has more than 2^32 segments");
  b->ngroups = (uint32_t)groups.size();
  b->nlong = (uint32_t)longs.size();
  if (b->ngroups == 0) return TM_OK;
  hipError_t e;
  if (b->ngroups > b->cap_groups) {
    (void)hipFree(b->d_groups); (void)hipFree(b->d_gmap); (void)hipFree(b->d_group_entry); (void)hipFree(b->d_group_base);
    b->d_groups = nullptr; b->d_gmap = nullptr; b->d_group_entry = nullptr; b->d_group_base = nullptr;
    b->cap_groups = b->ngroups + b->ngroups / 4 + 16;
    if ((e = hipMalloc((void**)&b->d_groups, (size_t)b->cap_groups * sizeof(Group))) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_gmap, (size_t)b->cap_groups * ENT * sizeof(uint4))) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_group_entry, b->cap_groups)) != hipSuccess ||
        (e = hipMalloc((void**)&b->d_group_base, (size_t)b->cap_groups * sizeof(uint4))) != hipSuccess)
      return hip_fail(e, "hipMalloc (segment groups)");
  }
  if (b->nlong > b->cap_long) {
    (void)hipFree(b->d_longs);
    b->d_longs = nullptr;
    b->cap_long = b->nlong + 16;
    if ((e = hipMalloc((void**)&b->d_longs, (size_t)b->cap_long * sizeof(LongDoc))) != hipSuccess) return hip_fail(e, "hipMalloc (long documents)");
  }
  if ((e = hipMemcpy(b->d_groups, groups.data(), groups.size() * sizeof(Group), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(b->d_longs, longs.data(), longs.size() * sizeof(LongDoc), hipMemcpyHostToDevice)) != hipSuccess)
    return hip_fail(e, "H2D segment groups");
  return TM_OK;
}

int run_pipeline(tm_batch* b, hipStream_t st, bool timed, float* ms, bool emit) {
  const tm_vocab* v = b->vocab;
  b->last_stream = st;
  hipError_t e;
  if (timed && !b->have_events) {
    for (auto& ev : b->ev) if ((e = hipEventCreate(&ev)) != hipSuccess) return hip_fail(e, "hipEventCreate");
    b->have_events = true;
  }
  auto mark = [&](int k) { if (timed) (void)hipEventRecord(b->ev[k], st); };
  (void)hipMemsetAsync(b->d_error, 0, 4, st);
  const uint32_t nd = b->ndocs;
  const uint64_t nseg = b->nseg;
  mark(0);
  if (nd > 0) {
    k_doc_nseg<<<(nd + 255) / 256, 256, 0, st>>>(b->d_doc_begin, b->d_doc_end, nd, b->d_doc_nseg);
    scan_u32(b->d_doc_nseg, nd, b->d_scan_tmp, b->d_totals + 0, b->d_doc_seg_start, st);
    if (nseg > 0) k_segments<<<(uint32_t)((nseg + 255) / 256), 256, 0, st>>>(b->d_doc_seg_start, nd, nseg, b->d_seg_doc);
  }
  mark(1);
  if (nseg > 0)
    k_match_branch<<<(uint32_t)((nseg + WAVES - 1) / WAVES), WAVES * 64, 0, st>>>(v->tables, b->d_text, b->d_doc_begin, b->d_doc_end, b->d_seg_doc,
                                                                                b->d_doc_seg_start, nseg, b->d_R, b->d_exitmap,
                                                                                getenv("TM_DBG") ? atoi(getenv("TM_DBG")) : 0);
  mark(2);
  if (nd > 0)
    k_resolve<<<(nd + 255) / 256, 256, 0, st>>>(b->d_exitmap, b->d_doc_seg_start, nd, b->d_seg_entry, b->d_seg_tokbase,
                                                b->d_doc_ntok, b->d_doc_events, b->d_doc_missing, b->d_error);
  if (b->ngroups > 0) {
    k_group_compose<<<b->ngroups, 128, 0, st>>>(b->d_exitmap, b->d_groups, b->d_gmap);
    k_long_top<<<(b->nlong + 63) / 64, 64, 0, st>>>(b->d_gmap, b->d_longs, b->nlong, b->d_group_entry, b->d_group_base, b->d_doc_ntok,
                                                    b->d_doc_events, b->d_doc_missing, b->d_error);
    k_group_expand<<<(b->ngroups + 63) / 64, 64, 0, st>>>(b->d_exitmap, b->d_groups, b->ngroups, b->d_group_entry, b->d_group_base,
                                                          b->d_seg_entry, b->d_seg_tokbase, b->d_error);
  }
  mark(3);
  if (nd > 0) scan_u32(b->d_doc_ntok, nd, b->d_scan_tmp, b->d_totals + 1, b->d_tok_offsets, st);
  else (void)hipMemsetAsync(b->d_tok_offsets, 0, 8, st);
  mark(4);
  if (emit && nseg > 0)
    k_chain<false, 4><<<(uint32_t)((nseg + 3) / 4), 256, 0, st>>>(b->d_R, b->d_text, b->d_doc_begin, b->d_doc_end, b->d_seg_doc, b->d_doc_seg_start,
                                                               nseg, b->d_seg_entry, b->d_seg_tokbase, b->d_tok_offsets, v->tables.delete_id,
                                                               b->out_cap, b->d_out, nullptr, nullptr, nullptr);
  mark(5);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "kernel launch");
  if (timed) {
    if ((e = hipEventSynchronize(b->ev[TM_NUM_KERNELS])) != hipSuccess) return hip_fail(e, "hipEventSynchronize");
    for (int k = 0; k < TM_NUM_KERNELS; k++) (void)hipEventElapsedTime(&ms[k], b->ev[k], b->ev[k + 1]);
  }
  return TM_OK;
}

// after a run: make sure the output buffer was large enough; if not, grow it and redo the emit stage
int ensure_output(tm_batch* b) {
  hipError_t e;
  uint64_t totals[3];
  uint32_t err = 0;
  if ((e = hipStreamSynchronize(b->last_stream)) != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  if ((e = hipMemcpy(totals, b->d_totals, sizeof totals, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "hipMemcpy totals");
  if ((e = hipMemcpy(&err, b->d_error, 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "hipMemcpy error flag");
  if (err != 0) return set_error(TM_E_HIP, "device pipeline inconsistency (unreachable segment entry state)");
  uint64_t total = b->ndocs ? totals[1] : 0;
  if (total > b->out_cap) {
    (void)hipFree(b->d_out);
    b->device_bytes -= b->out_cap * 4;
    b->d_out = nullptr;
    b->out_cap = total + 1024;
    if ((e = dalloc(b, &b->d_out, b->out_cap)) != hipSuccess) return hip_fail(e, "hipMalloc output");
    hipStream_t st = b->last_stream;
    k_chain<false, 4><<<(uint32_t)((b->nseg + 3) / 4), 256, 0, st>>>(b->d_R, b->d_text, b->d_doc_begin, b->d_doc_end, b->d_seg_doc, b->d_doc_seg_start,
                                                                  b->nseg, b->d_seg_entry, b->d_seg_tokbase, b->d_tok_offsets,
                                                                  b->vocab->tables.delete_id, b->out_cap, b->d_out, nullptr, nullptr, nullptr);
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "emit rerun");
  }
  return TM_OK;
}

}  // namespace

extern "C" {

const char* tm_kernel_name(int k) { return k >= 0 && k < TM_NUM_KERNELS ? kKernelNames[k] : ""; }

static int make_workspace(const tm_vocab* v, uint64_t max_bytes, uint32_t max_docs, bool own_text, bool with_output, tm_batch** out) {
  *out = nullptr;
  auto* b = new tm_batch();
  b->vocab = v;
  b->max_bytes = max_bytes;
  b->max_docs = max_docs;
  b->max_segs = max_bytes / SEG + max_docs + 1;
  b->out_cap = with_output ? max_bytes / 2 + 2ull * max_docs + 1024 : 0;   // grows on demand (worst case is 2 ids per byte)
  b->text_borrowed = !own_text;
  const uint64_t nd1 = (uint64_t)max_docs + 1;
  const uint64_t scan_blocks = std::max<uint64_t>((nd1 + SCAN_CH) / SCAN_CH + 1, 16);
  hipError_t e = hipSuccess;
  if ((own_text && (e = dalloc(b, &b->d_text, max_bytes + 256)) != hipSuccess) || (e = dalloc(b, &b->d_offsets, 2 * nd1)) != hipSuccess ||
      (e = dalloc(b, &b->d_doc_nseg, nd1)) != hipSuccess || (e = dalloc(b, &b->d_doc_seg_start, nd1 + 1)) != hipSuccess ||
      (e = dalloc(b, &b->d_seg_doc, b->max_segs)) != hipSuccess || (e = dalloc(b, &b->d_R, max_bytes + 64)) != hipSuccess ||
      (e = dalloc(b, &b->d_exitmap, b->max_segs * ENT)) != hipSuccess || (e = dalloc(b, &b->d_seg_entry, b->max_segs)) != hipSuccess ||
      (e = dalloc(b, &b->d_seg_tokbase, b->max_segs)) != hipSuccess || (e = dalloc(b, &b->d_doc_ntok, nd1)) != hipSuccess ||
      (e = dalloc(b, &b->d_doc_events, nd1)) != hipSuccess || (e = dalloc(b, &b->d_doc_missing, nd1)) != hipSuccess ||
      (e = dalloc(b, &b->d_tok_offsets, nd1 + 1)) != hipSuccess || (e = dalloc(b, &b->d_scan_tmp, scan_blocks)) != hipSuccess ||
      (e = dalloc(b, &b->d_totals, 4)) != hipSuccess || (e = dalloc(b, &b->d_error, 4)) != hipSuccess ||
      (e = dalloc(b, &b->d_out, b->out_cap)) != hipSuccess) {
    tm_batch_free(b);
    return hip_fail(e, "hipMalloc (batch workspace)");
  }
  (void)hipMemset(b->d_totals, 0, 32);
  *out = b;
  return TM_OK;
}

int tm_batch_create(const tm_vocab* v, uint64_t max_bytes, uint32_t max_docs, tm_batch** out) {
  if (!v || !out) return set_error(TM_E_INVALID, "null argument");
  return make_workspace(v, max_bytes, max_docs, true, true, out);
}

void tm_batch_free(tm_batch* b) {
  if (!b) return;
  void* ptrs[] = {b->text_borrowed ? nullptr : (void*)b->d_text, b->d_offsets, b->d_doc_nseg, b->d_doc_seg_start, b->d_seg_doc, b->d_R, b->d_exitmap, b->d_seg_entry,
                  b->d_seg_tokbase, b->d_doc_ntok, b->d_doc_events, b->d_doc_missing, b->d_tok_offsets, b->d_scan_tmp, b->d_totals,
                  b->d_error, b->d_out, b->d_groups, b->d_longs, b->d_gmap, b->d_group_entry, b->d_group_base};
  for (void* p : ptrs) (void)hipFree(p);
  if (b->have_events) for (auto& ev : b->ev) (void)hipEventDestroy(ev);
  delete b;
}

int tm_batch_upload(tm_batch* b, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs) {
  if (!b || (ndocs && (!offsets))) return set_error(TM_E_INVALID, "null argument");
  if (ndocs > b->max_docs) return set_error(TM_E_LIMIT, "batch has %u documents, workspace sized for %u", ndocs, b->max_docs);
  uint64_t nbytes = ndocs ? offsets[ndocs] : 0;
  if (ndocs && offsets[0] != 0) return set_error(TM_E_INVALID, "offsets[0] must be 0");
  if (nbytes > b->max_bytes) return set_error(TM_E_LIMIT, "batch has %llu bytes, workspace sized for %llu", (unsigned long long)nbytes, (unsigned long long)b->max_bytes);
  uint64_t nseg = 0;
  for (uint32_t d = 0; d < ndocs; d++) {
    if (offsets[d + 1] < offsets[d]) return set_error(TM_E_INVALID, "offsets not monotone at document %u", d);
    nseg += (offsets[d + 1] - offsets[d] + SEG - 1) / SEG;
  }
  hipError_t e;
  if (nbytes && (e = hipMemcpy(b->d_text, text, nbytes, hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "H2D text");
  if (ndocs && (e = hipMemcpy(b->d_offsets, offsets, ((uint64_t)ndocs + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "H2D offsets");
  b->ndocs = ndocs;
  b->nbytes = nbytes;
  b->nseg = nseg;
  b->d_doc_begin = b->d_offsets;
  b->d_doc_end = b->d_offsets + 1;
  return build_groups(b, offsets, offsets + 1, ndocs);
}

int tm_batch_run(tm_batch* b, void* stream) {
  if (!b) return set_error(TM_E_INVALID, "null argument");
  return run_pipeline(b, (hipStream_t)stream, false, nullptr, true);
}

int tm_batch_run_timed(tm_batch* b, void* stream, float* ms) {
  if (!b || !ms) return set_error(TM_E_INVALID, "null argument");
  return run_pipeline(b, (hipStream_t)stream, true, ms, true);
}

int tm_batch_totals(tm_batch* b, uint64_t* total_tokens, uint64_t* total_missing) {
  if (!b) return set_error(TM_E_INVALID, "null argument");
  int rc = ensure_output(b);
  if (rc != TM_OK) return rc;
  hipError_t e;
  uint64_t totals[3] = {0, 0, 0};
  if ((e = hipMemcpy(totals, b->d_totals, sizeof totals, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "hipMemcpy totals");
  if (total_tokens) *total_tokens = b->ndocs ? totals[1] : 0;
  if (total_missing) {
    std::vector<uint32_t> miss(b->ndocs);
    if (b->ndocs && (e = hipMemcpy(miss.data(), b->d_doc_missing, (size_t)b->ndocs * 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "hipMemcpy missing");
    uint64_t m = 0;
    for (auto x : miss) m += x;
    *total_missing = m;
  }
  return TM_OK;
}

int tm_batch_download(tm_batch* b, uint32_t* tokens_out, uint64_t tokens_cap, uint64_t* tok_offsets, uint32_t* missing) {
  if (!b) return set_error(TM_E_INVALID, "null argument");
  int rc = ensure_output(b);
  if (rc != TM_OK) return rc;
  hipError_t e;
  std::vector<uint64_t> offs((size_t)b->ndocs + 1, 0);
  if (b->ndocs && (e = hipMemcpy(offs.data(), b->d_tok_offsets, offs.size() * 8, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H tok_offsets");
  if (tok_offsets) std::memcpy(tok_offsets, offs.data(), offs.size() * 8);
  if (missing && b->ndocs && (e = hipMemcpy(missing, b->d_doc_missing, (size_t)b->ndocs * 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H missing");
  uint64_t total = offs[b->ndocs];
  if (total > tokens_cap) return set_error(TM_E_NOSPACE, "tokens_cap %llu < %llu required", (unsigned long long)tokens_cap, (unsigned long long)total);
  if (total && (e = hipMemcpy(tokens_out, b->d_out, total * 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H tokens");
  return TM_OK;
}

const uint32_t* tm_batch_device_tokens(const tm_batch* b) { return b->d_out; }
const uint64_t* tm_batch_device_tok_offsets(const tm_batch* b) { return b->d_tok_offsets; }
uint64_t tm_batch_device_bytes(const tm_batch* b) { return b->device_bytes; }

// ---- host-buffer entry points ---------------------------------------------------------------------
static int with_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, tm_batch** pb, bool emit) {
  if (!v || (ndocs && !offsets)) return set_error(TM_E_INVALID, "null argument");
  uint64_t nbytes = ndocs ? offsets[ndocs] : 0;
  int rc = tm_batch_create(v, nbytes, ndocs, pb);
  if (rc != TM_OK) return rc;
  if ((rc = tm_batch_upload(*pb, text, offsets, ndocs)) != TM_OK) return rc;
  return run_pipeline(*pb, nullptr, false, nullptr, emit);
}

int tm_tokenize_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t* tokens_out,
                      uint64_t tokens_cap, uint64_t* tok_offsets, uint32_t* missing) {
  tm_batch* b = nullptr;
  int rc = with_batch(v, text, offsets, ndocs, &b, true);
  if (rc == TM_OK) rc = tm_batch_download(b, tokens_out, tokens_cap, tok_offsets, missing);
  tm_batch_free(b);
  return rc;
}

int tm_count_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint64_t* counts,
                   uint32_t* missing) {
  tm_batch* b = nullptr;
  int rc = with_batch(v, text, offsets, ndocs, &b, false);
  if (rc == TM_OK && ndocs) {
    hipError_t e;
    std::vector<uint32_t> ev(ndocs);
    uint32_t err = 0;
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) rc = hip_fail(e, "sync");
    else if ((e = hipMemcpy(&err, b->d_error, 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H");
    else if (err) rc = set_error(TM_E_HIP, "device pipeline inconsistency");
    else if ((e = hipMemcpy(ev.data(), b->d_doc_events, (size_t)ndocs * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H counts");
    else if (missing && (e = hipMemcpy(missing, b->d_doc_missing, (size_t)ndocs * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H missing");
    if (rc == TM_OK && counts) for (uint32_t d = 0; d < ndocs; d++) counts[d] = ev[d];
  }
  tm_batch_free(b);
  return rc;
}

int tm_tokenize_batch_serialized(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs,
                                 uint32_t encoding_length, uint8_t* bytes_out, uint64_t bytes_cap, uint64_t* byte_offsets,
                                 uint32_t* missing, uint32_t* encoding_length_used) {
  if (!v) return set_error(TM_E_INVALID, "null argument");
  if (encoding_length <= 1) encoding_length = v->host.n_ids <= 65536 ? 2 : 3;          // go :990-996
  if (encoding_length < 2 || encoding_length > 4) return set_error(TM_E_INVALID, "Invalid encoding length");   // go :1012
  if (encoding_length_used) *encoding_length_used = encoding_length;
  tm_batch* b = nullptr;
  int rc = with_batch(v, text, offsets, ndocs, &b, true);
  if (rc == TM_OK) rc = ensure_output(b);
  if (rc == TM_OK) {
    hipError_t e;
    std::vector<uint64_t> offs((size_t)ndocs + 1, 0);
    if (ndocs && (e = hipMemcpy(offs.data(), b->d_tok_offsets, offs.size() * 8, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H tok_offsets");
    if (rc == TM_OK) {
      uint64_t total = offs[ndocs];
      if (byte_offsets) for (size_t d = 0; d <= ndocs; d++) byte_offsets[d] = offs[d] * encoding_length;
      if (missing && ndocs && (e = hipMemcpy(missing, b->d_doc_missing, (size_t)ndocs * 4, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H missing");
      if (rc == TM_OK && total * encoding_length > bytes_cap) rc = set_error(TM_E_NOSPACE, "bytes_cap too small");
      if (rc == TM_OK && total) {
        uint8_t* d_bytes = nullptr;
        if ((e = hipMalloc((void**)&d_bytes, total * encoding_length)) != hipSuccess) rc = hip_fail(e, "hipMalloc");
        else {
          k_serialize<<<(uint32_t)((total + 255) / 256), 256>>>(b->d_out, total, encoding_length, d_bytes);
          if ((e = hipMemcpy(bytes_out, d_bytes, total * encoding_length, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_fail(e, "D2H bytes");
          (void)hipFree(d_bytes);
        }
      }
    }
  }
  tm_batch_free(b);
  return rc;
}

// ---- trainvocab scoring pass ----------------------------------------------------------------------
}  // extern "C"

struct tm_dataset {
  uint8_t* d_text = nullptr;
  uint64_t n = 0;
  tm_batch* ws = nullptr;          // workspace, created on first use and reused by every scoring pass
  uint32_t ws_docs = 0;
  uint32_t* d_hist = nullptr;      // scores | 4 token limbs | 256 missing counters
  uint64_t hist_words = 0;
  unsigned long long* d_tokens = nullptr;
  uint32_t* d_missing_bits = nullptr;
  int n_cu = 256;
};

static int score_run(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
                     hipStream_t st) {
  if (!v || !d) return set_error(TM_E_INVALID, "null argument");
  std::vector<uint64_t> be;
  uint64_t whole_off = 0, whole_len = d->n;
  if (n_strips == 0) { strip_off = &whole_off; strip_len = &whole_len; n_strips = 1; }
  be.resize(2ull * n_strips);
  uint64_t nseg = 0;
  for (uint32_t k = 0; k < n_strips; k++) {
    if (strip_off[k] > d->n || strip_len[k] > d->n - strip_off[k]) return set_error(TM_E_INVALID, "strip %u outside the dataset", k);
    be[k] = strip_off[k];
    be[n_strips + k] = strip_off[k] + strip_len[k];
    nseg += (strip_len[k] + SEG - 1) / SEG;
  }
  hipError_t e;
  if (d->ws && (d->ws->vocab != v || d->ws_docs < n_strips)) { tm_batch_free(d->ws); d->ws = nullptr; }
  if (!d->ws) {
    int rc = make_workspace(v, d->n, n_strips, false, false, &d->ws);
    if (rc != TM_OK) return rc;
    d->ws_docs = n_strips;
    d->ws->d_text = d->d_text;
  }
  tm_batch* b = d->ws;
  b->vocab = v;
  const uint64_t words = (uint64_t)v->host.n_ids + 4 + 256;
  if (d->hist_words != words) {
    (void)hipFree(d->d_hist);
    d->d_hist = nullptr;
    if ((e = hipMalloc((void**)&d->d_hist, words * 4)) != hipSuccess) return hip_fail(e, "hipMalloc histogram");
    d->hist_words = words;
  }
  if ((e = hipMemcpyAsync(b->d_offsets, be.data(), be.size() * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return hip_fail(e, "H2D strips");
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return hip_fail(e, "sync");   // `be` is a host temporary
  b->d_doc_begin = b->d_offsets;
  b->d_doc_end = b->d_offsets + n_strips;
  b->ndocs = n_strips;
  b->nbytes = d->n;
  b->nseg = nseg;
  { int grc = build_groups(b, be.data(), be.data() + n_strips, n_strips); if (grc != TM_OK) return grc; }
  (void)hipMemsetAsync(d->d_hist, 0, words * 4, st);
  (void)hipMemsetAsync(d->d_tokens, 0, 8, st);
  (void)hipMemsetAsync(d->d_missing_bits, 0, 32, st);
  int rc = run_pipeline(b, st, false, nullptr, false);
  if (rc != TM_OK) return rc;
  if (nseg > 0)
    k_chain<true, 16><<<(uint32_t)std::min<uint64_t>((nseg + 15) / 16, (uint64_t)d->n_cu), 1024, 0, st>>>(b->d_R, b->d_text, b->d_doc_begin, b->d_doc_end, b->d_seg_doc, b->d_doc_seg_start,
                                                              nseg, b->d_seg_entry, b->d_seg_tokbase, b->d_tok_offsets,
                                                              v->tables.has_delete ? v->tables.delete_id : 0, 0, nullptr, d->d_hist, d->d_tokens,
                                                              d->d_missing_bits);
  k_hist_finish<<<1, 256, 0, st>>>(d->d_tokens, d->d_missing_bits, d->d_hist + v->host.n_ids);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "kernel launch");
  return TM_OK;
}

extern "C" {

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
  { int dev = 0, cu = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) d->n_cu = cu; }
  *out = d;
  return TM_OK;
}

void tm_dataset_free(tm_dataset* d) {
  if (!d) return;
  tm_batch_free(d->ws);
  (void)hipFree(d->d_text); (void)hipFree(d->d_hist); (void)hipFree(d->d_tokens); (void)hipFree(d->d_missing_bits);
  delete d;
}

int tm_score_device(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
                    void* stream, uint32_t** dev_hist, uint64_t* n_words) {
  int rc = score_run(v, d, strip_off, strip_len, n_strips, (hipStream_t)stream);
  if (rc != TM_OK) return rc;
  if (dev_hist) *dev_hist = d->d_hist;
  if (n_words) *n_words = d->hist_words;
  return TM_OK;
}

int tm_score_device_into(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
                         void* stream, uint32_t* dst_device, uint64_t dst_words) {
  if (!dst_device) return set_error(TM_E_INVALID, "null argument");
  int rc = score_run(v, d, strip_off, strip_len, n_strips, (hipStream_t)stream);
  if (rc != TM_OK) return rc;
  if (dst_words < d->hist_words) return set_error(TM_E_NOSPACE, "destination holds %llu words, histogram has %llu", (unsigned long long)dst_words, (unsigned long long)d->hist_words);
  hipError_t e = hipMemcpyAsync(dst_device, d->d_hist, d->hist_words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "D2D histogram");
  return TM_OK;
}

int tm_score(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len, uint32_t n_strips,
             uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]) {
  int rc = score_run(v, d, strip_off, strip_len, n_strips, nullptr);
  if (rc != TM_OK) return rc;
  hipError_t e;
  std::vector<uint32_t> h(d->hist_words);
  uint32_t err = 0;
  if ((e = hipMemcpy(h.data(), d->d_hist, h.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H histogram");
  if ((e = hipMemcpy(&err, d->ws->d_error, 4, hipMemcpyDeviceToHost)) != hipSuccess) return hip_fail(e, "D2H error flag");
  if (err) return set_error(TM_E_HIP, "device pipeline inconsistency (unreachable segment entry state)");
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

}  // extern "C"
