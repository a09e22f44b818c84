// tm_vocab.hip — .vocab parser and device-table flattener (host side of tm_vocab_load).
//
// Replaces the table construction of Load, go/tokenmonster.go:2656-2736 (== tokenmonster-cpp
// src/tokenmonster.cpp:1287-1359): reads the records in file order (= pansearch index order),
// resolves alt lengths / ids from earlier records exactly as :2703-2712 does, and instead of
// pansearch.Fast builds the byte trie described in tm_tables.h.
#include <hip/hip_runtime.h>
#include <chrono>
#include <mutex>
#include <map>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "tm_device.h"

namespace tmh {

namespace {
thread_local char g_err[512];
uint32_t rd24(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }
}  // namespace

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
const char* last_error() { return g_err; }

int hip_fail(hipError_t e, const char* what) {
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
    return set_error(TM_E_NODEVICE, "%s: %s", what, hipGetErrorString(e));
  return set_error(TM_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

void note_table_use(const tm_vocab* v, hipStream_t st) {
  // a fresh event every time: a stream handle may be a NEW stream by now (a lane that rebuilt its workspace), and re-recording an event
  // whose last stream is gone is an error on this runtime
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
  if (hipEventRecord(ev, st) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev); return; }
  std::lock_guard<std::mutex> g(v->use_mu);
  for (auto& u : v->last_use) if (u.first == st) { (void)hipEventDestroy(u.second); u.second = ev; return; }
  v->last_use.emplace_back(st, ev);
}

int enter_device(const tm_vocab* v) {
  if (!v) return set_error(TM_E_INVALID, "null argument");
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == v->device) return TM_OK;
  const hipError_t e = hipSetDevice(v->device);
  return e == hipSuccess ? TM_OK : hip_fail(e, "hipSetDevice (device of the vocabulary)");
}


namespace {
struct StageTimer {           // TM_TRACE_BUILD=1: wall time of the stages of a table build on stderr (tools/build_profile.cpp)
  const bool on = getenv("TM_TRACE_BUILD") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "  [tables] %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};
}  // namespace

// ---- .vocab bytes -> records (go/tokenmonster.go:2656-2736; layout: SURVEY.md Appendix A) ---------------------------------------------
int parse_records(const uint8_t* f, size_t n, HostVocab& hv) {
  size_t pos = 0;
#define NEED(k) do { if (pos + (size_t)(k) > n) return set_error(TM_E_INVALID, "truncated .vocab at byte %zu", pos); } while (0)
  NEED(24);
  hv.capcode = f[0]; hv.charset = f[1]; hv.norm_flag = f[2]; hv.level = f[3]; hv.reserve = f[4];
  if (hv.charset > 2 || hv.capcode > 2) return set_error(TM_E_INVALID, "not a valid TokenMonster vocabulary");  // go :2675
  hv.unk = rd24(f + 8); hv.vocab_size = rd24(f + 11); hv.n_ids = rd24(f + 14); hv.n_info = rd24(f + 17);
  hv.delete_id = rd24(f + 20); hv.max_len = f[23];
  pos = 24;
  // header fields the kernels later use as indices or bit fields are checked here, before anything is allocated from them
  if (hv.unk != TM_NONE && hv.unk >= hv.n_ids) return set_error(TM_E_INVALID, "unk token id %u out of range (%u ids)", hv.unk, hv.n_ids);
  if (hv.delete_id != TM_NONE && hv.delete_id >= hv.n_ids) return set_error(TM_E_INVALID, "deleteToken id %u out of range (%u ids)", hv.delete_id, hv.n_ids);
  if (hv.max_len > 40) return set_error(TM_E_INVALID, "maxTokenLength %u > 40", hv.max_len);                      // go :2695
  if (hv.n_ids > kRowIdMask) return set_error(TM_E_LIMIT, "%u ids: the device tables hold at most %u", hv.n_ids, kRowIdMask);
  if (hv.n_info >= kMaxNodes) return set_error(TM_E_LIMIT, "%u index records: the walk tables hold fewer than %u trie nodes", hv.n_info, kMaxNodes);
  if ((uint64_t)hv.n_info * 16 > n) return set_error(TM_E_INVALID, "truncated .vocab: %u records do not fit %zu bytes", hv.n_info, n);
  const uint32_t n_info = hv.n_info;
  hv.keys.clear(); hv.key_off.assign(1, 0);
  hv.keys.reserve(n);
  hv.key_off.reserve((size_t)n_info + 1);
  hv.rec_flag.resize(n_info); hv.rec_nwords.resize(n_info); hv.rec_id.resize(n_info); hv.rec_index1.resize(n_info); hv.rec_index2.resize(n_info); hv.rec_score.resize(n_info);
  uint32_t prev_len = 0;
  for (uint32_t i = 0; i < n_info; i++) {
    NEED(1);
    uint32_t kl = f[pos++];
    if (kl == 0 || kl > 40) return set_error(TM_E_INVALID, "record %u: key length %u", i, kl);     // go :2695
    NEED(kl + 15);
    if (kl < prev_len || (kl == prev_len && i > 0 && std::memcmp(&hv.keys[hv.key_off[i - 1]], f + pos, kl) >= 0))
      return set_error(TM_E_INVALID, "record %u out of (length, bytewise) order", i);               // tokenmonster.cpp:1352-1357
    prev_len = kl;
    hv.keys.insert(hv.keys.end(), f + pos, f + pos + kl);
    hv.key_off.push_back((uint32_t)hv.keys.size());
    pos += kl;
    const uint32_t flag = f[pos], nw = f[pos + 1], index1 = rd24(f + pos + 2), index2 = rd24(f + pos + 5), id = rd24(f + pos + 8);
    std::memcpy(&hv.rec_score[i], f + pos + 11, 4);
    pos += 15;
    if (id >= hv.n_ids) return set_error(TM_E_INVALID, "record %u: id %u out of range", i, id);
    if (nw > 31) return set_error(TM_E_LIMIT, "record %u: nWords %u > 31", i, nw);
    if (index1 != TM_NONE && index1 >= i) return set_error(TM_E_INVALID, "record %u: alternative does not precede it", i);   // go :2703-2706
    if (index2 != TM_NONE && index2 >= i) return set_error(TM_E_INVALID, "record %u: alternative does not precede it", i);   // go :2708-2711
    hv.rec_flag[i] = (uint8_t)flag; hv.rec_nwords[i] = (uint8_t)nw; hv.rec_id[i] = id; hv.rec_index1[i] = index1; hv.rec_index2[i] = index2;
  }
  NEED(256);
  std::memcpy(hv.begin_byte, f + pos, 256);
  pos += 256;
  NEED(3);
  uint32_t nd = rd24(f + pos);
  pos += 3;
  for (uint32_t i = 0; i < nd; i++) { NEED(1); uint32_t l = f[pos++]; NEED(l + 7); pos += l + 7; }
  if (pos != n) return set_error(TM_E_INVALID, "trailing bytes after .vocab payload");   // go :2731
#undef NEED
  return TM_OK;
}

// ---- the trie: accepting node id == record ordinal; internal nodes numbered from n_info ------------------------------------------------
int build_trie(const HostVocab& hv, Trie& t, TrieVisitor* on_key) {
  const uint32_t n_info = hv.n_info;
  const uint8_t* K = hv.keys.data();
  const uint32_t* ko = hv.key_off.data();
  // the records lie by length, bytewise within a length: one run per length, merged into plain lexicographic order through a small heap
  struct Run { uint32_t cur, end; };
  std::vector<Run> runs;
  for (uint32_t i = 0; i < n_info;) {
    const uint32_t L = ko[i + 1] - ko[i];
    uint32_t j = i + 1;
    while (j < n_info && ko[j + 1] - ko[j] == L) j++;
    runs.push_back(Run{i, j});
    i = j;
  }
  auto less = [&](uint32_t a, uint32_t b) {             // key a < key b, lexicographically (a prefix before what extends it)
    const uint32_t la = ko[a + 1] - ko[a], lb = ko[b + 1] - ko[b];
    const int c = std::memcmp(K + ko[a], K + ko[b], la < lb ? la : lb);
    return c != 0 ? c < 0 : la < lb;
  };
  std::vector<uint32_t> heap;                            // run numbers, the run whose current key is smallest on top
  auto run_less = [&](uint32_t x, uint32_t y) { return less(runs[y].cur, runs[x].cur); };     // (std:: heaps are max-heaps)
  for (uint32_t r = 0; r < runs.size(); r++) heap.push_back(r);
  std::make_heap(heap.begin(), heap.end(), run_less);

  t.n_info = n_info;
  t.depth_of.assign(n_info, 0); t.byte_of.assign(n_info, 0); t.parent_of.assign(n_info, 0);
  t.depth_of.reserve(hv.keys.size() / 2 + n_info); t.byte_of.reserve(hv.keys.size() / 2 + n_info); t.parent_of.reserve(hv.keys.size() / 2 + n_info);
  for (auto& c : t.root_child) c = kNone;
  std::vector<uint32_t> e_parent, e_kid;                 // edges in order of creation: per parent their bytes ascend
  e_parent.reserve(hv.keys.size() / 2 + n_info); e_kid.reserve(hv.keys.size() / 2 + n_info);
  uint32_t next_internal = n_info;
  uint32_t path[41], path_ord[41];                       // node / key ordinal of the first d + 1 bytes of the key before
  const uint8_t* prev = nullptr;
  uint32_t prev_len = 0;
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), run_less);
    Run& r = runs[heap.back()];
    const uint32_t i = r.cur++;
    if (r.cur < r.end) std::push_heap(heap.begin(), heap.end(), run_less); else heap.pop_back();
    const uint8_t* k = K + ko[i];
    const uint32_t kl = ko[i + 1] - ko[i];
    uint32_t common = 0;
    while (common < prev_len && common < kl && k[common] == prev[common]) common++;
    if (common == kl) return set_error(TM_E_INVALID, "record %u repeats a key", i);           // (equal keys: the order check of the records rules it out)
    uint32_t node = common ? path[common - 1] : Trie::kRoot;
    for (uint32_t d = common; d < kl; d++) {
      uint32_t c;
      if (d + 1 == kl) c = i;                            // the key's own node: its prefixes were visited before it, so it is new
      else {
        if (next_internal >= kMaxNodes) return set_error(TM_E_LIMIT, "vocabulary needs more than %u trie nodes", kMaxNodes);
        c = next_internal++;
        t.depth_of.push_back(0); t.byte_of.push_back(0); t.parent_of.push_back(0);
      }
      t.depth_of[c] = (uint8_t)(d + 1); t.byte_of[c] = k[d]; t.parent_of[c] = node;
      if (node == Trie::kRoot) t.root_child[k[d]] = c;
      else { e_parent.push_back(node); e_kid.push_back(((uint32_t)k[d] << 24) | c); }
      path[d] = c; path_ord[d] = d + 1 == kl ? i : kNone;
      node = c;
    }
    if (on_key) on_key->key(i, path_ord);
    prev = k; prev_len = kl;
  }
  t.n_nodes = next_internal;
  // children per node, bytes ascending (a stable counting sort of the edges by parent keeps the order of creation)
  t.kid_start.assign((size_t)t.n_nodes + 1, 0);
  for (uint32_t p : e_parent) t.kid_start[p + 1]++;
  for (uint32_t n = 0; n < t.n_nodes; n++) t.kid_start[n + 1] += t.kid_start[n];
  t.kid.resize(e_kid.size());
  { std::vector<uint32_t> fill(t.kid_start.begin(), t.kid_start.end() - 1);
    for (size_t q = 0; q < e_kid.size(); q++) t.kid[fill[e_parent[q]]++] = e_kid[q]; }
  return TM_OK;
}

// ---- records + trie -> the tables of tm_tables.h -----------------------------------------------------------------------------------------
int build_tables(HostVocab& hv, const Trie& t, const std::vector<uint32_t>* perm) {
  StageTimer st;
  // node ids as the tables carry them: the trie's own (accepting node == record ordinal, tm_device.h) unless a layout by use renumbers them
  // (tm_vocab_tune: `perm` maps accepting nodes onto [0, n_info) and the others onto [n_info, n_nodes)).  Everything that is INDEXED by a
  // node id (rows, space-prefix links, values, suffix links) or CARRIES one (node values, check words, link targets) goes through P();
  // cmask / base_of / depth_of / has_child below stay indexed by the trie's ids.
  auto P = [&](uint32_t n) -> uint32_t { return perm ? (*perm)[n] : n; };
  const uint32_t n_info = hv.n_info, n_nodes = t.n_nodes;
  const uint32_t kRoot = Trie::kRoot;
  const std::vector<uint8_t>& depth_of = t.depth_of;
  const std::vector<uint8_t>& flags = hv.rec_flag;
  const std::vector<uint8_t>& nwords = hv.rec_nwords;
  const std::vector<uint32_t>& ids = hv.rec_id;
  auto klen = [&](uint32_t i) { return hv.key_off[i + 1] - hv.key_off[i]; };
  // rows: first-token constants (tm_tables.h): allLetters + max0(nWords-1) + nWords*100 (+ the length, for the alternatives), alternatives resolved as Load does (go :2703-2712)
  hv.rows.resize(n_info);
  for (uint32_t i = 0; i < n_info; i++) {
    const uint32_t flag = flags[i], nw = nwords[i], id = ids[i], index1 = hv.rec_index1[i], index2 = hv.rec_index2[i];
    uint32_t id1 = 0, id2 = 0, len1 = 0, len2 = 0, nw1 = 0, nw2 = 0, fl1 = 0, fl2 = 0;
    if (index1 != TM_NONE) { len1 = klen(index1); id1 = ids[index1]; nw1 = nwords[index1]; fl1 = flags[index1]; }
    if (index2 != TM_NONE) { len2 = klen(index2); id2 = ids[index2]; nw2 = nwords[index2]; fl2 = flags[index2]; }
    auto fconst = [](uint32_t fl, uint32_t nwk) { return ((fl >> 7) & 1u) + (nwk > 0 ? nwk - 1 : 0u) + nwk * 100u; };
    Row& r = hv.rows[P(i)];
    r.x = id | (fconst(flag, nw) << kRowIdBits);
    r.y = id1 | ((len1 ? len1 + fconst(fl1, nw1) : 0u) << kRowIdBits);
    r.z = id2 | ((len2 ? len2 + fconst(fl2, nw2) : 0u) << kRowIdBits);
    r.w = len1 | (len2 << 6) | ((flag & 1u) << 12) | ((fl1 & 1u) << 13) | ((fl2 & 1u) << 14) | (((flag >> 3) & 1u) << 15) | (((fl1 >> 3) & 1u) << 16) |
          (((fl2 >> 3) & 1u) << 17) | ((nw >= 2 ? 1u : 0u) << 18) | ((nw1 >= 2 ? 1u : 0u) << 19) | ((nw2 >= 2 ? 1u : 0u) << 20) | (((flag >> 5) & 1u) << 21);
  }
  st.mark("rows");
  std::vector<uint8_t> has_child(n_nodes, 0);
  std::vector<uint32_t> cmask(n_nodes, 0);
  // child-byte filter of every node: bit (b & 31) is set if the node has a child over byte b.  A walk only probes for a byte
  // whose bit is set, so a probe that cannot hit (half of all positions end on one) is almost never issued: most nodes have one child.
  for (uint32_t n = 0; n < n_nodes; n++) {
    uint32_t m = 0;
    for (uint32_t q = t.kid_start[n]; q < t.kid_start[n + 1]; q++) m |= 1u << ((t.kid[q] >> 24) & 31u);
    cmask[n] = m;
    has_child[n] = t.kid_start[n + 1] > t.kid_start[n];
  }
  std::vector<uint32_t> by_depth(n_nodes);      // counting sort by depth (<= 40): parents before children
  {
    uint32_t start[66] = {0};
    for (uint32_t i = 0; i < n_nodes; i++) start[depth_of[i] + 1]++;
    for (int d = 1; d < 66; d++) start[d] += start[d - 1];
    for (uint32_t i = 0; i < n_nodes; i++) by_depth[start[depth_of[i]]++] = i;
  }
  st.mark("filters, depth order");
  // Forward-delete hint (tm_tables.h): can the walk of ' '+key (the probe of go :1088-1095) end on something longer than
  // ' '+key itself?  Only then is the probe worth starting.  The hint rides in the begins-with-space bit of tokens that
  // begin with a letter (the two are mutually exclusive in any vocabulary the reference's builder writes); if a file
  // ever carries both bits on one record the hint is switched off and the kernels probe every eligible position.
  // Where ' ' + s stands in the trie follows, for EVERY node s, from where ' ' + parent(s) stands (one child look-up per node instead of a
  // walk per key): sp_node[s] = the node reached, sp_full[s] = all of s was consumed, sp_best[s] = the deepest accepting node on the way.
  const uint32_t spl_off = hv.charset == 2 ? 2u : 1u;
  uint32_t spl_start = t.root_child[' '];
  if (spl_start != kNone && spl_off == 2) spl_start = t.find(spl_start, 0u);
  std::vector<uint32_t> sp_node, sp_best;
  std::vector<uint8_t> sp_full;
  hv.spl_hint = 1;
  for (uint32_t i = 0; i < n_info; i++) if ((flags[i] & 2u) && (flags[i] & 4u)) hv.spl_hint = 0;
  if (spl_start != kNone) {
    sp_node.assign(n_nodes, kNone); sp_best.assign(n_nodes, kNone); sp_full.assign(n_nodes, 0);
    const uint32_t best0 = spl_start < n_info ? spl_start : kNone;
    for (uint32_t n : by_depth) {
      const uint32_t par = t.parent_of[n];
      const uint32_t pn = par == kRoot ? spl_start : sp_node[par], pb = par == kRoot ? best0 : sp_best[par];
      const bool pfull = par == kRoot ? true : sp_full[par] != 0;
      uint32_t c = kNone;
      if (pfull && depth_of[pn] < hv.max_len) c = t.find(pn, t.byte_of[n]);
      if (c != kNone) { sp_node[n] = c; sp_full[n] = 1; sp_best[n] = c < n_info ? c : pb; }
      else { sp_node[n] = pn; sp_full[n] = 0; sp_best[n] = pb; }
    }
  }
  auto spl_cont = [&](uint32_t i) -> uint32_t { return (sp_full[i] && has_child[sp_node[i]] && depth_of[sp_node[i]] < hv.max_len) ? 1u : 0u; };
  auto value_of = [&](uint32_t id) {
    uint32_t v = P(id) | (has_child[id] ? kHasChildren : 0);
    if (id < n_info) {
      uint32_t f5 = flag8_to_flag5(flags[id]);
      if (hv.spl_hint && (f5 & 2u)) {
        const bool hint = spl_start != kNone && (spl_cont(id) || (sp_best[id] != kNone && depth_of[sp_best[id]] > klen(id) + 1));
        f5 = (f5 & ~4u) | (hint ? 4u : 0u);
      }
      v |= ((uint32_t)nwords[id] << 22) | (f5 << 27);
    }
    return v;
  };
  st.mark("space-prefix links");
  hv.root.assign(256, kNone);
  // ---- the double array (tm_tables.h): a base for every node at depth >= 2 that has children, such that the entries base + b of
  // its children are free.  Parents are placed shallow first (the shallow end of the trie is where most walks are, and it ends up
  // together at the front of the array) by first fit from the lowest free entry; a parent that does not fit after a bounded number
  // of tries goes behind everything placed so far, and the holes it skipped are filled by the one-child parents that make up most
  // of the deeper levels: the array ends up > 95 % full.
  std::vector<uint32_t> base_of(n_nodes, 0);
  std::vector<uint8_t> chain;                   // one-child chains (tm_tables.h)
  std::vector<uint32_t> tailw(n_nodes, 0);      // chain word of a node that has a record, else 0
  uint32_t n_tail_records = 0;
  auto base_word = [&](uint32_t n) -> uint32_t { return tailw[n] ? tailw[n] : base_of[n]; };      // what an entry says about where to go on from node n
  std::vector<uint4> da;
  {
    const std::vector<uint32_t>& kid_start = t.kid_start;
    const std::vector<uint32_t>& kid = t.kid;
    size_t n_edges = 0;
    for (uint32_t n = 0; n < n_nodes; n++) if (depth_of[n] >= 2) n_edges += kid_start[n + 1] - kid_start[n];
    const uint32_t first = 256;                                  // entries below stay empty: base = entry - byte is never negative
    std::vector<uint8_t> used(n_edges + n_edges / 4 + 1024, 0);
    uint32_t cursor = first, frontier = first;                    // lowest free entry / one past the highest used one
    auto grow = [&](size_t need) { if (need > used.size()) used.resize(need + need / 4, 0); };
    for (uint32_t n : by_depth) {
      if (depth_of[n] < 2) continue;
      const uint32_t k0 = kid_start[n], k1 = kid_start[n + 1];
      if (k0 == k1) continue;
      const uint32_t b_lo = kid[k0] >> 24, span = (kid[k1 - 1] >> 24) - b_lo;
      while (used[cursor]) cursor++;
      uint32_t f = cursor, tries = 0;
      for (;;) {                                                  // f = entry of the child over the smallest byte
        if (f >= frontier || ++tries > 48) { f = std::max(f, frontier); grow((size_t)f + span + 2); break; }
        grow((size_t)f + span + 2);
        bool ok = true;
        for (uint32_t q = k0 + 1; q < k1 && ok; q++) ok = !used[f + (kid[q] >> 24) - b_lo];
        if (ok) break;
        do f++; while (used[f]);
      }
      const uint32_t base = f - b_lo;
      base_of[n] = base;
      for (uint32_t q = k0; q < k1; q++) used[base + (kid[q] >> 24)] = 1;
      frontier = std::max(frontier, f + span + 1);
    }
    hv.n_da = frontier + 256;                                    // base + 255 stays inside for every base
    // ---- one-child chains (tm_tables.h): chain[n] = bytes from n down to the first key below it if the trie is a chain of one-child
    // nodes until there, else 0.  Every node whose chain is at least kTailMin long gets a record (its first kTailMax bytes at most: a longer
    // chain goes on from the record's end node, which has a record of its own); the record's place in tab is known once the links' size is.
    chain.assign(n_nodes, 0);
    for (size_t q = by_depth.size(); q-- > 0;) {
      const uint32_t n = by_depth[q];
      if (kid_start[n + 1] - kid_start[n] != 1) continue;
      const uint32_t k = kid[kid_start[n]] & 0xFFFFFFu;
      if (k < n_info) chain[n] = 1; else if (chain[k] > 0) chain[n] = (uint8_t)(chain[k] + 1);
    }
    {
      const size_t first_rec = ((size_t)hv.n_da + 1) + kL2Size + n_nodes;      // in 16-byte entries: double array | empty entry | direct map | links | records
      // (in the order of the node ids the tables carry: under a layout by use the records of the chains that are taken most lie together)
      std::vector<uint32_t> by_id(n_nodes);
      for (uint32_t n = 0; n < n_nodes; n++) by_id[P(n)] = n;
      uint32_t nrec = 0;
      for (uint32_t n : by_id)
        if (depth_of[n] >= 2 && chain[n] >= kTailMin) tailw[n] = kTailFlag | (uint32_t)(first_rec + 3 * (size_t)nrec++);
      n_tail_records = nrec;
    }
    da.assign((size_t)hv.n_da + 1, uint4{kNone, kNone, 0u, 0u});
    for (uint32_t n = 0; n < n_nodes; n++) {
      if (depth_of[n] < 2) continue;
      for (uint32_t q = kid_start[n]; q < kid_start[n + 1]; q++) {
        const uint32_t c = kid[q] & 0xFFFFFFu;
        da[base_of[n] + (kid[q] >> 24)] = uint4{P(n), value_of(c), cmask[c], base_word(c)};
      }
    }
  }
  st.mark("double array");
  hv.idle_off = hv.n_da * 16u;
  // one allocation (tm_tables.h): double array | always-empty entry | direct map | suffix links
  const size_t direct_base = 2 * ((size_t)hv.n_da + 1);                       // in 8-byte units
  const size_t link_base = direct_base + kDirectSlots;
  hv.direct_off = (uint32_t)(direct_base * sizeof(uint2));
  hv.link_off = (uint32_t)(link_base * sizeof(uint2));
  const size_t rec_base = link_base + 2 * (size_t)n_nodes;                  // (8-byte units) chain records; two spare entries behind them (nothing reads them: room for a reader that looks ahead)
  hv.tab.assign(rec_base + 2 * (3 * (size_t)n_tail_records + 2), uint2{kNone, kNone});
  memcpy(hv.tab.data(), da.data(), da.size() * sizeof(uint4));
  std::vector<uint32_t> l2v(kL2Size, kNone), l2n(kL2Size, kNone);       // value / trie id of the depth-2 node b0b1
  for (uint32_t b0 = 0; b0 < 256; b0++) {
    const uint32_t c1 = t.root_child[b0];
    if (c1 == kNone) continue;
    hv.root[b0] = value_of(c1);
    for (uint32_t q = t.kid_start[c1]; q < t.kid_start[c1 + 1]; q++) { l2v[(b0 << 8) | (t.kid[q] >> 24)] = value_of(t.kid[q] & 0xFFFFFFu); l2n[(b0 << 8) | (t.kid[q] >> 24)] = t.kid[q] & 0xFFFFFFu; }
  }
  st.mark("table layout, depth 1 and 2");
  // suffix links (tm_tables.h): where the walk of text[p+1:] stands once the walk of text[p:] has ended on node n.
  // link(n) follows from link(parent(n)) as in Aho-Corasick, in order of depth; best[m] = deepest accepting node on the
  // path root..m.  Only links of nodes at depth >= 3 are ever read by the kernels.
  {
    std::vector<uint32_t> best(n_nodes, kNone);          // best accepting ancestor-or-self
    std::vector<uint32_t> lnode(n_nodes, kRoot);          // link target (kRoot = depth 0)
    std::vector<uint8_t> lfull(n_nodes, 0);
    auto depth_at = [&](uint32_t n) -> uint32_t { return n == kRoot ? 0u : depth_of[n]; };
    for (uint32_t n : by_depth) {
      const uint32_t par = t.parent_of[n];
      best[n] = n < n_info ? n : (par == kRoot ? kNone : best[par]);
      if (par == kRoot) { lnode[n] = kRoot; lfull[n] = 1; continue; }                 // s[1:] is empty
      const uint32_t pm = lnode[par];
      if (!lfull[par]) { lnode[n] = pm; lfull[n] = 0; continue; }                     // already fell off the trie
      const uint32_t c = t.find(pm, t.byte_of[n]);
      if (c == kNone) { lnode[n] = pm; lfull[n] = 0; } else { lnode[n] = c; lfull[n] = 1; }
    }
    uint2* lt = hv.tab.data() + link_base;
    for (uint32_t n = 0; n < n_nodes; n++) {
      const uint32_t m = lnode[n], dm = depth_at(m);
      const uint32_t hc = (m != kRoot && has_child[m]) ? 1u : 0u;
      const uint32_t b = m == kRoot ? kNone : best[m];
      const size_t at = P(n);
      lt[2 * at] = uint2{((m == kRoot ? m : P(m)) & kLinkNodeMask) | (dm << 20) | ((b != kNone ? (uint32_t)depth_of[b] : 0u) << 26), b != kNone ? value_of(b) : 0u};
      lt[2 * at + 1] = uint2{(lfull[n] && hc) ? cmask[m] : 0u, (lfull[n] && hc) ? base_word(m) : 0u};
    }
  }
  // chain records: header in link format + 32 bytes of string, for every node with a chain word
  if (n_tail_records) {
    uint4* tab16 = reinterpret_cast<uint4*>(hv.tab.data());
    for (uint32_t n = 0; n < n_nodes; n++) {
      if (!tailw[n]) continue;
      uint4* r = tab16 + tail_record(tailw[n]);
      uint8_t str[32] = {0};
      const uint32_t len = std::min<uint32_t>(chain[n], kTailMax);
      uint32_t e = n;
      for (uint32_t k = 0; k < len; k++) { const uint32_t kq = t.kid[t.kid_start[e]]; str[k] = (uint8_t)(kq >> 24); e = kq & 0xFFFFFFu; }
      const bool go_on = has_child[e] && depth_of[e] < hv.max_len;
      r[0] = uint4{(P(e) & kLinkNodeMask) | (len << 20), e < n_info ? value_of(e) : 0u, go_on ? cmask[e] : 0u, go_on ? base_word(e) : 0u};
      std::memcpy(&r[1], str, 32);
    }
  }
  if (st.on) fprintf(stderr, "  [tables] %u nodes, %u double-array entries, %u chain records, tab %.2f MB\n", n_nodes, hv.n_da, n_tail_records, hv.tab.size() * 8 / 1e6);
  st.mark("suffix links");
  // reverse table for decoding: reverse[id] = key of the LAST record carrying that id (go/tokenmonster.go:2715, quirk Q3)
  {
    std::vector<uint32_t> last(hv.n_ids, kNone);
    for (uint32_t i = 0; i < n_info; i++) last[ids[i]] = i;
    hv.rev_off.assign((size_t)hv.n_ids + 1, 0);
    hv.rev_bytes.clear();
    hv.rev_bytes.reserve(hv.keys.size());
    for (uint32_t id = 0; id < hv.n_ids; id++) {
      hv.rev_off[id] = (uint32_t)hv.rev_bytes.size();
      if (last[id] != kNone) hv.rev_bytes.insert(hv.rev_bytes.end(), hv.keys.begin() + hv.key_off[last[id]], hv.keys.begin() + hv.key_off[last[id] + 1]);
    }
    hv.rev_off[hv.n_ids] = (uint32_t)hv.rev_bytes.size();
    // ... and as the device reads it: place and length of a key in ONE word (place | length << 26), so that a decode kernel gathers one word per id
    // and not two (the gathers are what k_dec_tile_len / k_dec_gather wait for)
    if (hv.rev_bytes.size() >= (1u << kRevPlaceBits)) return set_error(TM_E_LIMIT, "the keys of the reverse table take %zu bytes, more than the device form holds", hv.rev_bytes.size());
    hv.rev_pack.assign((size_t)hv.n_ids + 1, 0);
    for (uint32_t id = 0; id < hv.n_ids; id++) hv.rev_pack[id] = hv.rev_off[id] | ((hv.rev_off[id + 1] - hv.rev_off[id]) << kRevPlaceBits);
    hv.rev_pack[hv.n_ids] = hv.rev_off[hv.n_ids];
  }
  // space-prefix links: x = node reached | continue << 21 | best accepting depth << 22, y = that node's value
  hv.vals.resize(n_info);
  for (uint32_t i = 0; i < n_info; i++) hv.vals[P(i)] = value_of(i);
  hv.spl.assign(n_info, uint4{kNone, 0u, 0u, 0u});
  if (spl_start != kNone)
    for (uint32_t i = 0; i < n_info; i++) {
      const uint32_t cont = spl_cont(i), bestn = sp_best[i];
      hv.spl[P(i)] = uint4{P(sp_node[i]) | (cont << 21) | ((bestn != kNone ? (uint32_t)depth_of[bestn] : 0u) << 22), bestn != kNone ? value_of(bestn) : 0u,
                        cont ? cmask[sp_node[i]] : 0u, cont ? base_word(sp_node[i]) : 0u};
    }
  st.mark("reverse, values, space-prefix entries");
  // direct map: one 16-byte entry (same format as a suffix link) resolves the first two bytes of any walk, the
  // depth-1 answer folded in
  for (uint32_t b0 = 0; b0 < 256; b0++) {
    const uint32_t r = hv.root[b0];
    for (uint32_t b1 = 0; b1 < 256; b1++) {
      uint32_t bestlen = 0, bestv = 0, cont = 0, id2 = 0, n2 = 0;
      const uint32_t v2 = l2v[(b0 << 8) | b1];
      if (r != kNone) {
        if (node_id(r) < n_info) { bestlen = 1; bestv = r; }
        if (v2 != kNone) {                        // node b0b1 exists
          if (node_id(v2) < n_info) { bestlen = 2; bestv = v2; }
          if (v2 & kHasChildren) { cont = 1; id2 = node_id(v2); n2 = l2n[(b0 << 8) | b1]; }
        }
      }
      uint2* e = hv.tab.data() + direct_base + 2 * (size_t)(b0 | (b1 << 8));      // indexed by the little-endian u16 at the position
      e[0] = uint2{id2 | ((cont ? 2u : 0u) << 20) | (bestlen << 26), bestv};
      e[1] = uint2{cont ? cmask[n2] : 0u, cont ? base_word(n2) : 0u};
    }
  }
  st.mark("direct map");
  hv.n_nodes = n_nodes;
  hv.off = hv.charset == 2 ? 2 : 1;
  hv.bstart = hv.root[' '];
  if (hv.off == 2 && hv.bstart != kNone) {
    // UTF-16: the virtual prefix is ' ' 0x00 (lilbufOffset 2, go :1031-1034): start from that depth-2 node
    const uint32_t v2 = l2v[(' ' << 8) | 0];
    hv.bstart = (v2 != kNone && (v2 & kHasChildren)) ? (node_id(v2) | kHasChildren) : kNone;
  }
  return TM_OK;
}

int parse_vocab(const uint8_t* f, size_t n, HostVocab& hv) {
  StageTimer st;
  int rc = parse_records(f, n, hv);
  if (rc != TM_OK) return rc;
  st.mark("records");
  Trie t;
  if ((rc = build_trie(hv, t, nullptr)) != TM_OK) return rc;
  st.mark("trie");
  return build_tables(hv, t);
}

// go/tokenmonster.go:2602-2653 (Save): the records back in the layout of SURVEY.md Appendix A
const std::vector<uint8_t>& vocab_image(const HostVocab& hv) {
  static std::mutex mu;
  std::lock_guard<std::mutex> g(mu);
  if (!hv.image.empty() || hv.key_off.empty()) return hv.image;
  std::vector<uint8_t>& o = hv.image;
  auto w24 = [&](uint32_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)(v >> 16)); };
  o.reserve(24 + hv.keys.size() + (size_t)hv.n_info * 16 + 300);
  o.push_back(hv.capcode); o.push_back(hv.charset); o.push_back(hv.norm_flag); o.push_back(hv.level); o.push_back(hv.reserve); o.push_back(0); o.push_back(0); o.push_back(0);
  w24(hv.unk); w24(hv.vocab_size); w24(hv.n_ids); w24(hv.n_info); w24(hv.delete_id);
  o.push_back((uint8_t)hv.max_len);
  for (uint32_t i = 0; i < hv.n_info; i++) {
    const uint32_t kl = hv.key_off[i + 1] - hv.key_off[i];
    o.push_back((uint8_t)kl);
    o.insert(o.end(), hv.keys.begin() + hv.key_off[i], hv.keys.begin() + hv.key_off[i + 1]);
    o.push_back(hv.rec_flag[i]); o.push_back(hv.rec_nwords[i]);
    w24(hv.rec_index1[i]); w24(hv.rec_index2[i]); w24(hv.rec_id[i]);
    uint32_t b; std::memcpy(&b, &hv.rec_score[i], 4);
    o.push_back((uint8_t)b); o.push_back((uint8_t)(b >> 8)); o.push_back((uint8_t)(b >> 16)); o.push_back((uint8_t)(b >> 24));
  }
  o.insert(o.end(), hv.begin_byte, hv.begin_byte + 256);
  w24(0);                                                      // deleted tokens: this library keeps none
  return o;
}

}  // namespace tmh

using namespace tmh;

extern "C" {

const char* tm_last_error(void) { return tmh::last_error(); }

int tm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tm_set_device(int device) {
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  return TM_OK;
}

namespace {
// The trainvocab worker loads and frees a vocabulary per candidate (trainvocab.go:530-907), several workers at once, while the GPU
// scores: hipMalloc / hipFree per table would synchronize the device under the other workers' scoring passes, and a pageable
// hipMemcpy would queue as a copy kernel behind them.  A vocabulary therefore lives in ONE device block, filled by ONE asynchronous
// copy from a pinned staging block on a copy-only stream; freed blocks of both kinds are kept for the next load (bounded).
// A parked device block may still be read by kernels that were in flight when its vocabulary was freed (the asynchronous entry points
// return before their kernels have run): it carries one event per stream the tables were used on, and the loader that takes the block
// waits for them before the copy that refills it.
struct Parked { void* p; std::vector<hipEvent_t> pending; };
struct BlockCache {
  std::mutex mu;
  std::multimap<size_t, Parked> dev_free, host_free;
  size_t dev_cached = 0, host_cached = 0;
  hipStream_t stream = nullptr;
};
constexpr size_t kDevCacheMax = 4ull << 30, kHostCacheMax = 512ull << 20;
extern "C++" BlockCache& cache_of(int device) {
  static BlockCache caches[64];
  return caches[device & 63];
}
size_t round_block(size_t bytes) { return (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1); }
void* block_get(int device, bool host, size_t bytes, size_t* got, hipError_t* err) {
  BlockCache& c = cache_of(device);
  Parked pk{nullptr, {}};
  bool found = false;
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto& fl = host ? c.host_free : c.dev_free;
    auto it = fl.lower_bound(bytes);
    if (it != fl.end() && it->first <= 2 * bytes + (4u << 20)) {
      pk = std::move(it->second); *got = it->first;
      (host ? c.host_cached : c.dev_cached) -= it->first;
      fl.erase(it);
      found = true;
    }
  }
  if (found) {
    // outside the lock: a kernel of the freed vocabulary may still be running, and the other loaders / frees of the device must not wait for it
    for (hipEvent_t ev : pk.pending) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
    if (!pk.pending.empty()) (void)hipGetLastError();
    return pk.p;
  }
  void* p = nullptr;
  *got = round_block(bytes);
  *err = host ? hipHostMalloc(&p, *got, hipHostMallocDefault) : hipMalloc(&p, *got);
  return *err == hipSuccess ? p : nullptr;
}
void block_put(int device, bool host, void* p, size_t bytes, std::vector<hipEvent_t> pending = {}) {
  if (!p) return;
  BlockCache& c = cache_of(device);
  {
    std::lock_guard<std::mutex> g(c.mu);
    size_t& cached = host ? c.host_cached : c.dev_cached;
    if (cached + bytes <= (host ? kHostCacheMax : kDevCacheMax)) {
      (host ? c.host_free : c.dev_free).emplace(bytes, Parked{p, std::move(pending)});
      cached += bytes;
      return;
    }
  }
  for (hipEvent_t ev : pending) (void)hipEventDestroy(ev);
  if (host) (void)hipHostFree(p); else (void)hipFree(p);      // (hipFree waits for the device)
}
}  // namespace

static void set_tables(tm_vocab* v) {
  const HostVocab& hv = v->host;
  Tables& t = v->tables;
  t.root = v->d_root; t.tab = v->d_tab; t.spl = v->d_spl; t.vals = v->d_vals; t.rows = v->d_rows; t.begin_byte = v->d_begin_byte;
  t.idle_off = hv.idle_off; t.n_da = hv.n_da; t.n_info = hv.n_info; t.max_len = hv.max_len;
  t.off = hv.off; t.bstart = hv.bstart; t.spl_hint = hv.spl_hint; t.link_off = hv.link_off; t.direct_off = hv.direct_off;
  t.has_delete = hv.delete_id != TM_NONE; t.delete_id = hv.delete_id; t.unk_id = hv.unk;
  t.last_rec = (uint32_t)(v->part_bytes[1] / 16 >= 3 ? v->part_bytes[1] / 16 - 3 : 0);
}

// "current device" is a property of the calling OS thread; a caller whose threads are not its own (a goroutine under cgo) names the device
int tm_vocab_load_on(const uint8_t* vocab_file, size_t n, int device, tm_vocab** out) {
  const int rc = tm_set_device(device);
  return rc == TM_OK ? tm_vocab_load(vocab_file, n, out) : rc;
}

static int upload_tables(tm_vocab* v);
static int reupload_tables(tm_vocab* v);

int tm_vocab_load(const uint8_t* vocab_file, size_t n, tm_vocab** out) {
  if (!vocab_file || !out) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  auto* v = new tm_vocab();
  int rc = parse_vocab(vocab_file, n, v->host);
  if (rc != TM_OK) { delete v; return rc; }
  v->host.image.assign(vocab_file, vocab_file + n);
  if ((rc = upload_tables(v)) != TM_OK) return rc;       // (frees v on failure)
  *out = v;
  return TM_OK;
}

// A candidate of the trainvocab loop (training/trainvocab.go:530-907 builds the tables of every candidate in place): token list ->
// records + trie (tm_build.cpp) -> tables -> device, without the .vocab image in between that tm_build_vocab + tm_vocab_load write and
// parse again.  Same tables, bit for bit, as loading the image tm_build_vocab makes of the same list (tests/test_builder_normalizer.py).
int tm_vocab_build(const uint8_t* blob, const uint32_t* off, uint32_t n_tokens, const uint8_t* special, uint32_t capcode, uint32_t charset, uint32_t norm_flag,
                   uint32_t level, int with_unk, int device, tm_vocab** out) {
  if (!out || (n_tokens && (!blob || !off))) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  { const int rc = tm_set_device(device); if (rc != TM_OK) return rc; }
  std::vector<std::string> toks(n_tokens);
  std::vector<uint8_t> sp(special ? n_tokens : 0, 0);
  for (uint32_t k = 0; k < n_tokens; k++) {
    toks[k].assign((const char*)blob + off[k], off[k + 1] - off[k]);
    if (special) sp[k] = special[k];
  }
  auto* v = new tm_vocab();
  Trie trie;
  int rc = build_vocab_records(toks, sp, capcode, charset, norm_flag, level, with_unk != 0, v->host, trie);
  if (rc == TM_OK) rc = build_tables(v->host, trie);
  if (rc != TM_OK) { delete v; return rc; }
  if ((rc = upload_tables(v)) != TM_OK) return rc;
  *out = v;
  return TM_OK;
}

struct Part { void** dst; const void* src; size_t bytes, at; };
static void table_parts(tm_vocab* v, Part (&parts)[8]) {
  HostVocab& hv = v->host;
  const Part p[8] = {{(void**)&v->d_root, hv.root.data(), 256 * 4, 0}, {(void**)&v->d_tab, hv.tab.data(), hv.tab.size() * sizeof(uint2), 0},
                     {(void**)&v->d_rows, hv.rows.data(), hv.rows.size() * sizeof(Row), 0}, {(void**)&v->d_spl, hv.spl.data(), hv.spl.size() * sizeof(uint4), 0},
                     {(void**)&v->d_vals, hv.vals.data(), hv.vals.size() * 4, 0}, {(void**)&v->d_rev_off, hv.rev_pack.data(), hv.rev_pack.size() * 4, 0},
                     {(void**)&v->d_rev_bytes, hv.rev_bytes.data(), hv.rev_bytes.size(), 0}, {(void**)&v->d_begin_byte, hv.begin_byte, 256, 0}};
  for (int k = 0; k < 8; k++) parts[k] = p[k];
}

// the tables of v->host once more into the block they already lie in (tm_vocab_tune: same sizes, other contents)
static int reupload_tables(tm_vocab* v) {
  Part parts[8];
  table_parts(v, parts);
  size_t total = 0;
  for (int k = 0; k < 8; k++) {
    if (parts[k].bytes != v->part_bytes[k]) return set_error(TM_E_INTERNAL, "table %d changed size", k);
    parts[k].at = total; total += (parts[k].bytes + 255) & ~(size_t)255;
  }
  total += 256;
  hipError_t e = hipSuccess;
  size_t stage_bytes = 0;
  void* stage = block_get(v->device, true, total, &stage_bytes, &e);
  if (!stage) return hip_fail(e, "vocabulary staging");
  for (Part& q : parts) if (q.bytes) std::memcpy((uint8_t*)stage + q.at, q.src, q.bytes);
  BlockCache& c = cache_of(v->device);
  {
    std::lock_guard<std::mutex> g(c.mu);
    if (!c.stream) e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(v->d_block, stage, total, hipMemcpyHostToDevice, c.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  block_put(v->device, true, stage, stage_bytes);
  if (e != hipSuccess) return hip_fail(e, "vocabulary upload");
  set_tables(v);
  return TM_OK;
}

// the tables of v->host into ONE device block on the current device; deletes v on failure
static int upload_tables(tm_vocab* v) {
  hipError_t e;
  int dev = 0;
  if ((e = hipGetDevice(&dev)) != hipSuccess) { delete v; return hip_fail(e, "hipGetDevice"); }
  v->device = dev;
  Part parts[8];
  table_parts(v, parts);
  size_t total = 0;
  for (int k = 0; k < 8; k++) { Part& q = parts[k]; q.at = total; total += (q.bytes + 255) & ~(size_t)255; v->device_bytes += q.bytes; v->part_bytes[k] = q.bytes; }
  total += 256;
  e = hipSuccess;
  size_t stage_bytes = 0;
  void* stage = block_get(dev, true, total, &stage_bytes, &e);
  v->d_block = stage ? block_get(dev, false, total, &v->block_bytes, &e) : nullptr;
  if (v->d_block) {
    for (Part& q : parts) {
      if (q.bytes) std::memcpy((uint8_t*)stage + q.at, q.src, q.bytes);
      *q.dst = (uint8_t*)v->d_block + q.at;
    }
    BlockCache& c = cache_of(dev);
    {
      std::lock_guard<std::mutex> g(c.mu);
      if (!c.stream) e = hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(v->d_block, stage, total, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
  }
  block_put(dev, true, stage, stage_bytes);
  if (!v->d_block || e != hipSuccess) {
    tm_vocab_free(v);
    return hip_fail(e, "vocabulary upload");
  }
  set_tables(v);
  return TM_OK;
}

// ---- the device block of a vocabulary handed from process to process -------------------------------------------------------------
// In the data-parallel scoring mode (every rank scores its byte range of the dataset against the SAME candidate) only one
// rank has to turn a candidate's token list into tables (tm_build_vocab + tm_vocab_load: ~50 ms of one host thread); the others take the
// finished block - a few MB, one broadcast over xGMI - and the ~200 bytes that say what lies where in it.
extern "C++" void tmh_vocab_quiesce(tm_vocab* v);
// ---- tables laid out by use (tm_vocab_tune) ---------------------------------------------------------------------------------------
// How often the match kernel would gather the entry of every node - its suffix link when a walk has ended on it, its row / space-prefix
// link when it is the longest match at a position - on `text`: the kernel's step A1 replayed on the host over the tables as they are
// (direct map on two bytes, suffix links, double-array probes behind the child filters).  use[] is indexed by the CURRENT node ids.
static void count_node_use(const HostVocab& hv, const uint8_t* text, uint64_t n, std::vector<uint64_t>& use) {
  use.assign(hv.n_nodes, 0);
  const uint2* tab = hv.tab.data();
  const uint4* da = reinterpret_cast<const uint4*>(tab);
  const size_t direct16 = hv.direct_off / 16, link16 = hv.link_off / 16;
  const int Lmax = (int)hv.max_len;
  auto at = [&](uint64_t i) -> uint32_t { return i < n ? text[i] : 0u; };
  int depth = 0;
  uint32_t node = 0;
  for (uint64_t pos = 0; pos + 1 < n; pos++) {
    const int limit = (int)std::min<uint64_t>(n - pos, (uint64_t)Lmax);
    const uint2* e;
    if (pos > 0 && depth >= 3) { use[node]++; e = tab + 2 * (link16 + node); }
    else e = tab + 2 * (direct16 + (at(pos) | (at(pos + 1) << 8)));
    const uint32_t src = e[0].x;
    uint32_t filt = e[1].x, bestv = e[0].y, base = e[1].y;
    depth = (int)link_depth(src); node = link_node(src);
    while (depth < limit) {
      const uint32_t c = at(pos + depth);
      if (!((filt >> (c & 31u)) & 1u)) break;
      if (is_tail_word(base)) {                                   // a one-child chain (tm_tables.h): the whole chain or nothing
        const uint4* r = da + tail_record(base);
        const int len = (int)tail_len(r[0].x);
        const uint8_t* str = reinterpret_cast<const uint8_t*>(r + 1);
        bool same = depth + len <= limit;
        for (int k = 0; k < len && same; k++) same = at(pos + depth + k) == str[k];
        if (!same) break;
        depth += len; node = link_node(r[0].x);
        if (r[0].y != 0) bestv = r[0].y;
        filt = r[0].z; base = r[0].w;
        if (filt == 0) break;
        continue;
      }
      const uint4 d = da[base + c];
      if (d.x != node) break;
      depth++; node = node_id(d.y);
      if (node < hv.n_info) bestv = d.y;
      filt = d.z; base = d.w;
      if (!(d.y & kHasChildren)) break;
    }
    if (bestv != 0 && node_id(bestv) < hv.n_info) use[node_id(bestv)]++;
  }
}

// the host tables of hv (in the trie's own numbering) once more, laid out by their use on `sample`
static int layout_by_use(HostVocab& hv, const Trie& trie, const uint8_t* sample, uint64_t n) {
  std::vector<uint64_t> use;
  count_node_use(hv, sample, n, use);
  // accepting nodes among themselves (they index the rows and space-prefix links too), the others among themselves; ties keep their order
  std::vector<uint32_t> order(hv.n_nodes), perm(hv.n_nodes);
  for (uint32_t i = 0; i < hv.n_nodes; i++) order[i] = i;
  std::stable_sort(order.begin(), order.begin() + hv.n_info, [&](uint32_t a, uint32_t b) { return use[a] > use[b]; });
  std::stable_sort(order.begin() + hv.n_info, order.end(), [&](uint32_t a, uint32_t b) { return use[a] > use[b]; });
  for (uint32_t i = 0; i < hv.n_nodes; i++) perm[order[i]] = i;
  const uint32_t n_da = hv.n_da;
  int rc = build_tables(hv, trie, &perm);
  if (rc == TM_OK && hv.n_da != n_da) rc = set_error(TM_E_INTERNAL, "the double array changed size under a renumbering of the nodes");
  return rc;
}

extern "C" int tm_vocab_tune(tm_vocab* v, const uint8_t* normalized_sample, uint64_t n) {
  if (!v || (n && !normalized_sample)) return set_error(TM_E_INVALID, "null argument");
  HostVocab& hv = v->host;
  if (hv.key_off.empty() || hv.rec_id.empty()) return set_error(TM_E_INVALID, "a vocabulary made from an imported block has no records to lay out again");
  { int rc = enter_device(v); if (rc != TM_OK) return rc; }
  // this vocabulary's kernels first (as tm_vocab_free does): the tables change under them otherwise
  tmh_vocab_quiesce(v);
  Trie trie;
  int rc = build_trie(hv, trie, nullptr);
  if (rc == TM_OK && v->tuned) rc = build_tables(hv, trie);            // count over the trie's own numbering
  if (rc != TM_OK) return rc;
  // The host tables are rewritten in place; whatever goes wrong from here on, host and device must describe the SAME layout when the call
  // returns (a later tm_vocab_block_export, a replica adopting the block or a second tune replays the host's view): on a failure the
  // natural layout is rebuilt and uploaded, and the error of the failed step is what the caller sees.
  rc = layout_by_use(hv, trie, normalized_sample, n);
  if (rc == TM_OK) rc = reupload_tables(v);
  if (rc == TM_OK) { v->tuned = true; return TM_OK; }
  const std::string keep = last_error();
  if (build_tables(hv, trie) == TM_OK && reupload_tables(v) == TM_OK) v->tuned = false;
  return set_error(rc, "%s", keep.c_str());
}

// tm_vocab_load with the layout by use from the start: the tables are written for the device once, in the order the sample asks for
extern "C" int tm_vocab_load_sample(const uint8_t* vocab_file, size_t n, const uint8_t* normalized_sample, uint64_t sample_n, tm_vocab** out) {
  if (!vocab_file || !out || (sample_n && !normalized_sample)) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  auto* v = new tm_vocab();
  int rc = parse_vocab(vocab_file, n, v->host);
  if (rc == TM_OK && sample_n) {
    Trie trie;
    rc = build_trie(v->host, trie, nullptr);
    if (rc == TM_OK) rc = layout_by_use(v->host, trie, normalized_sample, sample_n);
    if (rc == TM_OK) v->tuned = true;
  }
  if (rc != TM_OK) { delete v; return rc; }
  v->host.image.assign(vocab_file, vocab_file + n);
  if ((rc = upload_tables(v)) != TM_OK) return rc;       // (frees v on failure)
  *out = v;
  return TM_OK;
}

int tm_vocab_block_export(const tm_vocab* v, tm_vocab_block* m, void** device_ptr) {
  if (!v || !m) return set_error(TM_E_INVALID, "null argument");
  const HostVocab& hv = v->host;
  std::memset(m, 0, sizeof(*m));
  m->bytes = 256;                  // what the tables occupy (the parked block they lie in may be larger: that is nobody else's business, and not worth sending)
  for (int k = 0; k < 8; k++) { m->part_bytes[k] = v->part_bytes[k]; m->bytes += (v->part_bytes[k] + 255) & ~(uint64_t)255; }
  m->idle_off = hv.idle_off; m->n_da = hv.n_da; m->n_info = hv.n_info; m->max_len = hv.max_len; m->off = hv.off; m->bstart = hv.bstart;
  m->spl_hint = hv.spl_hint; m->link_off = hv.link_off; m->direct_off = hv.direct_off; m->delete_id = hv.delete_id; m->unk_id = hv.unk;
  m->n_ids = hv.n_ids; m->vocab_size = hv.vocab_size; m->capcode = hv.capcode; m->charset = hv.charset; m->norm_flag = hv.norm_flag; m->level = hv.level;
  m->reserve = hv.reserve; m->n_nodes = hv.n_nodes; m->pad = TM_VOCAB_BLOCK_FORMAT;
  if (device_ptr) *device_ptr = v->d_block;
  return TM_OK;
}

int tm_device_copy(void* dst_device, const void* src_device, uint64_t bytes) {
  if (bytes && (!dst_device || !src_device)) return set_error(TM_E_INVALID, "null argument");
  const hipError_t e = bytes ? hipMemcpy(dst_device, src_device, bytes, hipMemcpyDefault) : hipSuccess;      // (unified addressing: either side may also be page-locked host memory)
  return e == hipSuccess ? TM_OK : hip_fail(e, "device-to-device copy");
}

static void adopt_scalars(HostVocab& hv, const tm_vocab_block* m) {
  hv.idle_off = m->idle_off; hv.n_da = m->n_da; hv.n_info = m->n_info; hv.max_len = m->max_len; hv.off = m->off; hv.bstart = m->bstart;
  hv.spl_hint = m->spl_hint; hv.link_off = m->link_off; hv.direct_off = m->direct_off; hv.delete_id = m->delete_id; hv.unk = m->unk_id;
  hv.n_ids = m->n_ids; hv.vocab_size = m->vocab_size; hv.capcode = (uint8_t)m->capcode; hv.charset = (uint8_t)m->charset; hv.norm_flag = (uint8_t)m->norm_flag;
  hv.level = (uint8_t)m->level; hv.reserve = (uint8_t)m->reserve; hv.n_nodes = m->n_nodes;
}

// for a replica whose block is being overwritten with the exporter's retuned one (tm_vocab_set_tune): wait for the replica's own kernels,
// and afterwards take over the scalars that ride in the description (node values such as bstart move with a renumbering)
extern "C++" void tmh_vocab_quiesce(tm_vocab* v) {
  std::vector<hipEvent_t> pending;
  { std::lock_guard<std::mutex> g(v->use_mu); for (auto& u : v->last_use) pending.push_back(u.second); v->last_use.clear(); }
  for (hipEvent_t ev : pending) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
  (void)hipGetLastError();
}
extern "C++" int tmh_vocab_adopt(tm_vocab* v, const tm_vocab_block* m) {
  for (int k = 0; k < 8; k++) if (m->part_bytes[k] != v->part_bytes[k]) return set_error(TM_E_INVALID, "vocabulary block of another shape");
  adopt_scalars(v->host, m);
  set_tables(v);
  return TM_OK;
}

int tm_vocab_block_import(const tm_vocab_block* m, int device, tm_vocab** out, void** device_ptr) {
  if (!m || !out || !device_ptr) return set_error(TM_E_INVALID, "null argument");
  *out = nullptr;
  size_t total = 256;
  for (int k = 0; k < 8; k++) total += (m->part_bytes[k] + 255) & ~(uint64_t)255;
  if (m->pad != TM_VOCAB_BLOCK_FORMAT) return set_error(TM_E_INVALID, "vocabulary block of table format %u, this library reads format %u (exporter and importer are different builds)", m->pad, (unsigned)TM_VOCAB_BLOCK_FORMAT);
  // every offset the kernels add to a table pointer is checked against the part it indexes: a stale or foreign description must not make K1 gather outside the block
  const uint64_t tab = m->part_bytes[1];
  if (total > m->bytes || m->part_bytes[0] != 256 * 4 || m->part_bytes[7] != 256 || m->n_ids > kRowIdMask + 1 || m->n_info >= kMaxNodes || m->n_nodes < m->n_info ||
      m->n_nodes >= kMaxNodes || m->max_len > 40 || (m->off != 1 && m->off != 2) || m->idle_off != (uint64_t)m->n_da * 16 || ((uint64_t)m->n_da + 1) * 16 != m->direct_off ||
      (uint64_t)m->direct_off + (uint64_t)kDirectSlots * sizeof(uint2) != m->link_off || (uint64_t)m->link_off + 16ull * m->n_nodes > tab || (tab - ((uint64_t)m->link_off + 16ull * m->n_nodes)) % 48 != 32 ||        /* the chain records behind the links: three entries each + two spare */
      m->part_bytes[2] != 16ull * m->n_info || m->part_bytes[3] != 16ull * m->n_info || m->part_bytes[4] != 4ull * m->n_info ||
      m->part_bytes[5] != 4ull * ((uint64_t)m->n_ids + 1) || (m->delete_id != TM_NONE && m->delete_id >= m->n_ids) || (m->unk_id != TM_NONE && m->unk_id >= m->n_ids))
    return set_error(TM_E_INVALID, "vocabulary block description is inconsistent");
  { int rc = tm_set_device(device); if (rc != TM_OK) return rc; }
  auto* v = new tm_vocab();
  v->device = device;
  adopt_scalars(v->host, m);               // (scalars only: an imported vocabulary has no host tables - no Save, no host-side decode)
  hipError_t e = hipSuccess;
  v->d_block = block_get(device, false, (size_t)m->bytes, &v->block_bytes, &e);
  if (!v->d_block) { delete v; return hip_fail(e, "vocabulary block"); }
  void** dst[8] = {(void**)&v->d_root, (void**)&v->d_tab, (void**)&v->d_rows, (void**)&v->d_spl, (void**)&v->d_vals, (void**)&v->d_rev_off, (void**)&v->d_rev_bytes,
                   (void**)&v->d_begin_byte};
  size_t at = 0;
  for (int k = 0; k < 8; k++) { *dst[k] = (uint8_t*)v->d_block + at; v->part_bytes[k] = m->part_bytes[k]; v->device_bytes += m->part_bytes[k]; at += (m->part_bytes[k] + 255) & ~(uint64_t)255; }
  set_tables(v);
  *out = v;
  *device_ptr = v->d_block;      // the caller fills bytes [0, m->bytes) - e.g. the destination of an RCCL broadcast - before the first use
  return TM_OK;
}

void tm_vocab_free(tm_vocab* v) {
  if (!v) return;
  { int cur = -1; (void)hipGetDevice(&cur); if (cur != v->device) (void)hipSetDevice(v->device); }
  // The last table-reading kernel of every stream is waited for HERE, while the streams it may have run on - the lanes of this vocabulary's
  // pool, destroyed two lines below - still exist: a free waits for the vocabulary's own work (not for the device, as hipFree would) and the
  // block is parked with nothing pending.  (Until round 4 the events travelled with the parked block and the next load waited for them: by
  // then their streams were gone, and this runtime answers hipEventSynchronize on such an event with whatever the freed stream object holds
  // - "operation not permitted when stream is capturing" among others -, which the next launch check then took for its own error.)
  std::vector<hipEvent_t> pending;
  {
    std::lock_guard<std::mutex> g(v->use_mu);
    for (auto& u : v->last_use) pending.push_back(u.second);
    v->last_use.clear();
  }
  for (hipEvent_t ev : pending) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
  pending.clear();
  (void)hipGetLastError();             // (an event of a caller's stream that the caller has destroyed already: nothing to wait for)
  tmh::pool_destroy(v->pool);
  if (v->d_block) block_put(v->device, false, v->d_block, v->block_bytes, std::move(pending));
  else for (hipEvent_t ev : pending) (void)hipEventDestroy(ev);
  delete v;
}

uint32_t tm_vocab_size(const tm_vocab* v) { return v->host.vocab_size; }
uint32_t tm_vocab_n_info(const tm_vocab* v) { return v->host.n_info; }
uint32_t tm_vocab_n_ids(const tm_vocab* v) { return v->host.n_ids; }
uint32_t tm_vocab_max_token_length(const tm_vocab* v) { return v->host.max_len; }
uint32_t tm_vocab_capcode(const tm_vocab* v) { return v->host.capcode; }
uint32_t tm_vocab_charset(const tm_vocab* v) { return v->host.charset; }
uint32_t tm_vocab_normalization(const tm_vocab* v) { return v->host.norm_flag; }
uint32_t tm_vocab_unk(const tm_vocab* v) { return v->host.unk; }
uint32_t tm_vocab_delete_token(const tm_vocab* v) { return v->host.delete_id; }
uint64_t tm_vocab_device_bytes(const tm_vocab* v) { return v->device_bytes; }

}  // extern "C"
