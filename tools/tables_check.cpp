// tables_check.cpp — CPU check of the walk tables tm_vocab_load uploads (tokenmonster_amd/csrc/tm_tables.h), without a GPU:
//   1. structural invariants the kernels rely on (bucket fill order, every key reachable by the kernel's probe sequence, child
//      filters are supersets of the real child sets and empty exactly when there are no children, "no match => value 0" in the
//      link-format entries, which the unconditional descriptor store of k_match_branch depends on);
//   2. the walk of k_match_branch step A1 — direct map for the first position of a run, suffix links afterwards, probes only
//      where the child filter allows — replayed on synthetic text against a brute-force longest-prefix search over the keys of
//      the .vocab file: same (length, record ordinal) at every position (pansearch LongestSubstring semantics,
//      tokenmonster-cpp/src/tokenmonster.cpp:786-877).
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/tables_check.cpp -o /tmp/tables_check \
//         -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
// exit code 0 = all checks passed
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_device.h"

using namespace tmh;

namespace {
uint64_t g_bad = 0;
void fail(const char* what, uint64_t a = 0, uint64_t b = 0) {
  if (++g_bad <= 10) fprintf(stderr, "FAIL: %s (%llu, %llu)\n", what, (unsigned long long)a, (unsigned long long)b);
}

bool check_vocab(uint32_t kind, uint32_t vsize, uint32_t capcode, uint64_t seed, uint64_t text_bytes) {
  const uint64_t bad0 = g_bad;
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(kind, vsize, capcode, 1, 3, seed, 0, &img, &img_n) != 0) { fail("tm_synth_vocab"); return false; }
  HostVocab hv;
  if (parse_vocab(img, img_n, hv) != 0) { fail("parse_vocab"); tm_free(img); return false; }
  const uint2* tab = hv.tab.data();
  const uint2* direct = tab + hv.direct_off / 8;
  const uint2* link = tab + hv.link_off / 8;
  const uint32_t nb = hv.edge_mask + 1;

  // ---- 1a. buckets: slot 0 fills first; the bucket behind the table is empty; every key is where the probe sequence finds it
  std::vector<uint32_t> children(hv.n_nodes, 0);     // real child-byte sets, folded to 32 bits (bit b & 31)
  uint64_t nkeys = 0;
  for (uint32_t b = 0; b <= nb; b++) {
    const uint2 s0 = tab[2 * (size_t)b], s1 = tab[2 * (size_t)b + 1];
    if (b == nb) { if (s0.x != kNone || s1.x != kNone) fail("the bucket behind the table is not empty"); break; }
    if (s0.x == kNone && s1.x != kNone) fail("slot 1 of a bucket filled before slot 0", b);
    for (int q = 0; q < 2; q++) {
      const uint2 s = q ? s1 : s0;
      if (s.x == kNone) continue;
      nkeys++;
      const uint32_t key = s.x & kKeyMask, parent = key >> 8, byte = key & 0xFF;
      if (parent >= hv.n_nodes) { fail("parent node out of range", parent); continue; }
      children[parent] |= 1u << (byte & 31u);
      uint32_t h = edge_hash(parent, byte) >> hv.edge_shift;
      bool found = false;
      for (uint32_t step = 0; step <= nb; step++) {
        const uint2 t0 = tab[2 * (size_t)h], t1 = tab[2 * (size_t)h + 1];
        if ((t0.x & kKeyMask) == key || (t1.x & kKeyMask) == key) { found = h == b; break; }
        if (t1.x == kNone) break;                     // the kernel stops here: key reported absent
        h = (h + 1) & hv.edge_mask;
      }
      if (!found) fail("a key is not reachable by the probe sequence", key, b);
    }
  }
  // ---- 1b. child filters: 4 bits in every slot (of the node the edge leads to), 32 bits in every link-format entry
  auto fold4 = [](uint32_t m) { uint32_t f = 0; for (uint32_t q = 0; q < 32; q++) if ((m >> q) & 1u) f |= 1u << (q & 3u); return f; };
  for (uint32_t b = 0; b < nb; b++)
    for (int q = 0; q < 2; q++) {
      const uint2 s = tab[2 * (size_t)b + q];
      if (s.x == kNone) continue;
      const uint32_t child = node_id(s.y);
      if (child >= hv.n_nodes) { fail("child node out of range", child); continue; }
      if ((s.x >> 28) != fold4(children[child])) fail("4-bit child filter of a slot", s.x >> 28, fold4(children[child]));
      if (((s.y & kHasChildren) != 0) != (children[child] != 0)) fail("has-children bit of a node value", child);
    }
  auto check_link_format = [&](const uint2* e, const char* what) {
    const uint32_t x = e[0].x, y = e[0].y, filt = e[1].x, bestlen = e[1].y;
    const bool go = (x >> 21) & 1u;
    if (bestlen == 0 && y != 0) fail(what, 1, y);                                   // k_match_branch stores the descriptor unconditionally
    if (bestlen != 0 && (node_id(y) >= hv.n_info || bestlen > 40)) fail(what, 2, bestlen);
    if (go != (filt != 0)) fail(what, 3, filt);
    // (links that lead to a node of depth < 2 belong to nodes of depth < 3 and are never read: the walk takes the direct map there)
    if (go && ((x >> 23) & 63u) >= 2 && filt != children[x & kNodeMask]) fail(what, 4, filt);
  };
  for (uint32_t i = 0; i < kL2Size; i++) check_link_format(direct + 2 * (size_t)i, "direct map entry");
  for (uint32_t n = 0; n < hv.n_nodes; n++) check_link_format(link + 2 * (size_t)n, "suffix link entry");

  // ---- 2. the A1 walk against brute force ------------------------------------------------------------------------------
  std::unordered_map<std::string, uint32_t> keys;
  keys.reserve(hv.n_info * 2);
  for (uint32_t i = 0; i < hv.n_info; i++) keys.emplace(std::string((const char*)&hv.keys[hv.key_off[i]], hv.key_off[i + 1] - hv.key_off[i]), i);
  std::vector<uint8_t> raw(text_bytes + 70000);
  std::vector<uint64_t> roff(text_bytes / 64 + 17);
  uint32_t nd = 0; uint64_t nbytes = 0;
  tm_synth_corpus(kind, seed + 77, text_bytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nbytes);
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), nd, capcode, 1, 0, &text, off.data()) != 0) { fail("tm_normalize_batch"); tm_free(img); return false; }
  const int Lmax = (int)hv.max_len;
  uint64_t npos = 0, gathers = 0;
  for (uint32_t d = 0; d < nd; d++) {
    const uint8_t* t = text + off[d];
    const int dl = (int)(off[d + 1] - off[d]);
    auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
    int depth = 0; uint32_t node = 0;
    for (int pos = 0; pos + 1 < dl; pos++) {                       // (the last byte of a document is looked up in root[], not walked)
      const bool first = (pos % 5) == 0;                           // a lane's run: 5 positions, as with 256-byte segments
      const int limit = std::min(dl - pos, Lmax);
      const uint2* e = (!first && depth >= 3) ? link + 2 * (size_t)node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
      gathers++;
      uint32_t bestv = e[0].y; int bestlen = (int)e[1].y;
      depth = (int)((e[0].x >> 23) & 63u);
      node = e[0].x & kNodeMask;
      bool go = depth < limit && ((e[1].x >> (at(pos + depth) & 31u)) & 1u);
      while (go) {
        const uint32_t c = at(pos + depth), key = (node << 8) | c;
        uint32_t h = edge_hash(node, c) >> hv.edge_shift;
        uint2 hit{kNone, 0};
        for (;;) {
          gathers++;
          const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1];
          if ((s0.x & kKeyMask) == key) { hit = s0; break; }
          if ((s1.x & kKeyMask) == key) { hit = s1; break; }
          if (s1.x == kNone) break;
          h = (h + 1) & hv.edge_mask;
        }
        if (hit.x == kNone) break;
        depth++;
        node = node_id(hit.y);
        if (node < hv.n_info) { bestv = hit.y; bestlen = depth; }
        go = depth < limit && (((hit.x >> 28) >> (at(pos + depth) & 3u)) & 1u);
      }
      // brute force: the longest prefix of text[pos : pos + limit] that is a key
      int exp_len = 0; uint32_t exp_id = 0;
      for (int l = limit; l >= 1; l--) {
        auto it = keys.find(std::string((const char*)t + pos, (size_t)l));
        if (it != keys.end()) { exp_len = l; exp_id = it->second; break; }
      }
      if (bestlen != exp_len || (exp_len && node_id(bestv) != exp_id)) fail("longest match differs from brute force", off[d] + pos, (uint64_t)bestlen << 32 | (uint32_t)exp_len);
      npos++;
    }
  }
  printf("kind %u, %u ids, capcode %u: %u records, %u nodes, %llu edges in %u buckets; %llu positions walked, %.2f gathers each: %s\n", kind, vsize, capcode,
         hv.n_info, hv.n_nodes, (unsigned long long)nkeys, nb, (unsigned long long)npos, (double)gathers / (double)std::max<uint64_t>(npos, 1), g_bad == bad0 ? "ok" : "FAILED");
  tm_free(text); tm_free(img);
  return g_bad == bad0;
}
}  // namespace

int main(int argc, char** argv) {
  const uint64_t text_bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1u << 20);
  check_vocab(TM_KIND_ENGLISHCODE, 6000, 2, 0x544D0002, text_bytes);
  check_vocab(TM_KIND_ENGLISH, 3000, 2, 0x544D0001, text_bytes);
  check_vocab(TM_KIND_CODE, 4096, 0, 0x544D0004, text_bytes);
  printf("%llu failures\n", (unsigned long long)g_bad);
  return g_bad ? 1 : 0;
}
