// tables_check.cpp — CPU check of the walk tables tm_vocab_load uploads (tokenmonster_amd/csrc/tm_tables.h), without a GPU:
//   1. structural invariants the kernels rely on (every double-array entry lies at base(parent) + byte with base(parent) + 255 inside
//      the array, the entry behind the array is empty, child filters are exactly the real child sets folded to 32 bits and empty
//      when there are no children, "no match => value 0" in the link-format entries, which the unconditional descriptor store of
//      k_match_branch depends on);
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
  const uint4* da = reinterpret_cast<const uint4*>(tab);
  const uint32_t nb = hv.n_da;

  // ---- 1a. the double array: an entry's parent is known by its check word; base and byte follow from where the parent's other
  // children and the entries that lead to the parent say they are (checked through the walk below and the bases collected here)
  std::vector<uint32_t> children(hv.n_nodes, 0);     // real child-byte sets, folded to 32 bits (bit b & 31)
  std::vector<uint32_t> base_seen(hv.n_nodes, kNone), base_said(hv.n_nodes, kNone);
  uint64_t nkeys = 0;
  if (hv.idle_off != nb * 16u || da[nb].x != kNone) fail("the entry behind the array is not empty");
  for (uint32_t t = 0; t < nb; t++) {
    const uint4 e = da[t];
    if (e.x == kNone) continue;
    nkeys++;
    const uint32_t parent = e.x, child = node_id(e.y);
    if (parent >= hv.n_nodes || child >= hv.n_nodes) { fail("node out of range", parent, child); continue; }
    if (is_tail_word(e.w)) continue;                            // (a chain word instead of a base: checked through the walk below)
    base_said[child] = e.w;
    if (e.z != 0 && e.w + 255 >= nb + 1) fail("base + 255 leaves the array", e.w);
  }
  // every parent's entries share one base: entry index - byte, the byte being the last key byte of the child (accepting children)
  for (uint32_t t = 0; t < nb; t++) {
    const uint4 e = da[t];
    if (e.x == kNone || e.x >= hv.n_nodes) continue;
    const uint32_t child = node_id(e.y);
    if (child < hv.n_info) {
      const uint32_t byte = hv.keys[hv.key_off[child + 1] - 1];
      if (t < byte) { fail("entry below its byte", t, byte); continue; }
      if (base_seen[e.x] == kNone) base_seen[e.x] = t - byte; else if (base_seen[e.x] != t - byte) fail("two bases for one parent", e.x);
      children[e.x] |= 1u << (byte & 31u);
    }
  }
  for (uint32_t n = 0; n < hv.n_nodes; n++) if (base_seen[n] != kNone && base_said[n] != kNone && base_seen[n] != base_said[n]) fail("the base an entry gives for its node is not where the children are", n);
  // ---- 1b. child filters (of the node the entry leads to): supersets of the accepting children seen above; has-children bit
  for (uint32_t t = 0; t < nb; t++) {
    const uint4 e = da[t];
    if (e.x == kNone) continue;
    const uint32_t child = node_id(e.y);
    if (child >= hv.n_nodes) continue;
    if ((e.z & children[child]) != children[child]) fail("child filter of an entry misses a child", e.z, children[child]);
    if (((e.y & kHasChildren) != 0) != (e.z != 0)) fail("has-children bit of a node value", child);
  }
  auto check_link_format = [&](const uint2* e, const char* what) {
    const uint32_t x = e[0].x, y = e[0].y, filt = e[1].x, bestlen = link_bestlen(x);
    if (bestlen == 0 && y != 0) fail(what, 1, y);                                   // k_match_branch stores the descriptor unconditionally
    if (bestlen != 0 && (node_id(y) >= hv.n_info || bestlen > 40)) fail(what, 2, bestlen);
    // (links that lead to a node of depth < 2 belong to nodes of depth < 3 and are never read: the walk takes the direct map there)
    if (filt != 0 && link_depth(x) >= 2 && ((filt & children[link_node(x)]) != children[link_node(x)] || !is_tail_word(e[1].y) && e[1].y != base_said[link_node(x)] && base_said[link_node(x)] != kNone)) fail(what, 4, filt);
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
  uint64_t npos = 0, gathers = 0, ntail = 0;
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
      uint32_t bestv = e[0].y; int bestlen = (int)link_bestlen(e[0].x);
      depth = (int)link_depth(e[0].x);
      node = link_node(e[0].x);
      uint32_t base = e[1].y;
      bool go = depth < limit && ((e[1].x >> (at(pos + depth) & 31u)) & 1u);
      while (go) {
        const uint32_t c = at(pos + depth);
        gathers++;
        if (is_tail_word(base)) {                                  // a one-child chain: the record's string against the text, all of it or nothing (tm_tables.h)
          const uint4* r = da + tail_record(base);
          const int len = (int)tail_len(r[0].x);
          if (tail_record(base) + 3 > hv.tab.size() / 2 || len < (int)kTailMin || len > (int)kTailMax) { fail("chain record", base, (uint64_t)len); break; }
          const uint8_t* str = reinterpret_cast<const uint8_t*>(r + 1);
          bool same = depth + len <= limit;
          for (int k = 0; k < len && same; k++) same = at(pos + depth + k) == str[k];
          ntail++;
          if (!same) break;                                        // the walk ends on the chain's head: nothing between it and the end node is a key
          depth += len;
          node = link_node(r[0].x);
          if (r[0].y != 0) { bestv = r[0].y; bestlen = depth; if (node_id(r[0].y) != node) fail("chain record: value of another node", node); }
          base = r[0].w;
          go = depth < limit && ((r[0].z >> (at(pos + depth) & 31u)) & 1u);
          continue;
        }
        if (base + c > nb) { fail("probe outside the array", base, c); break; }
        const uint4 h = da[base + c];
        if (h.x != node) break;
        depth++;
        node = node_id(h.y);
        if (node < hv.n_info) { bestv = h.y; bestlen = depth; }
        base = h.w;
        go = depth < limit && ((h.z >> (at(pos + depth) & 31u)) & 1u);
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
  printf("kind %u, %u ids, capcode %u: %u records, %u nodes, %llu edges in %u entries; %llu positions walked, %.2f gathers each, %llu chain compares: %s\n", kind, vsize, capcode,
         hv.n_info, hv.n_nodes, (unsigned long long)nkeys, nb, (unsigned long long)npos, (double)gathers / (double)std::max<uint64_t>(npos, 1), (unsigned long long)ntail, g_bad == bad0 ? "ok" : "FAILED");
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
