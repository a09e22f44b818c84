// hot_sim.cpp — development aid: how much of K1's table traffic a small, statically chosen HOT table could absorb.
// Replays step A1 of k_match_branch (direct map / suffix links / two-slot buckets with the child filters, as the kernel walks them)
// over a synthetic corpus, counts how often every 16-byte table entry and every row is gathered, and prints the share of the
// gathers the K most frequently used entries cover (K = what fits 4 .. 128 KB).
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/hot_sim.cpp -o /tmp/hot_sim -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_device.h"
#include "tm_pipeline.h"

using namespace tmh;

static void coverage(const char* what, std::vector<uint64_t> cnt, size_t entry_bytes) {
  uint64_t total = 0;
  for (auto c : cnt) total += c;
  std::sort(cnt.begin(), cnt.end(), [](uint64_t a, uint64_t b) { return a > b; });
  size_t used = 0;
  for (auto c : cnt) used += c != 0;
  printf("%-28s total %11llu  distinct %8zu:", what, (unsigned long long)total, used);
  for (size_t kb : {4, 8, 16, 32, 64, 128}) {
    const size_t k = std::min(cnt.size(), kb * 1024 / entry_bytes);
    uint64_t s = 0;
    for (size_t i = 0; i < k; i++) s += cnt[i];
    printf("  %zuK %.3f", kb, total ? (double)s / total : 0.0);
  }
  printf("\n");
}

// working set in cache lines: how many MB of 128-byte lines cover a share of the gathers, with the entries where they lie and
// with the entries packed in order of use (what a layout by traffic could reach at best)
static void lines(const char* what, const std::vector<uint64_t>& cnt, size_t entry_bytes) {
  const size_t per = 128 / entry_bytes;
  uint64_t total = 0;
  for (auto c : cnt) total += c;
  std::vector<uint64_t> asis((cnt.size() + per - 1) / per, 0), sorted = cnt;
  for (size_t i = 0; i < cnt.size(); i++) asis[i / per] += cnt[i];
  std::sort(sorted.begin(), sorted.end(), [](uint64_t a, uint64_t b) { return a > b; });
  std::vector<uint64_t> packed(asis.size(), 0);
  for (size_t i = 0; i < sorted.size(); i++) packed[i / per] += sorted[i];
  std::sort(asis.begin(), asis.end(), [](uint64_t a, uint64_t b) { return a > b; });
  printf("%-28s %.2f MB, MB of 128-byte lines for a share of the gathers (as laid out / packed by use):", what, cnt.size() * entry_bytes / 1048576.0);
  for (double share : {0.8, 0.9, 0.95, 0.98, 0.99}) {
    auto need = [&](const std::vector<uint64_t>& v) { uint64_t s = 0; size_t k = 0; while (k < v.size() && s < share * total) s += v[k++]; return k * 128 / 1048576.0; };
    printf("  %.0f%% %.2f / %.2f", share * 100, need(asis), need(packed));
  }
  printf("\n");
}

int main(int argc, char** argv) {
  const uint32_t kind = argc > 1 ? atoi(argv[1]) : TM_KIND_ENGLISHCODE;
  const uint32_t vsize = argc > 2 ? atoi(argv[2]) : 32000;
  const uint64_t nbytes = argc > 3 ? atoll(argv[3]) : (8ull << 20);
  const uint32_t capcode = argc > 4 ? atoi(argv[4]) : 2;
  const int seg = argc > 5 ? atoi(argv[5]) : SEG;
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(kind, vsize, capcode, 1, 3, 0x544D0002, 0, &img, &img_n) != 0) return 1;
  HostVocab hv;
  if (parse_vocab(img, img_n, hv) != 0) { fprintf(stderr, "parse failed: %s\n", last_error()); return 1; }
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint32_t nd = 0; uint64_t nb = 0;
  tm_synth_corpus(kind, 0x434F5250 + 2, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nb);
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), nd, capcode, 1, 0, &text, off.data()) != 0) return 1;
  printf("vocab %u ids, n_info %u, nodes %u, tab %zu bytes; corpus %llu bytes in %u docs; segment %d\n", hv.n_ids, hv.n_info, hv.n_nodes, hv.tab.size() * 8,
         (unsigned long long)off[nd], nd, seg);
  const uint2* tab = hv.tab.data();
  const size_t n16 = hv.tab.size() / 2;                        // 16-byte entries of the gather buffer
  std::vector<uint64_t> c_parent(hv.n_nodes + 1, 0);          // probes issued below each node (whatever bucket they ended in)
  std::vector<uint64_t> c_direct(n16, 0), c_link(n16, 0), c_bucket(n16, 0), c_all(n16, 0), c_row(hv.n_info, 0), c_pair(65536, 0);
  const size_t direct16 = hv.direct_off / 16, link16 = hv.link_off / 16;
  // TAIL model: a chain of one-child, non-accepting nodes that ends in an accepting leaf, entered at its head c: tail_len[c] = edges from c to the leaf
  std::vector<uint32_t> tail_len(hv.n_nodes + 1, 0);
  {
    const uint32_t nn = hv.n_nodes + 1;
    std::vector<uint32_t> nch(nn, 0), only(nn, kNone);
    std::vector<uint8_t> haskids(nn, 0);
    for (size_t i = 0; i < hv.n_da; i++) { const uint4 d = reinterpret_cast<const uint4*>(tab)[i]; if (d.x != kNone && d.x < nn) { nch[d.x]++; only[d.x] = node_id(d.y); } }
    // memo from the leaves up: process nodes repeatedly along chains (chains are short: <= 40)
    for (uint32_t c = 0; c < nn; c++) {
      if (c < hv.n_info || nch[c] != 1) continue;                 // heads must be plain
      uint32_t n = c, L = 0; bool ok = false;
      for (int step = 0; step < 64; step++) {
        if (n >= nn) break;
        if (n < hv.n_info) { ok = nch[n] == 0; break; }            // accepting: a leaf ends the chain, anything else breaks it
        if (nch[n] != 1) break;
        n = only[n]; L++;
      }
      if (ok) tail_len[c] = L;
    }
    uint64_t heads = 0, sumL = 0;
    for (uint32_t c = 0; c < nn; c++) if (tail_len[c] >= 3) { heads++; sumL += tail_len[c]; }
    printf("TAIL model: %llu nodes head a chain of >= 3 one-child nodes to an accepting leaf (mean length %.1f)\n", (unsigned long long)heads, heads ? (double)sumL / heads : 0.0);
  }
  uint64_t tot_rounds_t[3] = {0, 0, 0}, tot_maxpos = 0;                            // chains of >= 3 / >= 5 / >= 8 as tails
  const int Lmax = (int)hv.max_len;
  uint64_t npos = 0, nrow = 0, nwaves = 0, tot_rounds = 0, sum_lane_rounds = 0, tot_rounds_c[4] = {0, 0, 0, 0};
  for (uint32_t d = 0; d < nd; d++) {
    const uint64_t b0 = off[d], e0 = off[d + 1];
    for (uint64_t begin = b0; begin < e0; begin += seg) {
      const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
      const uint8_t* t = text + begin;
      auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
      const int np = seg + 40;
      const int ntask = std::min(np, dl);
      const int nwalkpos = dl <= np ? ntask - 1 : ntask;
      const int run = (std::max(nwalkpos, 0) + 63) >> 6;
      int wave_rounds = 0, wave_rounds_c[4] = {0, 0, 0, 0}, wave_rounds_t[3] = {0, 0, 0}, wave_maxpos = 0;
      for (int lane = 0; lane < 64; lane++) {
        const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
        int depth = 0; uint32_t node = 0; bool first = true;
        int lane_rounds = 0, lane_rounds_c[4] = {0, 0, 0, 0}, lane_rounds_t[3] = {0, 0, 0};
        for (int pos = lane * run; pos < end; pos++) {
          const int limit = std::min(dl - pos, Lmax);
          size_t e16;
          if (!first && depth >= 3) { e16 = link16 + node; c_link[e16]++; }
          else { const uint32_t pr = at(pos) | (at(pos + 1) << 8); e16 = direct16 + pr; c_direct[e16]++; c_pair[pr]++; }
          c_all[e16]++;
          const uint2* e = tab + 2 * e16;
          uint32_t src = e[0].x, filt = e[1].x, bestv = e[0].y, base = e[1].y;
          depth = (int)link_depth(src); node = link_node(src);
          int rounds = 1, rounds_c[4] = {1, 1, 1, 1};       // this position's rounds: as built / with unary non-accepting chains walked 2, 3, 4 bytes per probe
          int chain = 0;                                      // probes since the last node that is accepting or branches
          int rounds_t[3] = {1, 1, 1}; bool in_tail[3] = {false, false, false};
          bool go = depth < limit;
          while (go) {
            const uint32_t c = at(pos + depth);
            if (!((filt >> (c & 31u)) & 1u)) break;
            if (is_tail_word(base)) {                                   // a one-child chain (tm_tables.h): one gather of its record, the whole chain or nothing
              const uint4* r = reinterpret_cast<const uint4*>(tab) + tail_record(base);
              c_bucket[tail_record(base)]++; c_all[tail_record(base)]++;
              rounds++; for (int q = 0; q < 4; q++) rounds_c[q]++; for (int q = 0; q < 3; q++) rounds_t[q]++;
              const int len = (int)tmh::tail_len(r[0].x);
              const uint8_t* str = reinterpret_cast<const uint8_t*>(r + 1);
              bool same = depth + len <= limit;
              for (int k = 0; k < len && same; k++) same = at(pos + depth + k) == str[k];
              if (!same) break;
              depth += len; node = link_node(r[0].x);
              if (r[0].y != 0) bestv = r[0].y;
              filt = r[0].z; base = r[0].w; chain = 0;
              go = filt != 0 && depth < limit;
              continue;
            }
            const size_t h = (size_t)base + c;
            c_parent[node]++; c_bucket[h]++; c_all[h]++;
            rounds++;
            const uint4 d = reinterpret_cast<const uint4*>(tab)[h];
            if (d.x != node) { for (int q = 0; q < 4; q++) rounds_c[q]++; for (int q = 0; q < 3; q++) if (!in_tail[q]) rounds_t[q]++; break; }
            for (int q = 0; q < 3; q++) {
              if (in_tail[q]) continue;
              rounds_t[q]++;
              const uint32_t tl = tail_len[node_id(d.y)], need = q == 0 ? 3u : q == 1 ? 5u : 8u;
              if (depth >= 2 && tl >= need) { in_tail[q] = true; rounds_t[q] += (int)((tl + 15) / 16); }
            }
            // a probe is free in the compressed models if it continues a chain: the previous node had one child and was not accepting
            const bool unary_prev = depth >= 3 && chain > 0;
            for (int q = 0; q < 4; q++) if (!(unary_prev && (chain % (q == 0 ? 16 : q + 1)) != 0)) rounds_c[q]++;
            depth++; node = node_id(d.y);
            if (node < hv.n_info) bestv = d.y;
            const bool plain = node >= hv.n_info && d.z != 0 && (d.z & (d.z - 1)) == 0;    // not accepting, children over one residue (mostly: one child)
            chain = plain ? chain + 1 : 0;
            filt = d.z; base = d.w;
            go = (d.y & kHasChildren) != 0 && depth < limit;
          }
          lane_rounds += rounds; wave_maxpos = std::max(wave_maxpos, rounds);
          for (int q = 0; q < 4; q++) lane_rounds_c[q] += rounds_c[q];
          for (int q = 0; q < 3; q++) lane_rounds_t[q] += rounds_t[q];
          if (pos < seg && bestv != 0 && node_id(bestv) < hv.n_info) { c_row[node_id(bestv)]++; nrow++; }
          first = false; npos++;
        }
        wave_rounds = std::max(wave_rounds, lane_rounds); sum_lane_rounds += lane_rounds;
        for (int q = 0; q < 4; q++) wave_rounds_c[q] = std::max(wave_rounds_c[q], lane_rounds_c[q]);
        for (int q = 0; q < 3; q++) wave_rounds_t[q] = std::max(wave_rounds_t[q], lane_rounds_t[q]);
      }
      nwaves++; tot_rounds += wave_rounds; tot_maxpos += wave_maxpos;
      for (int q = 0; q < 4; q++) tot_rounds_c[q] += wave_rounds_c[q];
      for (int q = 0; q < 3; q++) tot_rounds_t[q] += wave_rounds_t[q];
    }
  }
  printf("positions walked %llu (%.3f per byte), row gathers %.3f per byte\n", (unsigned long long)npos, (double)npos / off[nd], (double)nrow / off[nd]);
  printf("step A1: %.2f rounds per wavefront (the slowest lane), %.2f per lane on average; with unary chains walked k bytes per probe: k=2 %.2f, k=3 %.2f, k=4 %.2f, k=16 %.2f\n",
         (double)tot_rounds / nwaves, (double)sum_lane_rounds / nwaves / 64, (double)tot_rounds_c[1] / nwaves, (double)tot_rounds_c[2] / nwaves, (double)tot_rounds_c[3] / nwaves, (double)tot_rounds_c[0] / nwaves);
  printf("step A1: the deepest single position of a wavefront takes %.2f rounds (no re-balancing of the lanes' runs can go below that)\n", (double)tot_maxpos / nwaves);
  printf("step A1 with chains to an accepting leaf compared 16 bytes per gather (head probe + ceil(L / 16) rounds): chains >= 3: %.2f, >= 5: %.2f, >= 8: %.2f rounds per wavefront\n",
         (double)tot_rounds_t[0] / nwaves, (double)tot_rounds_t[1] / nwaves, (double)tot_rounds_t[2] / nwaves);
  coverage("SET via direct map (16 B)", c_direct, 16);
  coverage("SET via suffix link (16 B)", c_link, 16);
  coverage("PROBE entries (16 B)", c_bucket, 16);
  coverage("all A1 gathers (16 B)", c_all, 16);
  coverage("rows (16 B)", c_row, 16);
  lines("all A1 gathers", c_all, 16);
  lines("  suffix links", c_link, 16);
  lines("  buckets", c_bucket, 16);
  lines("  direct map", c_direct, 16);
  lines("rows", c_row, 16);
  // a HOT REGION of the edge hash: the children of the parents a static rule picks live in a small table of their own.  Share of
  // the probes that go there, for the rule "by measured traffic" (the best any rule can do) and for rules that only see the trie.
  {
    const uint32_t nn = hv.n_nodes + 1;
    std::vector<uint32_t> par(nn, kNone), nchild(nn, 0), depth(nn, 0), sub(nn, 1);
    for (size_t i = 0; i < hv.n_da; i++) { const uint4 d = reinterpret_cast<const uint4*>(tab)[i]; if (d.x != kNone) { const uint32_t pnode = d.x, c = node_id(d.y); if (c < nn && pnode < nn) { par[c] = pnode; nchild[pnode]++; } } }
    std::vector<uint32_t> order;                                 // parents first
    { std::vector<std::vector<uint32_t>> kids(nn);
      for (uint32_t c = 0; c < nn; c++) if (par[c] != kNone) kids[par[c]].push_back(c);
      for (uint32_t r = 0; r < nn; r++) if (par[r] == kNone && nchild[r]) { depth[r] = 2; order.push_back(r); }
      for (size_t i = 0; i < order.size(); i++) for (uint32_t c : kids[order[i]]) { depth[c] = depth[order[i]] + 1; order.push_back(c); } }
    for (size_t i = order.size(); i-- > 0;) if (par[order[i]] != kNone) sub[par[order[i]]] += sub[order[i]];
    std::vector<uint32_t> parents;
    uint64_t total = 0, edges = 0;
    for (uint32_t n = 0; n < nn; n++) if (nchild[n]) { parents.push_back(n); total += c_parent[n]; edges += nchild[n]; }
    printf("edge hash: %zu parents, %llu edges, %llu probes\n", parents.size(), (unsigned long long)edges, (unsigned long long)total);
    auto report = [&](const char* rule, auto less) {
      std::sort(parents.begin(), parents.end(), less);
      printf("  %-34s", rule);
      for (size_t kb : {256, 512, 1024, 2048}) {                   // hot region of kb KiB at 0.3 edges per slot
        const uint64_t cap = (uint64_t)(kb * 1024 / 8 * 0.3);
        uint64_t e = 0, pr = 0;
        for (uint32_t n : parents) { if (e + nchild[n] > cap) break; e += nchild[n]; pr += c_parent[n]; }
        printf("  %zuK: %.3f", kb, (double)pr / total);
      }
      printf("\n");
    };
    report("by measured traffic per edge", [&](uint32_t a, uint32_t b) { return (double)c_parent[a] / nchild[a] > (double)c_parent[b] / nchild[b]; });
    report("by depth, then subtree size", [&](uint32_t a, uint32_t b) { return depth[a] != depth[b] ? depth[a] < depth[b] : sub[a] > sub[b]; });
    report("by subtree size", [&](uint32_t a, uint32_t b) { return sub[a] != sub[b] ? sub[a] > sub[b] : depth[a] < depth[b]; });
    report("by subtree size per child", [&](uint32_t a, uint32_t b) { return (double)sub[a] / nchild[a] > (double)sub[b] / nchild[b]; });
    report("by subtree size / 2^depth", [&](uint32_t a, uint32_t b) { return (double)sub[a] / (1u << std::min(depth[a], 20u)) > (double)sub[b] / (1u << std::min(depth[b], 20u)); });
  }
  // the direct map as a rank-mapped square: bytes sorted by how often they occur in a looked-up pair; share of the look-ups whose
  // two bytes are both among the R most frequent ones (R*R entries)
  {
    std::vector<uint64_t> bc(256, 0);
    for (uint32_t pr = 0; pr < 65536; pr++) { bc[pr & 255] += c_pair[pr]; bc[pr >> 8] += c_pair[pr]; }
    std::vector<int> order(256);
    for (int i = 0; i < 256; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return bc[a] > bc[b]; });
    std::vector<int> rank(256);
    for (int i = 0; i < 256; i++) rank[order[i]] = i;
    uint64_t tot = 0;
    for (auto c : c_pair) tot += c;
    printf("direct map as a square of the R most frequent bytes:");
    for (int R : {16, 24, 32, 48, 64}) {
      uint64_t s = 0;
      for (uint32_t pr = 0; pr < 65536; pr++) if (rank[pr & 255] < R && rank[pr >> 8] < R) s += c_pair[pr];
      printf("  R=%d (%d KB at 16 B) %.3f", R, R * R * 16 / 1024, (double)s / tot);
    }
    printf("\n");
  }
  tm_free(text); tm_free(img);
  return 0;
}
