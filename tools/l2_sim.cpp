// l2_sim.cpp — host model of what the 4 MB L2 of ONE XCD sees while k_match_branch runs: the table gathers of step A1 (direct map, suffix
// links, edge hash — walked on the real tables like tools/a1_sim.cpp), the row gathers of step B and the space-prefix gathers of step A3
// (approximated: one row per position with a match, one space-prefix entry per eligible-looking position), interleaved with the
// kernel's streaming traffic (text in; T(p,0) rows, side lists and exit maps out).  Wavefronts of many segments run at once on an XCD, so
// their accesses are interleaved round by round across WAVES_IN_FLIGHT segments.  LRU, 16 ways, 128-byte lines.
// Development aid, written after the device showed that K1's time follows the tables' cache footprint (profiles/r02b_k1_variants_ab.txt):
// run it against the product library and, with LD_PRELOAD, against variant libraries whose tables differ.
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/l2_sim.cpp -o /tmp/l2_sim -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
//   /tmp/l2_sim [kind=1] [vocab=32000] [bytes=16777216] [capcode=2] ; LD_PRELOAD=$PWD/variants/sparse/libtokenmonster_hip.so /tmp/l2_sim
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

struct Cache {
  static constexpr int WAYS = 16, LINE = 128;
  size_t nsets;
  std::vector<uint64_t> tag;      // nsets x WAYS, most recent first
  uint64_t hits = 0, misses = 0;
  explicit Cache(size_t bytes) : nsets(bytes / LINE / WAYS), tag(bytes / LINE, ~0ull) {}
  bool access(uint64_t addr, bool allocate = true) {
    const uint64_t line = addr / LINE;
    uint64_t* s = &tag[(line % nsets) * WAYS];
    for (int w = 0; w < WAYS; w++) if (s[w] == line) { for (int k = w; k > 0; k--) s[k] = s[k - 1]; s[0] = line; hits++; return true; }
    misses++;
    if (allocate) { for (int k = WAYS - 1; k > 0; k--) s[k] = s[k - 1]; s[0] = line; }
    return false;
  }
};

int main(int argc, char** argv) {
  const uint32_t kind = argc > 1 ? atoi(argv[1]) : TM_KIND_ENGLISHCODE;
  const uint32_t vsize = argc > 2 ? atoi(argv[2]) : 32000;
  const uint64_t nbytes = argc > 3 ? atoll(argv[3]) : (16ull << 20);
  const uint32_t capcode = argc > 4 ? atoi(argv[4]) : 2;
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(kind, vsize, capcode, 1, 3, 0x544D0002, 0, &img, &img_n) != 0) { fprintf(stderr, "synth_vocab failed\n"); return 1; }
  HostVocab hv;
  if (parse_vocab(img, img_n, hv) != 0) { fprintf(stderr, "parse failed: %s\n", last_error()); return 1; }
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint32_t nd = 0; uint64_t nb = 0;
  tm_synth_corpus(kind, 0x434F5250 + 2, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nb);
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), nd, capcode, 1, 0, &text, off.data()) != 0) { fprintf(stderr, "normalize failed\n"); return 1; }
  const uint2* tab = hv.tab.data();
  const uint2* direct = tab + hv.direct_off / 8;
  const uint2* link = tab + hv.link_off / 8;
  const int Lmax = (int)hv.max_len;
  // device address map: tab | rows | spl | (streams far away)
  const uint64_t A_TAB = 0, A_ROWS = (hv.tab.size() * 8 + 4095) & ~4095ull, A_SPL = A_ROWS + ((hv.rows.size() * 16 + 4095) & ~4095ull), A_STREAM = 1ull << 40;
  printf("tables: edge hash %.2f MB, direct map %.2f MB, links %.2f MB, rows %.2f MB, space-prefix links %.2f MB\n", (hv.edge_mask + 2) * 16 / 1e6, kDirectSlots * 8 / 1e6,
         hv.n_nodes * 16 / 1e6, hv.rows.size() * 16 / 1e6, hv.spl.size() * 16 / 1e6);

  // the segments of one XCD (every 8th workgroup of 4 segments), their accesses round by round
  struct Seg { std::vector<std::vector<uint64_t>> rounds; uint64_t stream_addr; };
  std::vector<Seg> segs;
  uint64_t g = 0, total_rounds = 0;
  for (uint32_t d = 0; d < nd; d++) {
    const uint64_t b0 = off[d], e0 = off[d + 1];
    for (uint64_t begin = b0; begin < e0; begin += SEG, g++) {
      if (((g / WAVES) & 7) != 0) continue;
      const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
      const uint8_t* t = text + begin;
      auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
      const int ntask = std::min(NPOS, dl);
      const int nwalkpos = dl <= NPOS ? ntask - 1 : ntask;
      const int run = (std::max(nwalkpos, 0) + 63) >> 6;
      Seg sg; sg.stream_addr = A_STREAM + g * 4096;
      std::vector<uint64_t> rows_spl;
      for (int lane = 0; lane < 64; lane++) {
        const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
        size_t r = 0; int depth = 0; uint32_t node = 0; bool first = true;
        auto touch = [&](uint64_t a) { if (sg.rounds.size() <= r) sg.rounds.resize(r + 1); sg.rounds[r].push_back(a); r++; };
        for (int pos = lane * run; pos < end; pos++) {
          const int limit = std::min(dl - pos, Lmax);
          const uint2* e = (!first && depth >= 3) ? link + 2 * (size_t)node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
          touch(A_TAB + (uint64_t)((const uint8_t*)e - (const uint8_t*)tab));
          uint32_t src = e[0].x, filt = e[1].x, best = e[0].y;
          depth = (int)((src >> 23) & 63u); node = src & kNodeMask;
          bool from_set = true, go = (src & kHasChildren) != 0 && depth < limit;
          while (go) {
            const uint32_t c = at(pos + depth);
            if (from_set ? !((filt >> (c & 31u)) & 1u) : !((filt >> (c & 3u)) & 1u)) break;
            const uint32_t key = (node << 8) | c;
            uint32_t h = edge_hash(node, c) >> hv.edge_shift;
            bool hit = false;
            for (;;) { touch(A_TAB + (uint64_t)h * 16); const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1];
              if ((s0.x & kKeyMask) == key) { hit = true; src = s0.y; filt = s0.x >> 28; from_set = false; break; }
              if ((s1.x & kKeyMask) == key) { hit = true; src = s1.y; filt = s1.x >> 28; from_set = false; break; }
              if (s1.x == kNone) break; h = (h + 1) & hv.edge_mask; }
            if (!hit) break;
            depth++; node = src & kNodeMask; if (node < hv.n_info) best = src; go = (src & kHasChildren) != 0 && depth < limit;
          }
          if (best != 0 && (best & kNodeMask) < hv.n_info) {
            if (pos < SEG) rows_spl.push_back(A_ROWS + (uint64_t)(best & kNodeMask) * 16);                               // step B: the row of the match
            if (((best >> 28) & 1u) && ((best >> 22) & 31u) == 0) rows_spl.push_back(A_SPL + (uint64_t)(best & kNodeMask) * 16);   // step A3 (begins with a letter, no word boundary: an upper bound)
          }
          first = false;
        }
      }
      sg.rounds.push_back(rows_spl);        // after the walks
      total_rounds += sg.rounds.size();
      segs.push_back(std::move(sg));
    }
  }
  printf("%zu segments on this XCD, %.1f rounds each\n", segs.size(), (double)total_rounds / segs.size());

  for (int nt = 0; nt < 2; nt++) for (size_t l2 : {(size_t)4 << 20}) {
    Cache c(l2);
    uint64_t tab_acc = 0, tab_miss = 0;
    const size_t W = 1024;                    // wavefronts in flight on an XCD: 32 CUs x 32
    for (size_t base = 0; base < segs.size(); base += W) {
      const size_t n = std::min(W, segs.size() - base);
      size_t maxr = 0;
      for (size_t k = 0; k < n; k++) maxr = std::max(maxr, segs[base + k].rounds.size());
      for (size_t k = 0; k < n; k++) for (int q = 0; q < 3; q++) c.access(segs[base + k].stream_addr + 128 * q, !nt);          // text in (352 bytes)
      for (size_t r = 0; r < maxr; r++)
        for (size_t k = 0; k < n; k++) {
          const Seg& s = segs[base + k];
          if (r >= s.rounds.size()) continue;
          for (uint64_t a : s.rounds[r]) { tab_acc++; if (!c.access(a)) tab_miss++; }
          if (r + 1 == s.rounds.size()) for (int q = 0; q < 12; q++) c.access(s.stream_addr + 512 + 128 * q, !nt);              // rows + side list + exit map out (1.5 KB)
        }
    }
    printf("L2 %zu MB, streams %s: table accesses per segment %.0f, misses per segment %.1f (%.2f %%)\n", l2 >> 20, nt ? "bypass (non-temporal)" : "allocate",
           (double)tab_acc / segs.size(), (double)tab_miss / segs.size(), 100.0 * tab_miss / tab_acc);
  }
  tm_free(text); tm_free(img);
  return 0;
}
