// tail_sim.cpp — development aid: rounds of step A1 per wavefront if the DEEP part of a walk took several bytes per gather.
// The A1 loop of k_match_branch runs as many rounds as its slowest lane needs, and that lane is slow because of ONE deep walk (a
// multi-word token of up to 40 bytes, a byte per round: tools/hot_sim.cpp).  Walking one-child chains several bytes per probe in
// EVERY round was built in round 3 and lost (+22 %: the compare costs ~20 vector instructions in every round of every lane).  This
// model asks what a TWO-LOOP form would give: the loop as it is for the first R rounds, then — only for what is still walking —
// a heavier round in which a lane that stands on the head of a run of one-child, non-accepting nodes consumes up to K bytes of
// the run with one gather of a "run entry" (the run's bytes inline, its nodes numbered consecutively).
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/tail_sim.cpp -o /tmp/tail_sim -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
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

int main(int argc, char** argv) {
  const uint32_t kind = argc > 1 ? atoi(argv[1]) : TM_KIND_ENGLISHCODE;
  const uint32_t vsize = argc > 2 ? atoi(argv[2]) : 32000;
  const uint64_t nbytes = argc > 3 ? atoll(argv[3]) : (4ull << 20);
  const uint32_t capcode = argc > 4 ? atoi(argv[4]) : 2;
  const uint64_t vseed = argc > 5 ? strtoull(argv[5], nullptr, 0) : 0x544D0002;
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(kind, vsize, capcode, 1, 3, vseed, 0, &img, &img_n) != 0) return 1;
  HostVocab hv;
  if (parse_vocab(img, img_n, hv) != 0) { fprintf(stderr, "parse failed: %s\n", last_error()); return 1; }
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint32_t nd = 0; uint64_t nb = 0;
  tm_synth_corpus(kind, 0x434F5250 + 2, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nb);
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), nd, capcode, 1, 0, &text, off.data()) != 0) return 1;
  const uint2* tab = hv.tab.data();
  const uint4* da = reinterpret_cast<const uint4*>(tab);
  const size_t direct16 = hv.direct_off / 16, link16 = hv.link_off / 16;
  const uint32_t nn = hv.n_nodes + 1;
  // run_len[n]: number of edges of the one-child chain that starts BELOW node n and runs through non-accepting one-child nodes; it ends on the first node
  // that is accepting, branches or is a leaf (that node included as the last edge).  run_len[n] >= 2 means a run entry can take several bytes at once.
  std::vector<uint32_t> nch(nn, 0), only(nn, kNone);
  for (size_t i = 0; i < hv.n_da; i++) if (da[i].x != kNone && da[i].x < nn) { nch[da[i].x]++; only[da[i].x] = node_id(da[i].y); }
  std::vector<uint32_t> run_len(nn, 0);
  for (uint32_t n = 0; n < nn; n++) {
    if (nch[n] != 1) continue;
    uint32_t L = 0, c = n;
    while (nch[c] == 1) { c = only[c]; L++; if (c < hv.n_info || L >= 40) break; }
    run_len[n] = L;
  }
  const int Lmax = (int)hv.max_len;
  printf("vocab %u ids, %u nodes; corpus %llu bytes\n", hv.n_ids, hv.n_nodes, (unsigned long long)off[nd]);
  struct Cfg { int R, K; };
  const Cfg cfgs[] = {{1000, 1}, {8, 4}, {8, 8}, {10, 4}, {10, 8}, {12, 8}, {12, 16}, {14, 8}, {0, 8}, {0, 16}};
  for (const Cfg& cf : cfgs) {
    uint64_t nwaves = 0, tot_rounds = 0, tot_tail = 0, run_gathers = 0, probes = 0;
    for (uint32_t d = 0; d < nd; d++) {
      const uint64_t b0 = off[d], e0 = off[d + 1];
      uint32_t wv = 0;
      for (uint64_t begin = b0; begin < e0; begin += SEG, wv++) {
        const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
        const uint8_t* t = text + begin;
        auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
        const bool share = e0 - begin > SEG && (wv & 3) != 3;
        const int np = SEG + 40, ntask = share ? SEG : std::min(np, dl), nwalkpos = (!share && dl <= np) ? ntask - 1 : ntask, run = (std::max(nwalkpos, 0) + 63) >> 6;
        // lane state machines, stepped round by round together (the compressed step is only available from round R on)
        struct Lane { int pos, end, depth, limit; uint32_t node, filt, base; bool first, walking, done; };
        std::vector<Lane> ln(64);
        for (int l = 0; l < 64; l++) { ln[l] = Lane{l * run, std::max(std::min(l * run + run, nwalkpos), 0), 0, 0, 0, 0, 0, true, false, false}; ln[l].done = ln[l].pos >= ln[l].end; }
        int rounds = 0;
        for (;; rounds++) {
          bool any = false;
          for (auto& L : ln) {
            if (L.done) continue;
            any = true;
            if (!L.walking) {                         // SET round: direct map or suffix link
              const int pos = L.pos;
              L.limit = std::min(dl - pos, Lmax);
              const size_t e16 = (!L.first && L.depth >= 3) ? link16 + L.node : direct16 + (at(pos) | (at(pos + 1) << 8));
              const uint2* e = tab + 2 * e16;
              L.depth = (int)link_depth(e[0].x); L.node = link_node(e[0].x); L.filt = e[1].x; L.base = e[1].y;
              L.first = false;
              const uint32_t c = at(pos + L.depth);
              L.walking = L.depth < L.limit && ((L.filt >> (c & 31u)) & 1u);
              if (!L.walking) { L.pos++; if (L.pos >= L.end) L.done = true; }
              continue;
            }
            // PROBE round
            bool cont = false;
            if (rounds >= cf.R && cf.K > 1 && L.node < nn && run_len[L.node] >= 2) {
              // run entry: up to K bytes of the one-child run below the node
              run_gathers++;
              uint32_t k = 0, n = L.node;
              const uint32_t want = std::min<uint32_t>({run_len[L.node], (uint32_t)cf.K, (uint32_t)(L.limit - L.depth)});
              bool broke = false;
              while (k < want) {
                const uint32_t c = at(L.pos + L.depth);
                // the child of n over c (n has one child): find its entry
                const uint32_t child = only[n];
                // which byte leads to the child?  look it up in the double array: entry base(n) + c must have check == n
                uint32_t bn = 0;
                { // base of n: scan is not available here; use the invariant that the only child's entry is at base(n) + byte, found via the filter/base carried for the FIRST step and re-derived below
                  bn = (k == 0) ? L.base : L.base; }
                const uint4 dd = da[(size_t)bn + c];
                if (dd.x != n) { broke = true; break; }
                L.depth++; n = node_id(dd.y); L.filt = dd.z; L.base = dd.w; k++;
                (void)child;
              }
              L.node = n;
              if (broke) cont = false;
              else {
                const uint32_t c = at(L.pos + L.depth);
                cont = nch[n] > 0 && L.depth < L.limit && ((L.filt >> (c & 31u)) & 1u);
              }
            } else {
              probes++;
              const uint32_t c = at(L.pos + L.depth);
              const uint4 dd = da[(size_t)L.base + c];
              if (dd.x == L.node) {
                L.depth++; L.node = node_id(dd.y); L.filt = dd.z; L.base = dd.w;
                const uint32_t c2 = at(L.pos + L.depth);
                cont = (dd.y & kHasChildren) != 0 && L.depth < L.limit && ((L.filt >> (c2 & 31u)) & 1u);
              }
            }
            if (!cont) { L.walking = false; L.pos++; if (L.pos >= L.end) L.done = true; }
          }
          if (!any) break;
        }
        nwaves++; tot_rounds += rounds; tot_tail += rounds > cf.R ? rounds - cf.R : 0;
      }
    }
    const double r = (double)tot_rounds / nwaves, tl = (double)tot_tail / nwaves;
    printf("first %4d rounds as built, then runs %2d bytes per gather: %.2f rounds per wavefront (%.2f of them in the heavy loop); instructions ~ %.0f (34 per plain, 55 per heavy round); run gathers %.1f per wavefront\n",
           cf.R, cf.K, r, tl, (r - tl) * 34 + tl * 55, (double)run_gathers / nwaves);
  }
  tm_free(text); tm_free(img);
  return 0;
}
