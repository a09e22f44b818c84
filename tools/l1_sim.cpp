// l1_sim.cpp — development aid: what a layout of the walk tables BY USE would do to the requests K1 sends to the L2.
// k_match_branch is closest to the L2's request rate (bench.py: roofline_l2), and a request is a gather that missed the 32 KiB
// vector L1 of its CU (256 lines of 128 bytes, shared by 32 wavefronts).  A 128-byte line holds eight 16-byte entries: if the
// entries a walk touches most lie eight to a line, the L1 holds eight times as many of them.  This model replays steps A1 and B
// (row gathers) of the kernel over a synthetic corpus exactly as the kernel walks (direct map, suffix links, double-array probes
// behind the child filters; tools/hot_sim.cpp), interleaves the rounds of 32 wavefronts as one CU issues them, and runs the line
// addresses through an LRU cache of 256 lines — once with the tables as they are laid out, once with every table's entries
// renumbered by how often a CALIBRATION corpus (another seed) used them: node ids (suffix links, rows) are free to choose as long
// as accepting nodes stay below n_info; double-array entries move only as whole families (the children of one parent keep their
// places relative to base(parent)), so there the model places the parents in order of use.
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/l1_sim.cpp -o /tmp/l1_sim -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <unordered_map>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_device.h"
#include "tm_pipeline.h"

using namespace tmh;

struct Lru {                      // fully associative, `cap` lines
  size_t cap;
  std::list<uint64_t> order;
  std::unordered_map<uint64_t, std::list<uint64_t>::iterator> where;
  uint64_t hits = 0, misses = 0;
  explicit Lru(size_t c) : cap(c) {}
  void touch(uint64_t line) {
    auto it = where.find(line);
    if (it != where.end()) { order.erase(it->second); order.push_front(line); it->second = order.begin(); hits++; return; }
    misses++;
    order.push_front(line); where[line] = order.begin();
    if (order.size() > cap) { where.erase(order.back()); order.pop_back(); }
  }
};

// one gather = (table, entry index); tables: 0 double array, 1 direct map, 2 suffix links, 3 rows
struct G { uint8_t tab; uint32_t idx; };
struct WaveTrace { std::vector<std::vector<G>> rounds; };

struct Corpus { uint8_t* text = nullptr; std::vector<uint64_t> off; uint32_t nd = 0; };
static Corpus make_corpus(uint32_t kind, uint64_t seed, uint64_t nbytes, uint32_t capcode) {
  Corpus c;
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint64_t nb = 0;
  tm_synth_corpus(kind, seed, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &c.nd, &nb);
  c.off.resize(c.nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), c.nd, capcode, 1, 0, &c.text, c.off.data()) != 0) exit(1);
  return c;
}

// the kernel's A1 walk of one segment -> per-lane list of gathers per round (a lane's k-th gather is issued in the wave's k-th round)
static void trace_segment(const HostVocab& hv, const uint8_t* t, int dl, bool share, WaveTrace& out, std::vector<uint64_t>* cnt /* [4] */) {
  const uint2* tab = hv.tab.data();
  const size_t direct16 = hv.direct_off / 16, link16 = hv.link_off / 16;
  const int Lmax = (int)hv.max_len, seg = SEG, np = seg + 40;
  auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
  const int ntask = share ? seg : std::min(np, dl);
  const int nwalkpos = (!share && dl <= np) ? ntask - 1 : ntask;
  const int run = (std::max(nwalkpos, 0) + 63) >> 6;
  std::vector<std::vector<G>> lanes(64);
  std::vector<G> rows;
  for (int lane = 0; lane < 64; lane++) {
    const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
    int depth = 0; uint32_t node = 0; bool first = true;
    for (int pos = lane * run; pos < end; pos++) {
      const int limit = std::min(dl - pos, Lmax);
      G g;
      if (!first && depth >= 3) g = G{2, node}; else g = G{1, at(pos) | (at(pos + 1) << 8)};
      lanes[lane].push_back(g);
      const uint2* e = tab + 2 * ((g.tab == 2 ? link16 : direct16) + g.idx);
      uint32_t src = e[0].x, filt = e[1].x, bestv = e[0].y, base = e[1].y;
      depth = (int)link_depth(src); node = link_node(src);
      bool go = depth < limit;
      while (go) {
        const uint32_t c = at(pos + depth);
        if (!((filt >> (c & 31u)) & 1u)) break;
        const uint32_t h = base + c;
        lanes[lane].push_back(G{0, h});
        const uint4 d = reinterpret_cast<const uint4*>(tab)[h];
        if (d.x != node) break;
        depth++; node = node_id(d.y);
        if (node < hv.n_info) bestv = d.y;
        filt = d.z; base = d.w;
        go = (d.y & kHasChildren) != 0 && depth < limit;
      }
      if (pos < seg && pos < dl && bestv != 0 && node_id(bestv) < hv.n_info) rows.push_back(G{3, node_id(bestv)});
      first = false;
    }
  }
  size_t nr = 0;
  for (auto& l : lanes) nr = std::max(nr, l.size());
  for (size_t r = 0; r < nr; r++) {
    std::vector<G> round;
    for (auto& l : lanes) if (r < l.size()) round.push_back(l[r]);
    out.rounds.push_back(std::move(round));
  }
  for (size_t i = 0; i < rows.size(); i += 64) out.rounds.emplace_back(rows.begin() + i, rows.begin() + std::min(rows.size(), i + 64));
  if (cnt) for (auto& r : out.rounds) for (auto& g : r) cnt[g.tab][g.idx]++;
}

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
  const size_t n16 = hv.tab.size() / 2;
  printf("vocab %u ids, n_info %u, nodes %u, double array %u entries, tab %.2f MB, rows %.2f MB\n", hv.n_ids, hv.n_info, hv.n_nodes, hv.n_da, n16 * 16 / 1048576.0, hv.n_info * 16 / 1048576.0);

  // ---- calibration corpus: how often is every entry used?
  std::vector<uint64_t> cnt[4];
  cnt[0].assign(hv.n_da + 1, 0); cnt[1].assign(65536, 0); cnt[2].assign(hv.n_nodes + 1, 0); cnt[3].assign(hv.n_info, 0);
  {
    Corpus cal = make_corpus(kind, 0x434F5250 + 77, nbytes, capcode);
    for (uint32_t d = 0; d < cal.nd; d++)
      for (uint64_t b = cal.off[d]; b < cal.off[d + 1]; b += SEG) {
        WaveTrace w;
        const int dl = (int)std::min<uint64_t>(cal.off[d + 1] - b, 1 << 20);
        trace_segment(hv, cal.text + b, dl, cal.off[d + 1] - b > SEG, w, cnt);
      }
  }
  // ---- layouts by use
  // node tables (links: all nodes; rows: accepting nodes): new index = rank by count
  auto rank_of = [](const std::vector<uint64_t>& c) {
    std::vector<uint32_t> order(c.size()), rank(c.size());
    for (uint32_t i = 0; i < c.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return c[a] > c[b]; });
    for (uint32_t i = 0; i < c.size(); i++) rank[order[i]] = i;
    return rank;
  };
  const std::vector<uint32_t> rank_link = rank_of(cnt[2]), rank_direct = rank_of(cnt[1]), rank_da_free = rank_of(cnt[0]);
  std::vector<uint32_t> rank_row = rank_of(cnt[3]);
  if (getenv("ROWS_BY_SCORE")) {          // a STATIC predictor of use: the score column of the .vocab file (what the trainer measured), nothing from a sample
    std::vector<uint64_t> sc(hv.n_info);
    for (uint32_t i = 0; i < hv.n_info; i++) sc[i] = (uint64_t)(std::max(0.0f, hv.rec_score[i]) * 1e15);
    rank_row = rank_of(sc);
  }
  // double array: families (children of one parent) re-placed by first fit in order of the family's use
  std::vector<uint32_t> da_new(hv.n_da + 1, 0);
  {
    const uint4* da = reinterpret_cast<const uint4*>(hv.tab.data());
    struct Fam { uint32_t parent; std::vector<uint32_t> entries; uint64_t use = 0; };
    std::unordered_map<uint32_t, size_t> of;
    std::vector<Fam> fams;
    for (uint32_t i = 0; i < hv.n_da; i++) if (da[i].x != kNone) {
      auto it = of.find(da[i].x);
      if (it == of.end()) { of[da[i].x] = fams.size(); fams.push_back(Fam{da[i].x, {}, 0}); it = of.find(da[i].x); }
      fams[it->second].entries.push_back(i); fams[it->second].use += cnt[0][i];
    }
    std::stable_sort(fams.begin(), fams.end(), [](const Fam& a, const Fam& b) { return a.use * b.entries.size() > b.use * a.entries.size(); });   // use per entry
    std::vector<uint8_t> used((size_t)hv.n_da * 2 + 1024, 0);
    uint32_t cursor = 256;
    for (auto& f : fams) {
      const uint32_t lo = *std::min_element(f.entries.begin(), f.entries.end());
      while (used[cursor]) cursor++;
      uint32_t at = cursor;
      for (int tries = 0;; tries++) {
        bool ok = true;
        for (uint32_t e : f.entries) if (used[at + (e - lo)]) { ok = false; break; }
        if (ok) break;
        do at++; while (used[at]);
        if (tries > 200) { while (true) { bool free_ = true; for (uint32_t e : f.entries) if (used[at + (e - lo)]) { free_ = false; break; } if (free_) break; at++; } break; }
      }
      for (uint32_t e : f.entries) { used[at + (e - lo)] = 1; da_new[e] = at + (e - lo); }
    }
  }
  // ---- evaluation corpus, 32 wavefronts of a CU interleaved round by round
  Corpus ev = make_corpus(kind, 0x434F5250 + 2, nbytes, capcode);
  std::vector<WaveTrace> segs;
  for (uint32_t d = 0; d < ev.nd; d++)
    for (uint64_t b = ev.off[d]; b < ev.off[d + 1]; b += SEG) {
      segs.emplace_back();
      const int dl = (int)std::min<uint64_t>(ev.off[d + 1] - b, 1 << 20);
      trace_segment(hv, ev.text + b, dl, ev.off[d + 1] - b > SEG && (segs.size() & 3) != 0, segs.back(), nullptr);
    }
  const uint64_t base_tab[4] = {0, hv.direct_off / 16, hv.link_off / 16, 1ull << 28};       // entry-index bases of the four tables (rows in their own space)
  struct Layout { const char* name; bool da, direct, link, row; };
  const Layout layouts[] = {{"as laid out", false, false, false, false}, {"rows by use", false, false, false, true}, {"links by use", false, false, true, false},
                            {"rows + links by use", false, false, true, true}, {"+ double-array families by use", true, false, true, true},
                            {"+ direct map by use (needs an index step: bound only)", true, true, true, true}};
  // (argv[6] / argv[7]: another cache - e.g. 32768 lines and 1024 wavefronts: the 4 MB L2 of an XCD under its 32 CUs)
  const size_t arg_lines = argc > 6 ? (size_t)atoll(argv[6]) : 0, arg_slots = argc > 7 ? (size_t)atoll(argv[7]) : 32;
  for (size_t lines : {arg_lines ? arg_lines : (size_t)256, arg_lines ? (size_t)0 : (size_t)128}) {
    if (!lines) continue;
    for (const Layout& L : layouts) {
      Lru l1(lines);
      uint64_t per_tab_miss[4] = {0, 0, 0, 0}, per_tab[4] = {0, 0, 0, 0}, lookups = 0;
      std::vector<uint64_t> uniq;
      // a CU: 32 slots; slot s works through segments s, s + 32 * (number of CUs) ... here simply consecutive blocks of 32 segments at a time, each slot
      // starting its next segment when it finishes the current one
      size_t next = 0;
      struct Slot { size_t seg; size_t round; bool live; };
      std::vector<Slot> slots(arg_slots, Slot{0, 0, false});
      size_t live = 0;
      for (auto& s : slots) if (next < segs.size()) { s = Slot{next++, 0, true}; live++; }
      while (live) {
        for (auto& s : slots) {
          if (!s.live) continue;
          const WaveTrace& w = segs[s.seg];
          if (s.round >= w.rounds.size()) { if (next < segs.size()) s = Slot{next++, 0, true}; else { s.live = false; live--; } continue; }
          uniq.clear();
          for (const G& g : w.rounds[s.round]) {
            uint64_t idx = g.idx;
            if (g.tab == 0 && L.da) idx = da_new[g.idx];
            if (g.tab == 1 && L.direct) idx = rank_direct[g.idx];
            if (g.tab == 2 && L.link) idx = rank_link[g.idx];
            if (g.tab == 3 && L.row) idx = rank_row[g.idx];
            const uint64_t before = l1.misses;
            l1.touch((base_tab[g.tab] + idx) >> 3);
            uniq.push_back((base_tab[g.tab] + idx) >> 3);
            per_tab[g.tab]++; per_tab_miss[g.tab] += l1.misses - before;
          }
          // the L1 looks a line up once per gather INSTRUCTION however many lanes want it
          std::sort(uniq.begin(), uniq.end());
          lookups += std::unique(uniq.begin(), uniq.end()) - uniq.begin();
          s.round++;
        }
      }
      const double nseg = (double)segs.size();
      printf("L1 %3zu lines  %-56s line look-ups per segment %7.1f  misses %7.1f  (double array %6.1f of %6.1f, direct %5.1f of %5.1f, links %6.1f of %6.1f, rows %6.1f of %6.1f)\n", lines, L.name,
             lookups / nseg, l1.misses / nseg, per_tab_miss[0] / nseg, per_tab[0] / nseg, per_tab_miss[1] / nseg, per_tab[1] / nseg, per_tab_miss[2] / nseg, per_tab[2] / nseg,
             per_tab_miss[3] / nseg, per_tab[3] / nseg);
    }
  }
  tm_free(ev.text); tm_free(img);
  return 0;
}
