// a1_sim.cpp — host-side model of step A1 of k_match_branch (tm_kernels.hip): counts the gathers every lane issues and the
// rounds a wavefront needs (max over its lanes), for the table layout as built by parse_vocab and for candidate layouts
// (child-byte filters that suppress probes which cannot hit).  Development aid: the kernel is bound by vector-instruction
// issue, rounds x instructions per round is its cost model, and this runs without a GPU.
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/a1_sim.cpp -o /tmp/a1_sim -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_device.h"
#include "tm_pipeline.h"

using namespace tmh;

struct Stats { uint64_t first_g = 0, first_n = 0; uint64_t pos = 0, set = 0, hit = 0, again = 0, miss = 0, filtered = 0, rounds = 0, waves = 0, lane_sum = 0; };

int main(int argc, char** argv) {
  const uint32_t kind = argc > 1 ? atoi(argv[1]) : TM_KIND_ENGLISHCODE;
  const uint32_t vsize = argc > 2 ? atoi(argv[2]) : 32000;
  const uint64_t nbytes = argc > 3 ? atoll(argv[3]) : (8ull << 20);
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
  const uint64_t N = off[nd];
  printf("vocab: n_info %u nodes %u edge buckets (2 slots each) %u (mask %x) tab bytes %zu; corpus %llu docs %u\n", hv.n_info, hv.n_nodes, hv.edge_mask + 1,
         hv.edge_mask, hv.tab.size() * 8, (unsigned long long)N, nd);
  // hash buckets of two slots {key | 4-bit child filter << 28, value}; link format {x, y | child filter, best depth}

  // child-byte masks per node (from the edge hash: all edges into depth >= 3)
  std::vector<uint64_t> cmask(hv.n_nodes, 0);
  std::vector<uint32_t> nchild(hv.n_nodes, 0);
  uint64_t nedges = 0;
  for (uint32_t s = 0; s < 2 * (hv.edge_mask + 1); s++) {
    const uint2 e = hv.tab[s];
    if (e.x == kNone) continue;
    cmask[(e.x & kKeyMask) >> 8] |= 1ull << (e.x & 63u);
    nchild[(e.x & kKeyMask) >> 8]++;
    nedges++;
  }
  {
    uint64_t hist[8] = {0};
    for (uint32_t n = 0; n < hv.n_nodes; n++) { uint32_t k = nchild[n]; hist[k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 2 : k <= 4 ? 3 : k <= 8 ? 4 : k <= 16 ? 5 : k <= 32 ? 6 : 7]++; }
    printf("edges %llu; nodes by #children 0:%llu 1:%llu 2:%llu 3-4:%llu 5-8:%llu 9-16:%llu 17-32:%llu >32:%llu\n", (unsigned long long)nedges,
           (unsigned long long)hist[0], (unsigned long long)hist[1], (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4],
           (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7]);
  }
  const uint2* tab = hv.tab.data();
  const uint2* direct = tab + hv.direct_off / 8;
  const uint2* link = tab + hv.link_off / 8;
  const int Lmax = (int)hv.max_len;

  // variants: 0 = has-children bit only (no filter); 1 = the filters stored in the tables (what the kernel does: 32 bits behind a
  // link-format entry, 4 bits in a hash slot); 2 = ideal 64-bit filter everywhere
  std::vector<uint64_t> sigs; uint64_t sig_bad = 0;
  for (int variant = 0; variant < 4; variant++) {   // 3: filter only in the link-format entries
    Stats st;
    std::vector<uint32_t> round_hist(128, 0);
    for (uint32_t d = 0; d < nd; d++) {
      const uint64_t b0 = off[d], e0 = off[d + 1];
      for (uint64_t begin = b0; begin < e0; begin += SEG) {
        const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
        const uint8_t* t = text + begin;
        auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
        const int ntask = std::min(NPOS, dl);
        const int nwalkpos = dl <= NPOS ? ntask - 1 : ntask;
        const int run = (std::max(nwalkpos, 0) + 63) >> 6;
        int wave_rounds = 0;
        for (int lane = 0; lane < 64; lane++) {
          int pos = lane * run;
          const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
          int rounds = 0;
          int depth = 0; uint32_t node = 0;
          bool first = true;
          while (pos < end) {
            const int limit = std::min(dl - pos, Lmax);
            const int rounds_before = rounds;
            // SET gather
            const uint2* e;
            if (!first && depth >= 3) e = link + 2 * (size_t)node;
            else e = direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
            // note: the kernel takes the link of the node the previous walk ENDED on and its depth-1
            rounds++; st.set++;
            uint32_t src = e[0].x;
            uint32_t filt = e[1].x;                                  // child filter of the node the entry leads to
            depth = (int)((src >> 23) & 63u);
            node = src & kNodeMask;
            int bestlen = (int)e[1].y;
            bool from_set = true;
            bool go = (src & kHasChildren) != 0 && depth < limit;
            while (go) {
              const uint32_t c = at(pos + depth);
              if (variant == 1) { if (from_set ? !((filt >> (c & 31u)) & 1u) : !((filt >> (c & 3u)) & 1u)) { st.filtered++; break; } }
              else if (variant == 3) { if (from_set && !((filt >> (c & 31u)) & 1u)) { st.filtered++; break; } }
              else if (variant == 2) { if (!((cmask[node] >> (c & 63u)) & 1ull)) { st.filtered++; break; } }
              const uint32_t key = (node << 8) | c;
              uint32_t h = edge_hash(node, c) >> hv.edge_shift;
              bool hit = false;
              for (;;) {                                                  // one gather = one bucket of two slots
                rounds++;
                const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1];
                if ((s0.x & kKeyMask) == key) { hit = true; st.hit++; src = s0.y; filt = s0.x >> 28; from_set = false; break; }
                if ((s1.x & kKeyMask) == key) { hit = true; st.hit++; src = s1.y; filt = s1.x >> 28; from_set = false; break; }
                if (s1.x == kNone) { st.miss++; break; }
                st.again++;
                h = (h + 1) & hv.edge_mask;
              }
              if (!hit) break;
              depth++;
              node = src & kNodeMask;
              if (node < hv.n_info) bestlen = depth;
              go = (src & kHasChildren) != 0 && depth < limit;
            }
            // every variant must end every position in the same state (the filter only suppresses probes that cannot hit)
            { const uint64_t sig = ((uint64_t)node << 16) | ((uint64_t)depth << 8) | (uint64_t)bestlen; if (variant == 0) sigs.push_back(sig); else if (sigs[st.pos] != sig) { if (++sig_bad < 5) fprintf(stderr, "STATE MISMATCH variant %d position %llu\n", variant, (unsigned long long)st.pos); } }
            if (first) { st.first_n++; st.first_g += rounds - rounds_before; }
            first = false;
            pos++;
            st.pos++;
          }
          wave_rounds = std::max(wave_rounds, rounds);
          st.lane_sum += rounds;
        }
        st.rounds += wave_rounds;
        st.waves++;
        round_hist[std::min(wave_rounds, 127)]++;
      }
    }
    const double g = (double)(st.set + st.hit + st.again + st.miss);
    printf("variant %d: positions %llu gathers/pos %.3f (set %.3f hit %.3f again %.3f miss %.3f) filtered/pos %.3f | rounds/wave %.2f  lane-mean %.2f  efficiency %.2f\n",
           variant, (unsigned long long)st.pos, g / st.pos, (double)st.set / st.pos, (double)st.hit / st.pos, (double)st.again / st.pos,
           (double)st.miss / st.pos, (double)st.filtered / st.pos, (double)st.rounds / st.waves, (double)st.lane_sum / st.waves / 64.0,
           (double)st.lane_sum / 64.0 / st.rounds);
    printf("   state mismatches vs variant 0: %llu\n", (unsigned long long)sig_bad);
    printf("   first position of a run: %.2f gathers (others %.2f)\n", (double)st.first_g / st.first_n, (g - st.first_g) / (st.pos - st.first_n));
  }

  // lockstep model with run splitting: every `every` rounds a lane whose run is finished takes the back half of what its right
  // (mode 1) or right-else-left (mode 2) neighbour has not started yet; the stolen run starts from the direct map (no link).
  // Result on the englishcode-32000 shape: 30.6 rounds per wavefront without, 30.3 with (and 8 % more gathers): the rounds of a
  // wavefront are set by ONE deep walk (a multi-word token, one byte per round), which no hand-out can split.
  for (int mode = 0; mode < 3; mode++) for (int every : {1, 4}) {
    if (mode == 0 && every != 1) continue;
    uint64_t rounds_total = 0, waves = 0, gathers = 0, steals = 0;
    for (uint32_t d = 0; d < nd; d++) {
      const uint64_t b0 = off[d], e0 = off[d + 1];
      for (uint64_t begin = b0; begin < e0; begin += SEG) {
        const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
        const uint8_t* t = text + begin;
        auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
        const int ntask = std::min(NPOS, dl);
        const int nwalkpos = dl <= NPOS ? ntask - 1 : ntask;
        const int run = (std::max(nwalkpos, 0) + 63) >> 6;
        struct L { int pos, end, left, depth; uint32_t node; bool first; } ln[64];
        // gathers the walk at `pos` needs given the state the previous position of the run ended in
        auto cost = [&](L& l) {
          const int pos = l.pos, limit = std::min(dl - pos, Lmax);
          const uint2* e = (!l.first && l.depth >= 3) ? link + 2 * (size_t)l.node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
          int rounds = 1;
          uint32_t src = e[0].x, filt = e[1].x;
          int depth = (int)((src >> 23) & 63u); uint32_t node = src & kNodeMask;
          bool from_set = true, go = (src & kHasChildren) != 0 && depth < limit;
          while (go) {
            const uint32_t c = at(pos + depth);
            if (from_set ? !((filt >> (c & 31u)) & 1u) : !((filt >> (c & 3u)) & 1u)) break;
            const uint32_t key = (node << 8) | c;
            uint32_t h = edge_hash(node, c) >> hv.edge_shift;
            bool hit = false;
            for (;;) { rounds++; const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1];
              if ((s0.x & kKeyMask) == key) { hit = true; src = s0.y; filt = s0.x >> 28; from_set = false; break; }
              if ((s1.x & kKeyMask) == key) { hit = true; src = s1.y; filt = s1.x >> 28; from_set = false; break; }
              if (s1.x == kNone) break; h = (h + 1) & hv.edge_mask; }
            if (!hit) break;
            depth++; node = src & kNodeMask; go = (src & kHasChildren) != 0 && depth < limit;
          }
          l.depth = depth; l.node = node; l.first = false; l.left = rounds;
        };
        for (int lane = 0; lane < 64; lane++) {
          L& l = ln[lane];
          l.pos = lane * run; l.end = std::max(std::min(lane * run + run, nwalkpos), 0); l.first = true; l.depth = 0; l.node = 0; l.left = 0;
          if (l.pos < l.end) cost(l);
        }
        int tnow = 0;
        for (;;) {
          bool any = false;
          for (int lane = 0; lane < 64; lane++) any |= ln[lane].left > 0;
          if (!any) break;
          tnow++;
          for (int lane = 0; lane < 64; lane++) {
            L& l = ln[lane];
            if (l.left == 0) continue;
            gathers++;
            if (--l.left == 0) { l.pos++; if (l.pos < l.end) cost(l); }
          }
          if (mode != 0 && tnow % every == 0) {
            bool idle[64]; int vp[64], ve[64];
            for (int lane = 0; lane < 64; lane++) { idle[lane] = ln[lane].left == 0; vp[lane] = ln[lane].pos; ve[lane] = ln[lane].end; }
            bool taken[64] = {false};
            for (int lane = 0; lane < 64; lane++) {
              if (!idle[lane]) continue;
              int cand[2] = {lane + 1, mode == 2 ? lane - 1 : 64};
              for (int k = 0; k < 2; k++) {
                const int v = cand[k];
                if (v < 0 || v > 63 || idle[v] || taken[v]) continue;
                const int rem = ve[v] - vp[v] - 1;               // positions the victim has not started
                if (rem < 1) continue;
                const int m = ve[v] - (rem + 1) / 2;
                ln[lane].pos = m; ln[lane].end = ve[v]; ln[lane].first = true; cost(ln[lane]);
                ln[v].end = m; taken[v] = true; steals++;
                break;
              }
            }
          }
        }
        rounds_total += tnow; waves++;
      }
    }
    printf("lockstep mode %d (0 static, 1 steal right, 2 steal right/left) every %d rounds: rounds/wave %.2f gathers/wave %.1f steals/wave %.2f\n", mode, every,
           (double)rounds_total / waves, (double)gathers / waves, (double)steals / waves);
  }

  // skip edges (path compression): an edge carries up to K bytes of the non-accepting unary chain below its child and the node at
  // the chain's end; a probe that hits and finds those bytes in the text advances 1 + L levels in one round.  Collisions are
  // modelled with the existing two-slot buckets (a 16-byte single-slot table of twice the buckets behaves about the same).
  {
    std::vector<uint32_t> only_byte(hv.n_nodes, 0), only_child(hv.n_nodes, kNone);
    for (uint32_t sl = 0; sl < 2 * (hv.edge_mask + 1); sl++) {
      const uint2 e = hv.tab[sl];
      if (e.x == kNone) continue;
      const uint32_t parent = (e.x & kKeyMask) >> 8;
      if (nchild[parent] == 1) { only_byte[parent] = e.x & 0xFFu; only_child[parent] = e.y; }
    }
    for (int K : {0, 2, 3, 4, 8}) {
      uint64_t rounds_total = 0, waves = 0, gathers = 0, jumps = 0, jumped = 0;
      for (uint32_t d = 0; d < nd; d++) {
        const uint64_t b0 = off[d], e0 = off[d + 1];
        for (uint64_t begin = b0; begin < e0; begin += SEG) {
          const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
          const uint8_t* t = text + begin;
          auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
          const int ntask = std::min(NPOS, dl);
          const int nwalkpos = dl <= NPOS ? ntask - 1 : ntask;
          const int run = (std::max(nwalkpos, 0) + 63) >> 6;
          int wave_rounds = 0;
          for (int lane = 0; lane < 64; lane++) {
            int pos = lane * run;
            const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
            int rounds = 0, depth = 0; uint32_t node = 0; bool first = true;
            while (pos < end) {
              const int limit = std::min(dl - pos, Lmax);
              const uint2* e = (!first && depth >= 3) ? link + 2 * (size_t)node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
              rounds++;
              uint32_t src = e[0].x, filt = e[1].x;
              depth = (int)((src >> 23) & 63u); node = src & kNodeMask;
              bool from_set = true, go = (src & kHasChildren) != 0 && depth < limit;
              while (go) {
                const uint32_t c = at(pos + depth);
                if (from_set ? !((filt >> (c & 31u)) & 1u) : !((filt >> (c & 3u)) & 1u)) break;
                const uint32_t key = (node << 8) | c;
                uint32_t h = edge_hash(node, c) >> hv.edge_shift;
                bool hit = false;
                for (;;) { rounds++; const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1];
                  if ((s0.x & kKeyMask) == key) { hit = true; src = s0.y; filt = s0.x >> 28; from_set = false; break; }
                  if ((s1.x & kKeyMask) == key) { hit = true; src = s1.y; filt = s1.x >> 28; from_set = false; break; }
                  if (s1.x == kNone) break; h = (h + 1) & hv.edge_mask; }
                if (!hit) break;
                depth++; node = src & kNodeMask;
                // the chain below `node`: non-accepting nodes with one child each
                int L = 0; uint32_t cn = node, csrc = src;
                while (L < K && cn >= hv.n_info && nchild[cn] == 1 && depth + L < limit && at(pos + depth + L) == only_byte[cn]) { csrc = only_child[cn]; cn = csrc & kNodeMask; L++; }
                if (L > 0) { jumps++; jumped += L; depth += L; node = cn; src = csrc; filt = 0xF; /* (the end node's filter rides in the slot too) */ }
                go = (src & kHasChildren) != 0 && depth < limit;
              }
              first = false; pos++;
            }
            wave_rounds = std::max(wave_rounds, rounds); gathers += rounds;
          }
          rounds_total += wave_rounds; waves++;
        }
      }
      printf("skip edges, up to %2d chain bytes per slot: rounds/wave %.2f gathers/wave %.1f jumps/wave %.1f (%.2f levels each)\n", K, (double)rounds_total / waves,
             (double)gathers / waves, (double)jumps / waves, jumps ? (double)jumped / jumps : 0.0);
    }
  }

  // halo sharing: the 40 halo positions of a segment are the first 40 of the next one; if the wavefront of the next segment (same
  // workgroup of WAVES = 4 consecutive segments, same document) hands them over through LDS, this wavefront walks 256 positions instead
  // of 296 (runs of 4 instead of 5 per lane)
  {
    uint64_t rounds_now = 0, rounds_shared = 0, waves = 0, sharing = 0, g = 0;
    for (uint32_t d = 0; d < nd; d++) {
      const uint64_t b0 = off[d], e0 = off[d + 1];
      for (uint64_t begin = b0; begin < e0; begin += SEG, g++) {
        const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
        const uint8_t* t = text + begin;
        auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
        const bool share = (g & 3) != 3 && begin + SEG < e0;
        auto wave_rounds = [&](int npos) {
          const int ntask = std::min(npos, dl);
          const int nwalkpos = (dl <= npos && npos == NPOS) ? ntask - 1 : ntask;
          const int run = (std::max(nwalkpos, 0) + 63) >> 6;
          int wr = 0;
          for (int lane = 0; lane < 64; lane++) {
            const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
            int rounds = 0, depth = 0; uint32_t node = 0; bool first = true;
            for (int pos = lane * run; pos < end; pos++) {
              const int limit = std::min(dl - pos, Lmax);
              const uint2* e = (!first && depth >= 3) ? link + 2 * (size_t)node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
              rounds++;
              uint32_t src = e[0].x, filt = e[1].x;
              depth = (int)((src >> 23) & 63u); node = src & kNodeMask;
              bool from_set = true, go = (src & kHasChildren) != 0 && depth < limit;
              while (go) {
                const uint32_t c = at(pos + depth);
                if (from_set ? !((filt >> (c & 31u)) & 1u) : !((filt >> (c & 3u)) & 1u)) break;
                const uint32_t key = (node << 8) | c;
                uint32_t h = edge_hash(node, c) >> hv.edge_shift;
                bool hit = false;
                for (;;) { rounds++; const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1];
                  if ((s0.x & kKeyMask) == key) { hit = true; src = s0.y; filt = s0.x >> 28; from_set = false; break; }
                  if ((s1.x & kKeyMask) == key) { hit = true; src = s1.y; filt = s1.x >> 28; from_set = false; break; }
                  if (s1.x == kNone) break; h = (h + 1) & hv.edge_mask; }
                if (!hit) break;
                depth++; node = src & kNodeMask; go = (src & kHasChildren) != 0 && depth < limit;
              }
              first = false;
            }
            wr = std::max(wr, rounds);
          }
          return wr;
        };
        const int full = wave_rounds(NPOS);
        rounds_now += full;
        rounds_shared += share ? wave_rounds(SEG) : full;
        sharing += share; waves++;
      }
    }
    printf("halo sharing: %.1f %% of the wavefronts take their halo from the next one: rounds/wave %.2f -> %.2f\n", 100.0 * sharing / waves, (double)rounds_now / waves,
           (double)rounds_shared / waves);
  }

  // dynamic hand-out: runs of r consecutive positions, a lane that finishes takes the next unassigned run (from scratch)
  for (int variant = 0; variant < 2; variant++) for (int r = 1; r <= 6; r++) {
    uint64_t rounds_total = 0, waves = 0, gathers = 0, npos = 0;
    std::vector<int> cost;   // gathers per run
    for (uint32_t d = 0; d < nd; d++) {
      const uint64_t b0 = off[d], e0 = off[d + 1];
      for (uint64_t begin = b0; begin < e0; begin += SEG) {
        const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
        const uint8_t* t = text + begin;
        auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
        const int ntask = std::min(NPOS, dl);
        const int nwalkpos = dl <= NPOS ? ntask - 1 : ntask;
        cost.clear();
        for (int a = 0; a < nwalkpos; a += r) {
          int rounds = 0, depth = 0; uint32_t node = 0; bool first = true;
          for (int pos = a; pos < std::min(a + r, nwalkpos); pos++) {
            const int limit = std::min(dl - pos, Lmax);
            const uint2* e = (!first && depth >= 3) ? link + 2 * (size_t)node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
            rounds++;
            uint32_t src = e[0].x;
            depth = (int)((src >> 23) & 63u); node = src & kNodeMask;
            bool go = (src & kHasChildren) != 0 && depth < limit;
            while (go) {
              const uint32_t c = at(pos + depth);
              if (variant != 0 && !((cmask[node] >> (c & 63u)) & 1ull)) break;
              const uint32_t key = (node << 8) | c;
              uint32_t h = edge_hash(node, c) >> hv.edge_shift;
              bool hit = false;
              for (;;) { rounds++; const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1]; if ((s0.x & kKeyMask) == key) { hit = true; src = s0.y; break; } if ((s1.x & kKeyMask) == key) { hit = true; src = s1.y; break; } if (s1.x == kNone) break; h = (h + 1) & hv.edge_mask; }
              if (!hit) break;
              depth++; node = src & kNodeMask; go = (src & kHasChildren) != 0 && depth < limit;
            }
            first = false; npos++;
          }
          cost.push_back(rounds); gathers += rounds;
        }
        // list scheduling on 64 lanes
        int busy[64] = {0}; size_t next = 0; int tnow = 0;
        // event simulation: at each round, lanes with busy==0 take next run
        size_t remaining = cost.size();
        int active = 0;
        while (next < cost.size() || active > 0) {
          active = 0;
          for (int l = 0; l < 64; l++) { if (busy[l] == 0 && next < cost.size()) busy[l] = cost[next++]; if (busy[l] > 0) { busy[l]--; active++; } }
          if (active > 0) tnow++;
          for (int l = 0, a2 = 0; l < 64; l++) a2 += busy[l] > 0, active = std::max(active, a2);
          active = 0; for (int l = 0; l < 64; l++) active += busy[l] > 0;
        }
        (void)remaining;
        rounds_total += tnow; waves++;
      }
    }
    printf("dynamic variant %d run %d: gathers/pos %.3f rounds/wave %.2f\n", variant, r, (double)gathers / npos, (double)rounds_total / waves);
  }

  // static runs, other segment sizes (rounds per 256 bytes of text)
  for (int variant = 0; variant < 2; variant++) for (int seg : {192, 256, 384, 512, 768, 1024}) {
    uint64_t rounds_total = 0, lane_sum = 0, bytes = 0;
    for (uint32_t d = 0; d < nd; d++) {
      const uint64_t b0 = off[d], e0 = off[d + 1];
      for (uint64_t begin = b0; begin < e0; begin += seg) {
        const int dl = (int)std::min<uint64_t>(e0 - begin, 1 << 20);
        const uint8_t* t = text + begin;
        auto at = [&](int i) -> uint32_t { return i < dl ? t[i] : 0u; };
        const int npos = seg + 40;
        const int ntask = std::min(npos, dl);
        const int nwalkpos = dl <= npos ? ntask - 1 : ntask;
        const int run = (std::max(nwalkpos, 0) + 63) >> 6;
        int wave_rounds = 0;
        for (int lane = 0; lane < 64; lane++) {
          const int end = std::max(std::min(lane * run + run, nwalkpos), 0);
          int rounds = 0, depth = 0; uint32_t node = 0; bool first = true;
          for (int pos = lane * run; pos < end; pos++) {
            const int limit = std::min(dl - pos, Lmax);
            const uint2* e = (!first && depth >= 3) ? link + 2 * (size_t)node : direct + 2 * (size_t)(at(pos) | (at(pos + 1) << 8));
            rounds++;
            uint32_t src = e[0].x;
            depth = (int)((src >> 23) & 63u); node = src & kNodeMask;
            bool go = (src & kHasChildren) != 0 && depth < limit;
            while (go) {
              const uint32_t c = at(pos + depth);
              if (variant != 0 && !((cmask[node] >> (c & 63u)) & 1ull)) break;
              const uint32_t key = (node << 8) | c;
              uint32_t h = edge_hash(node, c) >> hv.edge_shift;
              bool hit = false;
              for (;;) { rounds++; const uint2 s0 = tab[2 * (size_t)h], s1 = tab[2 * (size_t)h + 1]; if ((s0.x & kKeyMask) == key) { hit = true; src = s0.y; break; } if ((s1.x & kKeyMask) == key) { hit = true; src = s1.y; break; } if (s1.x == kNone) break; h = (h + 1) & hv.edge_mask; }
              if (!hit) break;
              depth++; node = src & kNodeMask; go = (src & kHasChildren) != 0 && depth < limit;
            }
            first = false;
          }
          wave_rounds = std::max(wave_rounds, rounds); lane_sum += rounds;
        }
        rounds_total += wave_rounds; bytes += std::min(dl, seg);
      }
    }
    printf("static variant %d seg %d: rounds per 256 B %.2f (lane mean %.2f, efficiency %.2f)\n", variant, seg, 256.0 * rounds_total / bytes,
           256.0 * lane_sum / 64 / bytes, (double)lane_sum / 64 / rounds_total);
  }
  tm_free(text); tm_free(img);
  return 0;
}
