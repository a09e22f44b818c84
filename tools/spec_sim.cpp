// spec_sim.cpp — host model of a SPECULATIVE chain walker (development aid, decides whether the design is worth building):
// one lane per segment follows the ONE chain of its segment serially, started W bytes before the segment at an arbitrary byte in
// state fd = 0; the chain is usable when the true entry state of the segment lies on it.  Measures, on the synthetic corpus and
// the real tables: (1) how often it does not (mismatch rate by W), (2) gathers per token step and the rounds a wavefront of 64
// such lanes needs in lockstep (one token step per lane per iteration; walks of a step serial in one slot, or in parallel slots).
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc -I oracle tools/spec_sim.cpp -o /tmp/spec_sim -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Loracle -ltm_oracle -Wl,-rpath,$PWD/tokenmonster_amd -Wl,-rpath,$PWD/oracle
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_device.h"
#include "tm_pipeline.h"
#include "tm_oracle.h"

using namespace tmh;

static HostVocab hv;
static tmo_vocab* ov;
static const uint2 *tab, *direct_map;
static int Lmax, OFF;

// gathers of a from-scratch trie walk over s[0..n): direct map (1) + one per probe; *depth_out = trie depth reached
template <class At>
static int walk_gathers(At at, int n, int* depth_out, int start_depth = -1, uint32_t start_node = 0, uint32_t start_filt = 0, bool start_go = false) {
  int rounds = 0, depth; uint32_t node, src, filt; bool from_set = true, go;
  const int limit = std::min(n, Lmax);
  if (start_depth < 0) {
    if (n < 2) { *depth_out = n; return 0; }
    const uint2* e = direct_map + 2 * (size_t)(at(0) | (at(1) << 8));
    rounds = 1; src = e[0].x; filt = e[1].x; depth = (int)((src >> 23) & 63u); node = src & kNodeMask;
    go = (src & kHasChildren) != 0 && depth < limit;
  } else { depth = start_depth; node = start_node; filt = start_filt; go = start_go && depth < limit; }
  while (go) {
    const uint32_t c = at(depth);
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
  *depth_out = depth;
  return rounds;
}

struct St { int i, fd; uint32_t index, length; bool found; };
struct Cost { int nwalk = 0, g[6] = {0}; bool b[6] = {false}; int ids = 0; };   // gathers per walk of this step (b: a forward-delete walk, continues its main walk)

static const uint8_t* T; static int DL;     // current document
static uint32_t at_doc(int i) { return i < DL ? T[i] : 0u; }

static void match_at(int p, uint32_t* idx, uint32_t* len, bool* found, int* gathers) {
  const int rem = DL - p;
  *found = tmo_longest(ov, T + p, (size_t)std::min(rem, Lmax), idx, len) != 0;
  int d; *gathers = rem >= 2 ? walk_gathers([&](int k) { return at_doc(p + k); }, rem, &d) : 0;
}

// one token step from state s (go :1051-1276 as restated in oracle/tm_oracle.c); s.found/index/length valid on entry
static void step(St& s, Cost& c, const uint8_t* bb) {
  c = Cost();
  if (!s.found) { s.i++; s.fd = 0; c.ids = 1; int g; match_at(s.i, &s.index, &s.length, &s.found, &g); if (s.i < DL) { c.g[c.nwalk++] = g; } return; }
  const Row& O = hv.rows[s.index];
  const int len = (int)s.length, fd = s.fd, i = s.i;
  const int len1 = (int)(O.w & 63u), len2 = (int)((O.w >> 6) & 63u);
  const int i1 = i + len;
  uint32_t xs[6] = {0}, ls[6] = {0}; int sc[6]; for (int k = 0; k < 6; k++) sc[k] = NOSCORE;
  int flens[3] = {len, len1 - fd, len2 - fd};
  bool looked = false;
  if (i1 < DL && ((O.w & (1u << 21)) == 0 || bb[T[i1]] != 12)) {
    looked = true;
    const int nk = len1 == 0 ? 1 : (len2 == 0 ? 2 : 3);
    // scores via the oracle's formula need flags: take them from the oracle rows through a tiny re-derivation: use tmo_* ? simpler:
    // the kernel's folded constants (Row) — same as transition<FD>
    const int fpart[3] = {len + (int)(O.x >> kRowIdBits) - fd * (100 + (int)((O.w >> 18) & 1u)), (int)(O.y >> kRowIdBits) - fd * (101 + (int)((O.w >> 19) & 1u)),
                          (int)(O.z >> kRowIdBits) - fd * (101 + (int)((O.w >> 20) & 1u))};
    const uint32_t fbw[3] = {((O.w >> 12) & 1u) | (((O.w >> 15) & 1u) << 8), ((O.w >> 13) & 1u) | (((O.w >> 16) & 1u) << 8), ((O.w >> 14) & 1u) | (((O.w >> 17) & 1u) << 8)};
    for (int k = 0; k < nk; k++) {
      const int ik = i + flens[k];
      uint32_t xk, lk; bool fk; int g;
      match_at(ik, &xk, &lk, &fk, &g);
      c.g[c.nwalk] = g; c.b[c.nwalk++] = false;
      if (!fk) continue;
      const uint32_t v = hv.vals[xk];
      const uint32_t f5 = v >> 27, snw = (v >> 22) & 31u;
      const uint32_t nb = bb[at_doc(ik + (int)lk)];
      auto score = [&](uint32_t vv, int ll, uint32_t nbb, bool bvar) {
        const uint32_t f = vv >> 27, nw = (vv >> 22) & 31u;
        const int send = f & 1, sbegl = (f >> 1) & 1, sbegs = ((f >> 2) & 1) & ~(((f >> 1) & 1) & hv.spl_hint), sbegc = (f >> 3) & 1, sall = (f >> 4) & 1;
        const int S = ll + sall + std::max((int)nw - 1, 0) + (bvar ? 0 : sbegs) + (int)((nbb >> 2) & 1u) + ((int)nw + (int)(nbb >> 3)) * 100 - (send & (int)(nbb & 1u)) * 3;
        const int pen = bvar ? (int)(fbw[k] & 1u) * 103 + (int)((fbw[k] >> 8) & 1u & sbegc) * 100 + 1 : (int)(fbw[k] & 1u & sbegl) * 103 + (int)((fbw[k] >> 8) & 1u & sbegc) * 100;
        int r = fpart[k] + S - pen;
        if (k > 0) { const int BL = flens[k] + ll; r -= (BL < len ? 100 : 0) + (BL == len ? 10000 : 0); }
        return r;
      };
      sc[k] = score(v, (int)lk, nb, false); xs[k] = xk; ls[k] = lk;
      const bool hint = ((f5 >> 2) & 1u) | (hv.spl_hint ^ 1u);
      if (hv.delete_id != TM_NONE && hv.bstart != kNone && (f5 & 2u) && nb == 1 && snw == 0) {
        const int remk = DL - ik, m = std::max(std::min(remk, Lmax - OFF), 0);
        uint8_t lil[48]; memset(lil, 0, sizeof lil); lil[0] = 32; memcpy(lil + OFF, T + ik, (size_t)m);
        uint32_t xb, lb; tmo_longest(ov, lil, (size_t)(m + OFF), &xb, &lb);
        if (hint) {   // the walk is only issued when the token's hint bit allows it: one space-prefix link gather + what continues behind ' '+match
          int d; walk_gathers([&](int q) { return (uint32_t)lil[q]; }, m + OFF, &d);
          c.g[c.nwalk] = 1 + std::max(0, d - ((int)lk + OFF)) ; c.b[c.nwalk++] = true;
        }
        if ((int)lb > (int)lk + 1) {
          const int lbb = (int)lb - OFF;
          sc[3 + k] = score(hv.vals[xb], lbb, bb[at_doc(ik + lbb)], true); xs[3 + k] = xb; ls[3 + k] = (uint32_t)lbb;
        }
      }
    }
  }
  int best = NOSCORE, bk = -1;
  for (int k = 0; k < 6; k++) if (sc[k] > best) { best = sc[k]; bk = k; }
  (void)looked;
  if (bk < 0) {
    s.i = i1; s.fd = 0; c.ids = 1; int g; match_at(s.i, &s.index, &s.length, &s.found, &g);
    if (!looked && s.i < DL) { c.g[c.nwalk] = g; c.b[c.nwalk++] = false; }   // fast exit: the match at i1 is still needed (the looked case did it: k = 0)
    return;
  }
  const int k = bk % 3;
  s.i = i + flens[k]; s.fd = bk >= 3; s.index = xs[bk]; s.length = ls[bk]; s.found = true; c.ids = 1 + s.fd;
}

int main(int argc, char** argv) {
  const uint32_t kind = argc > 1 ? atoi(argv[1]) : TM_KIND_ENGLISHCODE;
  const uint32_t vsize = argc > 2 ? atoi(argv[2]) : 32000;
  const uint64_t nbytes = argc > 3 ? atoll(argv[3]) : (8ull << 20);
  const uint32_t capcode = argc > 4 ? atoi(argv[4]) : 2;
  const int S = argc > 5 ? atoi(argv[5]) : 256;
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(kind, vsize, capcode, 1, 3, 0x544D0002, 0, &img, &img_n) != 0) { fprintf(stderr, "synth_vocab failed\n"); return 1; }
  if (parse_vocab(img, img_n, hv) != 0) { fprintf(stderr, "parse failed: %s\n", last_error()); return 1; }
  ov = tmo_load(img, img_n);
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint32_t nd = 0; uint64_t nb = 0;
  tm_synth_corpus(kind, 0x434F5250 + 2, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nb);
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), nd, capcode, 1, 0, &text, off.data()) != 0) { fprintf(stderr, "normalize failed\n"); return 1; }
  tab = hv.tab.data(); direct_map = tab + hv.direct_off / 8; Lmax = (int)hv.max_len; OFF = (int)hv.off;
  const uint8_t* bb = hv.begin_byte;
  printf("vocab n_info %u; corpus %llu bytes %u docs; segment %d\n", hv.n_info, (unsigned long long)off[nd], nd, S);

  // true chains: mark[pos] bit fd
  std::vector<uint8_t> mark(off[nd] + 64, 0);
  uint64_t true_steps = 0, true_ids = 0, walks = 0, walk_g = 0, bw = 0, bwg = 0, nw_hist[7] = {0};
  for (uint32_t d = 0; d < nd; d++) {
    T = text + off[d]; DL = (int)(off[d + 1] - off[d]);
    if (DL == 0) continue;
    St s{0, 0, 0, 0, false}; int g; match_at(0, &s.index, &s.length, &s.found, &g);
    while (s.i < DL) {
      mark[off[d] + s.i] |= (uint8_t)(1u << s.fd);
      Cost c; step(s, c, bb); true_steps++; true_ids += c.ids;
      int n = 0; for (int k = 0; k < c.nwalk; k++) { if (c.b[k]) { bw++; bwg += c.g[k]; } else { walks++; walk_g += c.g[k]; n++; } }
      nw_hist[std::min(n, 6)]++;
    }
  }
  printf("true chain: %.3f bytes/step, %.3f ids/step; plain walks/step %.3f (%.2f gathers each), fd walks/step %.3f (%.2f gathers each); steps by #plain walks 0:%.3f 1:%.3f 2:%.3f 3:%.3f\n",
         (double)off[nd] / true_steps, (double)true_ids / true_steps, (double)walks / true_steps, (double)walk_g / walks, (double)bw / true_steps, bw ? (double)bwg / bw : 0.0,
         (double)nw_hist[0] / true_steps, (double)nw_hist[1] / true_steps, (double)nw_hist[2] / true_steps, (double)nw_hist[3] / true_steps);

  for (int W : {16, 32, 48, 64, 96}) for (int snap = 0; snap < 2; snap++) {
    // lanes: every segment of every document in order; a wavefront = 64 consecutive lanes
    struct Lane { uint32_t d; int s0, sb, se; };
    std::vector<Lane> lanes;
    for (uint32_t d = 0; d < nd; d++) {
      const int dl = (int)(off[d + 1] - off[d]);
      for (int b = 0; b < dl; b += S) {
        int s0 = std::max(0, b - W);
        if (snap && s0 > 0) { const uint8_t* t = text + off[d]; int q = s0; while (q < b && t[q] != ' ') q++; if (q < b) s0 = q; }   // start on the first space of the warm-up window
        lanes.push_back({d, s0, b, std::min(b + S, dl)});
      }
    }
    uint64_t spec = 0, mism = 0, iters = 0, rounds_ser = 0, rounds_par = 0, lane_steps = 0, lane_g = 0, useful_steps = 0;
    for (size_t w0 = 0; w0 < lanes.size(); w0 += 64) {
      const int nl = (int)std::min<size_t>(64, lanes.size() - w0);
      St st[64]; bool act[64]; bool hit[64]; int first_g[64];
      for (int l = 0; l < nl; l++) {
        const Lane& L = lanes[w0 + l];
        T = text + off[L.d]; DL = (int)(off[L.d + 1] - off[L.d]);
        st[l] = St{L.s0, 0, 0, 0, false}; match_at(L.s0, &st[l].index, &st[l].length, &st[l].found, &first_g[l]);
        act[l] = true; hit[l] = L.s0 == 0;
      }
      for (int it = 0;; it++) {
        int mser = 0, mpar = 0; bool any = false;
        for (int l = 0; l < nl; l++) {
          if (!act[l]) continue;
          const Lane& L = lanes[w0 + l];
          T = text + off[L.d]; DL = (int)(off[L.d + 1] - off[L.d]);
          if (st[l].i >= L.se) { act[l] = false; continue; }
          any = true;
          if (!hit[l] && st[l].i >= L.sb) {      // first state at or behind the segment start: is it the true chain's?
            // the true entry state is the first marked state >= sb; the spec chain is usable iff it passes through it
            int q = L.sb; while (q < DL && mark[off[L.d] + q] == 0) q++;
            if (st[l].i == q && (mark[off[L.d] + q] >> st[l].fd) & 1) hit[l] = true;
            else if (st[l].i > q || (st[l].i == q)) { hit[l] = true; mism++; }       // passed it without meeting it (counted once)
          }
          if (st[l].i >= L.sb) useful_steps++;
          Cost c; step(st[l], c, bb);
          int ser = 1 + (it == 0 ? first_g[l] : 0), par = 0, last_main = 0;
          for (int k = 0; k < c.nwalk; k++) { ser += c.g[k]; if (c.b[k]) { par = std::max(par, last_main + c.g[k]); } else { last_main = c.g[k]; par = std::max(par, c.g[k]); } }
          par += 1 + (it == 0 ? first_g[l] : 0);
          lane_steps++; lane_g += ser;
          mser = std::max(mser, ser); mpar = std::max(mpar, par);
        }
        if (!any) break;
        iters++; rounds_ser += mser; rounds_par += mpar;
      }
      for (int l = 0; l < nl; l++) if (lanes[w0 + l].s0 != 0) spec++;
    }
    const double nwave = (double)((lanes.size() + 63) / 64);
    printf("W %3d%s: speculative segments %llu, mismatches %llu (%.4f %%) | per wavefront: iterations %.1f, rounds serial-slot %.1f (%.2f / iteration), parallel-slots %.1f (%.2f) | lane steps %.1f of %.1f slots (efficiency %.2f), useful %.2f; gathers / lane step %.2f\n",
           W, snap ? " snap" : "     ", (unsigned long long)spec, (unsigned long long)mism, 100.0 * mism / std::max<uint64_t>(spec, 1), iters / nwave, rounds_ser / nwave, (double)rounds_ser / iters,
           rounds_par / nwave, (double)rounds_par / iters, lane_steps / nwave, 64.0 * iters / nwave, (double)lane_steps / (64.0 * iters), (double)useful_steps / lane_steps, (double)lane_g / lane_steps);
  }
  return 0;
}
