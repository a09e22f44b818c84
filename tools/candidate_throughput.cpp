// candidate_throughput.cpp — measurement for SURVEY 8f #3: the trainvocab loop around the scoring pass, through the C ABI only.
// For every candidate vocabulary the worker builds the tables (tm_build_vocab = training/trainvocab.go:548-907), uploads them
// (tm_vocab_load) and scores the dataset (tm_score).  With the scoring pass on the GPU, build + load is the Amdahl term of ONE
// worker — the reference runs one goroutine per candidate (trainvocab.go:1827-1829), and so does this: T host threads prepare
// candidates while the GPU scores one (tm_score serialises passes per dataset).  Prints the stage times of one candidate and the
// end-to-end candidates/s for 1..T threads.
//   hipcc -O2 -std=c++17 -I include tools/candidate_throughput.cpp -o /tmp/cand -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -lpthread -Wl,-rpath,$PWD/tokenmonster_amd
//   /tmp/cand [MiB of raw text = 256] [candidates = 48] [max threads = 16] [ranks = 0]
// ranks = N > 0: the data-parallel mode of DESIGN section 5 with N lanes of ONE GPU standing in for N GPUs — the dataset is cut into N byte
// ranges (each uploaded with its halo), every candidate is built and loaded ONCE (round-robin over the worker threads), its device block
// goes to the other N - 1 "ranks" (tm_vocab_block_export / _import + a device copy where N GPUs would do one RCCL broadcast), and each rank
// scores its range as a piece of the one whole-buffer walk (tm_score_begin / 80 exit states / tm_score_finish).  Prints what a candidate
// costs a rank on the host besides the pass, the rate at which the worker threads turn out candidates, and what N GPUs could sustain.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tokenmonster_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const uint64_t mib = argc > 1 ? atoll(argv[1]) : 256;
  const int ncand = argc > 2 ? atoi(argv[2]) : 48;
  const int maxt = argc > 3 ? atoi(argv[3]) : 16;
  const int nranks = argc > 4 ? atoi(argv[4]) : 0;
  if (tm_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(TM_KIND_ENGLISHCODE, 65536, 2, 1, 5, 0x544D0005, 0, &img, &img_n) != 0) { fprintf(stderr, "synth_vocab: %s\n", tm_last_error()); return 1; }
  // the token list of the image: records that are not "D "-duplicates (score >= -0.5, go/tokenmonster.go:2383)
  std::vector<std::string> singles, multi;
  {
    const uint32_t n_info = img[17] | (img[18] << 8) | (img[19] << 16);
    size_t pos = 24;
    for (uint32_t i = 0; i < n_info; i++) {
      const uint32_t kl = img[pos];
      float score; memcpy(&score, img + pos + 1 + kl + 11, 4);
      if (score >= -0.5f) (kl == 1 ? singles : multi).emplace_back((const char*)img + pos + 1, kl);
      pos += 1 + kl + 15;
    }
  }
  tm_free(img);
  const uint64_t nbytes = mib << 20;
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint32_t nd = 0; uint64_t nb = 0;
  tm_synth_corpus(TM_KIND_ENGLISHCODE, 0x434F5250 + 5, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nb);
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  if (tm_normalize_batch(raw.data(), roff.data(), nd, 2, 1, 0, &text, off.data()) != 0) return 1;
  tm_dataset* ds = nullptr;
  if (tm_dataset_upload(text, off[nd], &ds) != 0) { fprintf(stderr, "upload: %s\n", tm_last_error()); return 1; }
  printf("dataset %.1f MB normalized; %zu single-byte + %zu longer tokens\n", off[nd] / 1e6, singles.size(), multi.size());

  bool via_image = false;
  auto one = [&](int k, double* t) -> uint64_t {
    // a candidate set like trainvocab's (:2250-2263): the single bytes + a random 97 % of the other tokens
    double t0 = now();
    std::vector<uint8_t> blob; std::vector<uint32_t> o(1, 0);
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
    auto add = [&](const std::string& tk) { blob.insert(blob.end(), tk.begin(), tk.end()); o.push_back((uint32_t)blob.size()); };
    for (auto& tk : singles) add(tk);
    for (auto& tk : multi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; if ((s >> 11) % 100 < 97) add(tk); }
    tm_vocab* v = nullptr;
    double t1, t2;
    if (via_image) {             // rounds 2-3: the .vocab image in between
      uint8_t* im = nullptr; size_t im_n = 0;
      if (tm_build_vocab(blob.data(), o.data(), (uint32_t)o.size() - 1, nullptr, 2, 1, 1, 5, 0, &im, &im_n) != 0) { fprintf(stderr, "build: %s\n", tm_last_error()); exit(1); }
      t1 = now();
      if (tm_vocab_load(im, im_n, &v) != 0) { fprintf(stderr, "load: %s\n", tm_last_error()); exit(1); }
      tm_free(im);
      t2 = now();
    } else {                     // token list -> tables -> device in one call
      if (tm_vocab_build(blob.data(), o.data(), (uint32_t)o.size() - 1, nullptr, 2, 1, 1, 5, 0, 0, &v) != 0) { fprintf(stderr, "build: %s\n", tm_last_error()); exit(1); }
      t1 = t2 = now();
    }
    std::vector<uint32_t> scores(tm_vocab_n_ids(v));
    uint64_t tit = 0; uint8_t ms[32];
    if (tm_score(v, ds, nullptr, nullptr, 0, scores.data(), &tit, ms) != 0) { fprintf(stderr, "score: %s\n", tm_last_error()); exit(1); }
    double t3 = now();
    tm_vocab_free(v);
    if (t) { t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2; }
    return tit;
  };
  one(0, nullptr);
  double t[3] = {0, 0, 0};
  via_image = true;
  one(0, nullptr);
  for (int k = 0; k < 4; k++) one(k, t);
  printf("one candidate (about 63 600 ids), through the .vocab image: tm_build_vocab %.1f ms, tm_vocab_load %.1f ms, tm_score %.1f ms  -> build + load = %.2f x the scoring pass\n",
         t[0] / 4 * 1e3, t[1] / 4 * 1e3, t[2] / 4 * 1e3, (t[0] + t[1]) / t[2]);
  via_image = false;
  t[0] = t[1] = t[2] = 0;
  for (int k = 0; k < 4; k++) one(k, t);
  printf("one candidate (about 63 600 ids), tm_vocab_build (token list -> tables -> device): %.1f ms, tm_score %.1f ms  -> build = %.2f x the scoring pass\n",
         t[0] / 4 * 1e3, t[2] / 4 * 1e3, t[0] / t[2]);
  for (int nt = 1; nt <= maxt; nt *= 2) {
    std::atomic<int> next{0};
    const double t0 = now();
    std::vector<std::thread> th;
    std::vector<double> st(3 * nt, 0.0);
    for (int i = 0; i < nt; i++) th.emplace_back([&, i] { for (;;) { int k = next.fetch_add(1); if (k >= ncand) break; one(k, &st[3 * i]); } });
    for (auto& x : th) x.join();
    const double dt = now() - t0;
    double a[3] = {0, 0, 0};
    for (int i = 0; i < nt; i++) for (int q = 0; q < 3; q++) a[q] += st[3 * i + q];
    printf("%2d worker threads: %d candidates in %.2f s = %5.1f candidates/s  (the scoring pass alone allows %.1f/s); per candidate: build %.0f ms, load %.0f ms, score incl. waiting %.0f ms\n",
           nt, ncand, dt, ncand / dt, 4.0 / t[2], a[0] / ncand * 1e3, a[1] / ncand * 1e3, a[2] / ncand * 1e3);
  }
  if (nranks > 0) {
    // ---- N lanes of this GPU standing in for N GPUs ---------------------------------------------------------------------------
    const uint64_t N = off[nd], HALO = 128;
    std::vector<tm_dataset*> rds(nranks);
    std::vector<uint64_t> lo(nranks + 1);
    for (int r = 0; r <= nranks; r++) lo[r] = N * r / nranks;
    for (int r = 0; r < nranks; r++) {
      const uint64_t end = std::min<uint64_t>(N, lo[r + 1] + (r + 1 < nranks ? HALO : 0));
      if (tm_dataset_upload(text + lo[r], end - lo[r], &rds[r]) != 0) { fprintf(stderr, "upload: %s\n", tm_last_error()); return 1; }
    }
    auto build_load = [&](int k, tm_vocab** out) {
      std::vector<uint8_t> blob; std::vector<uint32_t> o(1, 0);
      uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
      auto add = [&](const std::string& tk) { blob.insert(blob.end(), tk.begin(), tk.end()); o.push_back((uint32_t)blob.size()); };
      for (auto& tk : singles) add(tk);
      for (auto& tk : multi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; if ((s >> 11) % 100 < 97) add(tk); }
      if (tm_vocab_build(blob.data(), o.data(), (uint32_t)o.size() - 1, nullptr, 2, 1, 1, 5, 0, 0, out) != 0) { fprintf(stderr, "build: %s\n", tm_last_error()); exit(1); }
    };
    // (1) how fast the host turns out candidates, GPU otherwise idle
    for (int nt = 1; nt <= maxt; nt *= 2) {
      std::atomic<int> next{0};
      const double t0 = now();
      std::vector<std::thread> th;
      for (int i = 0; i < nt; i++) th.emplace_back([&] { for (;;) { int k = next.fetch_add(1); if (k >= ncand) break; tm_vocab* v = nullptr; build_load(k, &v); tm_vocab_free(v); } });
      for (auto& x : th) x.join();
      printf("tm_vocab_build only, %2d host threads: %5.1f candidates/s\n", nt, ncand / (now() - t0));
    }
    // (2) one candidate through all ranks: what the ranks that did NOT build it pay, and the pass over a range
    double t_imp = 0, t_pass = 0, t_first = 0, block_mb = 0;
    uint64_t tokens_total = 0;
    const int reps = 6;
    for (int k = 0; k < reps; k++) {
      tm_vocab* v0 = nullptr;
      build_load(k, &v0);
      tm_vocab_block meta; void* src = nullptr;
      if (tm_vocab_block_export(v0, &meta, &src) != 0) return 1;
      block_mb = meta.bytes / 1e6;
      std::vector<tm_vocab*> vs(nranks, nullptr);
      vs[0] = v0;
      const double a = now();
      for (int r = 1; r < nranks; r++) {
        void* dst = nullptr;
        if (tm_vocab_block_import(&meta, 0, &vs[r], &dst) != 0 || tm_device_copy(dst, src, meta.bytes) != 0) { fprintf(stderr, "import: %s\n", tm_last_error()); return 1; }
      }
      const double b = now();
      uint32_t entry = 0;
      uint64_t tok = 0;
      for (int r = 0; r < nranks; r++) {
        uint8_t exits[80];
        const double c = now();
        if (tm_score_begin(vs[r], rds[r], 0, lo[r + 1] - lo[r], r + 1 < nranks ? 1 : 0, nullptr, exits) != 0 ||
            tm_score_finish(vs[r], rds[r], entry, nullptr, nullptr, 0) != 0) { fprintf(stderr, "range pass: %s\n", tm_last_error()); return 1; }
        std::vector<uint32_t> sc(tm_vocab_n_ids(vs[r])); uint64_t tit = 0; uint8_t ms[32];
        if (tm_score_read(vs[r], rds[r], sc.data(), &tit, ms) != 0) return 1;
        if (k > 0) { t_pass += now() - c; if (r == 0) t_first += now() - c; }
        tok += tit;
        entry = exits[entry];
      }
      if (k > 0) t_imp += (b - a) / std::max(1, nranks - 1);
      tokens_total = tok;
      for (auto* v : vs) tm_vocab_free(v);
    }
    const double pass_ms = t_pass / (reps - 1) / nranks * 1e3;
    printf("%d ranks: a rank that did not build the candidate pays %.2f ms (import + block copy, %.1f MB) instead of build + load (%.1f ms); pass over 1/%d of the dataset %.2f ms "
           "(incl. the histogram read); tokens of the whole walk %llu\n", nranks, t_imp / (reps - 1) * 1e3, block_mb, (t[0] + t[1]) / 4 * 1e3, nranks, pass_ms, (unsigned long long)tokens_total);
    printf("  => %d GPUs in the data-parallel mode need a new candidate every %.2f ms = %.0f candidates/s; the builders above must turn out that many (round-robin over the ranks' host threads)\n",
           nranks, pass_ms, 1e3 / pass_ms);
    for (auto* d : rds) tm_dataset_free(d);
  }
  tm_dataset_free(ds);
  tm_free(text);
  return 0;
}
