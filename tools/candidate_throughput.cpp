// candidate_throughput.cpp — measurement for SURVEY 8f #3: the trainvocab loop around the scoring pass, through the C ABI only.
// For every candidate vocabulary the worker builds the tables (tm_build_vocab = training/trainvocab.go:548-907), uploads them
// (tm_vocab_load) and scores the dataset (tm_score).  With the scoring pass on the GPU, build + load is the Amdahl term of ONE
// worker — the reference runs one goroutine per candidate (trainvocab.go:1827-1829), and so does this: T host threads prepare
// candidates while the GPU scores one (tm_score serialises passes per dataset).  Prints the stage times of one candidate and the
// end-to-end candidates/s for 1..T threads.
//   hipcc -O2 -std=c++17 -I include tools/candidate_throughput.cpp -o /tmp/cand -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -lpthread -Wl,-rpath,$PWD/tokenmonster_amd
//   /tmp/cand [MiB of raw text = 256] [candidates = 48] [max threads = 16]
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

  auto one = [&](int k, double* t) -> uint64_t {
    // a candidate set like trainvocab's (:2250-2263): the single bytes + a random 97 % of the other tokens
    double t0 = now();
    std::vector<uint8_t> blob; std::vector<uint32_t> o(1, 0);
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);
    auto add = [&](const std::string& tk) { blob.insert(blob.end(), tk.begin(), tk.end()); o.push_back((uint32_t)blob.size()); };
    for (auto& tk : singles) add(tk);
    for (auto& tk : multi) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; if ((s >> 11) % 100 < 97) add(tk); }
    uint8_t* im = nullptr; size_t im_n = 0;
    if (tm_build_vocab(blob.data(), o.data(), (uint32_t)o.size() - 1, nullptr, 2, 1, 1, 5, 0, &im, &im_n) != 0) { fprintf(stderr, "build: %s\n", tm_last_error()); exit(1); }
    double t1 = now();
    tm_vocab* v = nullptr;
    if (tm_vocab_load(im, im_n, &v) != 0) { fprintf(stderr, "load: %s\n", tm_last_error()); exit(1); }
    tm_free(im);
    double t2 = now();
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
  for (int k = 0; k < 4; k++) one(k, t);
  printf("one candidate (about 63 600 ids): tm_build_vocab %.1f ms, tm_vocab_load %.1f ms, tm_score %.1f ms  -> build + load = %.2f x the scoring pass\n",
         t[0] / 4 * 1e3, t[1] / 4 * 1e3, t[2] / 4 * 1e3, (t[0] + t[1]) / t[2]);
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
  tm_dataset_free(ds);
  tm_free(text);
  return 0;
}
