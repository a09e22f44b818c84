// tools/emu/tsan_host.cpp — the HOST layer of the library (lanes, worker pool, pinned mailbox, block cache: tm_host.hip, tm_kernels.hip,
// tm_vocab.hip, tm_normalize.cpp) under ThreadSanitizer, with the kernels running on the emulated device (tools/emu).  Development aid:
//   bash tools/emu/tsan_host.sh
// Several threads call the host-buffer entry points (tokenize, count, pipeline, decode) of ONE vocabulary at once (what goroutines of tokenmonsterserver do,
// training/tokenmonsterserver.go:363-378), two of them the chunked pipeline, one loads and frees further vocabularies meanwhile (the
// trainvocab worker's pattern); every result is compared with a single-threaded run.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tokenmonster_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != TM_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, tm_last_error()); std::exit(1); } } while (0)

int main() {
  uint8_t* img = nullptr; size_t img_n = 0;
  CHECK(tm_synth_vocab(TM_KIND_ENGLISHCODE, 2000, 2, 1, 3, 0x544D0007, 0, &img, &img_n));
  tm_vocab* v = nullptr;
  CHECK(tm_vocab_load(img, img_n, &v));
  const uint64_t nbytes = 600000;
  std::vector<uint8_t> raw(nbytes + 70000);
  std::vector<uint64_t> roff(nbytes / 64 + 17);
  uint32_t nd = 0; uint64_t nb = 0;
  CHECK(tm_synth_corpus(TM_KIND_ENGLISHCODE, 0x434F5250 + 9, nbytes, 2048, raw.data(), roff.data(), (uint32_t)roff.size() - 1, &nd, &nb));
  uint8_t* text = nullptr; std::vector<uint64_t> off(nd + 1);
  CHECK(tm_normalize_batch(raw.data(), roff.data(), nd, 2, 1, 0, &text, off.data()));
  // reference results, one thread
  std::vector<uint32_t> ids(off[nd] + 64); std::vector<uint64_t> toff(nd + 1); std::vector<uint32_t> miss(nd + 1);
  CHECK(tm_tokenize_batch(v, text, off.data(), nd, ids.data(), ids.size(), toff.data(), miss.data()));
  std::vector<uint8_t> ser(2 * toff[nd] + 64); std::vector<uint64_t> soff(nd + 1); uint32_t enc = 0;
  CHECK(tm_tokenize_pipeline(v, raw.data(), roff.data(), nd, 1, 2, 64 << 10, 3, ser.data(), ser.size(), soff.data(), miss.data(), &enc, nullptr));
  if (soff[nd] != 2 * toff[nd]) { std::fprintf(stderr, "pipeline: %llu bytes, batch %llu ids\n", (unsigned long long)soff[nd], (unsigned long long)toff[nd]); return 1; }
  for (uint64_t k = 0; k < toff[nd]; k++) if ((uint32_t)(ser[2 * k] | (ser[2 * k + 1] << 8)) != ids[k]) { std::fprintf(stderr, "pipeline id %llu differs\n", (unsigned long long)k); return 1; }

  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 6; t++) th.emplace_back([&, t] {
    // a slice of the documents per caller, several rounds (lanes are taken, given back, taken again)
    for (int round = 0; round < 3; round++) {
      const uint32_t d0 = (uint32_t)((uint64_t)nd * t / 6), d1 = (uint32_t)((uint64_t)nd * (t + 1) / 6), n = d1 - d0;
      std::vector<uint64_t> o(n + 1);
      for (uint32_t k = 0; k <= n; k++) o[k] = off[d0 + k] - off[d0];
      std::vector<uint32_t> out(o[n] + 64); std::vector<uint64_t> to(n + 1); std::vector<uint32_t> ms(n + 1);
      if (round == 1) {
        std::vector<uint64_t> cnt(n + 1);
        if (tm_count_batch(v, text + off[d0], o.data(), n, cnt.data(), ms.data()) != TM_OK) { bad++; return; }
        continue;
      }
      if (tm_tokenize_batch(v, text + off[d0], o.data(), n, out.data(), out.size(), to.data(), ms.data()) != TM_OK) { bad++; return; }
      if (to[n] != toff[d1] - toff[d0] || std::memcmp(out.data(), ids.data() + toff[d0], to[n] * 4) != 0) { bad++; return; }
    }
  });
  for (int t = 0; t < 2; t++) th.emplace_back([&] {
    std::vector<uint8_t> s2(ser.size()); std::vector<uint64_t> so(nd + 1); std::vector<uint32_t> ms(nd + 1); uint32_t e2 = 0;
    if (tm_tokenize_pipeline(v, raw.data(), roff.data(), nd, 1, 2, 96 << 10, 2, s2.data(), s2.size(), so.data(), ms.data(), &e2, nullptr) != TM_OK) { bad++; return; }
    if (so[nd] != soff[nd] || std::memcmp(s2.data(), ser.data(), so[nd]) != 0) bad++;
  });
  // decode jobs beside the tokenize jobs (tokenmonsterserver jobs 2-9): same lanes, same answers as a single-threaded decode
  std::vector<uint8_t> dec(off[nd] * 2 + 4096); std::vector<uint64_t> doff(nd + 1);
  CHECK(tm_decode_batch(v, ids.data(), toff.data(), nd, 0, dec.data(), dec.size(), doff.data()));
  for (int t = 0; t < 2; t++) th.emplace_back([&] {
    for (int round = 0; round < 3; round++) {
      std::vector<uint8_t> d2(dec.size()); std::vector<uint64_t> o2(nd + 1);
      if (tm_decode_batch(v, ids.data(), toff.data(), nd, round == 1, d2.data(), d2.size(), o2.data()) != TM_OK) { bad++; return; }
      if (round != 1 && (o2[nd] != doff[nd] || std::memcmp(d2.data(), dec.data(), o2[nd]) != 0)) bad++;
    }
  });
  th.emplace_back([&] {
    for (int k = 0; k < 4; k++) {
      tm_vocab* w = nullptr;
      if (tm_vocab_load(img, img_n, &w) != TM_OK) { bad++; return; }
      uint32_t one[8]; uint64_t to[2]; uint32_t ms[2]; const uint64_t o[2] = {0, 5};
      if (tm_tokenize_batch(w, (const uint8_t*)"hello", o, 1, one, 8, to, ms) != TM_OK) bad++;
      tm_vocab_free(w);
    }
  });
  // the multi-device driver beside all of that (tm_multi.hip: a host thread per member, its meeting points, the sum of the members' histograms):
  // three members on the one emulated device, the chunked pipeline over all their lanes and two whole-buffer scoring passes, from two callers
  std::vector<uint32_t> sc_ref(tm_vocab_n_ids(v)); uint64_t tit_ref = 0; uint8_t ms_ref[32];
  {
    tm_dataset* d1 = nullptr;
    CHECK(tm_dataset_upload(text, off[nd], &d1));
    const uint64_t so0 = 0, sl0 = off[nd];
    CHECK(tm_score(v, d1, &so0, &sl0, 1, sc_ref.data(), &tit_ref, ms_ref));
    tm_dataset_free(d1);
  }
  for (int t = 0; t < 2; t++) th.emplace_back([&] {
    const int devs[3] = {0, 0, 0};
    tm_devices* g = nullptr; tm_vocab_set* vs = nullptr; tm_dataset_set* ds = nullptr;
    if (tm_devices_open_list(devs, 3, &g) != TM_OK || tm_vocab_load_all(g, img, img_n, &vs) != TM_OK) { bad++; return; }
    std::vector<uint8_t> s3(ser.size()); std::vector<uint64_t> so(nd + 1); std::vector<uint32_t> ms(nd + 1); uint32_t e3 = 0;
    if (tm_tokenize_pipeline_multi(vs, raw.data(), roff.data(), nd, 1, 2, 48 << 10, 2, s3.data(), s3.size(), so.data(), ms.data(), &e3, nullptr) != TM_OK ||
        so[nd] != soff[nd] || std::memcmp(s3.data(), ser.data(), so[nd]) != 0) bad++;
    if (tm_dataset_upload_sharded(g, text, off[nd], &ds) != TM_OK) { bad++; return; }
    for (int round = 0; round < 2; round++) {
      std::vector<uint32_t> sc(sc_ref.size()); uint64_t tit = 0; uint8_t m3[32];
      if (tm_score_multi(vs, ds, sc.data(), &tit, m3) != TM_OK || tit != tit_ref || sc != sc_ref || std::memcmp(m3, ms_ref, 32) != 0) bad++;
    }
    tm_dataset_set_free(ds); tm_vocab_set_free(vs); tm_devices_close(g);
  });
  for (auto& t : th) t.join();
  tm_vocab_free(v);
  tm_free(text); tm_free(img);
  if (bad.load()) { std::fprintf(stderr, "%d caller(s) got a wrong result\n", bad.load()); return 1; }
  std::printf("tsan_host ok: %u documents, %llu ids, 13 concurrent callers (two of them through the multi-device driver)\n", nd, (unsigned long long)toff[nd]);
  return 0;
}
