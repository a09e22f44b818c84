// build_profile.cpp — development aid (host only, no GPU): where the time of one candidate's table construction goes — tm_build_vocab
// (rules of trainvocab.go:548-907 -> .vocab image) and the host half of tm_vocab_load (parse_vocab: records -> trie -> double array ->
// links), the Amdahl term of the trainvocab loop once the scoring pass runs on the GPU.  TM_TRACE=1 prints the stages.
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/build_profile.cpp -o /tmp/build_profile -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_device.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const uint32_t vsize = argc > 1 ? atoi(argv[1]) : 65536;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  uint8_t* img = nullptr; size_t img_n = 0;
  if (tm_synth_vocab(TM_KIND_ENGLISHCODE, vsize, 2, 1, 5, 0x544D0005, 0, &img, &img_n) != 0) return 1;
  std::vector<uint8_t> blob; std::vector<uint32_t> o(1, 0);
  {
    const uint32_t n_info = img[17] | (img[18] << 8) | (img[19] << 16);
    size_t pos = 24;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (uint32_t i = 0; i < n_info; i++) {
      const uint32_t kl = img[pos];
      float score; memcpy(&score, img + pos + 1 + kl + 11, 4);
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      if (score >= -0.5f && (kl == 1 || (s >> 11) % 100 < 97)) { blob.insert(blob.end(), img + pos + 1, img + pos + 1 + kl); o.push_back((uint32_t)blob.size()); }
      pos += 1 + kl + 15;
    }
  }
  tm_free(img);
  double tb = 0, tp = 0;
  for (int r = 0; r < reps; r++) {
    uint8_t* im = nullptr; size_t im_n = 0;
    double t0 = now();
    if (tm_build_vocab(blob.data(), o.data(), (uint32_t)o.size() - 1, nullptr, 2, 1, 1, 5, 0, &im, &im_n) != 0) return 1;
    double t1 = now();
    tmh::HostVocab hv;
    if (tmh::parse_vocab(im, im_n, hv) != 0) return 1;
    double t2 = now();
    if (r) { tb += t1 - t0; tp += t2 - t1; }
    if (r == reps - 1) printf("%zu tokens -> %u records, %u trie nodes, double array %u entries, image %.2f MB\n", o.size() - 1, hv.n_info, hv.n_nodes, hv.n_da, im_n / 1048576.0);
    tm_free(im);
  }
  printf("tm_build_vocab %.1f ms, parse_vocab (host half of tm_vocab_load) %.1f ms\n", tb / (reps - 1) * 1e3, tp / (reps - 1) * 1e3);
  return 0;
}
