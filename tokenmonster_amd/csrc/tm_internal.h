// tm_internal.h — shared between the host translation units of libtokenmonster_hip.so.
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace tmh {

// thread-local error string behind tm_last_error()
int set_error(int code, const char* fmt, ...);
const char* last_error();

// tm_build.cpp (build_vocab_records, which fills a HostVocab and its trie directly, is declared in tm_device.h)
int build_vocab_image(const std::vector<std::string>& tokens, const std::vector<uint8_t>& special, uint32_t capcode,
                      uint32_t charset, uint32_t norm_flag, uint32_t level, bool with_unk,
                      std::vector<uint8_t>& image, const std::vector<float>* token_scores = nullptr);

// tm_normalize.cpp
struct NmTwo;
void build_two_table(uint32_t norm_flag, NmTwo* out);     // NM_TWO_SIZE entries: the device normalizer's table of the two-byte characters (tm_norm_masks.h)
void build_three_tables(uint32_t norm_flag, uint32_t* blk, uint32_t* cp);    // NM_BLK_WORDS + NM_CP_WORDS words: the three-byte characters it passes through
void build_four_table(uint32_t norm_flag, uint32_t* blk4);                   // NM_BLK4_WORDS words: the blocks of four-byte characters it passes through
struct NmLea;
void build_ccc_table(uint32_t norm_flag, bool marks, uint8_t* out);                      // NM_CCC_SIZE entries: canonical class of the three-byte marks of U+0800..U+1FFF that NFD leaves alone (0: not one)
void build_dec3_table(uint32_t* out);                                        // NM_DEC3_SIZE entries: the three-byte characters of U+0900..U+1BFF that NFD splits in two three-byte ones
void build_kana_table(uint16_t* out);                                        // NM_KANA_SIZE entries: the voiced kana of U+3040..U+30FF under NFD, a kana + U+3099 / U+309A
void build_lea_table(uint32_t norm_flag, NmLea* out);                        // NM_LEA_SIZE entries: Latin Extended Additional under NFD, a letter + one or two marks
void build_accent_table(uint32_t* out);                                      // NM_TWO_SIZE words: what flag 4 `accents` leaves of the two-byte characters (the filter pass)
void normalize_bytes(const uint8_t* data, size_t n, uint32_t capcode, uint32_t norm_flag, std::vector<uint8_t>& out);

int normalize_batch_into(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode, uint32_t norm_flag,
                         uint32_t threads, uint64_t* out_offsets, const std::function<uint8_t*(uint64_t)>& alloc);

// state of a capcode decoder between two calls (javascript/tokenmonster.js:1008-1013)
struct CapcodeState { bool in_word = false, in_char = false, del = false, ignore = false; };
void capcode_decode_stream(CapcodeState& st, const uint8_t* in, size_t n, std::vector<uint8_t>& out);     // appends
void nocapcode_decode_stream(CapcodeState& st, const uint8_t* in, size_t n, std::vector<uint8_t>& out);   // appends

// the device decoder's tables (tm_decode.hip): the two-byte characters U+0080..U+07FF, then block and code-point codes of the three-byte ones
constexpr uint32_t DEC_TWO = 0x780, DEC_BLK_WORDS = 64, DEC_CP_WORDS = 4096, DEC_BLK4_WORDS = 1024, DEC_TABLE_WORDS = DEC_TWO + DEC_BLK_WORDS + DEC_CP_WORDS + DEC_BLK4_WORDS;
void build_dec_tables(uint32_t* two, uint32_t* blk, uint32_t* cp, uint32_t* blk4);

void capcode_decode_batch(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode, uint32_t threads,
                          std::vector<std::vector<uint8_t>>& outs);

// runs `work` on the calling thread and on up to threads-1 pooled workers at once; `work` must pull from a shared queue
void run_on_workers(uint32_t threads, const std::function<void()>& work);

// small deterministic PRNG (splitmix64 seeding + xoshiro256**), used by the synthetic generators
struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (auto& x : s) {
      seed += 0x9E3779B97F4A7C15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      x = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  bool chance(double p) { return unit() < p; }
};

}  // namespace tmh
