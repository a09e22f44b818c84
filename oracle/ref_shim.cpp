// oracle/ref_shim.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" window onto the reference's own C++ runtime (tokenmonster-cpp/src/tokenmonster.cpp,
// compiled from where it lies under /root/reference by oracle/Makefile into oracle/_ref/).
// Nothing here restates an algorithm: every call forwards to tokenmonster::Vocab.
// `tokenize_normalized` is a private member of the reference class
// (tokenmonster-cpp/include/tokenmonster/tokenmonster.hpp:140); it is THE function the hot path
// must match (tokenmonster.cpp:1723-1991 == go/tokenmonster.go:1017-1279), so the header is
// included with access control disabled.  Standard headers are included first so the macro
// cannot touch them.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <memory>
#include <optional>
#include <span>
#include <stdexcept>
#include <string>
#include <atomic>
#include <string_view>
#include <thread>
#include <utility>
#include <vector>
#include <capcode/capcode.hpp>

#define private public
#include <tokenmonster/tokenmonster.hpp>
#undef private

namespace {
thread_local std::string g_err;
using tokenmonster::Vocab;
std::span<const std::uint8_t> sp(const std::uint8_t* p, std::size_t n) { return {p, n}; }
}  // namespace

extern "C" {

const char* tmref_last_error() { return g_err.c_str(); }

void* tmref_load(const char* path) {
  try {
    return new Vocab(Vocab::load(path));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}

void tmref_free(void* v) { delete static_cast<Vocab*>(v); }

int tmref_vocab_size(void* v) { return static_cast<Vocab*>(v)->size(); }
int tmref_max_token_length(void* v) { return static_cast<Vocab*>(v)->max_token_length(); }
int tmref_capcode(void* v) { return static_cast<Vocab*>(v)->capcode(); }
int tmref_charset(void* v) { return static_cast<Vocab*>(v)->charset(); }
int tmref_normalization(void* v) { return static_cast<Vocab*>(v)->normalization_code(); }
std::uint32_t tmref_unk(void* v) { return static_cast<Vocab*>(v)->unk(); }

// tokenmonster.cpp:1723  (Vocab::tokenize_normalized) — returns token count, or -needed if cap too small
long long tmref_tokenize_normalized(void* v, const std::uint8_t* data, std::size_t n, std::uint32_t* out,
                                    std::size_t cap, int* missing) {
  try {
    auto r = static_cast<Vocab*>(v)->tokenize_normalized(sp(data, n));
    if (missing) *missing = r.missing;
    if (r.tokens.size() > cap) return -(long long)r.tokens.size();
    if (!r.tokens.empty()) std::memcpy(out, r.tokens.data(), r.tokens.size() * 4);
    return (long long)r.tokens.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// tokenmonster.cpp:1368 (Vocab::tokenize = normalize + capcode + tokenize_normalized)
long long tmref_tokenize(void* v, const std::uint8_t* data, std::size_t n, std::uint32_t* out, std::size_t cap,
                         int* missing) {
  try {
    auto r = static_cast<Vocab*>(v)->tokenize(sp(data, n));
    if (missing) *missing = r.missing;
    if (r.tokens.size() > cap) return -(long long)r.tokens.size();
    if (!r.tokens.empty()) std::memcpy(out, r.tokens.data(), r.tokens.size() * 4);
    return (long long)r.tokens.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// tokenmonster.cpp:2257-ish (Vocab::tokenize_count_normalized) — Q2: b-branches count 1
long long tmref_count_normalized(void* v, const std::uint8_t* data, std::size_t n, int* missing) {
  try {
    auto r = static_cast<Vocab*>(v)->tokenize_count_normalized(sp(data, n));
    if (missing) *missing = r.missing;
    return r.tokens;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

long long tmref_normalize(void* v, const std::uint8_t* data, std::size_t n, std::uint8_t* out, std::size_t cap) {
  try {
    auto r = static_cast<Vocab*>(v)->normalize(sp(data, n));
    if (r.size() > cap) return -(long long)r.size();
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return (long long)r.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

long long tmref_decode(void* v, const std::uint32_t* toks, std::size_t n, std::uint8_t* out, std::size_t cap) {
  try {
    auto r = static_cast<Vocab*>(v)->decode(std::span<const std::uint32_t>(toks, n));
    if (r.size() > cap) return -(long long)r.size();
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return (long long)r.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// decode without the capcode/charset post-processing (tokenmonster.cpp Vocab::decode_raw)
long long tmref_decode_raw(void* v, const std::uint32_t* toks, std::size_t n, std::uint8_t* out, std::size_t cap) {
  try {
    auto r = static_cast<Vocab*>(v)->decode_raw(std::span<const std::uint32_t>(toks, n));
    if (r.size() > cap) return -(long long)r.size();
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return (long long)r.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// the reference's streaming Decoder (tokenmonster.cpp:1509-1721)
void* tmref_decoder_new(void* v) {
  try { return new tokenmonster::Decoder(static_cast<Vocab*>(v)->new_decoder()); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void tmref_decoder_free(void* d) { delete static_cast<tokenmonster::Decoder*>(d); }
long long tmref_decoder_decode(void* d, const std::uint32_t* toks, std::size_t n, std::uint8_t* out, std::size_t cap) {
  try {
    auto r = static_cast<tokenmonster::Decoder*>(d)->decode(std::span<const std::uint32_t>(toks, n));
    if (r.size() > cap) return -(long long)r.size();
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return (long long)r.size();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
long long tmref_decoder_decode_serialized(void* d, const std::uint8_t* data, std::size_t n, int enc, std::uint8_t* out, std::size_t cap) {
  try {
    auto r = static_cast<tokenmonster::Decoder*>(d)->decode_serialized(sp(data, n), (std::uint8_t)enc);
    if (r.size() > cap) return -(long long)r.size();
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return (long long)r.size();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
long long tmref_decoder_flush(void* d, std::uint8_t* out, std::size_t cap) {
  try {
    auto r = static_cast<tokenmonster::Decoder*>(d)->flush();
    if (r.size() > cap) return -(long long)r.size();
    if (!r.empty()) std::memcpy(out, r.data(), r.size());
    return (long long)r.size();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// All-core baseline for bench.py (the reference server's model: documents are independent, one goroutine each,
// training/tokenmonsterserver.go:363-378): `threads` std::threads pull documents from a shared counter and call
// Vocab::tokenize (raw != 0: normalize + capcode + walk) or Vocab::tokenize_normalized.  Returns the number of tokens.
long long tmref_tokenize_docs_mt(void* v, const std::uint8_t* text, const std::uint64_t* offsets, std::uint32_t ndocs, int raw,
                                 std::uint32_t threads) {
  try {
    auto* vocab = static_cast<Vocab*>(v);
    if (threads == 0) threads = 1;
    std::atomic<std::uint32_t> next{0};
    std::atomic<long long> total{0};
    std::atomic<bool> failed{false};
    auto work = [&]() {
      long long mine = 0;
      try {
        for (;;) {
          const std::uint32_t base = next.fetch_add(16);
          if (base >= ndocs) break;
          for (std::uint32_t d = base; d < ndocs && d < base + 16; d++) {
            auto doc = sp(text + offsets[d], (std::size_t)(offsets[d + 1] - offsets[d]));
            mine += (long long)(raw ? vocab->tokenize(doc) : vocab->tokenize_normalized(doc)).tokens.size();
          }
        }
      } catch (...) { failed = true; }
      total += mine;
    };
    std::vector<std::thread> th;
    for (std::uint32_t t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed) { g_err = "a worker thread failed"; return -1; }
    return total.load();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// The same fan-out as a CHECKER (bench.py's verification of the whole corpus, outside the timed region): every document is tokenized by the
// reference and its ids and `missing` are compared with what the device produced (ids[toff[d] .. toff[d+1]), missing[d]).  Returns the
// number of documents that differ (0 = all equal), -1 on failure; *first_bad = the lowest differing document; *ntok = the reference's
// token total.  Nothing is restated here either: the comparison is memcmp.
long long tmref_verify_docs_mt(void* v, const std::uint8_t* text, const std::uint64_t* offsets, std::uint32_t ndocs, int raw, std::uint32_t threads,
                               const std::uint32_t* ids, const std::uint64_t* toff, const std::uint32_t* missing, std::uint32_t* first_bad,
                               long long* ntok) {
  try {
    auto* vocab = static_cast<Vocab*>(v);
    if (threads == 0) threads = 1;
    std::atomic<std::uint32_t> next{0}, worst{0xFFFFFFFFu};
    std::atomic<long long> total{0}, bad{0};
    std::atomic<bool> failed{false};
    auto work = [&]() {
      long long mine = 0, mybad = 0;
      std::uint32_t myworst = 0xFFFFFFFFu;
      try {
        for (;;) {
          const std::uint32_t base = next.fetch_add(16);
          if (base >= ndocs) break;
          for (std::uint32_t d = base; d < ndocs && d < base + 16; d++) {
            auto doc = sp(text + offsets[d], (std::size_t)(offsets[d + 1] - offsets[d]));
            const auto r = raw ? vocab->tokenize(doc) : vocab->tokenize_normalized(doc);
            mine += (long long)r.tokens.size();
            const std::uint64_t n = toff[d + 1] - toff[d];
            const bool same = n == r.tokens.size() && (n == 0 || std::memcmp(ids + toff[d], r.tokens.data(), n * 4) == 0) &&
                              (!missing || (long long)missing[d] == (long long)r.missing);
            if (!same) { mybad++; if (d < myworst) myworst = d; }
          }
        }
      } catch (...) { failed = true; }
      total += mine; bad += mybad;
      std::uint32_t w = worst.load();
      while (myworst < w && !worst.compare_exchange_weak(w, myworst)) {}
    };
    std::vector<std::thread> th;
    for (std::uint32_t t = 1; t < threads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed) { g_err = "a worker thread failed"; return -1; }
    if (first_bad) *first_bad = worst.load();
    if (ntok) *ntok = total.load();
    return bad.load();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

}  // extern "C"
