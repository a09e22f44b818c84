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
#include <string_view>
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

}  // extern "C"
