// examples/cpp_port_shim/tokenmonster_shim.cpp — the port's public classes (include/tokenmonster/tokenmonster.hpp beside this file) over the
// C ABI of libtokenmonster_hip.so.  What each method replaces is cited from the reference's C++ port (tokenmonster-cpp/src/tokenmonster.cpp)
// and the Go library it translates; the work itself — normalization + capcode, the ungreedy walk, serialization, decoding — happens behind
// tm_* calls, on the GPU wherever the library puts it there.  Errors of the library become tokenmonster::Error with tm_last_error()'s text.
#include <tokenmonster/tokenmonster.hpp>

#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>

#include "tm_build.h"
#include "tokenmonster_hip.h"

namespace tokenmonster {

namespace {

using Bytes = std::vector<std::uint8_t>;

[[noreturn]] void fail(const char* what) { throw Error(std::string(what) + ": " + tm_last_error()); }
void check(int rc, const char* what) { if (rc != TM_OK) fail(what); }

std::uint32_t u24(const std::uint8_t* p) { return p[0] | (std::uint32_t)p[1] << 8 | (std::uint32_t)p[2] << 16; }

// one record of the .vocab image (go/tokenmonster.go:2656-2736)
struct Record { Bytes token; std::uint8_t flag = 0; std::uint32_t id = 0; float score = 0.0F; };

}  // namespace

struct Vocab::Impl {
  tm_vocab* v = nullptr;
  std::vector<Record> records;                           // file order
  std::vector<Bytes> reverse;                            // id -> bytes, the last record carrying the id wins (go :2715)
  std::unordered_map<std::string, std::uint32_t> by_key; // key -> record ordinal
  std::uint8_t level = 0;
  ~Impl() { if (v) tm_vocab_free(v); }

  // ids of ONE document; raw: normalization on the way (go :233-253), the pipeline's device normalizer
  TokenizeResult run(std::span<const std::uint8_t> data, bool raw) const {
    TokenizeResult r;
    const std::uint64_t offsets[2] = {0, data.size()};
    std::uint64_t tok_off[2] = {0, 0};
    std::uint32_t missing = 0;
    const std::uint8_t dummy = 0;
    const std::uint8_t* text = data.empty() ? &dummy : data.data();
    if (raw) {
      // Vocab.Tokenize on raw text = TokenizeToSerialized with four bytes per id (tm_tokenize_pipeline normalizes on the device)
      std::uint32_t used = 0;
      std::uint64_t cap = (std::uint64_t)data.size() * 4 + 64;
      for (;;) {
        Bytes out(cap);
        const int rc = tm_tokenize_pipeline(v, text, offsets, 1, 1, 4, 0, 0, out.data(), cap, tok_off, &missing, &used, nullptr);
        if (rc == TM_E_NOSPACE) { cap = tok_off[1]; continue; }
        check(rc, "tokenize");
        r.tokens.resize((std::size_t)(tok_off[1] / 4));
        if (!r.tokens.empty()) std::memcpy(r.tokens.data(), out.data(), r.tokens.size() * 4);
        break;
      }
    } else {
      std::uint64_t cap = data.size() + 16;
      for (;;) {
        r.tokens.resize((std::size_t)cap);
        const int rc = tm_tokenize_batch(v, text, offsets, 1, r.tokens.data(), cap, tok_off, &missing);
        if (rc == TM_E_NOSPACE) { cap = tok_off[1]; continue; }
        check(rc, "tokenize_normalized");
        r.tokens.resize((std::size_t)tok_off[1]);
        break;
      }
    }
    r.missing = (int)missing;
    return r;
  }
};

Vocab::Vocab() = default;
Vocab::~Vocab() = default;
Vocab::Vocab(Vocab&&) noexcept = default;
Vocab& Vocab::operator=(Vocab&&) noexcept = default;

// Load (go/tokenmonster.go:2656-2736; tokenmonster.cpp:1287-1359): the file's bytes go to tm_vocab_load, which checks them and builds
// the device tables; the token list behind tokens() / id_to_token / token_to_id is read back from the image the library keeps
Vocab Vocab::load(const std::filesystem::path& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw Error("cannot open vocabulary file: " + path.string());
  Bytes file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  if (file.empty()) throw Error("empty vocabulary file: " + path.string());
  Vocab out;
  out.impl_ = std::make_unique<Impl>();
  Impl& m = *out.impl_;
  check(tm_vocab_load(file.data(), file.size(), &m.v), "load");
  const std::uint8_t* f = file.data();           // (checked by the loader: lengths and counts are consistent)
  m.level = f[3];
  const std::uint32_t n_rev = u24(f + 14), n_info = u24(f + 17);
  m.reverse.resize(n_rev);
  m.records.reserve(n_info);
  std::size_t pos = 24;
  for (std::uint32_t i = 0; i < n_info; i++) {
    const std::size_t kl = f[pos];
    Record r;
    r.token.assign(f + pos + 1, f + pos + 1 + kl);
    const std::uint8_t* q = f + pos + 1 + kl;
    r.flag = q[0];
    r.id = u24(q + 8);
    std::memcpy(&r.score, q + 11, 4);
    if (r.id < n_rev) m.reverse[r.id] = r.token;
    m.by_key.emplace(std::string(r.token.begin(), r.token.end()), i);
    m.records.push_back(std::move(r));
    pos += 16 + kl;
  }
  return out;
}


#define TM_IMPL() ([this]() -> const Impl& { if (!impl_ || !impl_->v) throw Error("vocabulary not loaded"); return *impl_; }())

std::vector<std::uint8_t> Vocab::normalize(std::span<const std::uint8_t> data) const {          // go :233-253; tokenmonster.cpp:1360
  const Impl& m = TM_IMPL();
  std::uint8_t* out = nullptr;
  std::size_t n = 0;
  check(tm_normalize(data.data(), data.size(), tm_vocab_capcode(m.v), tm_vocab_normalization(m.v), &out, &n), "normalize");
  Bytes r(out, out + n);
  tm_free(out);
  return r;
}

TokenizeResult Vocab::tokenize(std::span<const std::uint8_t> data) const { return TM_IMPL().run(data, true); }                       // go :965
TokenizeResult Vocab::tokenize_normalized(std::span<const std::uint8_t> normalized) const { return TM_IMPL().run(normalized, false); } // go :1017

CountResult Vocab::count(std::span<const std::uint8_t> data) const {                              // go :971 (quirk Q2: a forward-delete pair counts 1)
  const Impl& m = TM_IMPL();
  const std::uint64_t offsets[2] = {0, data.size()};
  std::uint64_t n = 0;
  std::uint32_t missing = 0;
  const std::uint8_t dummy = 0;
  check(tm_count_batch_raw(m.v, data.empty() ? &dummy : data.data(), offsets, 1, &n, &missing), "count");
  return CountResult{(int)n, (int)missing};
}

CountResult Vocab::tokenize_count_normalized(std::span<const std::uint8_t> normalized) const {    // go :1281
  const Impl& m = TM_IMPL();
  const std::uint64_t offsets[2] = {0, normalized.size()};
  std::uint64_t n = 0;
  std::uint32_t missing = 0;
  const std::uint8_t dummy = 0;
  check(tm_count_batch(m.v, normalized.empty() ? &dummy : normalized.data(), offsets, 1, &n, &missing), "count");
  return CountResult{(int)n, (int)missing};
}

SerializedResult Vocab::tokenize_serialized(std::span<const std::uint8_t> data, std::uint8_t encoding_length) const {      // go :986
  const Impl& m = TM_IMPL();
  if (encoding_length == 1 || encoding_length > 4) throw Error("encoding_length must be 0, 2, 3 or 4");
  const std::uint64_t offsets[2] = {0, data.size()};
  std::uint64_t byte_off[2] = {0, 0};
  std::uint32_t missing = 0, used = 0;
  const std::uint8_t dummy = 0;
  SerializedResult r;
  std::uint64_t cap = (std::uint64_t)data.size() * 4 + 64;
  for (;;) {
    r.bytes.resize((std::size_t)cap);
    const int rc = tm_tokenize_pipeline(m.v, data.empty() ? &dummy : data.data(), offsets, 1, 1, encoding_length, 0, 0, r.bytes.data(), cap, byte_off,
                                        &missing, &used, nullptr);
    if (rc == TM_E_NOSPACE) { cap = byte_off[1]; continue; }
    check(rc, "tokenize_serialized");
    break;
  }
  r.bytes.resize((std::size_t)byte_off[1]);
  r.encoding_length = (std::uint8_t)used;
  r.missing = (int)missing;
  return r;
}

std::vector<std::uint8_t> Vocab::decode(std::span<const std::uint32_t> tokens) const {            // go :445; tokenmonster.cpp:1420-1425
  const Impl& m = TM_IMPL();
  const std::uint64_t tok_off[2] = {0, tokens.size()};
  std::uint64_t out_off[2] = {0, 0};
  const std::uint32_t dummy = 0;
  Bytes out(tokens.size() * 8 + 64);
  for (;;) {
    const int rc = tm_decode_batch(m.v, tokens.empty() ? &dummy : tokens.data(), tok_off, 1, 0, out.data(), out.size(), out_off);
    if (rc == TM_E_NOSPACE) { out.resize((std::size_t)out_off[1]); continue; }
    check(rc, "decode");
    break;
  }
  out.resize((std::size_t)out_off[1]);
  return out;
}

// little-endian ids of 2, 3 or 4 bytes; 0 picks 2 or 3 by len(reverse) (go :990-996); an incomplete id at the end is dropped
std::vector<std::uint32_t> Vocab::deserialize(std::span<const std::uint8_t> data, std::uint8_t encoding_length) const {
  const Impl& m = TM_IMPL();
  std::uint32_t w = encoding_length;
  if (w == 0) w = m.reverse.size() <= 65536 ? 2 : 3;
  std::vector<std::uint32_t> ids;
  if (w < 2 || w > 4) return ids;
  ids.reserve(data.size() / w);
  for (std::size_t i = 0; i + w <= data.size(); i += w) {
    std::uint32_t x = 0;
    for (std::uint32_t b = 0; b < w; b++) x |= (std::uint32_t)data[i + b] << (8 * b);
    ids.push_back(x);
  }
  return ids;
}

std::vector<std::uint8_t> Vocab::decode_serialized(std::span<const std::uint8_t> data, std::uint8_t encoding_length) const {   // go :464
  if (encoding_length == 1) encoding_length = 0;          // (the port's decode_serialized_raw treats <= 1 as automatic)
  return decode(deserialize(data, encoding_length));       // ids beyond len(reverse) are skipped by the decoder, as the port skips them
}

Decoder Vocab::new_decoder() const {                      // go :552
  const Impl& m = TM_IMPL();
  tm_decoder* d = nullptr;
  check(tm_decoder_new(m.v, &d), "new_decoder");
  Decoder out;
  out.vocab_ = this;
  out.state_ = std::shared_ptr<void>(d, [](void* p) { tm_decoder_free((tm_decoder*)p); });
  return out;
}

std::vector<std::uint8_t> Vocab::denormalize(std::span<const std::uint8_t> token) const {         // tokenmonster.cpp:3248
  const Impl& m = TM_IMPL();
  std::uint8_t* out = nullptr;
  std::size_t n = 0;
  check(tm_denormalize(token.data(), token.size(), tm_vocab_capcode(m.v), &out, &n), "denormalize");
  Bytes r(out, out + n);
  tm_free(out);
  return r;
}

// the "D "-duplicates of the index (score < -0.5, go :3460) are not tokens of the vocabulary
std::vector<std::vector<std::uint8_t>> Vocab::tokens() const {
  const Impl& m = TM_IMPL();
  std::vector<Bytes> list;
  for (const Record& r : m.records) if (r.score > -0.5F) list.push_back(r.token);
  return list;
}

std::vector<Info> Vocab::tokens_detailed() const {        // go :2500-2560; tokenmonster.cpp:3186
  const Impl& m = TM_IMPL();
  std::vector<Info> list;
  for (const Record& r : m.records) {
    if (r.score < -0.5F) continue;
    Info i;
    i.id = r.id; i.token = r.token; i.token_decoded = denormalize(r.token); i.score = r.score;
    i.type = r.token.size() == 1 ? 1 : ((r.flag & 64) ? 2 : 0);
    list.push_back(std::move(i));
  }
  if (has_unk()) { Info i; i.id = unk(); i.type = 3; list.push_back(std::move(i)); }
  if ((int)list.size() < size()) list.resize((std::size_t)size());
  return list;
}

std::vector<Info> Vocab::special_tokens() const {
  const Impl& m = TM_IMPL();
  std::vector<Info> list;
  for (const Record& r : m.records)
    if ((r.flag & 64) && r.score >= -0.5F) {
      Info i;
      i.id = r.id; i.type = 2; i.token = r.token; i.token_decoded = denormalize(r.token); i.score = r.score;
      list.push_back(std::move(i));
    }
  return list;
}

std::optional<std::vector<std::uint8_t>> Vocab::id_to_token(std::uint32_t id) const {
  const Impl& m = TM_IMPL();
  if (id >= m.reverse.size()) return std::nullopt;
  return m.reverse[id];
}

std::optional<std::uint32_t> Vocab::token_to_id(std::span<const std::uint8_t> token) const {
  const Impl& m = TM_IMPL();
  const auto it = m.by_key.find(std::string(token.begin(), token.end()));
  if (it == m.by_key.end()) return std::nullopt;
  return m.records[it->second].id;
}

std::uint32_t Vocab::unk() const { return tm_vocab_unk(TM_IMPL().v); }
int Vocab::size() const { return (int)tm_vocab_size(TM_IMPL().v); }
int Vocab::max_token_length() const { return (int)tm_vocab_max_token_length(TM_IMPL().v); }
std::uint8_t Vocab::charset() const { return (std::uint8_t)tm_vocab_charset(TM_IMPL().v); }
std::uint8_t Vocab::capcode() const { return (std::uint8_t)tm_vocab_capcode(TM_IMPL().v); }
std::uint8_t Vocab::mode() const { return TM_IMPL().level; }
std::uint8_t Vocab::normalization_code() const { return (std::uint8_t)tm_vocab_normalization(TM_IMPL().v); }
int Vocab::highest_token_id() const { return (int)tm_vocab_n_ids(TM_IMPL().v) - 1; }

// ---- Decoder (go :552-700; tokenmonster.cpp:1512-1721): tm_decoder_* keeps the incomplete character and the capcode state between calls ----
namespace {
template <class Call> Bytes drain(tm_decoder* d, std::size_t guess, Call&& first) {
  Bytes out(guess + 64);
  std::uint64_t n = 0;
  int rc = first(out.data(), out.size(), &n);
  if (rc == TM_E_NOSPACE) {                    // the ids have been consumed and the text is kept: fetch it with a buffer of the size reported
    out.resize((std::size_t)n);
    rc = tm_decoder_decode(d, nullptr, 0, out.data(), out.size(), &n);
  }
  check(rc, "decoder");
  out.resize((std::size_t)n);
  return out;
}
}  // namespace

std::vector<std::uint8_t> Decoder::decode(std::span<const std::uint32_t> tokens) {
  if (!vocab_ || !state_) throw Error("decoder has no vocabulary");
  tm_decoder* d = (tm_decoder*)state_.get();
  return drain(d, tokens.size() * 8, [&](std::uint8_t* o, std::uint64_t cap, std::uint64_t* n) {
    return tm_decoder_decode(d, tokens.data(), tokens.size(), o, cap, n);
  });
}

std::vector<std::uint8_t> Decoder::decode_serialized(std::span<const std::uint8_t> data, std::uint8_t encoding_length) {
  if (!vocab_ || !state_) throw Error("decoder has no vocabulary");
  tm_decoder* d = (tm_decoder*)state_.get();
  return drain(d, data.size() * 4, [&](std::uint8_t* o, std::uint64_t cap, std::uint64_t* n) {
    return tm_decoder_decode_serialized(d, data.data(), data.size(), encoding_length, o, cap, n);
  });
}

std::vector<std::uint32_t> Decoder::deserialize(std::span<const std::uint8_t> data, std::uint8_t encoding_length) const {
  if (!vocab_) throw Error("decoder has no vocabulary");
  return vocab_->deserialize(data, encoding_length);
}

std::vector<std::uint8_t> Decoder::flush() {
  if (!vocab_ || !state_) throw Error("decoder has no vocabulary");
  tm_decoder* d = (tm_decoder*)state_.get();
  Bytes out(64);
  std::uint64_t n = 0;
  int rc = tm_decoder_flush(d, out.data(), out.size(), &n);
  if (rc == TM_E_NOSPACE) { out.resize((std::size_t)n); rc = tm_decoder_flush(d, out.data(), out.size(), &n); }
  check(rc, "flush");
  out.resize((std::size_t)n);
  return out;
}

}  // namespace tokenmonster
