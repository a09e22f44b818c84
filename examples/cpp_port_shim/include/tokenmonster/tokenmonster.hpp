// examples/cpp_port_shim/include/tokenmonster/tokenmonster.hpp — the public C++ interface of the reference's own C++ port
// (tokenmonster-cpp/include/tokenmonster/tokenmonster.hpp:52-115: namespace tokenmonster, Vocab::load / tokenize / count /
// tokenize_serialized / decode / new_decoder ..., Decoder, the result structs), re-declared over libtokenmonster_hip.so's C ABI
// (include/tokenmonster_hip.h, include/tm_build.h).  A program written against the port — its own tests/unit.cpp and tests/bench.cpp
// compile UNMODIFIED against this header (examples/cpp_port_shim/Makefile) — links tokenmonster_shim.cpp + the HIP library instead of
// tokenmonster.cpp and runs the tokenizer on the GPU.  Same names, argument meaning and error behaviour (tokenmonster::Error); nothing
// here includes or copies the port's sources, and no ICU / capcode-cpp is needed by the caller: normalization, capcode and the walk
// all live behind the C ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <memory>
#include <optional>
#include <span>
#include <stdexcept>
#include <vector>

namespace tokenmonster {

constexpr std::uint32_t does_not_exist = 0xFFFFFFU;      // go/tokenmonster.go:32 DOES_NOT_EXIST

class Error : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

struct TokenizeResult { std::vector<std::uint32_t> tokens; int missing = 0; };
struct CountResult { int tokens = 0; int missing = 0; };
struct SerializedResult { std::vector<std::uint8_t> bytes; std::uint8_t encoding_length = 0; int missing = 0; };
struct Info {
  std::uint32_t id = 0;
  std::vector<std::uint8_t> token, token_decoded;
  std::uint8_t type = 0;      // 0 regular, 1 single byte, 2 special, 3 unk
  float score = 0.0F;
};

class Vocab;

// streaming decoder (go/tokenmonster.go:552-700): tm_decoder_* behind it
class Decoder {
 public:
  Decoder() = default;
  std::vector<std::uint8_t> decode(std::span<const std::uint32_t> tokens);
  std::vector<std::uint8_t> decode_serialized(std::span<const std::uint8_t> data, std::uint8_t encoding_length = 0);
  std::vector<std::uint32_t> deserialize(std::span<const std::uint8_t> data, std::uint8_t encoding_length = 0) const;
  std::vector<std::uint8_t> flush();

 private:
  friend class Vocab;
  const Vocab* vocab_ = nullptr;
  std::shared_ptr<void> state_;       // tm_decoder*, freed with the last copy
};

class Vocab {
 public:
  Vocab();
  ~Vocab();
  Vocab(Vocab&&) noexcept;
  Vocab& operator=(Vocab&&) noexcept;
  Vocab(const Vocab&) = delete;
  Vocab& operator=(const Vocab&) = delete;

  static Vocab load(const std::filesystem::path& path);

  std::vector<std::uint8_t> normalize(std::span<const std::uint8_t> data) const;
  TokenizeResult tokenize(std::span<const std::uint8_t> data) const;
  CountResult count(std::span<const std::uint8_t> data) const;
  SerializedResult tokenize_serialized(std::span<const std::uint8_t> data, std::uint8_t encoding_length = 0) const;

  std::vector<std::uint8_t> decode(std::span<const std::uint32_t> tokens) const;
  std::vector<std::uint8_t> decode_serialized(std::span<const std::uint8_t> data, std::uint8_t encoding_length = 0) const;
  std::vector<std::uint32_t> deserialize(std::span<const std::uint8_t> data, std::uint8_t encoding_length = 0) const;

  Decoder new_decoder() const;

  std::vector<Info> tokens_detailed() const;
  std::vector<Info> special_tokens() const;
  std::vector<std::vector<std::uint8_t>> tokens() const;
  std::optional<std::vector<std::uint8_t>> id_to_token(std::uint32_t id) const;
  std::optional<std::uint32_t> token_to_id(std::span<const std::uint8_t> token) const;
  std::vector<std::uint8_t> denormalize(std::span<const std::uint8_t> token) const;

  std::uint32_t unk() const;
  bool has_unk() const { return unk() != does_not_exist; }
  int size() const;
  int max_token_length() const;
  std::uint8_t charset() const;
  std::uint8_t capcode() const;
  std::uint8_t mode() const;
  std::uint8_t normalization_code() const;
  int highest_token_id() const;

  // the walk on text that is normalized already (private in the port; its bench reaches them through `#define private public`)
  TokenizeResult tokenize_normalized(std::span<const std::uint8_t> normalized) const;
  CountResult tokenize_count_normalized(std::span<const std::uint8_t> normalized) const;

 private:
  friend class Decoder;
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace tokenmonster
