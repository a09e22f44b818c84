// oracle/capcode/capcode.hpp — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// The reference's C++ runtime (tokenmonster-cpp/src/tokenmonster.cpp:3, :478-479, :1422-1423,
// :1537-1538) consumes capcode as the external `capcode-cpp` library, which is NOT in
// /root/reference (tokenmonster-cpp/CMakeLists.txt:17-25).  To compile the reference sources
// into oracle/_ref we therefore have to supply <capcode/capcode.hpp> ourselves.
//
// This is a restatement of the only in-tree statement of capcode (level 2):
//   javascript/tokenmonster.js:872-1005  (capcode_encode)
//   javascript/tokenmonster.js:1007-1065 (CapcodeDecoder)
// on UTF-8 code points, with Unicode classes \p{Lu} \p{Ll} \p{L} \p{Nd} \p{M} taken from ICU
// (javascript/tokenmonster.js:880-898).
//
// PARITY UNPINNED: the Go module github.com/alasdairforsythe/capcode/go (no pinned version, no
// source in tree) is what the Go reference actually calls (go/tokenmonster.go:235-237); there is
// no golden vector for it anywhere in the reference.  Level 1 (NoCapcodeEncode/Decode, marker
// 0x7F) has no in-tree statement at all; it is restated here as "the level-2 rule for inserting
// the delete marker, with every case rule removed" and must be treated as a guess.
#pragma once

#include <cstdint>
#include <span>
#include <vector>

#include <unicode/uchar.h>

namespace capcode {

using Bytes = std::vector<std::uint8_t>;

namespace detail {

constexpr std::uint32_t kCharacterToken = 'C';  // javascript/tokenmonster.js:874
constexpr std::uint32_t kWordToken = 'W';       // :875
constexpr std::uint32_t kDeleteToken = 'D';     // :876
constexpr std::uint32_t kApostrophe = '\'';     // :877
constexpr std::uint32_t kApostrophe2 = 0x2019;  // :878
constexpr std::uint32_t kRawByte = 0x80000000U; // an undecodable byte carried through verbatim
constexpr std::uint32_t kNoCapcodeDelete = 0x7F;

inline bool is_upper(std::uint32_t r) { return r < kRawByte && u_charType((UChar32)r) == U_UPPERCASE_LETTER; }
inline bool is_lower(std::uint32_t r) { return r < kRawByte && u_charType((UChar32)r) == U_LOWERCASE_LETTER; }
inline bool is_letter(std::uint32_t r) {
  if (r >= kRawByte) return false;
  switch (u_charType((UChar32)r)) {
    case U_UPPERCASE_LETTER: case U_LOWERCASE_LETTER: case U_TITLECASE_LETTER:
    case U_MODIFIER_LETTER: case U_OTHER_LETTER: return true;
    default: return false;
  }
}
inline bool is_number(std::uint32_t r) { return r < kRawByte && u_charType((UChar32)r) == U_DECIMAL_DIGIT_NUMBER; }
inline bool is_modifier(std::uint32_t r) {
  if (r >= kRawByte) return false;
  switch (u_charType((UChar32)r)) {
    case U_NON_SPACING_MARK: case U_ENCLOSING_MARK: case U_COMBINING_SPACING_MARK: return true;
    default: return false;
  }
}
inline std::uint32_t to_lower(std::uint32_t r) { return r < kRawByte ? (std::uint32_t)u_tolower((UChar32)r) : r; }
inline std::uint32_t to_upper(std::uint32_t r) { return r < kRawByte ? (std::uint32_t)u_toupper((UChar32)r) : r; }

inline std::vector<std::uint32_t> decode_utf8(std::span<const std::uint8_t> s) {
  std::vector<std::uint32_t> out;
  out.reserve(s.size());
  std::size_t i = 0, n = s.size();
  while (i < n) {
    const std::uint8_t b0 = s[i];
    if (b0 < 0x80) { out.push_back(b0); i++; continue; }
    std::size_t need = 0;
    std::uint32_t cp = 0;
    if (b0 >= 0xC2 && b0 < 0xE0) { need = 1; cp = b0 & 0x1F; }
    else if (b0 >= 0xE0 && b0 < 0xF0) { need = 2; cp = b0 & 0x0F; }
    else if (b0 >= 0xF0 && b0 < 0xF5) { need = 3; cp = b0 & 0x07; }
    bool ok = need > 0 && i + need < n;
    for (std::size_t k = 1; ok && k <= need; k++) {
      if ((s[i + k] & 0xC0) != 0x80) ok = false;
      else cp = (cp << 6) | (s[i + k] & 0x3F);
    }
    if (ok && ((need == 2 && cp < 0x800) || (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) ||
               (cp >= 0xD800 && cp <= 0xDFFF))) ok = false;
    if (ok) { out.push_back(cp); i += 1 + need; }
    else { out.push_back(kRawByte | b0); i++; }
  }
  return out;
}

inline void append_utf8(Bytes& out, std::uint32_t r) {
  if (r >= kRawByte) { out.push_back((std::uint8_t)(r & 0xFF)); return; }
  if (r < 0x80) out.push_back((std::uint8_t)r);
  else if (r < 0x800) { out.push_back(0xC0 | (r >> 6)); out.push_back(0x80 | (r & 0x3F)); }
  else if (r < 0x10000) { out.push_back(0xE0 | (r >> 12)); out.push_back(0x80 | ((r >> 6) & 0x3F)); out.push_back(0x80 | (r & 0x3F)); }
  else { out.push_back(0xF0 | (r >> 18)); out.push_back(0x80 | ((r >> 12) & 0x3F)); out.push_back(0x80 | ((r >> 6) & 0x3F)); out.push_back(0x80 | (r & 0x3F)); }
}

inline Bytes encode_utf8(const std::vector<std::uint32_t>& v) {
  Bytes out;
  out.reserve(v.size() + v.size() / 4);
  for (auto r : v) append_utf8(out, r);
  return out;
}

inline bool word_joiner(std::uint32_t rlast) {
  // javascript/tokenmonster.js:915, :954 — "isLetter(rlast) || apostrophe || apostrophe2 || isModifier(rlast)"
  return is_letter(rlast) || rlast == kApostrophe || rlast == kApostrophe2 || is_modifier(rlast);
}

}  // namespace detail

// javascript/tokenmonster.js:900-1005
inline Bytes encode(std::span<const std::uint8_t> data) {
  using namespace detail;
  const auto in = decode_utf8(data);
  std::vector<std::uint32_t> buf;
  buf.reserve(in.size() + in.size() / 2 + 8);
  std::size_t gobackPos = 0, wordTokenPos = 0;
  std::uint32_t rlast = '.', rlast2 = '.';
  bool inWord = false, multiLetter = false;

  for (std::uint32_t r : in) {
    if (inWord) {
      if (is_upper(r)) {                                              // :913-919
        if (!word_joiner(rlast)) { buf.push_back(kDeleteToken); buf.push_back(' '); }
        multiLetter = true;
        buf.push_back(to_lower(r));
      } else {
        if (is_lower(r)) {                                            // :921
          inWord = false;
          buf[wordTokenPos] = kCharacterToken;                        // :923
          if (multiLetter) {                                          // :924-951
            for (std::size_t i2 = gobackPos; i2 < buf.size(); i2++) {
              if (buf[i2] == kDeleteToken && i2 + 1 < buf.size() && buf[i2 + 1] == ' ') {
                if (i2 + 2 < buf.size() && is_lower(buf[i2 + 2])) {
                  // "D x" -> "DC x": one marker inserted after the D
                  buf.insert(buf.begin() + (std::ptrdiff_t)(i2 + 1), kCharacterToken);
                  i2++;
                }
                i2 += 2;
              } else if (is_lower(buf[i2])) {
                // "x" -> "DC x"
                const std::uint32_t ins[3] = {kDeleteToken, kCharacterToken, ' '};
                buf.insert(buf.begin() + (std::ptrdiff_t)i2, ins, ins + 3);
                i2 += 3;
              }
            }
          }
          if (!word_joiner(rlast)) { buf.push_back(kDeleteToken); buf.push_back(' '); }  // :952-955
        } else {
          if (is_number(r)) {                                         // :957-961
            if (!is_number(rlast)) { buf.push_back(kDeleteToken); buf.push_back(' '); }
          } else if (!(r == kApostrophe || r == kApostrophe2 || is_modifier(r))) {
            inWord = false;                                           // :962-964
          }
        }
        buf.push_back(r);                                             // :966
      }
    } else {
      if (is_lower(r)) {                                              // :969-974
        if (!(rlast == ' ' || is_letter(rlast) ||
              (is_letter(rlast2) && (rlast == kApostrophe || rlast == kApostrophe2)) || is_modifier(rlast))) {
          buf.push_back(kDeleteToken); buf.push_back(' ');
        }
        buf.push_back(r);
      } else if (is_upper(r)) {                                       // :975-990
        if (rlast == ' ') {
          wordTokenPos = buf.size() - 1;
          buf[wordTokenPos] = kWordToken;
          buf.push_back(' ');
        } else {
          buf.push_back(kDeleteToken);
          wordTokenPos = buf.size();
          buf.push_back(kWordToken);
          buf.push_back(' ');
        }
        buf.push_back(to_lower(r));
        gobackPos = buf.size();
        multiLetter = false;
        inWord = true;
      } else if (is_number(r)) {                                      // :991-996
        if (!(rlast == ' ' || is_number(rlast))) { buf.push_back(kDeleteToken); buf.push_back(' '); }
        buf.push_back(r);
      } else {
        buf.push_back(r);                                             // :997-999
      }
    }
    rlast2 = rlast;
    rlast = r;
  }
  return encode_utf8(buf);
}

// Level 1: NO in-tree statement (see header).  Delete-marker insertion only, case untouched.
inline Bytes no_capcode_encode(std::span<const std::uint8_t> data) {
  using namespace detail;
  const auto in = decode_utf8(data);
  std::vector<std::uint32_t> buf;
  buf.reserve(in.size() + in.size() / 4 + 8);
  std::uint32_t rlast = '.', rlast2 = '.';
  for (std::uint32_t r : in) {
    if (is_letter(r)) {
      if (!(rlast == ' ' || is_letter(rlast) ||
            (is_letter(rlast2) && (rlast == kApostrophe || rlast == kApostrophe2)) || is_modifier(rlast))) {
        buf.push_back(kNoCapcodeDelete); buf.push_back(' ');
      }
    } else if (is_number(r)) {
      if (!(rlast == ' ' || is_number(rlast))) { buf.push_back(kNoCapcodeDelete); buf.push_back(' '); }
    }
    buf.push_back(r);
    rlast2 = rlast;
    rlast = r;
  }
  return encode_utf8(buf);
}

// javascript/tokenmonster.js:1007-1065
struct Decoder {
  bool inWord = false, inChar = false, del = false, ignore = false;

  Bytes decode(const Bytes& data) {
    using namespace detail;
    Bytes out;
    out.reserve(data.size());
    for (std::uint32_t r : decode_utf8(data)) {
      if (r == kCharacterToken) { inChar = true; inWord = false; continue; }
      if (r == kWordToken) { inWord = true; inChar = false; ignore = true; continue; }
      if (r == kDeleteToken) { del = true; continue; }
      if (r == ' ') {
        if (del) { del = false; }
        else { out.push_back(' '); if (!ignore) inWord = false; }
      } else {
        if (del) { del = false; }
        else if (inChar) { inChar = false; append_utf8(out, to_upper(r)); }
        else if (inWord) {
          if (is_lower(r) || is_upper(r)) append_utf8(out, to_upper(r));
          else {
            append_utf8(out, r);
            if (!(is_number(r) || r == kApostrophe || r == kApostrophe2 || is_modifier(r))) inWord = false;
          }
        } else append_utf8(out, r);
      }
      ignore = false;
    }
    return out;
  }

  Bytes no_capcode_decode(const Bytes& data) {
    using namespace detail;
    Bytes out;
    out.reserve(data.size());
    for (std::uint8_t b : data) {
      if (b == kNoCapcodeDelete) { del = true; continue; }
      if (del) { del = false; continue; }
      out.push_back(b);
    }
    return out;
  }
};

inline Bytes decode(Bytes data) { Decoder d; return d.decode(data); }
inline Bytes no_capcode_decode(Bytes data) { Decoder d; return d.no_capcode_decode(data); }

}  // namespace capcode
