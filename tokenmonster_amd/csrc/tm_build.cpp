// tm_build.cpp — vocabulary builder: per-token metadata rules of the reference restated in C++.
//
// Follows go/tokenmonster.go:3423-3793 (== training/trainvocab.go:548-907, the per-candidate table
// build of the trainvocab worker) and writes go/tokenmonster.go:2602-2653's .vocab layout.
// Unicode classes come from ICU (Go: unicode.IsLetter / IsNumber / Mn,Mc,Me / IsSpace).
#include "tm_build.h"
#include "tokenmonster_hip.h"
#include "tm_internal.h"
#include "tm_device.h"

#include <unicode/uchar.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace tmh {

namespace {

constexpr uint32_t kRuneError = 0xFFFD;

// utf8.DecodeRune semantics: (RuneError,0) on empty, (RuneError,1) on invalid
struct Rune { uint32_t r; int n; };

Rune decode_utf8(const uint8_t* b, size_t len) {
  if (len == 0) return {kRuneError, 0};
  uint8_t b0 = b[0];
  if (b0 < 0x80) return {b0, 1};
  if (b0 < 0xC2 || b0 > 0xF4) return {kRuneError, 1};
  int need = b0 < 0xE0 ? 1 : (b0 < 0xF0 ? 2 : 3);
  if ((size_t)need + 1 > len) return {kRuneError, 1};
  uint32_t cp = b0 & (need == 1 ? 0x1F : need == 2 ? 0x0F : 0x07);
  for (int k = 1; k <= need; k++) {
    if ((b[k] & 0xC0) != 0x80) return {kRuneError, 1};
    cp = (cp << 6) | (b[k] & 0x3F);
  }
  if ((need == 2 && cp < 0x800) || (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF))
    return {kRuneError, 1};
  return {cp, need + 1};
}

// go/tokenmonster.go:371-400
Rune decode_rune(const uint8_t* b, size_t len, uint32_t charset) {
  if (charset != 2) return decode_utf8(b, len);
  if (len < 2) return {kRuneError, 0};
  uint32_t u = b[0] | (b[1] << 8);
  if (u >= 0xD800 && u <= 0xDBFF) {
    if (len < 4) return {kRuneError, 0};
    uint32_t u2 = b[2] | (b[3] << 8);
    if (u2 < 0xDC00 || u2 > 0xDFFF) return {kRuneError, 0};
    return {0x10000 + ((u - 0xD800) << 10) + (u2 - 0xDC00), 4};
  }
  return {u, 2};
}

// go/tokenmonster.go:402-430 ; utf8.DecodeLastRune
uint32_t decode_last_rune(const uint8_t* b, size_t len, uint32_t charset) {
  if (charset == 2) {
    if (len < 2) return kRuneError;
    uint32_t u = b[len - 2] | (b[len - 1] << 8);
    if (u >= 0xDC00 && u <= 0xDFFF) {
      if (len < 4) return kRuneError;
      uint32_t u2 = b[len - 4] | (b[len - 3] << 8);
      if (u2 < 0xD800 || u2 > 0xDBFF) return kRuneError;
      return 0x10000 + ((u2 - 0xD800) << 10) + (u - 0xDC00);
    }
    return u;
  }
  if (len == 0) return kRuneError;
  size_t start = len - 1;
  size_t lim = len >= 4 ? len - 4 : 0;
  while (start > lim && (b[start] & 0xC0) == 0x80) start--;
  Rune r = decode_utf8(b + start, len - start);
  if (start + (size_t)r.n != len) return kRuneError;
  return r.r;
}

// (ASCII answered without the ICU property trie: most runes of a vocabulary are ASCII and the builder classifies each several times)
inline bool uni_letter(uint32_t r) { return r < 0x80 ? ((r | 0x20u) - 'a') < 26u : (U_GET_GC_MASK((UChar32)r) & U_GC_L_MASK) != 0; }
inline bool uni_number(uint32_t r) { return r < 0x80 ? (r - '0') < 10u : (U_GET_GC_MASK((UChar32)r) & U_GC_N_MASK) != 0; }
inline bool uni_mark(uint32_t r) { return r < 0x80 ? false : (U_GET_GC_MASK((UChar32)r) & U_GC_M_MASK) != 0; }
bool uni_space(uint32_t r) {
  if (r < 0x100) return r == '\t' || r == '\n' || r == '\v' || r == '\f' || r == '\r' || r == ' ' || r == 0x85 || r == 0xA0;
  return u_hasBinaryProperty((UChar32)r, UCHAR_WHITE_SPACE);
}

// go/tokenmonster.go:359-369
bool is_letter(uint32_t r, uint32_t cc) {
  return (uni_letter(r) && (cc != 2 || (r != 'W' && r != 'C' && r != 'D'))) || uni_mark(r);
}
bool is_alnum(uint32_t r, uint32_t cc) { return is_letter(r, cc) || uni_number(r); }
bool is_capcode(uint32_t r, uint32_t cc) {
  return (cc == 1 && r == 0x7F) || (cc == 2 && (r == 'C' || r == 'W' || r == 'D'));
}

bool key_less(std::string_view a, std::string_view b) {
  if (a.size() != b.size()) return a.size() < b.size();
  return std::memcmp(a.data(), b.data(), a.size()) < 0;
}

// go/tokenmonster.go:287-299 with ungreedySuffixes {"'s", "’s"} (:3157)
int has_suffix_pos(std::string_view key, uint32_t charset, uint32_t cc) {
  static const std::string_view sfx[2] = {"'s", "\xE2\x80\x99s"};
  for (const auto& s : sfx) {
    if (key.size() >= s.size() && key.compare(key.size() - s.size(), s.size(), s) == 0) {
      if (s.size() < key.size()) {
        uint32_t r = decode_last_rune((const uint8_t*)key.data(), key.size() - s.size(), charset);
        if (is_letter(r, cc)) return (int)(key.size() - s.size());
      }
    }
  }
  return -1;
}

}  // namespace

// The builder proper: token list -> the records of a .vocab file (keys in pansearch order with flag, nWords, the two alternatives, id,
// score; header; beginByte) in `hv`, and - because the search for a token's alternatives is a search among its own prefixes (go :3597) -
// the byte trie of the keys in `trie`, built on the way: the trie's path to a key knows which of its prefixes are keys, so the search costs
// no look-ups of its own, and tm_vocab_build hands the same trie on to the table construction (tm_vocab.hip: build_tables) instead of
// serializing an image and parsing it again.
int build_vocab_records(const std::vector<std::string>& tokens_in, const std::vector<uint8_t>& special_in, uint32_t capcode, uint32_t charset,
                        uint32_t norm_flag, uint32_t level, bool with_unk, HostVocab& hv, Trie& trie, const std::vector<float>* token_scores) {
  if (capcode > 2 || charset > 2) return set_error(TM_E_INVALID, "capcode/charset out of range");
  const bool trace = getenv("TM_TRACE_BUILD") != nullptr;
  auto tprev = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (!trace) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "  [build]  %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - tprev).count());
    tprev = n;
  };
  // ---- dic1: unique tokens in (length, bytewise) order  (go :3364-3380) -------------------------
  struct Tok { std::string_view s; bool special; };
  std::vector<Tok> dic1;
  dic1.reserve(tokens_in.size());
  std::unordered_map<std::string_view, float> given_scores;   // the score column of the .vocab records (go :2636); 1.0 when not given
  for (size_t k = 0; k < tokens_in.size(); k++) {
    if (tokens_in[k].empty()) continue;
    if (tokens_in[k].size() > 40) return set_error(TM_E_INVALID, "token longer than 40 bytes");
    dic1.push_back(Tok{std::string_view(tokens_in[k]), k < special_in.size() && special_in[k] != 0});
    if (token_scores && k < token_scores->size()) given_scores[std::string_view(tokens_in[k])] = std::max(0.0f, (*token_scores)[k]);
  }
  std::stable_sort(dic1.begin(), dic1.end(), [](const Tok& a, const Tok& b) { return key_less(a.s, b.s); });
  {   // unique; a token listed twice is special if any of its listings is (the first listing decided before; listings are not repeated in practice)
    size_t w = 0;
    for (size_t r = 0; r < dic1.size(); r++) { if (w && dic1[w - 1].s == dic1[r].s) continue; dic1[w++] = dic1[r]; }
    dic1.resize(w);
  }
  if (dic1.size() >= TM_NONE - 2) return set_error(TM_E_LIMIT, "too many tokens");
  mark("sort tokens");

  // ---- IDs + "D "-duplicates  (go :3423-3470) ---------------------------------------------------
  // A token that begins with a letter or digit is also listed as "D " + token under the same id (go :3450-3462).  Real tokens R and
  // duplicates D are both in (length, bytewise) order - the prefix is a constant - so the dictionary's key list is a merge of the two, and
  // what the Go code keeps in idsMap / scoresMap / specialMap falls out of the merge: a real token takes the next free id unless it IS the
  // duplicate of an earlier (shorter) token, whose id it then shares.
  const char add0 = capcode == 1 ? (char)0x7F : 'D';
  std::vector<char> arena;                              // the "D "-prefixed copies (reserved up front: views into it stay valid)
  { size_t need = 0; for (auto& d : dic1) need += d.s.size() + 2; arena.reserve(need); }
  struct Dup { std::string_view s; uint32_t base; };
  std::vector<Dup> dups;
  size_t n_single = 0;
  for (uint32_t k = 0; k < dic1.size(); k++) {
    const std::string_view tok = dic1[k].s;
    if (tok.size() == 1) n_single++;
    Rune r = decode_rune((const uint8_t*)tok.data(), tok.size(), charset);
    if (capcode != 0 && is_alnum(r.r, capcode) && tok.size() + 2 <= 40) {
      const size_t at = arena.size();
      arena.push_back(add0); arena.push_back(' ');
      arena.insert(arena.end(), tok.begin(), tok.end());
      dups.push_back(Dup{std::string_view(arena.data() + at, tok.size() + 2), k});
    }
  }
  struct Key { std::string_view s; uint32_t id; bool neg, special; };
  std::vector<Key> keys;
  keys.reserve(dic1.size() + dups.size());
  std::vector<uint32_t> rid(dic1.size(), TM_NONE);      // id of every real token
  uint32_t next_id = 0;
  {
    size_t ir = 0, id_ = 0;
    while (ir < dic1.size() || id_ < dups.size()) {
      const bool has_r = ir < dic1.size(), has_d = id_ < dups.size();
      const bool same = has_r && has_d && dic1[ir].s == dups[id_].s;
      if (same) {                                        // a real token that is the duplicate of an earlier one: that one's id, no id of its own
        const uint32_t base = dups[id_].base;
        rid[ir] = rid[base];
        keys.push_back(Key{dic1[ir].s, rid[base], true, dic1[ir].special || dic1[base].special});
        ir++; id_++;
      } else if (has_r && (!has_d || key_less(dic1[ir].s, dups[id_].s))) {
        rid[ir] = next_id++;
        keys.push_back(Key{dic1[ir].s, rid[ir], false, dic1[ir].special});
        ir++;
      } else {
        const uint32_t base = dups[id_].base;            // (shorter than its duplicate: it has its id already)
        keys.push_back(Key{dups[id_].s, rid[base], true, dic1[base].special});
        id_++;
      }
    }
  }
  const uint32_t n_tokens = next_id;
  // unk: id = number of tokens (go :3382-3398); canHaveUnkToken (go :437-442)
  uint32_t unk = TM_NONE;
  if (with_unk && ((n_single < 256 && capcode != 2) || n_single < 233)) unk = n_tokens;
  const uint32_t vocab_size = n_tokens + (unk != TM_NONE ? 1 : 0);
  const uint32_t n_info = (uint32_t)keys.size();
  if (n_info >= kMaxNodes) return set_error(TM_E_LIMIT, "%u index records: the walk tables hold fewer than %u trie nodes", n_info, kMaxNodes);
  mark("ids + duplicates, key order");

  // ---- the records' keys; header -----------------------------------------------------------------
  hv = HostVocab();
  hv.capcode = (uint8_t)capcode; hv.charset = (uint8_t)charset; hv.norm_flag = (uint8_t)norm_flag; hv.level = (uint8_t)level; hv.reserve = 0;
  hv.unk = unk; hv.vocab_size = vocab_size; hv.n_ids = vocab_size; hv.n_info = n_info;
  if (hv.n_ids > kRowIdMask) return set_error(TM_E_LIMIT, "%u ids: the device tables hold at most %u", hv.n_ids, kRowIdMask);
  hv.key_off.assign(1, 0);
  hv.key_off.reserve((size_t)n_info + 1);
  { size_t tot = 0; for (auto& k : keys) tot += k.s.size(); hv.keys.reserve(tot); }
  uint32_t max_len = 0;
  for (auto& k : keys) {
    hv.keys.insert(hv.keys.end(), k.s.begin(), k.s.end());
    hv.key_off.push_back((uint32_t)hv.keys.size());
    max_len = std::max<uint32_t>(max_len, (uint32_t)k.s.size());
  }
  hv.max_len = max_len;
  hv.rec_flag.assign(n_info, 0); hv.rec_nwords.assign(n_info, 0); hv.rec_id.resize(n_info); hv.rec_index1.assign(n_info, TM_NONE); hv.rec_index2.assign(n_info, TM_NONE);
  hv.rec_score.resize(n_info);
  // deleteToken index (go :3474-3483): the one-byte key 'D' (0x7F with capcode 1)
  uint32_t delete_index = TM_NONE;
  if (capcode != 0) for (uint32_t i = 0; i < n_info && keys[i].s.size() == 1; i++) if (keys[i].s[0] == add0) delete_index = i;

  // ---- per-record metadata (go :3486-3777), one key at a time as the trie reaches it ---------------
  struct Meta : TrieVisitor {
    const std::vector<Key>& keys; HostVocab& hv; uint32_t capcode, charset;
    const std::unordered_map<std::string_view, float>* given;
    uint32_t begin_count[256][4];
    Meta(const std::vector<Key>& k, HostVocab& h, uint32_t cc, uint32_t cs, const std::unordered_map<std::string_view, float>* g) : keys(k), hv(h), capcode(cc), charset(cs), given(g) {
      std::memset(begin_count, 0, sizeof begin_count);
    }
    void key(uint32_t on, const uint32_t* path_ord) override {
      const Key& self = keys[on];
      const std::string_view token = self.s;
      const uint8_t* t = (const uint8_t*)token.data();
      const size_t tl = token.size();
      hv.rec_id[on] = self.id;
      float score = self.neg ? -1.0f : 1.0f;
      if (!self.neg && given) { auto g = given->find(token); if (g != given->end()) score = g->second; }
      hv.rec_score[on] = score;
      if (self.special) { hv.rec_flag[on] = 64; return; }   // go :3504-3511
      uint8_t flag = 0, n_words = 0, priority1 = 0, priority2 = 0;
      int min_alt = 1, alt_len1 = 0, alt_len2 = 0;
      bool only_letter_space = false, only_number_space = false, only_punc = false;
      Rune d1 = decode_rune(t, tl, charset);
      Rune d2 = decode_rune(t + d1.n, tl - d1.n, charset);
      uint32_t r = d1.r, r2 = d2.r;
      int n = d1.n, n2 = d2.n;
      // beginning of token (go :3522-3542)
      if (r == ' ') {
        flag = 4; begin_count[t[0]][0]++;
        if (is_alnum(r2, capcode)) { n_words++; min_alt = 2; }
      } else if (is_letter(r, capcode)) {
        flag = 2; begin_count[t[0]][1]++;
      } else if (is_capcode(r, capcode)) {
        if (r == 'C' || r == 'W') flag = 4;
        flag |= 16; begin_count[t[0]][3]++;
      } else if (uni_number(r)) {
        begin_count[t[0]][2]++;
      } else {
        begin_count[t[0]][3]++;
      }
      // words in token (go :3544-3572)
      if (tl == 1) {
        only_punc = true;
      } else {
        if ((r == ' ' || is_letter(r, capcode)) && is_letter(r2, capcode)) only_letter_space = true;
        else if ((r == ' ' || uni_number(r)) && uni_number(r2)) only_number_space = true;
        else if (!is_alnum(r, capcode) && !is_alnum(r2, capcode)) only_punc = true;
        for (size_t i = (size_t)(n + n2); i < tl; i += (size_t)n2) {
          r = r2; n = n2;
          Rune d = decode_rune(t + i, tl - i, charset);
          r2 = d.r; n2 = d.n;
          if (n2 <= 0) break;  // (Go would spin on a malformed UTF-16 tail; nothing to classify)
          if (r == ' ' && is_alnum(r2, capcode)) n_words++;
          if (is_letter(r2, capcode)) { only_punc = false; only_number_space = false; }
          else if (uni_number(r2)) { only_punc = false; only_letter_space = false; }
          else if (r2 != ' ') { only_letter_space = false; only_number_space = false; }
        }
      }
      // go :3575-3593
      r = decode_last_rune(t, tl, charset);
      if (min_alt == 2 && is_letter(r, capcode) && only_letter_space && n_words == 1) flag |= 32;
      if (min_alt == 2 && n_words <= 1) min_alt = 1;
      if (is_capcode(r, capcode)) flag |= 8;
      if (is_letter(r, capcode)) flag |= 1;
      if (only_letter_space || only_number_space || only_punc) flag |= 128;

      const int has_suffix = has_suffix_pos(token, charset, capcode);
      uint32_t index1 = TM_NONE, index2 = TM_NONE;
      // slot choice shared by every rule: go :3606 etc.
      auto offer = [&](uint8_t prio, uint32_t index, int length) {
        if (priority1 < priority2 || (priority1 == priority2 && alt_len1 <= alt_len2)) {
          if (priority1 < prio) { index1 = index; alt_len1 = length; priority1 = prio; }
        } else {
          if (priority2 < prio) { index2 = index; alt_len2 = length; priority2 = prio; }
        }
      };
      for (int length = (int)tl - 1; length >= min_alt; length--) {      // go :3597
        const uint32_t index = path_ord[length - 1];                     // dictionary.Find(token[:length]): the trie's path knows (kNone: not a key)
        if (index == kNone) continue;
        // anything | space + letter-or-number (go :3602-3621)
        if (length <= (int)tl - 2 && t[length] == ' ') {
          Rune d = decode_rune(t + length + 1, tl - (size_t)length - 1, charset);
          if (is_letter(d.r, capcode) || uni_number(d.r)) { offer(10, index, length); continue; }
        }
        uint32_t ra = decode_last_rune(t, (size_t)length, charset);                     // go :3624
        uint32_t rb = decode_rune(t + length, tl - (size_t)length, charset).r;         // go :3625
        if (capcode == 0) {                                                              // go :3627-3647
          if (((!is_letter(ra, capcode) && ra != '_') && (is_letter(rb, capcode) || rb == '_')) ||
              (!uni_number(ra) && uni_number(rb))) { offer(9, index, length); continue; }
        }
        if (((is_letter(ra, capcode) || ra == '_') && (!is_letter(rb, capcode) && rb != '_')) ||
            (uni_number(ra) && !uni_number(rb))) { offer(9, index, length); continue; }   // go :3651-3668
        if (uni_space(ra) && !uni_space(rb)) { offer(7, index, length); continue; }        // go :3670
        if (!uni_space(ra) && uni_space(rb)) { offer(8, index, length); continue; }        // go :3686
        if (is_capcode(rb, capcode)) { offer(9, index, length); continue; }                // go :3702
        if (length == has_suffix) { offer(8, index, length); break; }                      // go :3720-3735 (Q7: break)
        offer(1, index, length);                                                           // go :3738-3750
      }
      // go :3761-3764
      if (alt_len2 > 0 && (priority2 > priority1 || (priority2 == priority1 && alt_len2 > alt_len1))) {
        std::swap(index1, index2); std::swap(alt_len1, alt_len2);
      }
      hv.rec_flag[on] = flag; hv.rec_nwords[on] = n_words;
      hv.rec_index1[on] = alt_len1 > 0 ? index1 : TM_NONE;
      hv.rec_index2[on] = alt_len2 > 0 ? index2 : TM_NONE;
    }
  } meta(keys, hv, capcode, charset, token_scores ? &given_scores : nullptr);
  { int rc = build_trie(hv, trie, &meta); if (rc != TM_OK) return rc; }
  for (uint32_t i = 0; i < n_info; i++) if (hv.rec_nwords[i] > 31) return set_error(TM_E_LIMIT, "record %u: nWords %u > 31", i, hv.rec_nwords[i]);
  mark("trie + metadata + alternatives");

  // ---- beginByte (go :3779-3788) ----------------------------------------------------------------
  for (int i = 0; i < 256; i++) {
    const uint32_t* c = meta.begin_count[i];
    hv.begin_byte[i] = 0;
    if (c[1] > c[0] && c[1] > c[2] && c[1] > c[3] && c[1] > 2) hv.begin_byte[i] = 1;
    else if (c[0] > c[1] && c[0] > c[2] && c[0] > c[3] && c[0] > 2) hv.begin_byte[i] = 12;
    else if (c[3] > c[0] && c[3] > c[1] && c[3] > c[2] && c[3] > 2) hv.begin_byte[i] = 10;
  }
  hv.delete_id = delete_index != TM_NONE ? hv.rec_id[delete_index] : TM_NONE;  // go :3791-3793
  return TM_OK;
}

// the same as the bytes of a .vocab file (go :2602-2653)
int build_vocab_image(const std::vector<std::string>& tokens_in, const std::vector<uint8_t>& special_in,
                      uint32_t capcode, uint32_t charset, uint32_t norm_flag, uint32_t level, bool with_unk,
                      std::vector<uint8_t>& image, const std::vector<float>* token_scores) {
  HostVocab hv;
  Trie trie;
  int rc = build_vocab_records(tokens_in, special_in, capcode, charset, norm_flag, level, with_unk, hv, trie, token_scores);
  if (rc != TM_OK) return rc;
  image = vocab_image(hv);
  return TM_OK;
}

}  // namespace tmh

extern "C" {

void tm_free(void* p) { std::free(p); }

int tm_build_vocab(const uint8_t* blob, const uint32_t* off, uint32_t n_tokens, const uint8_t* special,
                   uint32_t capcode, uint32_t charset, uint32_t norm_flag, uint32_t level, int with_unk,
                   uint8_t** out, size_t* out_n) {
  if (!out || !out_n || (n_tokens && (!blob || !off))) return tmh::set_error(TM_E_INVALID, "null argument");
  std::vector<std::string> toks(n_tokens);
  std::vector<uint8_t> sp(n_tokens, 0);
  for (uint32_t k = 0; k < n_tokens; k++) {
    toks[k].assign((const char*)blob + off[k], off[k + 1] - off[k]);
    if (special) sp[k] = special[k];
  }
  std::vector<uint8_t> image;
  int rc = tmh::build_vocab_image(toks, sp, capcode, charset, norm_flag, level, with_unk != 0, image);
  if (rc != TM_OK) return rc;
  *out = (uint8_t*)std::malloc(image.size());
  std::memcpy(*out, image.data(), image.size());
  *out_n = image.size();
  return TM_OK;
}

}  // extern "C"
