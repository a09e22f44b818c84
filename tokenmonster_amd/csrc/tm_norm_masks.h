// tm_norm_masks.h — the parts of capcode level 2 (javascript/tokenmonster.js:900-1005) that the device normalizer
// (tm_norm.hip, k_norm_emit2) does NOT evaluate lane by lane:
//   * the two facts about a byte that depend on an unbounded stretch of text — W "inWord" and T "the block ends in a
//     lower-case letter" — are flood fills on the 64-bit ballots of a chunk's character classes (a handful of scalar
//     instructions per 64 bytes instead of per-lane popcount / find-first arithmetic);
//   * the rule table itself: what a byte turns into is a function of (its class, the class before it, the class before an
//     apostrophe before it, W, T) — 2048 cases, tabulated once at compile time (nm_lut_entry) and read from LDS.
// Everything here is __host__ __device__ (and constexpr where it is a table) so that the exact code the kernel runs is
// checked on the CPU against the host normalizer: tools/norm_masks_check.cpp, run by tests/test_builder_normalizer.py.
//
// Vocabulary (see tm_norm.hip): a *block* is a maximal run of {capital, digit, apostrophe} bytes;
//   W[i]  there is a capital in the block bytes immediately before byte i            (the encoder's inWord state)
//   T[i]  for a block byte: the first byte after its block is a lower-case letter   ('C' marker, else 'W')
// Bit i of a mask is byte i of the chunk; carries between chunks are single bits.
#pragma once
#include <cstdint>

#if defined(__HIP__)
#define TM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TM_HD inline
#endif

namespace tmh {

// character classes (3 bits; bit 2 = member of a block) and flag bits of a classified byte
//   O other   L lower-case letter   SP space   LO letter that is neither upper nor lower case (ª º: isLetter only)
//   U capital   N digit   AP apostrophe (' and U+2019)   M combining mark (isModifier: here always the second half of a decomposed letter)
enum : uint32_t { NC_O = 0, NC_L = 1, NC_SP = 2, NC_LO = 3, NC_U = 4, NC_N = 5, NC_AP = 6, NC_M = 7 };
constexpr uint32_t NF_CLASS = 7u, NF_BLOCK = 4u, NF_CONT = 8u, NF_TERML = 16u, NF_UA_SHIFT = 5, NF_BAD = 0x80u;

// ---- two-byte UTF-8 characters U+0080..U+07FF (lead bytes C2..DF: Latin-1 Supplement to NKo - accented Latin, IPA, the combining
// marks, Greek, Cyrillic, Armenian, Hebrew, Arabic ...) ---------------------------------------------------------------------------------
// What NFD / lowercase / capcode do to them is a table of 30 x 64 entries (one per lead byte x second byte), built on the HOST from the
// host normalizer's own building blocks (ICU; tm_normalize.cpp: build_two_table) for the vocabulary's normalization flags, so that
// the device cannot disagree with it.  A character either stays one two-byte character (Æ ß « я ω ...: lead lane emits its first byte,
// continuation lane its second), or decomposes into an ASCII letter and one two-byte combining mark (é -> e + U+0301: lead lane takes
// the letter's role - class, markers, the letter - and the continuation lane, class M, emits the two bytes of the mark), or into a
// two-byte letter and one two-byte mark (й -> и + U+0306, ά -> α + U+0301: lead lane the first byte of the letter, continuation lane,
// class M, its second byte and the two bytes of the mark).  Anything else (three code points, a lower-case form of another length)
// stays without NT_OK: its document takes the host path.
//   a: class of the first code point [0..2] | NT_OK [3] | NT_DECOMP [4] | NT_DECOMP2 [5] | out0 [8..15] | out1 [16..23] | out2 [24..31]
//   b: low0 [0..7] | low1 [8..15] | out3 [16..23]
//      stays one character: out0 out1 (capcode lower-cases a capital to low0 low1);  NT_DECOMP: letter out0 (low0), mark out1 out2;
//      NT_DECOMP2: letter out0 out1 (low0 low1), mark out2 out3
// The entries of the leads C2..C5 (U+0080..U+017F: what European text is made of) are staged in LDS, the others are read where they lie.
struct NmTwo { uint32_t a, b; };
constexpr int NM_TWO_LEADS = 30, NM_TWO_SIZE = NM_TWO_LEADS * 64, NM_TWO_FAST = 256;
constexpr uint32_t NT_OK = 8u, NT_DECOMP = 16u, NT_DECOMP2 = 32u;
TM_HD bool nm_two_lead(uint32_t b) { return b - 0xC2u < (uint32_t)NM_TWO_LEADS; }
TM_HD bool nm_three_lead(uint32_t b) { return (b & 0xF0u) == 0xE0u; }
TM_HD bool nm_cont_byte(uint32_t b) { return (b & 0xC0u) == 0x80u; }
TM_HD uint32_t nm_two_index(uint32_t lead, uint32_t second) { return ((lead - 0xC2u) << 6) | (second & 63u); }
TM_HD bool nm_punct3(uint32_t b1, uint32_t b2) {   // E2 b1 b2 in the supported (NFD-stable) General Punctuation ranges U+2010..U+2027, U+2030..U+205E
  return (b1 == 0x80u && ((b2 >= 0x90u && b2 <= 0xA7u) || (b2 >= 0xB0u && b2 <= 0xBFu))) || (b1 == 0x81u && b2 >= 0x80u && b2 <= 0x9Eu);
}
// ---- three-byte characters U+0800..U+FFFF that the normalizer leaves alone ------------------------------------------------------------
// CJK ideographs, most kana, symbols, box drawing ...: NFD-stable, caseless, and to capcode either a letter that is neither upper nor
// lower case (class LO: \p{L}) or "other" (class O).  Two bits per code point, from the host normalizer's own functions
// (tm_normalize.cpp: build_three_tables): 0 = the document takes the host path (the character decomposes, has case, is a mark, a digit
// or a surrogate: voiced kana, Latin Extended Additional ...; Hangul syllables: below), 1 = class O, 2 = class LO, 3 (per code point only) = a
// combining mark of canonical class 0 (variation selectors, the enclosing keycap, spacing vowel signs: class M, never reordered).  First a table
// of the 1 024 blocks of 64 code points (256 bytes, staged in LDS): 0 / 1 / 2 when the whole block agrees, 3 = look the code point up (16 KB, in HBM).
constexpr int NM_BLK_WORDS = 64, NM_CP_WORDS = 4096;
// ---- four-byte characters U+10000..U+10FFFF (lead bytes F0..F4): emoji, pictographs, historic scripts, mathematical alphanumerics ... --
// The same two-bit codes per BLOCK of 64 code points (16 384 blocks of the planes 1..16, 4 KB, read where they lie: the characters are rare
// and the blocks of one text few), again from the host normalizer's own functions (tm_normalize.cpp: build_four_table): 1 = class O, 2 =
// class LO when every code point of the block is NFD-stable, caseless, neither digit nor mark - the emoji and symbol blocks, the
// ideographs of plane 2, Linear B, cuneiform, hieroglyphs ...; 0 = the document takes the host path (Deseret, Adlam and the other cased
// scripts of plane 1, the musical symbols that decompose, blocks with digits or combining marks, blocks that mix classes).
constexpr int NM_BLK4_WORDS = 1024;
TM_HD bool nm_four_lead(uint32_t b) { return b - 0xF0u < 5u; }
// ---- Hangul syllables U+AC00..U+D7A3 under NFD: the conjoining jamo L V (T) by arithmetic (Unicode 3.12) ----------------------------------
// 11 172 characters that DO decompose, but by rule instead of by table: s = cp - 0xAC00, L = U+1100 + s / 588, V = U+1161 + s % 588 / 28,
// T = U+11A7 + s % 28 (none when s % 28 == 0).  Three bytes in, six or nine out, which fits the one-lane-per-input-byte scheme: the lane
// of the syllable's first byte emits L, the second V, the third T or NOTHING (the only lanes of the pass that emit no byte at all).  To
// capcode syllable and jamo alike are letters without case (class LO).  Only with the NFD flag and capcode 2 (NM_MISC_HANGUL in the
// tables' last word: without capcode the pass keeps lengths, and without NFD the syllables are three-byte characters like any other).
constexpr uint32_t NM_MISC_HANGUL = 1u;
TM_HD bool nm_hangul_lead(uint32_t b) { return b - 0xEAu < 4u; }                      // EA B0 80 (U+AC00) .. ED 9E A3 (U+D7A3)
TM_HD bool nm_hangul(uint32_t cp) { return cp - 0xAC00u < 11172u; }
TM_HD uint32_t nm_cp3(uint32_t lead, uint32_t b1, uint32_t b2) { return ((lead & 15u) << 12) | ((b1 & 63u) << 6) | (b2 & 63u); }
// the bytes the lane of byte number `role` (0, 1, 2) of the syllable emits: returns 3 (*o1 *o2 *o3) or 0
TM_HD uint32_t nm_hangul_out(uint32_t cp, uint32_t role, uint32_t* o1, uint32_t* o2, uint32_t* o3) {
  const uint32_t s = cp - 0xAC00u, t = s % 28u;
  if (role == 2u && t == 0u) return 0u;
  const uint32_t j = role == 0u ? 0x1100u + s / 588u : (role == 1u ? 0x1161u + (s % 588u) / 28u : 0x11A7u + t);
  *o1 = 0xE1u; *o2 = 0x80u | ((j >> 6) & 63u); *o3 = 0x80u | (j & 63u);
  return 3u;
}
// which byte of a Hangul syllable the byte b between m2 m1 and p1 p2 is (0, 1, 2), and the syllable: false if it is none
TM_HD bool nm_hangul_role(uint32_t b, uint32_t m1, uint32_t m2, uint32_t p1, uint32_t p2, uint32_t* role, uint32_t* cp) {
  uint32_t lead, b1, b2;
  if (nm_hangul_lead(b)) { lead = b; b1 = p1; b2 = p2; *role = 0u; }
  else if (!nm_cont_byte(b)) return false;
  else if (nm_hangul_lead(m1)) { lead = m1; b1 = b; b2 = p1; *role = 1u; }
  else if (nm_cont_byte(m1) && nm_hangul_lead(m2)) { lead = m2; b1 = m1; b2 = b; *role = 2u; }
  else return false;
  if (!nm_cont_byte(b1) || !nm_cont_byte(b2)) return false;
  *cp = nm_cp3(lead, b1, b2);
  return nm_hangul(*cp);
}
// ---- Latin Extended Additional U+1E00..U+1EFF under NFD (round 6): what Vietnamese is written in beside the two-byte letters --------------------
// 256 three-byte characters (E1 B8..BB xx) that decompose into an ASCII letter and ONE or TWO two-byte combining marks (ế -> e + U+0302 +
// U+0301; ớ -> o + U+031B + U+0301), already in canonical order.  One lane per input byte as everywhere: the lane of the first byte takes the
// letter's role (class, markers, the letter), the second emits the first mark, the third the second mark - or nothing.  A table from the host
// normalizer's own functions (tm_normalize.cpp: build_lea_table), only with the NFD flag and capcode 2 (NM_MISC_LEA):
//   a: class of the letter [0..2] | NT_OK [3] | the letter [8..15] | its lower-case form [16..23] | number of marks [24..25]
//   b: first mark [0..15] | second mark [16..31]
// An entry without NT_OK (a two-byte base letter: ẛ; no decomposition: ẞ ỿ) leaves the document to the host, as before.
struct NmLea { uint32_t a, b; };
constexpr int NM_LEA_SIZE = 256;
constexpr uint32_t NM_MISC_LEA = 2u;
// which byte of such a character the byte b between m2 m1 and p1 p2 is (0, 1, 2) and the character's index in the table: false if it is none
TM_HD bool nm_lea_role(uint32_t b, uint32_t m1, uint32_t m2, uint32_t p1, uint32_t p2, uint32_t* role, uint32_t* idx) {
  uint32_t b1, b2;
  if (b == 0xE1u) { b1 = p1; b2 = p2; *role = 0u; }
  else if ((b & 0xC0u) != 0x80u) return false;
  else if (m1 == 0xE1u) { b1 = b; b2 = p1; *role = 1u; }
  else if (m2 == 0xE1u) { b1 = m1; b2 = b; *role = 2u; }
  else return false;
  if (b1 - 0xB8u >= 4u || (b2 & 0xC0u) != 0x80u) return false;
  *idx = ((b1 & 3u) << 6) | (b2 & 63u);
  return true;
}
// ---- the voiced kana of U+3040..U+30FF under NFD (round 6): が -> か + U+3099, ぱ -> は + U+309A, ヴ ヷ ヸ ヹ ヺ ゞ ヾ - what sent every Japanese document to the host ----
// Three bytes in, six out: the lane of the first byte emits the base kana (class LO, a letter without case), the second the combining
// mark (class M), the third nothing - the scheme of a Latin Extended Additional letter with one mark.  192 entries from the host
// normalizer's own functions (tm_normalize.cpp: build_kana_table), only with the NFD flag and capcode 2 and without `accents` (NM_MISC_KANA):
//   NK_OK | NK_SEMI (the mark is U+309A, else U+3099) | low byte of the base kana's code point (it lies in U+30xx like the character)
// The marks THEMSELVES in a text (canonical class 8: they may have to change places with a neighbour) stay the host's, and so does a two-byte
// mark right behind a voiced kana.
constexpr int NM_KANA_SIZE = 192;
constexpr uint32_t NM_MISC_KANA = 4u, NK_OK = 0x200u, NK_SEMI = 0x100u;
// which byte of a character of U+3040..U+30FF the byte b between m2 m1 and p1 p2 is (0, 1, 2) and the character's index in the table: false if it is none
TM_HD bool nm_kana_role(uint32_t b, uint32_t m1, uint32_t m2, uint32_t p1, uint32_t p2, uint32_t* role, uint32_t* idx) {
  uint32_t b1, b2;
  if (b == 0xE3u) { b1 = p1; b2 = p2; *role = 0u; }
  else if ((b & 0xC0u) != 0x80u) return false;
  else if (m1 == 0xE3u) { b1 = b; b2 = p1; *role = 1u; }
  else if (m2 == 0xE3u) { b1 = m1; b2 = b; *role = 2u; }
  else return false;
  if (b1 - 0x81u >= 3u || (b2 & 0xC0u) != 0x80u) return false;
  *idx = ((b1 - 0x81u) << 6) | (b2 & 63u);
  return true;
}
// the bytes the lane of byte number `role` of a voiced kana emits: returns 3 (*o1 *o2 *o3) or 0
TM_HD uint32_t nm_kana_out(uint32_t e, uint32_t role, uint32_t* o1, uint32_t* o2, uint32_t* o3) {
  if (role == 2u) return 0u;
  const uint32_t cp = role == 0u ? 0x3000u | (e & 0xFFu) : ((e & NK_SEMI) ? 0x309Au : 0x3099u);
  *o1 = 0xE3u; *o2 = 0x80u | ((cp >> 6) & 63u); *o3 = 0x80u | (cp & 63u);
  return 3u;
}
// what a kernel knows the tables by: the fast part of the two-byte table and the block table (LDS), the full tables (global memory)
// (The later tables - kana, classes, split characters - are reached from `lea` instead of through pointers of their own: while the out-of-line
// functions of tm_norm.hip took this struct by value, a 64-byte form of it made the kernels fault on the device.  They now get two pointers and
// put it together themselves - tm_norm.hip: tabs_from -, and the struct is kept small all the same.)
struct NmTabs { const NmTwo* two_fast; const NmTwo* two_all; const uint32_t* blk; const uint32_t* cp; const uint32_t* blk4; uint32_t misc; const NmLea* lea; };
static_assert(sizeof(NmTabs) <= 56, "the later tables are reached from lea");
TM_HD const uint16_t* nm_kana_tab(const NmTabs& t) { return reinterpret_cast<const uint16_t*>(t.lea + NM_LEA_SIZE); }      // the kana entries lie behind those of Latin Extended Additional
// ---- three-byte combining marks of canonical class > 0 in U+0800..U+1FFF under NFD (round 6): the virama and nukta of the Indic scripts, the
// tone marks and the vowels below of Thai and Lao, Tibetan, Myanmar, Khmer ... - what sent every Hindi or Thai document to the host ----------
// NFD leaves such a mark alone unless it has to change places with a neighbouring mark (canonical ordering), and that is decided where the
// LATER of two marks stands: a mark of class c behind a character that ends in a mark of class p > c (or in a mark whose class the device
// does not know: a two-byte one, the last of a decomposition) sends the document to the host; p <= c is in order already.  One byte per code
// point from the host normalizer's own functions (tm_normalize.cpp: build_ccc_table): the class, 0 = not such a mark.  Only with NM_MISC_CCC
// (the NFD flag without `accents`; without NFD every mark is inert and build_three_tables says so).  The table lies behind the kana entries.
// The same table names the decimal digits of the range (NM_CCC_DIGIT: no canonical class is 255) - Devanagari, Bengali, Thai ... digits are
// class N to capcode like the ASCII and the two-byte ones, whatever the flags.
// ... and its lower-case letters of three bytes (NM_CCC_LOWER: Georgian, the phonetic extensions ...), class L: nothing ever changes them.
constexpr uint32_t NM_CCC_BASE = 0x800u, NM_CCC_SIZE = 0x1800u, NM_MISC_CCC = 8u, NM_CCC_DIGIT = 255u, NM_CCC_LOWER = 254u;
TM_HD const uint8_t* nm_ccc_tab(const NmTabs& t) { return reinterpret_cast<const uint8_t*>(nm_kana_tab(t) + NM_KANA_SIZE); }
// ---- three-byte characters of U+0900..U+1BFF that NFD splits in TWO three-byte characters (round 6): the two-part vowel signs of Bengali, Tamil,
// Malayalam, Oriya ... (ো -> ে + া), the nukta letters written as one code point (क़ -> क + ़), Myanmar ဦ, Balinese ... -----------------------------
// Three bytes in, six out, the scheme of the voiced kana: the lane of the first byte emits the first character (a letter without case, or a mark
// of class 0), the second the second (a mark), the third nothing.  One word per code point from the host normalizer's own functions
// (tm_normalize.cpp: build_dec3_table), only with the NFD flag and capcode 2 and without `accents` (NM_MISC_DEC3):
//   ND_OK | ND_LETTER (the first character is a letter, else a mark) | first - 0x800 [0..12] | second - 0x800 [13..25] | [26..29]: 0, or which of
//   the NM_DEC3_THIRDS words behind the table holds a THIRD character (Kannada ೋ, Sinhala ෝ: three parts - the third lane emits it)
// A mark of class > 0 behind such a character is compared with the class of its last part like with any other mark (nm_classify_high).
constexpr uint32_t NM_DEC3_BASE = 0x900u, NM_DEC3_SIZE = 0x1300u, NM_DEC3_THIRDS = 16u, NM_MISC_DEC3 = 16u, ND_OK = 1u << 31, ND_LETTER = 1u << 30;
TM_HD const uint32_t* nm_dec3_tab(const NmTabs& t) { return reinterpret_cast<const uint32_t*>(nm_ccc_tab(t) + NM_CCC_SIZE); }
// the last character such an entry ends in (what a mark behind it is compared with)
TM_HD uint32_t nm_dec3_last(const NmTabs& t, uint32_t e) { const uint32_t k = (e >> 26) & 15u; return k ? nm_dec3_tab(t)[NM_DEC3_SIZE + k] : 0x800u + ((e >> 13) & 0x1FFFu); }
TM_HD uint32_t nm_dec3(const NmTabs& t, uint32_t cp) { return ((t.misc & NM_MISC_DEC3) && cp - NM_DEC3_BASE < NM_DEC3_SIZE) ? nm_dec3_tab(t)[cp - NM_DEC3_BASE] : 0u; }
// which byte of a three-byte character the byte b between m2 m1 and p1 p2 is (0, 1, 2) and the character: false if it is none
TM_HD bool nm_three_role(uint32_t b, uint32_t m1, uint32_t m2, uint32_t p1, uint32_t p2, uint32_t* role, uint32_t* cp) {
  uint32_t lead, b1, b2;
  if (nm_three_lead(b)) { lead = b; b1 = p1; b2 = p2; *role = 0u; }
  else if (!nm_cont_byte(b)) return false;
  else if (nm_three_lead(m1)) { lead = m1; b1 = b; b2 = p1; *role = 1u; }
  else if (nm_cont_byte(m1) && nm_three_lead(m2)) { lead = m2; b1 = m1; b2 = b; *role = 2u; }
  else return false;
  if (!nm_cont_byte(b1) || !nm_cont_byte(b2)) return false;
  *cp = nm_cp3(lead, b1, b2);
  return true;
}
// the bytes the lane of byte number `role` of such a character emits: returns 3 (*o1 *o2 *o3) or 0
TM_HD uint32_t nm_dec3_out(const NmTabs& t, uint32_t e, uint32_t role, uint32_t* o1, uint32_t* o2, uint32_t* o3) {
  if (role == 2u && ((e >> 26) & 15u) == 0u) return 0u;
  const uint32_t cp = role == 2u ? nm_dec3_last(t, e) : 0x800u + ((role == 0u ? e : e >> 13) & 0x1FFFu);
  *o1 = 0xE0u | (cp >> 12); *o2 = 0x80u | ((cp >> 6) & 63u); *o3 = 0x80u | (cp & 63u);
  return 3u;
}
TM_HD uint32_t nm_ccc3(const NmTabs& t, uint32_t cp) { return ((t.misc & NM_MISC_CCC) && cp - NM_CCC_BASE < NM_CCC_SIZE) ? (uint32_t)nm_ccc_tab(t)[cp - NM_CCC_BASE] : 0u; }
TM_HD NmTwo nm_two_get(const NmTabs& t, uint32_t idx) { return idx < (uint32_t)NM_TWO_FAST ? t.two_fast[idx] : t.two_all[idx]; }
TM_HD uint32_t nm_three_code(const NmTabs& t, uint32_t cp) {
  const uint32_t bc = (t.blk[cp >> 10] >> (2u * ((cp >> 6) & 15u))) & 3u;
  return bc != 3u ? bc : ((t.cp[cp >> 4] >> (2u * (cp & 15u))) & 3u);
}
TM_HD uint32_t nm_four_code(const NmTabs& t, uint32_t lead, uint32_t b1, uint32_t b2, uint32_t b3) {       // 0 also for anything that is not a well-formed character of the planes 1..16
  if (!nm_cont_byte(b1) || !nm_cont_byte(b2) || !nm_cont_byte(b3)) return 0u;
  const uint32_t cp = ((lead & 7u) << 18) | ((b1 & 63u) << 12) | ((b2 & 63u) << 6) | (b3 & 63u);
  if (cp - 0x10000u >= 0x100000u) return 0u;                                                             // overlong, or beyond U+10FFFF
  const uint32_t blk = (cp - 0x10000u) >> 6;
  const uint32_t code = (t.blk4[blk >> 4] >> (2u * (blk & 15u))) & 3u;
  return code == 3u ? 0u : code;
}
// class byte of the non-ASCII byte b between m2 m1 and p1 p2 (NF_BAD: the document needs the host normalizer)
TM_HD uint32_t nm_classify_high(uint32_t b, uint32_t m1, uint32_t m2, uint32_t m3, uint32_t p1, uint32_t p2, uint32_t p3, const NmTabs& tabs) {
  if (nm_two_lead(b)) {
    if (!nm_cont_byte(p1)) return NF_BAD;
    const uint32_t a = nm_two_get(tabs, nm_two_index(b, p1)).a;
    if (!(a & NT_OK)) return NF_BAD;
    if ((a & NF_CLASS) == NC_M && nm_cont_byte(m1) && nm_two_lead(m2)) {
      // two combining marks in a row (the second of them possibly out of a decomposition) may have to change places under NFD
      // (canonical ordering): not on the device
      const uint32_t pa = nm_two_get(tabs, nm_two_index(m2, m1)).a;
      if ((pa & NF_CLASS) == NC_M || (pa & (NT_DECOMP | NT_DECOMP2))) return NF_BAD;
    }
    // ... the same behind a character of Latin Extended Additional, which ends in a mark of its own
    if ((a & NF_CLASS) == NC_M && (tabs.misc & NM_MISC_LEA) && m3 == 0xE1u && m2 - 0xB8u < 4u && nm_cont_byte(m1) && (tabs.lea[((m2 & 3u) << 6) | (m1 & 63u)].a & NT_OK)) return NF_BAD;
    // ... behind a three-byte mark of class > 0 (whose place a two-byte mark of unknown class might have to take)
    if ((a & NF_CLASS) == NC_M && nm_three_lead(m3) && nm_cont_byte(m2) && nm_cont_byte(m1) && nm_ccc3(tabs, nm_cp3(m3, m2, m1)) - 1u < NM_CCC_LOWER - 1u) return NF_BAD;
    // ... behind a three-byte character that ends in a mark of class > 0 of its own (क़)
    if ((a & NF_CLASS) == NC_M && nm_three_lead(m3) && nm_cont_byte(m2) && nm_cont_byte(m1)) {
      const uint32_t pe = nm_dec3(tabs, nm_cp3(m3, m2, m1));
      if ((pe & ND_OK) && nm_ccc3(tabs, nm_dec3_last(tabs, pe)) - 1u < NM_CCC_LOWER - 1u) return NF_BAD;
    }
    // ... and behind a voiced kana
    if ((a & NF_CLASS) == NC_M && (tabs.misc & NM_MISC_KANA) && m3 == 0xE3u && m2 - 0x81u < 3u && nm_cont_byte(m1) && (nm_kana_tab(tabs)[((m2 - 0x81u) << 6) | (m1 & 63u)] & NK_OK)) return NF_BAD;
    return a & NF_CLASS;
  }
  if (nm_cont_byte(b) && nm_two_lead(m1)) {
    const uint32_t a = nm_two_get(tabs, nm_two_index(m1, b)).a;
    return (a & NT_OK) ? ((a & (NT_DECOMP | NT_DECOMP2)) ? (uint32_t)NC_M : ((a & NF_CLASS) | NF_CONT)) : NF_BAD;
  }
  uint32_t lead = 0, b1 = 0, b2 = 0, cont = 0;
  if (nm_three_lead(b)) { lead = b; b1 = p1; b2 = p2; }
  else if (nm_cont_byte(b) && nm_three_lead(m1)) { lead = m1; b1 = b; b2 = p1; cont = NF_CONT; }
  else if (nm_cont_byte(b) && nm_cont_byte(m1) && nm_three_lead(m2)) { lead = m2; b1 = m1; b2 = b; cont = NF_CONT; }
  else {
    // one of the four bytes of a character beyond the Basic Multilingual Plane: every byte takes the class of the character's block
    uint32_t code = 0u;
    if (nm_four_lead(b)) code = nm_four_code(tabs, b, p1, p2, p3);
    else if (nm_cont_byte(b)) {
      cont = NF_CONT;
      if (nm_four_lead(m1)) code = nm_four_code(tabs, m1, b, p1, p2);
      else if (nm_four_lead(m2)) code = nm_four_code(tabs, m2, m1, b, p1);
      else if (nm_four_lead(m3)) code = nm_four_code(tabs, m3, m2, m1, b);
    }
    return code == 0u ? (uint32_t)NF_BAD : ((code == 2u ? (uint32_t)NC_LO : (uint32_t)NC_O) | cont);
  }
  if (!nm_cont_byte(b1) || !nm_cont_byte(b2)) return NF_BAD;
  if (lead == 0xE2u && nm_punct3(b1, b2)) return ((b1 == 0x80u && b2 == 0x99u) ? (uint32_t)NC_AP : (uint32_t)NC_O) | cont;      // U+2019 is an apostrophe (tokenmonster.js:878)
  const uint32_t cp3 = nm_cp3(lead, b1, b2);
  if ((tabs.misc & NM_MISC_LEA) && cp3 - 0x1E00u < (uint32_t)NM_LEA_SIZE) {      // a letter and its marks: the first lane is the letter, the others are marks
    const uint32_t a = tabs.lea[cp3 - 0x1E00u].a;
    if (a & NT_OK) return cont ? (uint32_t)NC_M : (a & NF_CLASS);
  }
  if ((tabs.misc & NM_MISC_KANA) && cp3 - 0x3040u < (uint32_t)NM_KANA_SIZE && (nm_kana_tab(tabs)[cp3 - 0x3040u] & NK_OK)) return cont ? (uint32_t)NC_M : (uint32_t)NC_LO;      // a voiced kana: the kana, its mark, nothing
  { const uint32_t de = nm_dec3(tabs, cp3);       // split in two by NFD: the first character, its mark, nothing
    if (de & ND_OK) return cont ? (uint32_t)NC_M : ((de & ND_LETTER) ? (uint32_t)NC_LO : (uint32_t)NC_M); }
  const uint32_t code = nm_three_code(tabs, cp3);
  if (code == 0u) {
    const uint32_t c = nm_ccc3(tabs, cp3);
    if (c == NM_CCC_DIGIT) return (uint32_t)NC_N | cont;
    if (c == NM_CCC_LOWER) return (uint32_t)NC_L | cont;
    if (c != 0u) {                                  // a mark of canonical class c > 0: in place unless the character in front of it ends in a mark that belongs behind it
      if (!cont) {
        if (nm_cont_byte(m1) && nm_two_lead(m2)) {
          const uint32_t pa = nm_two_get(tabs, nm_two_index(m2, m1)).a;
          if ((pa & NF_CLASS) == NC_M || (pa & (NT_DECOMP | NT_DECOMP2))) return NF_BAD;
        } else if (nm_cont_byte(m1) && nm_cont_byte(m2) && nm_three_lead(m3)) {
          const uint32_t pcp = nm_cp3(m3, m2, m1);
          const uint32_t pe = nm_dec3(tabs, pcp);      // (a character that is split in two ends in the mark that is its second half)
          const uint32_t pc = nm_ccc3(tabs, (pe & ND_OK) ? nm_dec3_last(tabs, pe) : pcp);
          if (pc < NM_CCC_LOWER && pc > c) return NF_BAD;
          if ((tabs.misc & NM_MISC_LEA) && pcp - 0x1E00u < (uint32_t)NM_LEA_SIZE && (tabs.lea[pcp - 0x1E00u].a & NT_OK)) return NF_BAD;
          if ((tabs.misc & NM_MISC_KANA) && pcp - 0x3040u < (uint32_t)NM_KANA_SIZE && (nm_kana_tab(tabs)[pcp - 0x3040u] & NK_OK)) return NF_BAD;
        }
      }
      return (uint32_t)NC_M | cont;
    }
  }
  if (code == 0u && (tabs.misc & NM_MISC_HANGUL) && nm_hangul(cp3)) return (uint32_t)NC_LO | cont;       // decomposes, but by arithmetic: nm_hangul_out
  return code == 0u ? (uint32_t)NF_BAD : ((code == 3u ? (uint32_t)NC_M : (code == 2u ? (uint32_t)NC_LO : (uint32_t)NC_O)) | cont);      // (3: a combining mark of class 0)
}
// The same per CHARACTER, at its first byte b (>= 0xC0): class of the first byte | class of its other bytes << 8 | its length << 16 | (the pass
// changes its bytes: a two-byte character - they come from the table -, one NFD splits, a Hangul syllable) << 24 - what
// nm_classify_high says of each of its bytes, for one decoding of the character instead of one per byte (a character that is not taken:
// NF_BAD in both, length 1; the bytes behind it keep the NF_BAD the caller has given every byte beyond ASCII beforehand).
TM_HD uint32_t nm_classify_char(uint32_t b, uint32_t m1, uint32_t m2, uint32_t m3, uint32_t p1, uint32_t p2, uint32_t p3, const NmTabs& tabs) {
  const uint32_t cls = nm_classify_high(b, m1, m2, m3, p1, p2, p3, tabs);
  if (cls == NF_BAD || !(nm_two_lead(b) || nm_three_lead(b) || nm_four_lead(b))) return NF_BAD | (NF_BAD << 8) | (1u << 16);
  uint32_t cont = cls | NF_CONT, n = 4u, changes = 0u;
  if (nm_two_lead(b)) {
    n = 2u; changes = 1u;
    if (nm_two_get(tabs, nm_two_index(b, p1)).a & (NT_DECOMP | NT_DECOMP2)) cont = NC_M;            // the second half of a character that decomposes emits the mark
  } else if (nm_three_lead(b)) {
    n = 3u;
    const uint32_t cp3 = nm_cp3(b, p1, p2);                                                          // a letter and its marks (Latin Extended Additional, a voiced kana, a character NFD splits): the other lanes are marks
    if (((tabs.misc & NM_MISC_LEA) && cp3 - 0x1E00u < (uint32_t)NM_LEA_SIZE && (tabs.lea[cp3 - 0x1E00u].a & NT_OK)) ||
        ((tabs.misc & NM_MISC_KANA) && cp3 - 0x3040u < (uint32_t)NM_KANA_SIZE && (nm_kana_tab(tabs)[cp3 - 0x3040u] & NK_OK)) || (nm_dec3(tabs, cp3) & ND_OK)) { cont = NC_M; changes = 1u; }
    else if ((tabs.misc & NM_MISC_HANGUL) && nm_hangul(cp3)) changes = 1u;
  }
  return cls | (cont << 8) | (n << 16) | (changes << 24);
}
// the bytes of a lane that holds one byte of a two-byte character: *o3 = its last output byte; returns how many bytes the lane emits
// IN FRONT of it: 0, 1 (*y: the second half of a character that decomposes into an ASCII letter and a mark emits the mark) or 2 (*m3 *y:
// the second half of one that decomposes into a two-byte letter and a mark emits the letter's second byte and the mark)
// (`lowered`: the rule table says capcode lower-cases this character - lead lane; `capcode`: level 2 is on, which lower-cases every capital)
TM_HD uint32_t nm_two_out(NmTwo e, bool cont, bool lowered, bool capcode, uint32_t* o3, uint32_t* y, uint32_t* m3) {
  if (!cont) { *o3 = lowered ? (e.b & 0xFFu) : ((e.a >> 8) & 0xFFu); return 0u; }
  const bool upper = capcode && (e.a & NF_CLASS) == NC_U;
  if (e.a & NT_DECOMP) { *y = (e.a >> 16) & 0xFFu; *o3 = e.a >> 24; return 1u; }
  if (e.a & NT_DECOMP2) { *m3 = upper ? ((e.b >> 8) & 0xFFu) : ((e.a >> 16) & 0xFFu); *y = e.a >> 24; *o3 = (e.b >> 16) & 0xFFu; return 2u; }
  *o3 = upper ? ((e.b >> 8) & 0xFFu) : ((e.a >> 16) & 0xFFu);
  return 0u;
}

TM_HD uint64_t nm_brev(uint64_t x) { return __builtin_bitreverse64(x); }   // s_brev_b64 on the device
// every bit of M reachable upwards from a seed through consecutive set bits of M (seeds outside M are ignored):
// the carry of M + S ripples from a seed to the end of its run
TM_HD uint64_t nm_flood_up(uint64_t M, uint64_t S) { S &= M; return ((M ^ (M + S)) | S) & M; }
// the same downwards, for seeds that are known to lie inside M (the carry ripples in the bit-reversed masks)
TM_HD uint64_t nm_flood_down_inside(uint64_t M, uint64_t S) { const uint64_t m = nm_brev(M), s = nm_brev(S); return nm_brev(((m ^ (m + s)) | s) & m); }
// bit i = bit (i + 1) of the byte stream (next0 = bit 0 of the following chunk)
TM_HD uint64_t nm_shr1(uint64_t cur, uint64_t next0) { return (cur >> 1) | (next0 << 63); }
// valid-byte mask of chunk c of a piece of m bytes
TM_HD uint64_t nm_valid(int c, int m) { const int r = m - 64 * c; return r >= 64 ? ~0ull : (r <= 0 ? 0ull : ((1ull << r) - 1ull)); }

// T of a chunk's block bytes WITHOUT a sweep from the end of the piece: what a chunk needs to know about everything behind it is one bit,
//   T0(next chunk) = "the first byte at or after the next chunk's first byte that is not in a block is a lower-case letter"
// (that byte decides T of a block that reaches this chunk's end), and T0 of a chunk is a function of its own two ballots unless all 64 of
// its bytes are in one block - only then of the chunk behind it (`beyond`).  k_norm_emit2 computes TX of a chunk from the ballots of the
// chunk and of the one behind it as it goes forward; bytes behind the end of the text are class O (no block, no letter): T0 = 0 there.
TM_HD uint64_t nm_t0(uint64_t B, uint64_t L, uint64_t beyond) {
  const uint64_t nb = ~B;
  return nb == 0ull ? (beyond & 1ull) : ((L >> __builtin_ctzll(nb)) & 1ull);
}
TM_HD uint64_t nm_tx(uint64_t B, uint64_t L, uint64_t t0_next) { return nm_flood_down_inside(B, B & nm_shr1(L, t0_next & 1ull)); }

// Forward sweep over a chunk: W of every byte.  B / U: ballots "in a block" / "capital" (masked to the piece here); w_in: W of byte 0
// of the chunk; *w_out: W of byte 0 of the next chunk.
TM_HD uint64_t nm_inword(uint64_t B, uint64_t U, uint64_t V, uint64_t w_in, uint64_t* w_out) {
  const uint64_t G = nm_flood_up(B & V, (U & V) | (w_in & 1ull));   // block bytes with a capital at or before them in their run
  *w_out = G >> 63;
  return (G << 1) | (w_in & 1ull);
}

// A space right before a capital becomes that capital's marker (:976-979): 'C' if the capital's block ends in a lower-case
// letter, else 'W'.  SP / U: ballots of the chunk (U unmasked: the capital may be the first byte after the piece); next_u0:
// bit 0 of the capital ballot of chunk c+1; tx / tx_next0: TX of this chunk, bit 0 of TX of the next.
TM_HD void nm_space_markers(uint64_t SP, uint64_t U, uint64_t next_u0, uint64_t V, uint64_t tx, uint64_t tx_next0, uint64_t* spC, uint64_t* spW) {
  const uint64_t spM = SP & V & nm_shr1(U, next_u0 & 1ull);
  const uint64_t nextT = nm_shr1(tx, tx_next0 & 1ull);
  *spC = spM & nextT;
  *spW = spM & ~nextT;
}

// The carries of a piece from its MARGINS instead of from a pass over the whole document (k_norm_emit2<false>): the 64 bytes before
// the piece say whether its first byte is inside a word — unless all 64 are one block without a capital, which may still have one
// further back —; the 64 bytes after a full piece are the "next chunk" of its last chunk (nm_t0) — unless all 64 still belong to one block.
// Bb / Ub: ballots "in a block" / "capital" of the bytes before; Ba: "in a block" of the bytes after (zero when the text ends with the
// piece).  *w_in: W of byte 0.  Returns false when the margins cannot tell: the exact path (summaries + carries) has to run.
TM_HD bool nm_margin_carries(uint64_t Bb, uint64_t Ub, uint64_t Ba, uint64_t* w_in) {
  uint64_t w_out;
  (void)nm_inword(Bb, Ub, ~0ull, 0ull, &w_out);
  *w_in = w_out;
  return !((Bb == ~0ull && Ub == 0ull) || Ba == ~0ull);
}

// ---- the rule table ------------------------------------------------------------------------------------------------------
// index: class [0..2] | class of the previous byte [3..5] | class two characters back, looked at only behind an apostrophe
//        [6..8] | W [9] | T [10]
// entry: bytes emitted - 1 [0..1] | the byte itself is lower-cased [2] | the byte that precedes "' ' + byte" when three or four
//        are emitted [8..15]: 'D' (three), 'C' or 'W' after a leading 'D' (four)
constexpr int NM_LUT_SIZE = 2048;
TM_HD constexpr uint32_t nm_lut_index(uint32_t cls, uint32_t prev, uint32_t prev2, uint32_t w, uint32_t t) {
  return (cls & 7u) | ((prev & 7u) << 3) | ((prev2 & 7u) << 6) | ((w & 1u) << 9) | ((t & 1u) << 10);
}
TM_HD constexpr uint16_t nm_lut_entry(uint32_t idx, bool lower_all) {
  const uint32_t cls = idx & 7u, P = (idx >> 3) & 7u, P2 = (idx >> 6) & 7u;
  const bool W = (idx >> 9) & 1u, T = (idx >> 10) & 1u;
  // isLetter(rlast) / isLetter(rlast2) of tokenmonster.js:883-885; a combining mark behind a letter joins like the letter (:915, :954, :970)
  const bool Pletter = P == NC_L || P == NC_U || P == NC_LO, P2letter = P2 == NC_L || P2 == NC_U || P2 == NC_LO;
  uint32_t len = 1, lower = 0, mark = 0;
  if (cls == NC_U) {                                         // :913-916, :924-951, :975-990
    lower = 1;
    if (!W) {                                                // first capital of a run
      if (P == NC_SP) len = 2;                               // the space before it has become the marker (nm_space_markers)
      else { len = 4; mark = T ? 'C' : 'W'; }
    } else if (T) { len = 4; mark = 'C'; }                   // every later capital of a 'C' run
    else if (P == NC_N) { len = 3; mark = 'D'; }
  } else if (cls == NC_L) {                                  // :952-955 (the letter that ends a run), :970
    lower = lower_all ? 1 : 0;
    const bool joined = W ? (P == NC_U || P == NC_AP || P == NC_M)
                          : (P == NC_SP || Pletter || P == NC_M || (P == NC_AP && P2letter));
    if (!joined) { len = 3; mark = 'D'; }
  } else if (cls == NC_N) {                                  // :958 / :992
    const bool joined = W ? (P == NC_N) : (P == NC_SP || P == NC_N);
    if (!joined) { len = 3; mark = 'D'; }
  }
  return (uint16_t)((len - 1) | (lower << 2) | (mark << 8));
}
struct NmLut { uint16_t e[2][NM_LUT_SIZE]; };
constexpr NmLut nm_make_lut() {
  NmLut t{};
  for (int la = 0; la < 2; la++)
    for (int i = 0; i < NM_LUT_SIZE; i++) t.e[la][i] = nm_lut_entry((uint32_t)i, la != 0);
  return t;
}

}  // namespace tmh
