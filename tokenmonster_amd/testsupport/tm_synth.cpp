// tm_synth.cpp — deterministic synthetic lexicon, corpora and vocabularies of the BASELINE.json shapes.
// TEST / BENCHMARK SUPPORT: built into libtm_testsupport.so (include/tm_testsupport.h), not into the product library.
//
// No pretrained .vocab and no dataset exists in /root/reference or in this image, and
// there is no network, so the named configurations (english-24000, englishcode-32000,
// englishcode-100256, code-4096-nocapcode, 65536-token candidate set) are reproduced as *shapes*:
//   corpus   raw UTF-8 documents: English-like prose (Zipf over a fixed function-word list + generated
//            syllable words and a Zipf-weighted set of multi-word collocations, sentence case, ~4% Title/ALL-CAPS
//            words, punctuation, numbers), source code (keywords, camelCase / snake_case identifiers, operators,
//            indentation, recurring idioms) and log/JSON lines.
//   vocab    the most valuable substrings of a normalized sample of that corpus (count x (length-1)),
//            in the spirit of getalltokens+trainvocab (training/getalltokens.go, trainvocab.go) but
//            selected in one greedy pass; then tm_build_vocab computes flags/alternatives exactly as
//            the reference's builder does.  The reference trains with -max-token-length 40 and says the algorithm
//            "is optimized for" it (training/README.md:151-153): the collocations and idioms are what gives these
//            vocabularies the multi-word tail (keys up to 40 bytes) real ones have.  The number of IDs is exactly the
//            size asked for; record scores are the fraction of sample bytes a token's occurrences cover
//            (trainvocab.go:438-442).
#include "tm_build.h"
#include "tm_testsupport.h"
#include "tokenmonster_hip.h"
#include "tm_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace tmh {
namespace {

const char* kFunctionWords[] = {
    "the", "of", "and", "to", "a", "in", "is", "that", "for", "it", "as", "was", "with", "be", "by", "on", "not",
    "he", "this", "are", "or", "his", "from", "at", "which", "but", "have", "an", "had", "they", "you", "were",
    "their", "one", "all", "we", "can", "her", "has", "there", "been", "if", "more", "when", "will", "would", "who",
    "so", "no", "she", "other", "its", "may", "these", "what", "them", "than", "some", "him", "time", "into", "only",
    "could", "new", "then", "do", "first", "any", "my", "now", "such", "like", "our", "over", "man", "me", "even",
    "most", "made", "after", "also", "did", "many", "before", "must", "through", "years", "where", "much", "your",
    "way", "well", "down", "should", "because", "each", "just", "those", "people", "how", "too", "little", "state",
    "good", "very", "make", "world", "still", "own", "see", "men", "work", "long", "get", "here", "between", "both",
    "life", "being", "under", "never", "day", "same", "another", "know", "while", "last", "might", "us", "great",
    "old", "year", "off", "come", "since", "against", "go", "came", "right", "used", "take", "three"};
const char* kOnsets[] = {"b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "st",
                         "tr", "pr", "ch", "sh", "th", "br", "cl", "gr", "pl", "sp", "fl", "cr", "", "", "qu", "z"};
const char* kVowels[] = {"a", "e", "i", "o", "u", "a", "e", "i", "o", "ea", "ou", "ai", "io", "ee", "oo", "ie", "y"};
const char* kCodas[] = {"", "", "", "n", "r", "s", "t", "l", "m", "d", "ng", "st", "nt", "ck", "ll", "rs", "x", "p"};
const char* kSuffixes[] = {"", "", "", "", "s", "ed", "ing", "ly", "er", "tion", "ment", "ness", "able", "al", "ive"};
const char* kKeywords[] = {"if", "else", "for", "while", "return", "function", "var", "let", "const", "int", "void",
                           "def", "class", "import", "from", "public", "private", "static", "new", "null", "true",
                           "false", "this", "self", "string", "float", "struct", "switch", "case", "break", "try",
                           "catch", "throw", "async", "await", "print", "len", "range", "None", "elif", "in", "not",
                           "and", "or", "continue", "package", "func", "type", "interface", "map", "bool", "char"};
const char* kOps[] = {" = ", " == ", " != ", " + ", " - ", " * ", " / ", " < ", " > ", " <= ", " >= ", " && ", " || ",
                      " += ", " -= ", "++", "--", "->", "::", ".", ", ", " => ", " % ", " & ", " | ", " << "};
const char* kJsonKeys[] = {"id", "name", "type", "value", "status", "ts", "level", "msg", "user", "host", "path",
                           "code", "count", "items", "error", "data", "time", "size", "tags", "url"};
const char* kLogLevels[] = {"INFO", "WARN", "ERROR", "DEBUG", "TRACE"};
const char* kIdioms[] = {"for (int i = 0; i < n; i++) {\n", "if (err != nil) {\n", "return nil, err\n", "import numpy as np\n", "def __init__(self, ",
                         "public static void main(String[] args) {\n", "#include <stdio.h>\n", "console.log(", "from typing import List, Optional\n",
                         "if __name__ == \"__main__\":\n", "} else if (", "for i in range(len(", "std::vector<std::string> ", "return false;\n",
                         "return true;\n", "self.assertEqual(", "throw new IllegalArgumentException(", "fmt.Println(", "const result = await ",
                         "} catch (Exception e) {\n", "static const unsigned int ", "print(f\"", "using namespace std;\n", "package main\n\nimport (\n"};

template <size_t N> constexpr size_t countof(const char* (&)[N]) { return N; }

struct Lexicon {
  std::vector<std::string> words;     // rank order
  std::vector<double> cdf;            // Zipf cumulative
  std::vector<std::string> phrases;   // collocations of 2..6 words, rank order ("of the", "as well as the", ...)
  std::vector<double> pcdf;
  explicit Lexicon(uint64_t seed, size_t n_words = 50000) {
    Rng rng(seed ^ 0x4C455849434F4Eull);
    std::unordered_map<std::string, int> seen;
    for (size_t k = 0; k < countof(kFunctionWords); k++) { words.push_back(kFunctionWords[k]); seen[kFunctionWords[k]] = 1; }
    while (words.size() < n_words) {
      size_t rank = words.size();
      int syll = 1 + (rank > 300) + (rank > 3000 ? (int)rng.below(2) : 0) + (rank > 15000 ? (int)rng.below(2) : 0) + (int)rng.below(2);
      std::string w;
      for (int s = 0; s < syll; s++) {
        w += kOnsets[rng.below((uint32_t)countof(kOnsets))];
        w += kVowels[rng.below((uint32_t)countof(kVowels))];
        if (s + 1 == syll || rng.chance(0.3)) w += kCodas[rng.below((uint32_t)countof(kCodas))];
      }
      if (rank > 500) w += kSuffixes[rng.below((uint32_t)countof(kSuffixes))];
      if (w.size() < 2 || w.size() > 18 || seen.count(w)) continue;
      seen[w] = 1;
      words.push_back(w);
    }
    cdf.resize(words.size());
    double acc = 0;
    for (size_t k = 0; k < words.size(); k++) { acc += 1.0 / std::pow((double)k + 2.7, 1.05); cdf[k] = acc; }
    for (auto& c : cdf) c /= acc;
    // collocations: mostly frequent words, the frequent ones short
    const size_t n_phrases = 6000;
    std::unordered_map<std::string, int> pseen;
    while (phrases.size() < n_phrases) {
      const size_t rank = phrases.size();
      const int nw = 2 + (int)rng.below(rank < 200 ? 2 : (rank < 2000 ? 4 : 5));
      std::string ph;
      for (int k = 0; k < nw; k++) { if (k) ph.push_back(' '); ph += sample(rng); }
      if (ph.size() > 60 || pseen.count(ph)) continue;
      pseen[ph] = 1;
      phrases.push_back(ph);
    }
    pcdf.resize(phrases.size());
    acc = 0;
    for (size_t k = 0; k < phrases.size(); k++) { acc += 1.0 / std::pow((double)k + 4.0, 0.95); pcdf[k] = acc; }
    for (auto& c : pcdf) c /= acc;
  }
  const std::string& sample_phrase(Rng& rng) const {
    double u = rng.unit();
    size_t k = (size_t)(std::lower_bound(pcdf.begin(), pcdf.end(), u) - pcdf.begin());
    if (k >= phrases.size()) k = phrases.size() - 1;
    return phrases[k];
  }
  const std::string& sample(Rng& rng) const {
    double u = rng.unit();
    size_t k = (size_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
    if (k >= words.size()) k = words.size() - 1;
    return words[k];
  }
};

void cap_first(std::string& w) { if (!w.empty() && w[0] >= 'a' && w[0] <= 'z') w[0] = (char)(w[0] - 32); }
void cap_all(std::string& w) { for (auto& c : w) if (c >= 'a' && c <= 'z') c = (char)(c - 32); }

void gen_number(Rng& rng, std::string& out) {
  int nd = 1 + (int)rng.below(5);
  for (int k = 0; k < nd; k++) out.push_back((char)('0' + rng.below(10)));
  if (rng.chance(0.15)) { out.push_back('.'); out.push_back((char)('0' + rng.below(10))); out.push_back((char)('0' + rng.below(10))); }
}

void gen_prose(const Lexicon& lex, Rng& rng, size_t target, std::string& out) {
  size_t start = out.size();
  while (out.size() - start < target) {
    int n_sent = 2 + (int)rng.below(6);
    for (int s = 0; s < n_sent; s++) {
      int n_words = 4 + (int)rng.below(18);
      bool quoted = rng.chance(0.06);
      if (quoted) out += rng.chance(0.5) ? "\"" : "\xE2\x80\x9C";
      for (int w = 0; w < n_words; w++) {
        std::string word = rng.chance(0.10) ? lex.sample_phrase(rng) : lex.sample(rng);
        if (w == 0) cap_first(word);
        else if (rng.chance(0.03)) cap_first(word);
        else if (rng.chance(0.008)) cap_all(word);
        if (w > 0) out.push_back(' ');
        if (rng.chance(0.02)) { gen_number(rng, out); continue; }
        out += word;
        if (rng.chance(0.012)) out += rng.chance(0.7) ? "'s" : "\xE2\x80\x99s";
        if (rng.chance(0.004)) { out += "-"; out += lex.sample(rng); }
        if (w + 1 < n_words) {
          if (rng.chance(0.07)) out.push_back(',');
          else if (rng.chance(0.008)) out.push_back(';');
          else if (rng.chance(0.006)) out += " \xE2\x80\x94";
          else if (rng.chance(0.005)) { out += " ("; out += lex.sample(rng); out += ")"; }
        }
      }
      if (quoted) out += rng.chance(0.5) ? "\"" : "\xE2\x80\x9D";
      uint32_t e = rng.below(100);
      out += e < 86 ? "." : (e < 93 ? "?" : "!");
      if (s + 1 < n_sent) out.push_back(' ');
    }
    out += rng.chance(0.8) ? "\n\n" : "\n";
    if (rng.chance(0.002)) out += "caf\xC3\xA9 na\xC3\xAFve r\xC3\xA9sum\xC3\xA9 \xC3\x9C" "ber\n";   // a little non-ASCII
  }
}

std::string gen_ident(const Lexicon& lex, Rng& rng) {
  int parts = 1 + (int)rng.below(3);
  uint32_t style = rng.below(10);
  std::string id;
  for (int p = 0; p < parts; p++) {
    std::string w = lex.words[rng.below(3000)];
    if (style < 5) { if (p > 0) cap_first(w); }               // camelCase
    else if (style < 8) { if (p > 0) id.push_back('_'); }     // snake_case
    else if (style < 9) cap_first(w);                          // PascalCase
    else { cap_all(w); if (p > 0) id.push_back('_'); }         // CONSTANT_CASE
    id += w;
  }
  if (rng.chance(0.08)) id.push_back((char)('0' + rng.below(10)));
  return id;
}

void gen_expr(const Lexicon& lex, Rng& rng, std::string& out, int depth) {
  uint32_t k = rng.below(10);
  if (k < 4) out += gen_ident(lex, rng);
  else if (k < 6) gen_number(rng, out);
  else if (k < 7) { out += rng.chance(0.5) ? "\"" : "'"; char q = out.back(); out += lex.sample(rng); if (rng.chance(0.4)) { out += " "; out += lex.sample(rng); } out.push_back(q); }
  else if (k < 9 && depth < 2) { out += gen_ident(lex, rng); out += "("; int na = (int)rng.below(3); for (int a = 0; a < na; a++) { if (a) out += ", "; gen_expr(lex, rng, out, depth + 1); } out += ")"; }
  else { out += gen_ident(lex, rng); out += "["; gen_expr(lex, rng, out, depth + 1); out += "]"; }
  if (depth < 2 && rng.chance(0.3)) { out += kOps[rng.below((uint32_t)countof(kOps))]; gen_expr(lex, rng, out, depth + 1); }
}

void gen_code(const Lexicon& lex, Rng& rng, size_t target, std::string& out) {
  size_t start = out.size();
  bool tabs = rng.chance(0.4);
  int indent = 0;
  while (out.size() - start < target) {
    auto ind = [&]() { for (int k = 0; k < indent; k++) out += tabs ? "\t" : "    "; };
    uint32_t k = rng.below(20);
    ind();
    if (rng.chance(0.06)) { const char* idiom = kIdioms[rng.below((uint32_t)countof(kIdioms))]; out += idiom; if (out.back() != '\n') { gen_expr(lex, rng, out, 1); out += ")\n"; } continue; }
    if (k < 2) { out += rng.chance(0.5) ? "// " : "# "; int nw = 2 + (int)rng.below(8); for (int w = 0; w < nw; w++) { if (w) out.push_back(' '); out += lex.sample(rng); } out.push_back('\n'); }
    else if (k < 5 && indent < 5) { out += kKeywords[rng.below(5)]; out += " ("; gen_expr(lex, rng, out, 0); out += ") {\n"; indent++; }
    else if (k < 7 && indent > 0) { indent--; out.resize(out.size() - (tabs ? 1 : 4)); out += "}\n"; }
    else if (k < 9) { out += kKeywords[5 + rng.below(8)]; out.push_back(' '); out += gen_ident(lex, rng); out += "("; int na = (int)rng.below(4); for (int a = 0; a < na; a++) { if (a) out += ", "; out += gen_ident(lex, rng); } out += ") {\n"; if (indent < 5) indent++; }
    else if (k < 10) { out += "return "; gen_expr(lex, rng, out, 0); out += ";\n"; }
    else if (k < 11) { out.push_back('\n'); }
    else { if (rng.chance(0.4)) { out += kKeywords[rng.below((uint32_t)countof(kKeywords))]; out.push_back(' '); } out += gen_ident(lex, rng); out += kOps[rng.below(13)]; gen_expr(lex, rng, out, 0); out += rng.chance(0.7) ? ";\n" : "\n"; }
  }
}

void gen_log(const Lexicon& lex, Rng& rng, size_t target, std::string& out) {
  size_t start = out.size();
  bool json = rng.chance(0.5);
  char tmp[64];
  while (out.size() - start < target) {
    if (json) {
      out += "{";
      int nk = 2 + (int)rng.below(6);
      for (int k = 0; k < nk; k++) {
        if (k) out += ", ";
        out += "\""; out += kJsonKeys[rng.below((uint32_t)countof(kJsonKeys))]; out += "\": ";
        uint32_t t = rng.below(4);
        if (t == 0) gen_number(rng, out);
        else if (t == 1) out += rng.chance(0.5) ? "true" : "null";
        else { out += "\""; out += lex.sample(rng); if (rng.chance(0.3)) { out += " "; out += lex.sample(rng); } out += "\""; }
      }
      out += "}\n";
    } else {
      snprintf(tmp, sizeof tmp, "2024-%02u-%02uT%02u:%02u:%02uZ ", 1 + rng.below(12), 1 + rng.below(28), rng.below(24), rng.below(60), rng.below(60));
      out += tmp;
      out += kLogLevels[rng.below((uint32_t)countof(kLogLevels))];
      out += " ["; out += gen_ident(lex, rng); out += "] ";
      int nw = 3 + (int)rng.below(9);
      for (int w = 0; w < nw; w++) { if (w) out.push_back(' '); if (rng.chance(0.1)) gen_number(rng, out); else out += lex.sample(rng); }
      if (rng.chance(0.3)) { snprintf(tmp, sizeof tmp, " id=%u", rng.below(1000000)); out += tmp; }
      out.push_back('\n');
    }
  }
}

void gen_doc(const Lexicon& lex, Rng& rng, uint32_t kind, size_t target, std::string& out) {
  uint32_t u = rng.below(100);
  if (kind == TM_KIND_ENGLISH || (kind == TM_KIND_ENGLISHCODE && u < 60)) gen_prose(lex, rng, target, out);
  else if (kind == TM_KIND_CODE || u < 90) gen_code(lex, rng, target, out);
  else gen_log(lex, rng, target, out);
}

// FNV-1a over a byte range, for the substring counter
inline uint64_t hash_bytes(const uint8_t* p, size_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h ^ (h >> 29);
}

}  // namespace

// Generates the raw corpus into `text`, filling offsets.  Returns number of documents.
uint32_t synth_corpus(uint32_t kind, uint64_t seed, uint64_t nbytes, uint32_t median_doc, std::string& text,
                      std::vector<uint64_t>& offsets, uint32_t max_docs) {
  Lexicon lex(0x544D4C58ull);   // the lexicon is shared by every corpus and vocabulary
  Rng rng(seed);
  text.clear();
  text.reserve((size_t)nbytes + 70000);
  offsets.clear();
  offsets.push_back(0);
  const double mu = std::log((double)median_doc), sigma = 1.0;
  while (text.size() < nbytes && offsets.size() <= max_docs) {
    // log-normal document length via Box-Muller
    double u1 = std::max(rng.unit(), 1e-12), u2 = rng.unit();
    double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    double len = std::exp(mu + sigma * z);
    size_t target = (size_t)std::min(65536.0, std::max(64.0, len));
    size_t start = text.size();
    gen_doc(lex, rng, kind, target, text);
    if (text.size() - start > 65536) text.resize(start + 65536);
    // never end a document in the middle of a UTF-8 sequence
    while (text.size() > start && ((uint8_t)text.back() & 0xC0) == 0x80) text.pop_back();
    if (text.size() > start && ((uint8_t)text.back() & 0xC0) == 0xC0) text.pop_back();
    offsets.push_back(text.size());
  }
  return (uint32_t)(offsets.size() - 1);
}

int synth_vocab_image(uint32_t kind, uint32_t vocab_size, uint32_t capcode, uint32_t norm_flag, uint32_t level,
                      uint64_t seed, bool with_unk, std::vector<uint8_t>& image) {
  if (capcode != 0 && capcode != 2) return set_error(TM_E_INVALID, "synthetic vocabularies support capcode 0 or 2");
  // 1. normalized sample
  std::string raw;
  std::vector<uint64_t> offs;
  const uint64_t sample_bytes = vocab_size > 50000 ? (12u << 20) : (6u << 20);
  synth_corpus(kind, seed ^ 0x53414D504C45ull, sample_bytes, 2048, raw, offs, 1u << 20);
  std::vector<uint8_t> norm, tmp;
  norm.reserve(raw.size() + raw.size() / 4);
  std::vector<size_t> doc_end;
  for (size_t d = 0; d + 1 < offs.size(); d++) {
    normalize_bytes((const uint8_t*)raw.data() + offs[d], (size_t)(offs[d + 1] - offs[d]), capcode, norm_flag, tmp);
    norm.insert(norm.end(), tmp.begin(), tmp.end());
    doc_end.push_back(norm.size());
  }
  // 2. count candidate substrings: every substring of length 2..10 inside a document, plus
  //    "atom"-aligned n-grams (atom = maximal run of [letters/digits] with its leading space or marker)
  //    up to 40 bytes.  Open-addressing counter keyed by 64-bit hash (collisions are harmless here:
  //    they only perturb which substrings are chosen, and the builder re-verifies everything).
  struct Slot { uint64_t h; uint32_t count; uint32_t pos; uint8_t len; };
  size_t cap = 1;
  while (cap < norm.size() * 6) cap <<= 1;
  std::vector<Slot> table(cap, Slot{0, 0, 0, 0});
  auto bump = [&](size_t pos, size_t len) {
    uint64_t h = hash_bytes(&norm[pos], len) | 1;
    size_t s = (size_t)(h * 0x9E3779B97F4A7C15ull >> 20) & (cap - 1);
    for (;;) {
      Slot& sl = table[s];
      if (sl.h == h && sl.len == len) { sl.count++; return; }
      if (sl.h == 0) { sl.h = h; sl.count = 1; sl.pos = (uint32_t)pos; sl.len = (uint8_t)len; return; }
      s = (s + 1) & (cap - 1);
    }
  };
  auto is_word_byte = [](uint8_t b) { return (b >= 'a' && b <= 'z') || (b >= '0' && b <= '9') || b >= 0x80 || (b >= 'A' && b <= 'Z' && b != 'C' && b != 'W' && b != 'D'); };
  size_t ds = 0;
  for (size_t de : doc_end) {
    // short substrings
    for (size_t p = ds; p < de; p++) {
      size_t maxl = std::min<size_t>(10, de - p);
      for (size_t l = 2; l <= maxl; l++) {
        // do not split a UTF-8 sequence at either end
        if ((norm[p] & 0xC0) == 0x80) break;
        if (p + l < de && (norm[p + l] & 0xC0) == 0x80) continue;
        bump(p, l);
      }
    }
    // atom-aligned longer n-grams
    std::vector<size_t> starts;
    for (size_t p = ds; p < de;) {
      starts.push_back(p);
      size_t q = p;
      if (capcode == 2 && (norm[q] == 'D' || norm[q] == 'C' || norm[q] == 'W')) { q++; if (q < de && (norm[q] == 'C' || norm[q] == 'W')) q++; }
      if (q < de && norm[q] == ' ') q++;
      size_t w = q;
      while (w < de && is_word_byte(norm[w])) w++;
      if (w == q) w = std::min(de, q + 1 > p ? q + 1 : p + 1);
      p = std::max(w, p + 1);
    }
    starts.push_back(de);
    for (size_t a = 0; a + 1 < starts.size(); a++)
      for (size_t b = a + 1; b < starts.size() && b <= a + 6; b++) {
        size_t l = starts[b] - starts[a];
        if (l > 40) break;
        if (l > 10) bump(starts[a], l);
      }
    ds = de;
  }
  // 3. pick the best
  struct Cand { double value; uint32_t pos; uint8_t len; };
  std::vector<Cand> cands;
  for (auto& sl : table)
    if (sl.h != 0 && sl.count >= 4) cands.push_back({(double)sl.count * (double)(sl.len - 1), sl.pos, sl.len});
  std::sort(cands.begin(), cands.end(), [&](const Cand& a, const Cand& b) {
    if (a.value != b.value) return a.value > b.value;
    if (a.len != b.len) return a.len < b.len;
    return std::memcmp(&norm[a.pos], &norm[b.pos], a.len) < 0;
  });
  std::vector<std::string> tokens;
  std::vector<float> tok_scores;
  std::vector<uint8_t> special;
  // single bytes (go/tokenmonster.go:200-232): capcode 2 / UTF-8 -> genUTF8bytes; capcode 0 -> all 256
  bool single[256];
  std::memset(single, 0, sizeof single);
  if (capcode == 0) { for (int i = 0; i < 256; i++) single[i] = true; }
  else {
    for (int i = 32; i < 127; i++) if (!(i >= 'A' && i <= 'Z' && i != 'C' && i != 'W' && i != 'D')) single[i] = true;
    single[9] = single[10] = single[13] = true;
    for (int i = 0x80; i <= 0xBF; i++) single[i] = true;
    for (int i = 0xC2; i <= 0xF4; i++) single[i] = true;
  }
  uint64_t byte_count[256] = {0};
  for (uint8_t b : norm) byte_count[b]++;
  const double total = (double)std::max<size_t>(norm.size(), 1);
  for (int i = 0; i < 256; i++) if (single[i]) { tokens.push_back(std::string(1, (char)i)); tok_scores.push_back((float)((double)byte_count[i] / total)); }
  const size_t n_single = tokens.size();
  // A chosen token that equals the "D "-duplicate of another one shares its ID (go/tokenmonster.go:3450-3462), so the number of
  // IDs is only known after the build: take candidates in value order until the image has exactly vocab_size IDs.
  std::unordered_map<std::string, int> chosen;
  size_t next_cand = 0;
  uint32_t want = vocab_size > n_single + (with_unk ? 1u : 0u) ? vocab_size - (uint32_t)n_single - (with_unk ? 1u : 0u) : 0;
  for (int pass = 0; pass < 12; pass++) {
    while (chosen.size() < want && next_cand < cands.size()) {
      const Cand& c = cands[next_cand++];
      std::string s((const char*)&norm[c.pos], c.len);
      if (chosen.emplace(s, 1).second) { tokens.push_back(std::move(s)); tok_scores.push_back((float)(c.value / (double)(c.len - 1) * (double)c.len / total)); }
    }
    special.assign(tokens.size(), 0);
    int rc = build_vocab_image(tokens, special, capcode, /*charset=*/1, norm_flag, level, with_unk, image, &tok_scores);
    if (rc != TM_OK) return rc;
    const uint32_t got = (uint32_t)image[11] | ((uint32_t)image[12] << 8) | ((uint32_t)image[13] << 16);   // vocabSize, Appendix A
    if (got >= vocab_size || next_cand >= cands.size()) break;
    want += vocab_size - got;
  }
  return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_synth_corpus(uint32_t kind, uint64_t seed, uint64_t nbytes, uint32_t median_doc, uint8_t* text_out,
                    uint64_t* offsets_out, uint32_t max_docs, uint32_t* ndocs_out, uint64_t* nbytes_out) {
  if (!text_out || !offsets_out || !ndocs_out || !nbytes_out || kind > 2) return tmh::set_error(TM_E_INVALID, "bad argument");
  std::string text;
  std::vector<uint64_t> offs;
  uint32_t nd = tmh::synth_corpus(kind, seed, nbytes, median_doc ? median_doc : 2048, text, offs, max_docs);
  std::memcpy(text_out, text.data(), text.size());
  std::memcpy(offsets_out, offs.data(), offs.size() * 8);
  *ndocs_out = nd;
  *nbytes_out = text.size();
  return TM_OK;
}

int tm_synth_vocab(uint32_t kind, uint32_t vocab_size, uint32_t capcode, uint32_t norm_flag, uint32_t level,
                   uint64_t seed, int with_unk, uint8_t** out, size_t* out_n) {
  if (!out || !out_n || kind > 2) return tmh::set_error(TM_E_INVALID, "bad argument");
  std::vector<uint8_t> image;
  int rc = tmh::synth_vocab_image(kind, vocab_size, capcode, norm_flag, level, seed, with_unk != 0, image);
  if (rc != TM_OK) return rc;
  *out = (uint8_t*)std::malloc(image.size());
  std::memcpy(*out, image.data(), image.size());
  *out_n = image.size();
  return TM_OK;
}

}  // extern "C"
