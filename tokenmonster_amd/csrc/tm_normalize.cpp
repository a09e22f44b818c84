// tm_normalize.cpp — host-side pre-step of Tokenize: norm.Normalize then capcode.Encode
// (go/tokenmonster.go:233-253).  Neither third-party module is in /root/reference; the byte
// transforms follow tokenmonster-cpp/src/tokenmonster.cpp:190-475 (normalizer) and the only in-tree
// statement of capcode level 2, javascript/tokenmonster.js:900-1005.  This implementation works on
// the UTF-8 byte stream directly (ASCII classified by table, everything else through ICU).
//
// Scope: all 256 values of the normalizer flag byte (NFD, lowercase, accents, quotemarks, collapse, trim, leadingspace, unixlines:
// training/README.md:110-123; checked against the reference runtime's normalize for every value, tests/test_builder_normalizer.py) and
// capcode 0 and 2.  Capcode level 1 has no statement in the reference tree and is refused (TM_E_INVALID) instead of being guessed.  This
// file is also where the device normalizer's and decoder's character tables come from (build_two_table, build_three_tables,
// build_dec_tables): made by the functions below, so the device cannot disagree with the host about a character.
#include "tm_build.h"
#include "tokenmonster_hip.h"
#include "tm_internal.h"
#include "tm_norm_masks.h"

#include <unicode/normalizer2.h>
#include <unicode/uchar.h>
#include <unicode/locid.h>
#include <unicode/unistr.h>

#include <atomic>
#include <condition_variable>
#include <algorithm>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace tmh {

// A small persistent worker pool: the host-fallback normalizer and the decode post-pass run once per batch on a few thousand
// documents, and spawning their threads every time cost more than the work (64 fresh threads: 4 ms for 2 ms of work).
// Workers are created on first use and never joined (the pool object is leaked on purpose: no destructor-order problems at
// exit).  `work` must be a loop that pulls from a shared queue: helpers that do not wake up in time are simply not used.
// Several jobs can be open at once (the lanes of the host-to-host pipeline each normalize a few fallback documents at the same
// time): an idle worker joins whichever open job still wants helpers.
namespace {
struct Job { const std::function<void()>* work; uint32_t want, started = 0, running = 0; };
struct WorkerPool {
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  size_t nthreads = 0;
  std::vector<Job*> jobs;                        // open jobs
  Job* pick() { for (Job* j : jobs) if (j->started < j->want) return j; return nullptr; }
  void worker() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      Job* j = nullptr;
      cv_work.wait(lk, [&] { return (j = pick()) != nullptr; });
      j->started++; j->running++;
      lk.unlock();
      (*j->work)();
      lk.lock();
      if (--j->running == 0) cv_done.notify_all();
    }
  }
};
WorkerPool* g_pool = nullptr;
std::mutex g_pool_create;
}  // namespace

void run_on_workers(uint32_t threads, const std::function<void()>& work) {
  if (threads <= 1) { work(); return; }
  { std::lock_guard<std::mutex> g(g_pool_create); if (!g_pool) g_pool = new WorkerPool(); }
  WorkerPool& p = *g_pool;
  Job job{&work, threads - 1};
  {
    std::unique_lock<std::mutex> lk(p.mu);
    size_t wanted = job.want;
    for (Job* j : p.jobs) wanted += j->want;
    const size_t cap = std::max<size_t>(job.want, std::max(1u, std::thread::hardware_concurrency()));
    while (p.nthreads < std::min(wanted, cap)) { std::thread([&p] { p.worker(); }).detach(); p.nthreads++; }
    p.jobs.push_back(&job);
  }
  if (job.want >= p.nthreads) p.cv_work.notify_all();
  else for (uint32_t k = 0; k < job.want; k++) p.cv_work.notify_one();   // (a pool that once served 256 threads is not woken for a job of 4)
  work();                                       // the caller works too
  std::unique_lock<std::mutex> lk(p.mu);
  job.want = job.started;                       // late wakers find nothing to do
  p.cv_done.wait(lk, [&] { return job.running == 0; });
  p.jobs.erase(std::find(p.jobs.begin(), p.jobs.end(), &job));
}

}  // namespace tmh

namespace tmh {
namespace {

struct Cp { uint32_t r; int n; bool raw; };   // raw: undecodable byte carried through

inline Cp next_cp(const uint8_t* b, size_t len) {
  uint8_t b0 = b[0];
  if (b0 < 0x80) return {b0, 1, false};
  int need = 0; uint32_t cp = 0;
  if (b0 >= 0xC2 && b0 < 0xE0) { need = 1; cp = b0 & 0x1F; }
  else if (b0 >= 0xE0 && b0 < 0xF0) { need = 2; cp = b0 & 0x0F; }
  else if (b0 >= 0xF0 && b0 < 0xF5) { need = 3; cp = b0 & 0x07; }
  if (need == 0 || (size_t)need >= len) return {b0, 1, true};
  for (int k = 1; k <= need; k++) {
    if ((b[k] & 0xC0) != 0x80) return {b0, 1, true};
    cp = (cp << 6) | (b[k] & 0x3F);
  }
  if ((need == 2 && cp < 0x800) || (need == 3 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF))
    return {b0, 1, true};
  return {cp, need + 1, false};
}

enum : uint8_t { kUpper = 1, kLower = 2, kLetter = 4, kDigit = 8, kMark = 16 };

inline uint8_t classify(const Cp& c) {
  if (c.raw) return 0;
  uint32_t r = c.r;
  if (r < 0x80) {
    if (r >= 'a' && r <= 'z') return kLower | kLetter;
    if (r >= 'A' && r <= 'Z') return kUpper | kLetter;
    if (r >= '0' && r <= '9') return kDigit;
    return 0;
  }
  switch (u_charType((UChar32)r)) {
    case U_UPPERCASE_LETTER: return kUpper | kLetter;
    case U_LOWERCASE_LETTER: return kLower | kLetter;
    case U_TITLECASE_LETTER: case U_MODIFIER_LETTER: case U_OTHER_LETTER: return kLetter;
    case U_DECIMAL_DIGIT_NUMBER: return kDigit;
    case U_NON_SPACING_MARK: case U_ENCLOSING_MARK: case U_COMBINING_SPACING_MARK: return kMark;
    default: return 0;
  }
}

inline void put_cp(std::vector<uint8_t>& o, uint32_t r) {
  if (r < 0x80) o.push_back((uint8_t)r);
  else if (r < 0x800) { o.push_back(0xC0 | (r >> 6)); o.push_back(0x80 | (r & 0x3F)); }
  else if (r < 0x10000) { o.push_back(0xE0 | (r >> 12)); o.push_back(0x80 | ((r >> 6) & 0x3F)); o.push_back(0x80 | (r & 0x3F)); }
  else { o.push_back(0xF0 | (r >> 18)); o.push_back(0x80 | ((r >> 12) & 0x3F)); o.push_back(0x80 | ((r >> 6) & 0x3F)); o.push_back(0x80 | (r & 0x3F)); }
}
inline void put_lower(std::vector<uint8_t>& o, const Cp& c) {
  if (c.r < 0x80) o.push_back((uint8_t)(c.r | 0x20));
  else put_cp(o, (uint32_t)u_tolower((UChar32)c.r));
}

// what the encoder remembers about the previous one/two input code points
struct Last {
  bool space = false, letter = false, apostrophe = false, mark = false, digit = false;
  bool joiner() const { return letter || apostrophe || mark; }      // tokenmonster.js:915, :954
};
inline Last describe(const Cp& c, uint8_t cls) {
  Last l;
  l.space = !c.raw && c.r == ' ';
  l.letter = (cls & kLetter) != 0;
  l.apostrophe = !c.raw && (c.r == '\'' || c.r == 0x2019);
  l.mark = (cls & kMark) != 0;
  l.digit = (cls & kDigit) != 0;
  return l;
}

// tokenmonster.js:924-951 — after a run of capitals turns out to be followed by lowercase, every
// letter of the run after the first gets its own "DC " marker.  `from` is the byte offset just after
// the first (lower-cased) letter of the run.
void mark_run_letters(std::vector<uint8_t>& buf, size_t from) {
  std::vector<uint8_t> tail(buf.begin() + (std::ptrdiff_t)from, buf.end());
  buf.resize(from);
  size_t i = 0, n = tail.size();
  while (i < n) {
    if (tail[i] == 'D' && i + 1 < n && tail[i + 1] == ' ') {
      // an existing "D " (digit->letter or punctuation->letter inside the run)
      bool lower_next = false; int ln = 0;
      if (i + 2 < n) { Cp c = next_cp(&tail[i + 2], n - (i + 2)); lower_next = (classify(c) & kLower) != 0; ln = c.n; }
      if (lower_next) {
        buf.push_back('D'); buf.push_back('C'); buf.push_back(' ');
        buf.insert(buf.end(), tail.begin() + (std::ptrdiff_t)(i + 2), tail.begin() + (std::ptrdiff_t)(i + 2 + ln));
        i += 2 + (size_t)ln;
      } else {
        // JS skips "D ", and the element after it, unexamined
        size_t skip = 2;
        if (i + 2 < n) skip += (size_t)next_cp(&tail[i + 2], n - (i + 2)).n;
        if (i + skip > n) skip = n - i;
        buf.insert(buf.end(), tail.begin() + (std::ptrdiff_t)i, tail.begin() + (std::ptrdiff_t)(i + skip));
        i += skip;
      }
      continue;
    }
    Cp c = next_cp(&tail[i], n - i);
    if (classify(c) & kLower) { buf.push_back('D'); buf.push_back('C'); buf.push_back(' '); }
    buf.insert(buf.end(), tail.begin() + (std::ptrdiff_t)i, tail.begin() + (std::ptrdiff_t)(i + (size_t)c.n));
    i += (size_t)c.n;
  }
}

// capcode level 2 encoder, tokenmonster.js:900-1005
void capcode_encode(const uint8_t* in, size_t n, std::vector<uint8_t>& buf) {
  buf.clear();
  buf.reserve(n + n / 2 + 8);
  size_t goback = 0, word_token_pos = 0;
  Last last, last2;   // rlast = rlast2 = '.' initially: every predicate false
  bool in_word = false, multi = false;
  size_t i = 0;
  while (i < n) {
    // fast path: outside a capital run, a stretch of lowercase ASCII words separated by single spaces needs no marker
    // (every letter follows a letter or a space, :970) and is copied in bulk
    if (!in_word && (last.space || last.letter) && in[i] - 'a' < 26u) {
      size_t j = i;
      while (j < n) {
        const uint8_t x = in[j];
        if (x - 'a' < 26u) j++;
        else if (x == ' ' && j + 1 < n && in[j + 1] - 'a' < 26u) j++;
        else break;
      }
      buf.insert(buf.end(), in + i, in + j);
      Last letter; letter.letter = true;
      Last space; space.space = true;
      if (j - i >= 2) last2 = in[j - 2] == ' ' ? space : letter; else last2 = last;
      last = letter;                                  // the stretch always ends on a letter
      i = j;
      continue;
    }
    Cp c = next_cp(in + i, n - i);
    uint8_t cls = classify(c);
    if (in_word) {
      if (cls & kUpper) {                                           // :913-919
        if (!last.joiner()) { buf.push_back('D'); buf.push_back(' '); }
        multi = true;
        put_lower(buf, c);
      } else {
        if (cls & kLower) {                                         // :921-955
          in_word = false;
          buf[word_token_pos] = 'C';
          if (multi) mark_run_letters(buf, goback);
          if (!last.joiner()) { buf.push_back('D'); buf.push_back(' '); }
        } else if (cls & kDigit) {                                  // :957-961
          if (!last.digit) { buf.push_back('D'); buf.push_back(' '); }
        } else if (!((!c.raw && (c.r == '\'' || c.r == 0x2019)) || (cls & kMark))) {
          in_word = false;                                          // :962-964
        }
        buf.insert(buf.end(), in + i, in + i + c.n);                // :966
      }
    } else {
      if (cls & kLower) {                                           // :969-974
        if (!(last.space || last.letter || (last2.letter && last.apostrophe) || last.mark)) {
          buf.push_back('D'); buf.push_back(' ');
        }
        buf.insert(buf.end(), in + i, in + i + c.n);
      } else if (cls & kUpper) {                                    // :975-990
        if (last.space) {
          word_token_pos = buf.size() - 1;
          buf[word_token_pos] = 'W';
          buf.push_back(' ');
        } else {
          buf.push_back('D');
          word_token_pos = buf.size();
          buf.push_back('W');
          buf.push_back(' ');
        }
        put_lower(buf, c);
        goback = buf.size();
        multi = false;
        in_word = true;
      } else if (cls & kDigit) {                                    // :991-996
        if (!(last.space || last.digit)) { buf.push_back('D'); buf.push_back(' '); }
        buf.insert(buf.end(), in + i, in + i + c.n);
      } else {
        buf.insert(buf.end(), in + i, in + i + c.n);                // :997-999
      }
    }
    last2 = last;
    last = describe(c, cls);
    i += (size_t)c.n;
  }
}

bool is_ascii(const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) if (d[i] & 0x80) return false;
  return true;
}

void nfd_bytes(std::vector<uint8_t>& b) {    // tokenmonster.cpp:190-212
  if (is_ascii(b.data(), b.size())) return;
  // NFD is local: ASCII is inert (every ASCII character is a starter with no decomposition and composition is not
  // involved), so only the maximal non-ASCII stretches need ICU; the ASCII in between is copied through.
  UErrorCode status = U_ZERO_ERROR;
  const icu::Normalizer2* nz = icu::Normalizer2::getNFDInstance(status);
  std::vector<uint8_t> out;
  out.reserve(b.size() + b.size() / 8 + 16);
  const size_t n = b.size();
  size_t i = 0;
  std::string tmp;
  while (i < n) {
    size_t j = i;
    while (j < n && !(b[j] & 0x80)) j++;
    out.insert(out.end(), b.begin() + (std::ptrdiff_t)i, b.begin() + (std::ptrdiff_t)j);
    if (j >= n) break;
    size_t k = j;
    while (k < n && (b[k] & 0x80)) k++;
    icu::UnicodeString u = icu::UnicodeString::fromUTF8(icu::StringPiece((const char*)b.data() + j, (int32_t)(k - j)));
    icu::UnicodeString o;
    nz->normalize(u, o, status);
    tmp.clear();
    o.toUTF8String(tmp);
    out.insert(out.end(), tmp.begin(), tmp.end());
    i = k;
  }
  b.swap(out);
}

void lower_bytes(std::vector<uint8_t>& b) {  // tokenmonster.cpp:214-229
  bool need = false;
  for (auto x : b) if ((x & 0x80) || (x >= 'A' && x <= 'Z')) { need = true; break; }
  if (!need) return;
  icu::UnicodeString u = icu::UnicodeString::fromUTF8(icu::StringPiece((const char*)b.data(), (int32_t)b.size()));
  u.toLower(icu::Locale::getRoot());
  std::string s;
  u.toUTF8String(s);
  b.assign(s.begin(), s.end());
}

}  // namespace

// capcode level 2 decoder, javascript/tokenmonster.js:1007-1065: the four flags of CapcodeDecoder live in `st` (tm_internal.h), so
// that a streaming Decoder (go/tokenmonster.go:552-700) carries them from call to call; appends to `out`
void capcode_decode_stream(CapcodeState& st, const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
  out.reserve(out.size() + n);
  bool in_word = st.in_word, in_char = st.in_char, del = st.del, ignore = st.ignore;
  size_t i = 0;
  while (i < n) {
    const Cp c = next_cp(in + i, n - i);
    i += (size_t)c.n;
    if (!c.raw && c.r == 'C') { in_char = true; in_word = false; continue; }
    if (!c.raw && c.r == 'W') { in_word = true; in_char = false; ignore = true; continue; }
    if (!c.raw && c.r == 'D') { del = true; continue; }
    if (!c.raw && c.r == ' ') {
      if (del) del = false;
      else { out.push_back(' '); if (!ignore) in_word = false; }
    } else {
      const uint8_t cls = classify(c);
      if (del) del = false;
      else if (in_char) { in_char = false; if (c.raw) out.push_back((uint8_t)c.r); else put_cp(out, c.r < 0x80 && (cls & kLower) ? c.r - 32 : (uint32_t)u_toupper((UChar32)c.r)); }
      else if (in_word) {
        if (cls & (kLower | kUpper)) put_cp(out, c.r < 0x80 ? (cls & kLower ? c.r - 32 : c.r) : (uint32_t)u_toupper((UChar32)c.r));
        else {
          if (c.raw) out.push_back((uint8_t)c.r); else put_cp(out, c.r);
          if (!((cls & kDigit) || (!c.raw && (c.r == '\'' || c.r == 0x2019)) || (cls & kMark))) in_word = false;
        }
      } else { if (c.raw) out.push_back((uint8_t)c.r); else put_cp(out, c.r); }
    }
    ignore = false;
  }
  st.in_word = in_word; st.in_char = in_char; st.del = del; st.ignore = ignore;
}
void capcode_decode(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
  out.clear();
  CapcodeState st;
  capcode_decode_stream(st, in, n, out);
}

// level 1 (marker 0x7F): no in-tree statement; delete the marker and the character after it
void nocapcode_decode_stream(CapcodeState& st, const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
  out.reserve(out.size() + n);
  bool del = st.del;
  for (size_t i = 0; i < n; i++) {
    if (in[i] == 0x7F) { del = true; continue; }
    if (del) { del = false; continue; }
    out.push_back(in[i]);
  }
  st.del = del;
}
void nocapcode_decode(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
  out.clear();
  CapcodeState st;
  nocapcode_decode_stream(st, in, n, out);
}

// The tables of the device normalizer for the two-byte characters U+0080..U+07FF (tm_norm_masks.h: NmTwo) and for the three-byte characters
// it passes through, made from this file's own building blocks for the flags {NFD, lowercase} of a vocabulary: the device cannot disagree
// with the host about them.  A character the tables cannot express (anything but "stays one character", "ASCII letter + one two-byte combining
// mark" or "two-byte letter + one two-byte combining mark"; a lower-case form of another length) is left without NT_OK: documents that contain
// it take the host path.
void build_two_table(uint32_t norm_flag, NmTwo* out) {
  for (int k = 0; k < NM_TWO_SIZE; k++) out[k] = NmTwo{0, 0};
  for (uint32_t cp = 0x80; cp < 0x800; cp++) {
    const uint32_t lead = 0xC0u | (cp >> 6), second = 0x80u | (cp & 0x3Fu);
    NmTwo& e = out[nm_two_index(lead, second)];
    std::vector<uint8_t> t = {(uint8_t)lead, (uint8_t)second};
    if (norm_flag & 1) nfd_bytes(t);
    if (norm_flag & 2) {
      // lower-casing a whole text looks at a letter's neighbours (the final sigma: Σ -> ς at the end of a word, σ elsewhere); a character
      // whose lower-case form depends on them is not for a table
      std::vector<uint8_t> mid = {'a'};
      mid.insert(mid.end(), t.begin(), t.end());
      std::vector<uint8_t> both = mid, after = t;
      both.push_back('a'); after.push_back('a');
      lower_bytes(t); lower_bytes(mid); lower_bytes(both); lower_bytes(after);
      if (mid.size() != t.size() + 1 || both.size() != t.size() + 2 || after.size() != t.size() + 1 || !std::equal(t.begin(), t.end(), mid.begin() + 1) ||
          !std::equal(t.begin(), t.end(), both.begin() + 1) || !std::equal(t.begin(), t.end(), after.begin())) continue;
    }
    if (t.empty()) continue;
    const Cp c1 = next_cp(t.data(), t.size());
    if (c1.raw) continue;
    const uint8_t k1 = classify(c1);
    uint32_t cls;
    if (k1 & kUpper) cls = NC_U; else if (k1 & kLower) cls = NC_L; else if (k1 & kLetter) cls = NC_LO; else if (k1 & kDigit) cls = NC_N; else if (k1 & kMark) cls = NC_M; else cls = NC_O;
    if (c1.r == ' ' || c1.r == '\'') continue;                            // (no character of the range turns into one of these)
    std::vector<uint8_t> low;
    put_lower(low, c1);                                                    // what capcode writes for a capital (:918, :986)
    // A capital whose lower-case form is not a lower-case letter (ϒ ϓ ϔ: upper-case symbols without one) is not for the device's rule table:
    // as a later capital of a run that ends in a lower-case letter the reference gives it no "DC " - its pass over the run (:924-951,
    // mark_run_letters above) looks at the letters AFTER lower-casing, and this one is still a capital.  Its documents take the host path.
    if ((k1 & kUpper) && !(classify(next_cp(low.data(), low.size())) & kLower)) continue;
    if ((size_t)c1.n == t.size() && c1.n == 2) {                           // stays one two-byte character
      if (low.size() != 2) continue;
      e.a = cls | NT_OK | ((uint32_t)t[0] << 8) | ((uint32_t)t[1] << 16);
      e.b = (uint32_t)low[0] | ((uint32_t)low[1] << 8);
    } else if (c1.n == 1 && t.size() == 3) {                               // ASCII letter + one combining mark
      const Cp c2 = next_cp(t.data() + 1, 2);
      if (c2.raw || c2.n != 2 || !(classify(c2) & kMark) || !(k1 & kLetter) || low.size() != 1) continue;
      e.a = cls | NT_OK | NT_DECOMP | ((uint32_t)t[0] << 8) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 24);
      e.b = (uint32_t)low[0] | ((uint32_t)t[1] << 8);
    } else if (c1.n == 2 && t.size() == 4) {                               // two-byte letter + one combining mark (й ё ά ...)
      const Cp c2 = next_cp(t.data() + 2, 2);
      if (c2.raw || c2.n != 2 || !(classify(c2) & kMark) || !(k1 & kLetter) || low.size() != 2) continue;
      e.a = cls | NT_OK | NT_DECOMP2 | ((uint32_t)t[0] << 8) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 24);
      e.b = (uint32_t)low[0] | ((uint32_t)low[1] << 8) | ((uint32_t)t[3] << 16);
    }
  }
}

// Latin Extended Additional U+1E00..U+1EFF (tm_norm_masks.h: NmLea): what NFD (+ lowercase) makes of each character, where that is an ASCII
// letter followed by one or two two-byte combining marks
void build_lea_table(uint32_t norm_flag, NmLea* out) {
  for (int k = 0; k < NM_LEA_SIZE; k++) out[k] = NmLea{0, 0};
  if (!(norm_flag & 1)) return;
  for (uint32_t cp = 0x1E00; cp < 0x1F00; cp++) {
    std::vector<uint8_t> t;
    put_cp(t, cp);
    nfd_bytes(t);
    if (norm_flag & 2) lower_bytes(t);
    if (t.size() != 3 && t.size() != 5) continue;
    const Cp c1 = next_cp(t.data(), t.size());
    if (c1.raw || c1.n != 1) continue;
    const uint8_t k1 = classify(c1);
    if (!(k1 & kLetter)) continue;
    bool ok = true;
    uint32_t marks[2] = {0, 0};
    const uint32_t nm = (uint32_t)(t.size() - 1) / 2;
    for (uint32_t j = 0; j < nm && ok; j++) {
      const Cp cm = next_cp(t.data() + 1 + 2 * j, t.size() - 1 - 2 * j);
      ok = !cm.raw && cm.n == 2 && (classify(cm) & kMark);
      marks[j] = (uint32_t)t[1 + 2 * j] | ((uint32_t)t[2 + 2 * j] << 8);
    }
    if (!ok) continue;
    std::vector<uint8_t> low;
    put_lower(low, c1);
    if (low.size() != 1) continue;
    const uint32_t cls = (k1 & kUpper) ? NC_U : (k1 & kLower) ? NC_L : NC_LO;
    out[cp - 0x1E00] = NmLea{cls | NT_OK | ((uint32_t)t[0] << 8) | ((uint32_t)low[0] << 16) | (nm << 24), marks[0] | (marks[1] << 16)};
  }
}

// The voiced kana of U+3040..U+30FF (tm_norm_masks.h: NM_KANA_SIZE): the characters NFD turns into a letter of the same range followed by
// U+3099 or U+309A
void build_kana_table(uint16_t* out) {
  for (int k = 0; k < NM_KANA_SIZE; k++) out[k] = 0;
  for (uint32_t cp = 0x3040; cp < 0x3040 + (uint32_t)NM_KANA_SIZE; cp++) {
    std::vector<uint8_t> t;
    put_cp(t, cp);
    nfd_bytes(t);
    if (t.size() != 6) continue;
    const Cp c1 = next_cp(t.data(), 6);
    if (c1.raw || c1.n != 3) continue;
    const Cp c2 = next_cp(t.data() + 3, 3);
    if (c2.raw || c2.n != 3 || (c2.r != 0x3099 && c2.r != 0x309A) || !(classify(c2) & kMark)) continue;
    const uint8_t k1 = classify(c1);
    if (!(k1 & kLetter) || (k1 & (kUpper | kLower)) || (uint32_t)c1.r - 0x3000u >= 0x100u) continue;
    out[cp - 0x3040] = (uint16_t)(NK_OK | (c2.r == 0x309A ? NK_SEMI : 0u) | ((uint32_t)c1.r & 0xFFu));
  }
}

// The three-byte characters of U+0900..U+1BFF that NFD splits in two three-byte characters (tm_norm_masks.h: NM_DEC3_SIZE): a letter without case
// or a mark of class 0, NFD-stable by itself, and a mark behind it
void build_dec3_table(uint32_t* out) {
  for (uint32_t k = 0; k < NM_DEC3_SIZE + NM_DEC3_THIRDS; k++) out[k] = 0;
  uint32_t nthird = 0;
  for (uint32_t cp = NM_DEC3_BASE; cp < NM_DEC3_BASE + NM_DEC3_SIZE; cp++) {
    std::vector<uint8_t> t;
    put_cp(t, cp);
    nfd_bytes(t);
    if (t.size() != 6 && t.size() != 9) continue;
    const Cp c1 = next_cp(t.data(), 3);
    if (c1.raw || c1.n != 3) continue;
    bool ok = true;
    uint32_t parts[3] = {(uint32_t)c1.r, 0, 0};
    for (size_t j = 1; j < t.size() / 3 && ok; j++) {      // every further part: a mark of three bytes inside the range the tables cover
      const Cp cj = next_cp(t.data() + 3 * j, 3);
      ok = !cj.raw && cj.n == 3 && classify(cj) == kMark && (uint32_t)cj.r - 0x800u < 0x1800u;
      parts[j] = (uint32_t)cj.r;
    }
    if (!ok || (uint32_t)c1.r - 0x800u >= 0x1800u) continue;
    const uint8_t k1 = classify(c1);
    const bool letter = (k1 & kLetter) && !(k1 & (kUpper | kLower));
    if (!letter && !(k1 == kMark && u_getCombiningClass((UChar32)c1.r) == 0)) continue;
    std::vector<uint8_t> low;
    put_lower(low, c1);
    if (low.size() != 3 || low[0] != t[0] || low[1] != t[1] || low[2] != t[2]) continue;
    uint32_t third = 0;
    if (t.size() == 9) {                                   // three parts (Kannada U+0CCB, Sinhala U+0DDD): the third in one of the words behind the table
      if (nthird + 1 >= NM_DEC3_THIRDS) continue;
      third = ++nthird;
      out[NM_DEC3_SIZE + third] = parts[2];
    }
    out[cp - NM_DEC3_BASE] = ND_OK | (letter ? ND_LETTER : 0u) | (parts[0] - 0x800u) | ((parts[1] - 0x800u) << 13) | (third << 26);
  }
}

// The three-byte combining marks of U+0800..U+1FFF with a canonical class > 0 that the flags leave alone (tm_norm_masks.h: NM_CCC_SIZE): the class
// (marks == false: only the digits, NM_CCC_DIGIT - decimal digits of three bytes, Devanagari to Tai Tham)
void build_ccc_table(uint32_t norm_flag, bool marks, uint8_t* out) {
  for (uint32_t k = 0; k < NM_CCC_SIZE; k++) out[k] = 0;
  for (uint32_t cp = NM_CCC_BASE; cp < NM_CCC_BASE + NM_CCC_SIZE; cp++) {
    std::vector<uint8_t> in, t;
    put_cp(in, cp);
    t = in;
    if (norm_flag & 1) nfd_bytes(t);
    if (norm_flag & 2) lower_bytes(t);
    const Cp c1 = next_cp(in.data(), in.size());
    if (c1.raw || c1.n != 3 || t != in) continue;
    std::vector<uint8_t> low;
    put_lower(low, c1);
    const int ccc = u_getCombiningClass((UChar32)cp);
    if (marks && low == in && classify(c1) == kMark && ccc > 0 && ccc < (int)NM_CCC_LOWER) out[cp - NM_CCC_BASE] = (uint8_t)ccc;
    else if (low == in && classify(c1) == kDigit) out[cp - NM_CCC_BASE] = (uint8_t)NM_CCC_DIGIT;
    // a lower-case letter of three bytes (Georgian Mkhedruli, the phonetic extensions, small Cherokee ...): capcode and the flags never change one
    else if (low == in && (classify(c1) & kLower) && !(classify(c1) & (kUpper | kDigit | kMark))) out[cp - NM_CCC_BASE] = (uint8_t)NM_CCC_LOWER;
  }
}

// blk[NM_BLK_WORDS], cp[NM_CP_WORDS]: two bits per block of 64 code points / per code point of U+0000..U+FFFF (tm_norm_masks.h)
void build_three_tables(uint32_t norm_flag, uint32_t* blk, uint32_t* cpt) {
  for (int k = 0; k < NM_BLK_WORDS; k++) blk[k] = 0;
  for (int k = 0; k < NM_CP_WORDS; k++) cpt[k] = 0;
  for (uint32_t block = 0x800 >> 6; block < 1024; block++) {
    uint32_t first = 4;
    bool same = true;
    for (uint32_t cp = block << 6; cp < (block + 1) << 6; cp++) {
      uint32_t code = 0;
      if (cp < 0xD800 || cp > 0xDFFF) {
        std::vector<uint8_t> in, t;
        put_cp(in, cp);
        t = in;
        if (norm_flag & 1) nfd_bytes(t);
        if (norm_flag & 2) lower_bytes(t);
        const Cp c1 = next_cp(in.data(), in.size());
        std::vector<uint8_t> low;
        if (!c1.raw) put_lower(low, c1);
        const uint8_t k1 = classify(c1);
        // untouched by the flags, nothing capcode would mark or lower-case, and to capcode a letter without case or "other"
        if (t == in && !c1.raw && c1.n == 3 && low == in && !(k1 & (kUpper | kLower | kDigit | kMark))) code = (k1 & kLetter) ? 2u : 1u;
        // a combining mark of class 0 - the variation selectors (U+FE0F behind an emoji), the enclosing keycap, the spacing vowel signs of the
        // Indic scripts ...: canonical ordering never moves it, NFD leaves it alone: class M on the device too (round 5)
        // (without the NFD flag nothing is ever reordered: every mark is of that kind)
        else if (t == in && !c1.raw && c1.n == 3 && low == in && k1 == kMark && (u_getCombiningClass((UChar32)cp) == 0 || !(norm_flag & 1))) code = 3u;
      }
      cpt[cp >> 4] |= code << (2u * (cp & 15u));
      if (first == 4) first = code; else if (code != first) same = false;
    }
    blk[block >> 4] |= (same ? first : 3u) << (2u * (block & 15u));
  }
}

// blk4[NM_BLK4_WORDS]: two bits per block of 64 code points of U+10000..U+10FFFF (tm_norm_masks.h): 1 / 2 when EVERY code point of the block
// is untouched by the flags, caseless, neither digit nor mark, and all of them are "other" (1) resp. all letters (2); 0 otherwise
void build_four_table(uint32_t norm_flag, uint32_t* blk4) {
  for (int k = 0; k < NM_BLK4_WORDS; k++) blk4[k] = 0;
  for (uint32_t block = 0; block < 16384; block++) {
    uint32_t first = 4;
    for (uint32_t cp = 0x10000u + (block << 6); cp < 0x10000u + ((block + 1) << 6) && first != 0; cp++) {
      uint32_t code = 0;
      std::vector<uint8_t> in, t;
      put_cp(in, cp);
      t = in;
      if (norm_flag & 1) nfd_bytes(t);
      if (norm_flag & 2) lower_bytes(t);
      const Cp c1 = next_cp(in.data(), in.size());
      std::vector<uint8_t> low;
      if (!c1.raw) put_lower(low, c1);
      const uint8_t k1 = classify(c1);
      if (t == in && !c1.raw && c1.n == 4 && low == in && !(k1 & (kUpper | kLower | kDigit | kMark))) code = (k1 & kLetter) ? 2u : 1u;
      if (first == 4) first = code; else if (code != first) first = 0;
    }
    blk4[block >> 4] |= (first & 3u) << (2u * (block & 15u));
  }
}

// The tables of the device capcode DECODER (tm_decode.hip: k_dec_capcode), made from the functions the host decoder above uses, so that the
// device cannot disagree with it.
//   two[(lead - 0xC2) << 6 | second & 63], the two-byte characters U+0080..U+07FF: bit 0 the device may decode the character, bit 1 upper- or
//     lower-case letter (a capitalised word capitalises it), bit 2 digit or mark (keeps a capitalised word going), bits 8..15 / 16..23 the two
//     bytes of its upper-case form.  A character whose upper-case form has another length (ÿ µ ı ſ ŉ ...) stays without bit 0: its document
//     is left to the host.
//   blk / cp, the three-byte characters, two bits per block of 64 code points / per code point as in the normalizer's tables: 0 = host (a
//     letter with case, a surrogate), 1 = a character the decoder only passes on (it ends a capitalised word), 2 = a digit or mark (keeps the
//     word going); blocks: 3 = look the code point up.
//   blk4, the four-byte characters, two bits per block of 64 code points of U+10000..U+10FFFF: 1 / 2 as above when the whole block agrees
//     (emoji, symbols, the ideographs of plane 2 ...), else 0 = host (Deseret, Adlam and the other cased scripts; blocks that mix digits or
//     marks with other characters).
void build_dec_tables(uint32_t* two, uint32_t* blk, uint32_t* cpt, uint32_t* blk4) {
  for (uint32_t k = 0; k < DEC_BLK4_WORDS; k++) blk4[k] = 0;
  for (uint32_t block = 0; block < 16384; block++) {
    uint32_t first = 4;
    for (uint32_t cp = 0x10000u + (block << 6); cp < 0x10000u + ((block + 1) << 6) && first != 0; cp++) {
      std::vector<uint8_t> in;
      put_cp(in, cp);
      const Cp c = next_cp(in.data(), in.size());
      const uint8_t cls = classify(c);
      uint32_t code = 0;
      if (!c.raw && c.n == 4 && !(cls & (kLower | kUpper)) && (uint32_t)u_toupper((UChar32)cp) == cp) code = (cls & (kDigit | kMark)) ? 2u : 1u;
      if (first == 4) first = code; else if (code != first) first = 0;
    }
    blk4[block >> 4] |= (first & 3u) << (2u * (block & 15u));
  }
  for (uint32_t cp = 0x80; cp < 0x800; cp++) {
    const uint8_t b[2] = {(uint8_t)(0xC0u | (cp >> 6)), (uint8_t)(0x80u | (cp & 0x3Fu))};
    uint32_t e = 0;
    const Cp c = next_cp(b, 2);
    if (!c.raw && c.n == 2) {
      const uint8_t cls = classify(c);
      std::vector<uint8_t> up;
      put_cp(up, (uint32_t)u_toupper((UChar32)c.r));
      if (up.size() == 2) e = 1u | ((cls & (kLower | kUpper)) ? 2u : 0u) | ((cls & (kDigit | kMark)) ? 4u : 0u) | ((uint32_t)up[0] << 8) | ((uint32_t)up[1] << 16);
    }
    two[cp - 0x80] = e;
  }
  for (int k = 0; k < NM_BLK_WORDS; k++) blk[k] = 0;
  for (int k = 0; k < NM_CP_WORDS; k++) cpt[k] = 0;
  for (uint32_t block = 0x800 >> 6; block < 1024; block++) {
    uint32_t first = 4;
    bool same = true;
    for (uint32_t cp = block << 6; cp < (block + 1) << 6; cp++) {
      uint32_t code = 0;
      if (cp < 0xD800 || cp > 0xDFFF) {
        std::vector<uint8_t> in;
        put_cp(in, cp);
        const Cp c = next_cp(in.data(), in.size());
        const uint8_t cls = classify(c);
        if (!c.raw && c.n == 3 && !(cls & (kLower | kUpper)) && (uint32_t)u_toupper((UChar32)cp) == cp) code = (cls & (kDigit | kMark)) ? 2u : 1u;
      }
      cpt[cp >> 4] |= code << (2u * (cp & 15u));
      if (first == 4) first = code; else if (code != first) same = false;
    }
    blk[block >> 4] |= (same ? first : 3u) << (2u * (block & 15u));
  }
}

// capcode level 1 has no statement in the reference tree: refused rather than guessed
bool normalize_supported(uint32_t capcode, uint32_t norm_flag) { return (capcode == 0 || capcode == 2) && norm_flag < 256; }
// what the DEVICE normalizer (tm_norm.hip) does itself; documents of vocabularies with further flags take the host path below
// (round 6: every flag - the byte-level ones and `accents` in a filter pass in front of it, k_pf_pass)
bool normalize_on_device(uint32_t capcode, uint32_t norm_flag) { return (capcode == 0 || capcode == 2) && norm_flag < 256; }

namespace {

// The byte-level flags 8 quotemarks, 16 collapse, 128 unixlines (training/README.md:110-123), as ONE left-to-right pass with the
// semantics of the reference's fused loops (tokenmonster.cpp:285-425): a space is dropped when the byte before it in the INPUT was a
// space; with `fused_unix` a '\n' that follows '\r' overwrites it; a curly quote E2 80 {98,99 | 9C,9D} collapses to ' or ".
void squeeze(std::vector<uint8_t>& b, bool quotes, bool collapse, bool fused_unix) {
  // IN PLACE, like the reference, and that is part of the behaviour: the two bytes in front of a quote byte are read from the buffer
  // that is being compacted, so after exactly one dropped byte the look-behind sees a byte that was already moved and the quote is
  // left alone (tests/test_builder_normalizer.py fuzzes this against the reference runtime)
  const size_t n = b.size();
  size_t on = 0;
  uint8_t last = 0;
  for (size_t i = 0; i < n; i++) {
    const uint8_t c = b[i];
    if (collapse && c == ' ') { if (last != ' ') b[on++] = ' '; last = ' '; continue; }
    if (fused_unix && c == '\n' && last == '\r') { b[on - 1] = '\n'; last = '\n'; continue; }
    last = c;
    if (quotes && (c == 0x98 || c == 0x99 || c == 0x9C || c == 0x9D) && i > 1 && b[i - 1] == 0x80 && b[i - 2] == 0xE2) {
      b[on - 2] = c < 0x9C ? '\'' : '"';
      on--;
      continue;
    }
    b[on++] = c;
  }
  b.resize(on);
}
// flag 128 on its own (tokenmonster.cpp:302-314): a '\r' directly before '\n' is dropped
void unix_lines(std::vector<uint8_t>& b) {
  const size_t n = b.size();
  if (n < 2) return;
  size_t on = 0;
  for (size_t i = 0; i + 1 < n; i++) if (!(b[i] == '\r' && b[i + 1] == '\n')) b[on++] = b[i];
  b[on++] = b[n - 1];
  b.resize(on);
}
// flags 32 trim / 64 leadingspace (tokenmonster.cpp:245-283), including what trim+leadingspace does to a text WITHOUT leading
// whitespace: its last non-blank byte goes as well (:274-277 cut at i2, not i2 + 1)
void trim_and_lead(std::vector<uint8_t>& b, bool trim, bool lead) {
  const ptrdiff_t n = (ptrdiff_t)b.size();
  auto add_lead = [](std::vector<uint8_t>& x) { if (!x.empty() && x[0] != ' ') x.insert(x.begin(), ' '); };
  if (!trim) { if (lead) add_lead(b); return; }
  ptrdiff_t i = 0, i2 = n - 1;
  while (i < n && b[(size_t)i] <= 32) i++;
  while (i2 >= 0 && b[(size_t)i2] <= 32) i2--;
  if (i2 < 0) { b.clear(); return; }
  if (!lead) { b = std::vector<uint8_t>(b.begin() + i, b.begin() + i2 + 1); return; }
  if (i == 0) { b.resize((size_t)i2); add_lead(b); return; }
  b[(size_t)(i - 1)] = ' ';
  b = std::vector<uint8_t>(b.begin() + (i - 1), b.begin() + i2 + 1);
}
// flag 4 accents (tokenmonster.cpp:231-243): NFD, then every non-spacing mark (Mn) goes
void remove_marks(std::vector<uint8_t>& b) {
  nfd_bytes(b);
  bool any = false;
  for (auto x : b) if (x & 0x80) { any = true; break; }
  if (!any) return;
  std::vector<uint8_t> out;
  out.reserve(b.size());
  for (size_t i = 0; i < b.size();) {
    const Cp c = next_cp(b.data() + i, b.size() - i);
    if (c.raw || u_charType((UChar32)c.r) != U_NON_SPACING_MARK) out.insert(out.end(), b.begin() + (ptrdiff_t)i, b.begin() + (ptrdiff_t)i + c.n);
    i += (size_t)c.n;
  }
  b.swap(out);
}

}  // namespace

// What flag 4 `accents` (NFD, then every non-spacing mark goes) leaves of the two-byte characters U+0080..U+07FF, for the device's filter pass
// (tm_norm.hip: k_pf_pass), from the function the host path uses: kind [0..1] - 0 the character stays as it is, 1 nothing is left of it (a
// non-spacing mark), 2 one byte, 3 two other bytes - | first byte << 8 | second byte << 16.  A character that leaves more than two bytes
// stays (kind 0): the normalizer pass behind the filter then finds something it may not decompose in this mode and hands the document to the host.
void build_accent_table(uint32_t* out) {
  for (int k = 0; k < NM_TWO_SIZE; k++) out[k] = 0;
  for (uint32_t cp = 0x80; cp < 0x800; cp++) {
    const uint32_t lead = 0xC0u | (cp >> 6), second = 0x80u | (cp & 0x3Fu);
    std::vector<uint8_t> t = {(uint8_t)lead, (uint8_t)second};
    remove_marks(t);
    uint32_t e = 0;
    if (t.empty()) e = 1u;
    else if (t.size() == 1) e = 2u | ((uint32_t)t[0] << 8);
    else if (t.size() == 2 && (t[0] != lead || t[1] != second)) e = 3u | ((uint32_t)t[0] << 8) | ((uint32_t)t[1] << 16);
    out[nm_two_index(lead, second)] = e;
  }
}

// norm.Normalize + capcode.Encode (go/tokenmonster.go:242-253); flag order as tokenmonster.cpp:428-475
void normalize_bytes(const uint8_t* data, size_t n, uint32_t capcode, uint32_t norm_flag, std::vector<uint8_t>& out) {
  std::vector<uint8_t> tmp(data, data + n);
  if (norm_flag > 1) {
    const bool q = norm_flag & 8, c = norm_flag & 16, u = norm_flag & 128;
    if (u && c) squeeze(tmp, q, true, true);                       // :433-441
    else {
      if (u) unix_lines(tmp);                                      // :443
      if (q || c) squeeze(tmp, q, c, false);                       // :445-453
    }
    if (norm_flag & (32 | 64)) trim_and_lead(tmp, norm_flag & 32, norm_flag & 64);   // :454-462
  }
  if (norm_flag & 4) { remove_marks(tmp); if (norm_flag & 2) lower_bytes(tmp); }     // :464-468
  else if (norm_flag & 2) { if (norm_flag & 1) nfd_bytes(tmp); lower_bytes(tmp); }   // :469-472
  else if (norm_flag & 1) nfd_bytes(tmp);                                            // :473
  if (capcode == 2) capcode_encode(tmp.data(), tmp.size(), out);
  else out.swap(tmp);
}

void capcode_decode_batch(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode, uint32_t threads,
                          std::vector<std::vector<uint8_t>>& outs) {
  outs.assign(ndocs, {});
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  threads = std::min<uint32_t>(threads, std::max(1u, ndocs));
  const uint32_t grab = std::max(1u, std::min(64u, ndocs / (threads * 4u)));
  std::atomic<uint32_t> next{0};
  auto work = [&]() {
    for (;;) {
      uint32_t base = next.fetch_add(grab);
      if (base >= ndocs) break;
      for (uint32_t d = base; d < std::min(ndocs, base + grab); d++) {
        const uint8_t* p = text + offsets[d];
        const size_t n = (size_t)(offsets[d + 1] - offsets[d]);
        if (capcode == 2) capcode_decode(p, n, outs[d]);
        else if (capcode == 1) nocapcode_decode(p, n, outs[d]);
        else outs[d].assign(p, p + n);
      }
    }
  };
  tmh::run_on_workers(threads, work);
}

}  // namespace tmh

namespace tmh {

// Normalizes every document on up to `threads` pooled workers (0 = one per hardware thread), fills out_offsets[ndocs+1], asks
// `alloc` for a destination of the total size (a malloc'd block, a pinned staging buffer ...) and places the documents there,
// also in parallel.
int normalize_batch_into(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode, uint32_t norm_flag,
                         uint32_t threads, uint64_t* out_offsets, const std::function<uint8_t*(uint64_t)>& alloc) {
  if (!normalize_supported(capcode, norm_flag))
    return set_error(TM_E_INVALID, "normalization flags %u / capcode %u not supported by the host normalizer", norm_flag, capcode);
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  threads = std::min<uint32_t>(threads, std::max(1u, ndocs));
  const uint32_t grab = std::max(1u, std::min(64u, ndocs / (threads * 4u)));
  std::vector<std::vector<uint8_t>> outs(ndocs);
  std::atomic<uint32_t> next{0};
  run_on_workers(threads, [&]() {
    for (;;) {
      const uint32_t base = next.fetch_add(grab);
      if (base >= ndocs) break;
      for (uint32_t d = base; d < std::min(ndocs, base + grab); d++)
        normalize_bytes(text + offsets[d], (size_t)(offsets[d + 1] - offsets[d]), capcode, norm_flag, outs[d]);
    }
  });
  uint64_t total = 0;
  for (uint32_t d = 0; d < ndocs; d++) { out_offsets[d] = total; total += outs[d].size(); }
  out_offsets[ndocs] = total;
  uint8_t* o = alloc(total);
  if (!o) return set_error(TM_E_INVALID, "no destination for %llu normalized bytes", (unsigned long long)total);
  next = 0;
  run_on_workers(total > (1u << 20) ? threads : 1u, [&]() {
    for (;;) {
      const uint32_t base = next.fetch_add(grab);
      if (base >= ndocs) break;
      for (uint32_t d = base; d < std::min(ndocs, base + grab); d++)
        if (!outs[d].empty()) std::memcpy(o + out_offsets[d], outs[d].data(), outs[d].size());
    }
  });
  return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_normalize(const uint8_t* data, size_t n, uint32_t capcode, uint32_t norm_flag, uint8_t** out, size_t* out_n) {
  if (!out || !out_n || (n && !data)) return tmh::set_error(TM_E_INVALID, "null argument");
  if (!tmh::normalize_supported(capcode, norm_flag))
    return tmh::set_error(TM_E_INVALID, "normalization flags %u / capcode %u not supported by the host normalizer", norm_flag, capcode);
  std::vector<uint8_t> o;
  tmh::normalize_bytes(data, n, capcode, norm_flag, o);
  *out = (uint8_t*)std::malloc(o.size() ? o.size() : 1);
  if (!*out) { *out_n = 0; return tmh::set_error(TM_E_HIP, "out of host memory (%zu bytes)", o.size()); }
  if (!o.empty()) std::memcpy(*out, o.data(), o.size());
  *out_n = o.size();
  return TM_OK;
}

int tm_denormalize(const uint8_t* data, size_t n, uint32_t capcode, uint8_t** out, size_t* out_n) {
  if (!out || !out_n || (n && !data)) return tmh::set_error(TM_E_INVALID, "null argument");
  if (capcode > 2) return tmh::set_error(TM_E_INVALID, "capcode %u", capcode);
  std::vector<uint8_t> o;
  tmh::CapcodeState st;
  if (capcode == 2) tmh::capcode_decode_stream(st, data, n, o);
  else if (capcode == 1) tmh::nocapcode_decode_stream(st, data, n, o);
  else o.assign(data, data + n);
  *out = (uint8_t*)std::malloc(o.size() ? o.size() : 1);
  if (!*out) { *out_n = 0; return tmh::set_error(TM_E_HIP, "out of host memory (%zu bytes)", o.size()); }
  if (!o.empty()) std::memcpy(*out, o.data(), o.size());
  *out_n = o.size();
  return TM_OK;
}

int tm_normalize_batch(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode,
                       uint32_t norm_flag, uint32_t threads, uint8_t** out_text, uint64_t* out_offsets) {
  if (!out_text || !out_offsets || (ndocs && (!text || !offsets))) return tmh::set_error(TM_E_INVALID, "null argument");
  uint8_t* o = nullptr;
  int rc = tmh::normalize_batch_into(text, offsets, ndocs, capcode, norm_flag, threads, out_offsets,
                                     [&](uint64_t total) { o = (uint8_t*)std::malloc(total ? total : 1); return o; });
  if (rc != TM_OK) { std::free(o); return rc; }
  *out_text = o;
  return TM_OK;
}

}  // extern "C"
