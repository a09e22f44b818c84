// norm_masks_check.cpp — CPU check of the mask algebra the device normalizer runs (tokenmonster_amd/csrc/tm_norm_masks.h):
// replays k_norm_emit2's per-piece / per-chunk schedule (class masks -> backward sweep -> forward sweep -> bytes) with the very
// same __host__ __device__ functions and compares every document with the host normalizer (tm_normalize).
//   hipcc -O2 -std=c++17 -I include -I tokenmonster_amd/csrc tools/norm_masks_check.cpp -o /tmp/norm_masks_check \
//         -Ltokenmonster_amd -ltokenmonster_hip -ltm_testsupport -Wl,-rpath,$PWD/tokenmonster_amd
// usage: norm_masks_check [random documents] [seed]      exit code 0 = all documents equal
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tm_build.h"
#include "tm_testsupport.h"
#include "tm_internal.h"
#include "tm_norm_masks.h"

using namespace tmh;

namespace {
constexpr int PIECE = 1024, SLAB = 2 * PIECE;

uint32_t ncls_ascii(uint32_t c, bool lower_all) {
  if (c - 'a' < 26u) return NC_L;
  if (c - 'A' < 26u) return lower_all ? NC_L : NC_U;
  if (c - '0' < 10u) return NC_N;
  if (c == '\'') return NC_AP;
  if (c == ' ') return NC_SP;
  return NC_O;
}
static uint64_t g_margin_ok = 0, g_margin_unknown = 0;
static NmTwo g_two[2][NM_TWO_SIZE];     // [lower_all]: flags 1 (NFD) and 3 (NFD + lowercase)
static uint32_t g_blk[2][NM_BLK_WORDS], g_cp[2][NM_CP_WORDS], g_blk4[2][NM_BLK4_WORDS];      // the three- and four-byte characters the pass leaves alone (tm_norm_masks.h)
struct LeaKana { NmLea lea[NM_LEA_SIZE]; uint16_t kana[NM_KANA_SIZE]; uint8_t ccc[NM_CCC_SIZE]; uint32_t dec3[NM_DEC3_SIZE + NM_DEC3_THIRDS]; };      // (the kana entries lie behind those of Latin Extended Additional: nm_kana_tab)
static LeaKana g_lk[2];
#define g_lea_of(k) (g_lk[k].lea)      // Latin Extended Additional under NFD: a letter + one or two marks (round 6)
#define g_kana (g_lk[0].kana)             // the voiced kana under NFD: a kana + U+3099 / U+309A (round 6)
static NmTabs tabs_of(bool lower_all) { const int k = lower_all ? 1 : 0; return NmTabs{g_two[k], g_two[k], g_blk[k], g_cp[k], g_blk4[k], NM_MISC_HANGUL | NM_MISC_LEA | NM_MISC_KANA | NM_MISC_CCC | NM_MISC_DEC3, g_lk[k].lea}; }
// class byte of every byte of a document as norm_load_piece computes it (bytes outside the document read as 0)
bool classify(const std::vector<uint8_t>& d, bool lower_all, std::vector<uint8_t>& f) {
  const int n = (int)d.size();
  auto at = [&](int i) -> uint32_t { return i >= 0 && i < n ? d[i] : 0u; };
  f.assign(n, 0);
  bool ok_all = true;
  // (as norm_load_piece does it: every byte beyond ASCII NF_BAD, then every character once, at its first byte, for all of its bytes)
  for (int i = 0; i < n; i++) f[i] = (uint8_t)(d[i] < 0x80u ? ncls_ascii(d[i], lower_all) : (uint32_t)NF_BAD);
  for (int i = 0; i < n; i++) {
    if ((d[i] & 0xC0u) != 0xC0u) continue;
    const uint32_t r = nm_classify_char(d[i], at(i - 1), at(i - 2), at(i - 3), at(i + 1), at(i + 2), at(i + 3), tabs_of(lower_all));
    if ((r & 0xFFu) == NF_BAD) continue;
    for (int j = 0; j < (int)((r >> 16) & 0xFFu) && i + j < n; j++) f[i + j] = (uint8_t)(j ? (r >> 8) : r);
  }
  for (int i = 0; i < n; i++) if (f[i] == NF_BAD) ok_all = false;
  return ok_all;
}
bool is_block(uint32_t cls) { return (cls & NF_BLOCK) != 0; }

// what k_norm_emit2 does for one piece: returns the bytes
static const NmLut kLut = nm_make_lut();
// (t0_beyond: nm_t0 of what lies behind the 64 bytes of margin after the piece - the exact path's carry; 0 when the margins are all there is: k_norm_emit2<false>)
void emit_piece(const std::vector<uint8_t>& d, const std::vector<uint8_t>& f, int pb, int m, bool w_in, uint64_t t0_beyond, bool lower_all,
                std::vector<uint8_t>& out) {
  const int n = (int)d.size();
  // class byte at piece-relative position rel, as the kernel's LDS holds it: classified from six bytes before the piece to five
  // after its 1024; bytes outside the document read as class O
  auto fat = [&](int rel) -> uint32_t {
    const int p = pb + rel;
    if (rel < -66 || rel > PIECE + 65) { fprintf(stderr, "harness: class byte %d outside the classified LDS range\n", rel); abort(); }
    return (p < 0 || p >= n) ? (uint32_t)NC_O : (uint32_t)f[p];
  };
  auto ballot = [&](int c, auto pred) -> uint64_t {
    uint64_t r = 0;
    for (int i = 0; i < 64; i++) {
      const int rel = 64 * c + i;
      if (rel < -66 || rel > PIECE + 65) continue;
      if (pred(fat(rel))) r |= 1ull << i;
    }
    return r;
  };
  auto is_b = [](uint32_t fl) { return (fl & NF_BLOCK) != 0; };
  auto is_l = [](uint32_t fl) { return (fl & NF_CLASS) == NC_L; };
  auto is_u = [](uint32_t fl) { return (fl & NF_CLASS) == NC_U; };
  auto is_sp = [](uint32_t fl) { return (fl & NF_CLASS) == NC_SP; };
  const int nch = (m + 63) / 64;
  // T of chunk c as the kernel finds it going forwards: from the ballots of the chunk and T0 of the one behind it
  uint64_t TX[18] = {0}, T0N[18] = {0};
  for (int c = 0; c < nch; c++) {
    const uint64_t Bn = ballot(c + 1, is_b);
    uint64_t t0n;
    if (Bn == ~0ull) {
      t0n = t0_beyond & 1ull;
      for (int j = c + 2; j <= 16; j++) { const uint64_t Bj = ballot(j, is_b); if (Bj != ~0ull) { t0n = nm_t0(Bj, ballot(j, is_l), 0ull); break; } }
    } else t0n = nm_t0(Bn, ballot(c + 1, is_l), 0ull);
    T0N[c] = t0n;
    TX[c] = nm_tx(ballot(c, is_b), ballot(c, is_l), t0n);
  }
  uint64_t w = w_in ? 1ull : 0ull;
  uint64_t Ucur = ballot(0, is_u);
  for (int c = 0; c < nch; c++) {
    const uint64_t Unext = ballot(c + 1, is_u);
    uint64_t w_out, spC, spW;
    const uint64_t V = nm_valid(c, m);      // (the kernel lets the lanes behind the end of the document emit their zero byte behind everything else and subtracts them)
    const uint64_t W = nm_inword(ballot(c, is_b), Ucur, ~0ull, w, &w_out);
    nm_space_markers(ballot(c, is_sp), Ucur, Unext, ~0ull, TX[c], T0N[c], &spC, &spW);
    w = w_out;
    for (int i = 0; i < 64; i++) {
      const uint64_t bit = 1ull << i;
      if (!(V & bit)) continue;
      const int rel = 64 * c + i;
      const uint32_t fl = fat(rel), fp = fat(rel - 1), f2 = fat(rel - 2), f4 = fat(rel - 4);
      const uint32_t idx = nm_lut_index((fl & NF_CONT) ? (uint32_t)NC_O : fl, fp, (fp & NF_CONT) ? f4 : f2, (uint32_t)((W >> i) & 1ull), (uint32_t)((TX[c] >> i) & 1ull));
      const uint32_t code = kLut.e[lower_all ? 1 : 0][idx];
      uint32_t len = (code & 3u) + 1u;
      const uint32_t b = d[pb + rel];
      uint32_t o3 = b | ((code & 4u) << 3), ysp = ' ', m3 = code >> 8;
      if (spC & bit) o3 = 'C';
      if (spW & bit) o3 = 'W';
      auto rawat = [&](int r) -> uint32_t { const int p = pb + r; return (p < 0 || p >= n) ? 0u : d[p]; };
      const uint32_t bm1 = rawat(rel - 1), bp1 = rawat(rel + 1);
      const bool lead2 = nm_two_lead(b), cont2 = nm_cont_byte(b) && nm_two_lead(bm1);
      if (lead2 || cont2) {
        const NmTwo e = g_two[lower_all ? 1 : 0][lead2 ? nm_two_index(b, bp1) : nm_two_index(bm1, b)];
        uint32_t y = 0, mm = 0;
        const uint32_t extra = nm_two_out(e, cont2, (code & 4u) != 0, true, &o3, &y, &mm);
        if (extra >= 1u) { len = 1u + extra; ysp = y; m3 = mm; }
      }
      uint32_t hrole, hcp;
      if (fl != NF_BAD && nm_hangul_role(b, bm1, rawat(rel - 2), bp1, rawat(rel + 2), &hrole, &hcp)) { len = nm_hangul_out(hcp, hrole, &m3, &ysp, &o3); if (len == 0) continue; }
      uint32_t lrole, lidx;
      if (fl != NF_BAD && nm_lea_role(b, bm1, rawat(rel - 2), bp1, rawat(rel + 2), &lrole, &lidx) && (g_lk[lower_all ? 1 : 0].lea[lidx].a & NT_OK)) {
        const NmLea le = g_lk[lower_all ? 1 : 0].lea[lidx];
        if (lrole == 0u) o3 = (code & 4u) ? ((le.a >> 16) & 0xFFu) : ((le.a >> 8) & 0xFFu);
        else if (lrole == 1u) { len = 2u; ysp = le.b & 0xFFu; o3 = (le.b >> 8) & 0xFFu; }
        else if (((le.a >> 24) & 3u) == 2u) { len = 2u; ysp = (le.b >> 16) & 0xFFu; o3 = le.b >> 24; }
        else continue;
      }
      uint32_t krole, kidx;
      if (fl != NF_BAD && nm_kana_role(b, bm1, rawat(rel - 2), bp1, rawat(rel + 2), &krole, &kidx) && (g_kana[kidx] & NK_OK)) { len = nm_kana_out(g_kana[kidx], krole, &m3, &ysp, &o3); if (len == 0) continue; }
      uint32_t drole, dcp;
      if (fl != NF_BAD && nm_three_role(b, bm1, rawat(rel - 2), bp1, rawat(rel + 2), &drole, &dcp) && (nm_dec3(tabs_of(lower_all), dcp) & ND_OK)) { len = nm_dec3_out(tabs_of(lower_all), nm_dec3(tabs_of(lower_all), dcp), drole, &m3, &ysp, &o3); if (len == 0) continue; }
      if (len == 4) out.push_back('D');
      if (len >= 3) out.push_back((uint8_t)m3);
      if (len >= 2) out.push_back((uint8_t)ysp);
      out.push_back((uint8_t)o3);
    }
    Ucur = Unext;
  }
}

bool check_doc(const std::vector<uint8_t>& d, uint32_t norm_flag, uint64_t* skipped) {
  const bool lower_all = (norm_flag & 2u) != 0;
  std::vector<uint8_t> f;
  if (!classify(d, lower_all, f)) { (*skipped)++; return true; }     // host-fallback document on the device too
  const int n = (int)d.size();
  std::vector<uint8_t> got;
  for (int pb = 0; pb < n; pb += PIECE) {
    const int m = std::min(PIECE, n - pb);
    bool w_in = false;
    for (int j = pb - 1; j >= 0 && is_block(f[j] & 7u); j--) if ((f[j] & 7u) == NC_U) { w_in = true; break; }
    bool carry_tl = false;
    {
      int j = pb + m;
      while (j < n && is_block(f[j] & 7u)) j++;
      carry_tl = j < n && (f[j] & 7u) == NC_L;
    }
    // the carries as k_norm_emit2<false> finds them: from the 64 bytes either side of the piece; when they cannot tell, the exact path
    // (the document-wide carries above) runs on the device too
    auto cls_at = [&](int p) -> uint32_t { return (p < 0 || p >= n) ? (uint32_t)NC_O : (uint32_t)f[p]; };
    uint64_t Bb = 0, Ub = 0, Ba = 0;
    for (int i = 0; i < 64; i++) {
      const uint32_t fb = cls_at(pb - 64 + i), fa = m == PIECE ? cls_at(pb + PIECE + i) : 0u;
      if (fb & NF_BLOCK) Bb |= 1ull << i;
      if ((fb & NF_CLASS) == NC_U) Ub |= 1ull << i;
      if (fa & NF_BLOCK) Ba |= 1ull << i;
    }
    uint64_t w_m;
    if (nm_margin_carries(Bb, Ub, Ba, &w_m)) {
      g_margin_ok++;
      if ((w_m != 0) != w_in) { fprintf(stderr, "MARGIN: inWord seed %d, the document says %d (piece at %d)\n", (int)w_m, (int)w_in, pb); return false; }
      emit_piece(d, f, pb, m, w_m != 0, 0ull, lower_all, got);
    } else {
      g_margin_unknown++;
      emit_piece(d, f, pb, m, w_in, carry_tl ? 1ull : 0ull, lower_all, got);
    }
  }
  uint8_t* exp = nullptr; size_t exp_n = 0;
  if (tm_normalize(d.data(), d.size(), 2, norm_flag, &exp, &exp_n) != 0) { fprintf(stderr, "tm_normalize failed\n"); return false; }
  const bool same = exp_n == got.size() && (exp_n == 0 || memcmp(exp, got.data(), exp_n) == 0);
  if (!same) {
    size_t k = 0;
    while (k < exp_n && k < got.size() && exp[k] == got[k]) k++;
    fprintf(stderr, "MISMATCH (flag %u, %zu bytes): expected %zu bytes, got %zu; first difference at output byte %zu\n  input : %.*s\n", norm_flag, d.size(),
            exp_n, got.size(), k, (int)std::min<size_t>(d.size(), 200), (const char*)d.data());
    if (const char* dump = getenv("NM_DUMP")) { FILE* f = fopen(dump, "wb"); if (f) { fwrite(d.data(), 1, d.size(), f); fclose(f); } }      // the document, for a closer look
    const size_t a = k > 20 ? k - 20 : 0;
    fprintf(stderr, "  expect: ...%.*s\n  got   : ...%.*s\n", (int)std::min<size_t>(exp_n - a, 60), (const char*)exp + a, (int)std::min<size_t>(got.size() - a, 60),
            (const char*)got.data() + a);
  }
  tm_free(exp);
  return same;
}
}  // namespace

int main(int argc, char** argv) {
  const int ndocs = argc > 1 ? atoi(argv[1]) : 20000;
  const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
  Rng rng(seed);
  uint64_t bad = 0, skipped = 0, total = 0;
  const char* ascii = "aBcDeFGhijKLMnop XYZ  '''1234567890.,-_()\n\tQ";
  const char* multi[] = {"\xE2\x80\x99", "\xE2\x80\x9C", "\xE2\x80\x9D", "\xE2\x80\x94", "\xE2\x80\xA6", "\xE2\x81\x80"};
  build_two_table(1, g_two[0]);
  build_two_table(3, g_two[1]);
  build_three_tables(1, g_blk[0], g_cp[0]);
  build_three_tables(3, g_blk[1], g_cp[1]);
  build_four_table(1, g_blk4[0]);
  build_four_table(3, g_blk4[1]);
  build_lea_table(1, g_lk[0].lea);
  build_lea_table(3, g_lk[1].lea);
  build_kana_table(g_lk[0].kana); build_kana_table(g_lk[1].kana);
  build_ccc_table(1, true, g_lk[0].ccc); build_ccc_table(3, true, g_lk[1].ccc);
  build_dec3_table(g_lk[0].dec3); build_dec3_table(g_lk[1].dec3);
  { int n = 0; for (uint32_t k = 0; k < NM_DEC3_SIZE; k++) n += (g_lk[0].dec3[k] & ND_OK) != 0; int n3 = 0; for (uint32_t k = 0; k < NM_DEC3_SIZE; k++) n3 += ((g_lk[0].dec3[k] >> 26) & 15u) != 0; printf("three-byte characters NFD splits in two or three, on the device: %d (%d of them in three)\n", n, n3); }
  { int n = 0; for (uint32_t k = 0; k < NM_CCC_SIZE; k++) n += g_lk[0].ccc[k] != 0 && g_lk[0].ccc[k] < NM_CCC_LOWER; printf("three-byte marks of canonical class > 0 in U+0800..U+1FFF on the device: %d\n", n); }
  { int ok = 0; for (int k = 0; k < NM_KANA_SIZE; k++) ok += (g_kana[k] & NK_OK) != 0; printf("voiced kana (NFD): %d characters of U+3040..U+30FF on the device\n", ok); }
  { int ok = 0, two = 0; for (int k = 0; k < NM_LEA_SIZE; k++) { ok += (g_lk[0].lea[k].a & NT_OK) != 0; two += (g_lk[0].lea[k].a & NT_OK) && ((g_lk[0].lea[k].a >> 24) & 3u) == 2u; }
    printf("Latin Extended Additional (NFD): %d of %d characters on the device, %d of them with two marks\n", ok, NM_LEA_SIZE, two); }
  { int ok = 0, dec = 0, dec2 = 0; for (int k = 0; k < NM_TWO_SIZE; k++) { ok += (g_two[0][k].a & NT_OK) != 0; dec += (g_two[0][k].a & NT_DECOMP) != 0; dec2 += (g_two[0][k].a & NT_DECOMP2) != 0; }
    int c1 = 0, c2 = 0, mixed = 0;
    for (uint32_t cp = 0x800; cp < 0x10000; cp++) { const uint32_t c = (g_cp[0][cp >> 4] >> (2 * (cp & 15))) & 3u; c1 += c == 1; c2 += c == 2; }
    for (uint32_t b = 0; b < 1024; b++) mixed += ((g_blk[0][b >> 4] >> (2 * (b & 15))) & 3u) == 3u;
    printf("two-byte table (NFD): %d of %d characters on the device, %d decompose into an ASCII letter + mark, %d into a two-byte letter + mark\n", ok, NM_TWO_SIZE, dec, dec2);
    { int b1 = 0, b2 = 0; for (uint32_t b = 0; b < 16384; b++) { const uint32_t c = (g_blk4[0][b >> 4] >> (2 * (b & 15))) & 3u; b1 += c == 1; b2 += c == 2; }
      printf("four-byte characters: %d of 16384 blocks of 64 code points pass as class O, %d as letters without case\n", b1, b2); }
    printf("three-byte characters left alone on the device: %d class O, %d letters without case; %d of 1024 blocks are mixed (per code point)\n", c1, c2, mixed); }
  std::vector<std::string> fixed = {"", "A", "a", "AB", "Ab", "aB", "ABc", "ABC", " ABC d", "HTTPServer2Go x", "X's Y'S it's 'a' I'M", "12AB34cd", "A1B2c",
                                    "X\xE2\x80\x99s Y\xE2\x80\x99S it\xE2\x80\x99s", std::string(200, 'A') + "b", std::string(200, 'A'),
                                    "a" + std::string(130, 'B') + " " + std::string(70, 'C') + "d", std::string(3000, 'A') + "b", std::string(5000, 'Q'),
                                    std::string(1023, 'x') + " Abc", std::string(1023, 'x') + "A" + "bc", std::string(1022, 'x') + " A" + std::string(1100, 'B') + "c",
                                    std::string(1024, 'A') + std::string(1024, '1') + "z", std::string(1020, ' ') + "AB'\xE2\x80\x99" + "cD",
                                    "\xC3\x89t\xC3\xA9 \xC3\x80 la carte", "\xC3\x89\xC3\x89\xC3\x89 x \xC3\x89\xC3\x89" "b", "na\xC3\xAFve caf\xC3\xA9's \xC3\x86on \xC3\x98L", "stra\xC3\x9F" "e \xC2\xAB" "a\xC2\xBB 1\xC2\xBA 2\xC2\xAA",
                                    "\xC5\x81\xC3\xB3" "d\xC5\xBA \xC4\x8C\xC4\x8D" "SR \xC4\xB0stanbul \xC4\xB1\xC5\xBF", std::string(1023, 'x') + "\xC3\x89" "b", std::string(1022, 'x') + " \xC3\x89" + std::string(40, 'A') + "c",
                                    std::string(63, 'a') + "\xC3\xA9\xC3\xA9", "l'\xC3\xA9t\xC3\xA9 d'\xC3\x89" "mile 3\xC3\xA8me \xC3\xA9's",
                                    u8"Привет, Мир! Ёжик и йод. МОСКВА Санкт-Петербург", u8"Καλημέρα κόσμε. ΑΘΗΝΑ Ελλάδα ά έ ή ί ό ύ ώ ΐ", u8"שלום עולם בְּרֵאשִׁית", u8"مرحبا بالعالم ١٢٣ كِتَاب",
                                    u8"中文文本，测试。Hello世界 ABC中文", u8"こんにちは世界 カタカナ がぎぐ パピプ", u8"한국어 텍스트", u8"가 각 힣 뷁 A가B 가a 1가 '가' 한글Hangul 가\u0301", std::string(1022, 'x') + u8"한국", std::string(1023, 'x') + u8"각", std::string(400, 'x') + std::string(u8"한국어텍스트가나다라마바사") + std::string(u8"아자차카타파하") + std::string(600, 'y'), u8"a\u0301 e\u0301\u0323 o\u0323\u0301 Ắ ǖ", u8"→ ★ ∑ √ ①②③ Ḁḁ ẞ",
                                    std::string(1023, 'x') + u8"й", std::string(1022, 'x') + u8"Йод", std::string(62, 'a') + u8"йй" + std::string(61, 'b') + u8"中文",
                                    u8"Hello 😀 World 🌍🚀 it's 👍🏽 A😀B c😀d 1😀2 '😀' 𝒜𝒷 𠀀𠀁 done", std::string(1021, 'x') + u8"😀Ab", std::string(1022, 'x') + u8"😀" + " Ab", std::string(1023, 'x') + u8"A😀b", std::string(61, 'A') + u8"😀😀" + std::string(70, 'b'),
                                    u8"𐐀𐐨 Deseret", u8"𝅗𝅥 half note", "\xF0\x9F\x98", "\xF4\x90\x80\x80 beyond", "\xF0\x80\x80\x80 overlong", u8"x😀", u8"I ❤️ U ☺️ 1️⃣ A️b a️B ❤️️", u8"की कि कु हिन्दी HINDI ह",
                                    u8"Việt Nam: tiếng Việt, Hà Nội và Thành phố Hồ Chí Minh. ĐƯỜNG Nguyễn Huệ, PHỞ bò; Ắ ắ Ế ế Ộ ộ Ự ự Ỹ ỹ", u8"TIẾNG VIỆT viết HOA và Thường; ớt's Ớt'S 1ế2 'ệ' Ḁḁ ẛ ẞ ỿ ế\u0301 e\u0302\u0301",
                                    std::string(1022, 'x') + u8"ếệ", std::string(1023, 'x') + u8"Ế" + "b", std::string(1021, 'x') + u8" Ệ" + std::string(40, 'A') + "c", std::string(61, 'A') + u8"ỆỆ" + std::string(70, 'b')};
  for (int lower = 0; lower < 2; lower++) {
    const uint32_t flag = lower ? 3u : 1u;
    for (const auto& s : fixed) { std::vector<uint8_t> d(s.begin(), s.end()); total++; if (!check_doc(d, flag, &skipped)) bad++; }
    for (int k = 0; k < ndocs; k++) {
      // lengths cluster around the piece and chunk boundaries
      const uint32_t mode = rng.below(4);
      size_t len = mode == 0 ? rng.below(200) : mode == 1 ? 1024u * (1 + rng.below(3)) - 40 + rng.below(80) : mode == 2 ? 64u * (1 + rng.below(40)) - 4 + rng.below(8) : rng.below(5000);
      std::vector<uint8_t> d;
      const uint32_t style = rng.below(5);      // 0 mixed, 1 capitals-heavy, 2 digits/apostrophes-heavy, 3 spaces + capitals, 4 long runs
      const bool latin = rng.below(2) != 0;     // half of the documents carry accented Latin letters, a few of them a lot
      // a third of the documents are written in another script: Greek, Cyrillic (with the letters that decompose: й ё ά ...), Hebrew, Arabic,
      // standalone combining marks, Chinese, Japanese (with voiced kana: a kana and its mark), Korean (Hangul syllables decompose by arithmetic), symbols, four-byte characters
      const uint32_t script = rng.below(3) == 0 ? 1 + rng.below(11) : 0;
      const uint32_t latin_share = rng.below(4) == 0 ? 40 : 6;
      while (d.size() < len) {
        const uint32_t r = rng.below(100);
        if (style == 4 && r < 30) { const char ch = "AB1'a "[rng.below(6)]; const uint32_t rep = 1 + rng.below(150); for (uint32_t q = 0; q < rep; q++) d.push_back((uint8_t)ch); continue; }
        if (r < 4) { const char* mchar = multi[rng.below(6)]; d.insert(d.end(), mchar, mchar + 3); continue; }
        if (r == 4 && rng.below(3) == 0) { const uint32_t e = 0x1F600 + rng.below(0x50); const uint8_t b4[4] = {0xF0, (uint8_t)(0x80 | ((e >> 12) & 0x3F)), (uint8_t)(0x80 | ((e >> 6) & 0x3F)), (uint8_t)(0x80 | (e & 0x3F))}; d.insert(d.end(), b4, b4 + 4); continue; }   // an emoticon anywhere
        if (script && r < 60) {
          uint32_t cp = 0;
          switch (script) {
            case 1: cp = 0x0386 + rng.below(0x48); break;                                   // Greek
            case 2: cp = 0x0400 + rng.below(0x60); break;                                   // Cyrillic
            case 3: cp = 0x05D0 + rng.below(0x1B); if (rng.below(8) == 0) cp = 0x05B0 + rng.below(0x10); break;   // Hebrew letters, now and then a point
            case 4: cp = 0x0621 + rng.below(0x2A); if (rng.below(8) == 0) cp = 0x064B + rng.below(8); if (rng.below(10) == 0) cp = 0x0660 + rng.below(10); break;   // Arabic, marks, digits
            case 5: cp = rng.below(4) ? 0x4E00 + rng.below(0x5000) : 0x3000 + rng.below(0x40); break;   // Chinese + CJK punctuation
            case 6: cp = rng.below(3) ? 0x3041 + rng.below(0x56) : (rng.below(2) ? 0x30A1 + rng.below(0x5E) : 0x4E00 + rng.below(0x5000));      // Japanese
                    if (rng.below(14) == 0) cp = rng.below(2) ? 0x3099 + rng.below(2) : 0x0300 + rng.below(4); break;   // ... now and then a voicing mark by itself or a Latin one (the host's behind a voiced kana)
            case 7: cp = rng.below(6) ? 0x0180 + rng.below(0x680) : 0x0300 + rng.below(0x70); break;   // anything two-byte, and stray combining marks
            case 10: cp = 0x0900 + rng.below(0x80); if (rng.below(5) == 0) cp = rng.below(2) ? 0x094D : (rng.below(2) ? 0x093C : 0x0951 + rng.below(4));      // Devanagari: virama 9, nukta 7, accents 230 / 220 - in and out of order, the letters that decompose
                     if (rng.below(40) == 0) cp = 0x0300 + rng.below(0x30); if (rng.below(30) == 0) cp = 0x0980 + rng.below(0x80); break;             // ... a Latin mark, Bengali (two-part vowels decompose)
            case 11: cp = 0x0E01 + rng.below(0x3A); if (rng.below(4) == 0) cp = rng.below(2) ? 0x0E38 + rng.below(3) : 0x0E48 + rng.below(4);                // Thai: vowels below 103, tone marks 107 (อยู่: 103 then 107, in order)
                     if (rng.below(30) == 0) cp = 0x0EB8 + rng.below(2); if (rng.below(30) == 0) cp = 0x0F71 + rng.below(0x14); if (rng.below(30) == 0) cp = 0x1037 + rng.below(4);
                     if (rng.below(12) == 0) cp = rng.below(8) ? 0x10D0 + rng.below(0x2B) : (rng.below(2) ? 0x1D00 + rng.below(0x80) : 0x1C90 + rng.below(0x2B)); break;   // ... Lao, Tibetan, Myanmar marks; Georgian (lower-case letters of three bytes), phonetic extensions, Mtavruli capitals (the host's)
            case 9: {                                                                        // four bytes: emoji and pictographs, plane-2 ideographs, mathematical letters; now and then what the host has to do
              const uint32_t q = rng.below(40);
              cp = q < 20 ? 0x1F300 + rng.below(0x700) : q < 28 ? 0x20000 + rng.below(0xA000) : q < 34 ? 0x1D400 + rng.below(0x400) : q < 36 ? 0x10000 + rng.below(0x100) :
                   q == 36 ? 0x10400 + rng.below(0x50) /* Deseret: case */ : q == 37 ? 0x1D15E + rng.below(7) /* musical symbols that decompose */ : q == 38 ? 0x1F100 + rng.below(0x10) : 0x10000 + rng.below(0x100000);
              break; }
            default: if (rng.below(6) == 0) { cp = rng.below(2) ? 0xFE0F : (rng.below(2) ? 0x20E3 : 0x093E + rng.below(0x10)); break; }      // three-byte marks: variation selector, keycap, Devanagari vowel signs (some of class 0, some not)
                     cp = rng.below(3) ? 0x2190 + rng.below(0x400) : (rng.below(2) ? 0xAC00 + rng.below(0x2BA4) : 0x1E00 + rng.below(0x100)); break;   // arrows / symbols; Hangul, Latin Extended Additional (host)
          }
          if (cp < 0x800) { d.push_back((uint8_t)(0xC0 | (cp >> 6))); d.push_back((uint8_t)(0x80 | (cp & 0x3F))); }
          else if (cp >= 0x10000) {
            d.push_back((uint8_t)(0xF0 | (cp >> 18))); d.push_back((uint8_t)(0x80 | ((cp >> 12) & 0x3F)));
            if (rng.below(300)) { d.push_back((uint8_t)(0x80 | ((cp >> 6) & 0x3F))); if (rng.below(300)) d.push_back((uint8_t)(0x80 | (cp & 0x3F))); }      // (now and then cut short)
          }
          else { d.push_back((uint8_t)(0xE0 | (cp >> 12))); d.push_back((uint8_t)(0x80 | ((cp >> 6) & 0x3F))); if (rng.below(400)) d.push_back((uint8_t)(0x80 | (cp & 0x3F))); }
          if (rng.below(5) == 0) d.push_back(' ');
          continue;
        }
        if (latin && r < 4 + latin_share) {      // a character of U+00A0..U+017F (now and then an unsupported or broken one: the document then takes the host path)
          const uint32_t cp = rng.below(40) == 0 ? 0x80 + rng.below(0x100) : (rng.below(3) ? 0xC0 + rng.below(0x40) : 0xA0 + rng.below(0xE0));
          d.push_back((uint8_t)(0xC0 | (cp >> 6))); if (rng.below(300)) d.push_back((uint8_t)(0x80 | (cp & 0x3F)));
          continue;
        }
        if (style == 1 && r < 60) { d.push_back((uint8_t)('A' + rng.below(26))); continue; }
        if (style == 2 && r < 60) { d.push_back((uint8_t)("0123456789''"[rng.below(12)])); continue; }
        if (style == 3 && r < 50) { d.push_back(rng.below(2) ? ' ' : (uint8_t)('A' + rng.below(26))); continue; }
        d.push_back((uint8_t)ascii[rng.below((uint32_t)strlen(ascii))]);
      }
      total++;
      if (!check_doc(d, flag, &skipped)) { bad++; if (bad > 5) break; }
    }
  }
  // documents of the synthetic bench corpus
  {
    std::vector<uint8_t> raw((4u << 20) + 70000);
    std::vector<uint64_t> off(70000);
    uint32_t nd = 0; uint64_t nb = 0;
    tm_synth_corpus(TM_KIND_ENGLISHCODE, 99, 4u << 20, 2048, raw.data(), off.data(), (uint32_t)off.size() - 1, &nd, &nb);
    for (uint32_t k = 0; k < nd && bad <= 5; k++) {
      std::vector<uint8_t> d(raw.begin() + off[k], raw.begin() + off[k + 1]);
      total++;
      if (!check_doc(d, 1, &skipped)) bad++;
    }
  }
  printf("pieces whose carries came from their margins: %llu; margins could not tell (exact path): %llu\n", (unsigned long long)g_margin_ok, (unsigned long long)g_margin_unknown);
  printf("%llu documents, %llu skipped (host-fallback class), %llu mismatches\n", (unsigned long long)total, (unsigned long long)skipped, (unsigned long long)bad);
  return bad ? 1 : 0;
}
