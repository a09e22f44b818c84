// tm_formats.hip — the two on-disk formats either side of the path, host only:
//   .vocab  the vocabulary file (go/tokenmonster.go:2602-2653 Save / :2656-2736 Load; layout SURVEY.md Appendix A).  This library never
//           mutates a vocabulary, so Save is "the image it was loaded from": tm_vocab_image hands it back, tm_vocab_save writes it.
//   .tok    the token dictionaries getalltokens writes and trainvocab reads and writes (training/trainvocab.go:412-480; reader with
//           scores training/exportvocab.go:20-60): a zlib stream of  u8 capcode, u8 charset, u8 normalization flag, u8 level, u8 reserve,
//           3 reserved bytes | u64 count | count x { u8 length, bytes } | optionally count x f32 score (= bytes covered / dataset size)
//           | optionally u32 nSpecial + nSpecial x { u8 length, bytes }, all little-endian.  tm_tok_read / tm_tok_write let candidate
//           token sets travel between the reference's tools and tm_build_vocab.
#include "tm_build.h"
#include "tokenmonster_hip.h"
#include "tm_device.h"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <vector>

using namespace tmh;

extern "C" {

int tm_vocab_image(const tm_vocab* v, const uint8_t** image, size_t* n) {
  if (!v || !image || !n) return set_error(TM_E_INVALID, "null argument");
  const std::vector<uint8_t>& img = tmh::vocab_image(v->host);      // (a vocabulary built from a token list writes its image on first request)
  if (img.empty()) return set_error(TM_E_INVALID, "an imported vocabulary (tm_vocab_block_import) has no file image");
  *image = img.data();
  *n = img.size();
  return TM_OK;
}

int tm_vocab_save(const tm_vocab* v, const char* path) {
  if (!v || !path) return set_error(TM_E_INVALID, "null argument");
  const std::vector<uint8_t>& img = tmh::vocab_image(v->host);
  if (img.empty()) return set_error(TM_E_INVALID, "an imported vocabulary (tm_vocab_block_import) has no file image");
  FILE* f = std::fopen(path, "wb");
  if (!f) return set_error(TM_E_INVALID, "cannot open %s for writing", path);
  const size_t n = img.size();
  const bool ok = std::fwrite(img.data(), 1, n, f) == n;
  return (std::fclose(f) == 0 && ok) ? TM_OK : set_error(TM_E_INVALID, "short write to %s", path);
}

int tm_tok_read(const uint8_t* file, size_t n, uint8_t header[5], uint8_t** blob, uint32_t** offsets, uint32_t* count, float** scores,
                uint8_t** special_blob, uint32_t** special_offsets, uint32_t* n_special) {
  if (!file || !header || !blob || !offsets || !count) return set_error(TM_E_INVALID, "null argument");
  *blob = nullptr; *offsets = nullptr; *count = 0;
  if (scores) *scores = nullptr;
  if (special_blob) *special_blob = nullptr;
  if (special_offsets) *special_offsets = nullptr;
  if (n_special) *n_special = 0;
  // inflate everything (dictionaries are a few megabytes)
  std::vector<uint8_t> raw;
  {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return set_error(TM_E_INVALID, "zlib: inflateInit failed");
    zs.next_in = const_cast<Bytef*>(file);
    zs.avail_in = (uInt)n;
    uint8_t buf[1 << 16];
    int zr;
    do {
      zs.next_out = buf;
      zs.avail_out = sizeof buf;
      zr = inflate(&zs, Z_NO_FLUSH);
      if (zr != Z_OK && zr != Z_STREAM_END) { inflateEnd(&zs); return set_error(TM_E_INVALID, "not a zlib stream (.tok files are zlib-compressed)"); }
      raw.insert(raw.end(), buf, buf + (sizeof buf - zs.avail_out));
    } while (zr != Z_STREAM_END);
    inflateEnd(&zs);
  }
  size_t pos = 0;
  auto need = [&](size_t k) { return pos + k <= raw.size(); };
  if (!need(16)) return set_error(TM_E_INVALID, "truncated .tok header");
  std::memcpy(header, raw.data(), 5);
  if (header[1] > 2 || header[3] > 5) return set_error(TM_E_INVALID, "not a token dictionary (charset %u, level %u)", header[1], header[3]);
  uint64_t cnt = 0;
  for (int k = 0; k < 8; k++) cnt |= (uint64_t)raw[8 + k] << (8 * k);
  pos = 16;
  if (cnt > raw.size()) return set_error(TM_E_INVALID, "token count %llu exceeds the file", (unsigned long long)cnt);
  std::vector<uint32_t> off(1, 0);
  std::vector<uint8_t> bl;
  for (uint64_t i = 0; i < cnt; i++) {
    if (!need(1)) return set_error(TM_E_INVALID, "truncated .tok at token %llu", (unsigned long long)i);
    const uint32_t l = raw[pos++];
    if (!need(l)) return set_error(TM_E_INVALID, "truncated .tok at token %llu", (unsigned long long)i);
    bl.insert(bl.end(), raw.begin() + (ptrdiff_t)pos, raw.begin() + (ptrdiff_t)(pos + l));
    pos += l;
    off.push_back((uint32_t)bl.size());
  }
  std::vector<float> sc;
  std::vector<uint32_t> soff(1, 0);
  std::vector<uint8_t> sbl;
  if (pos < raw.size()) {                                   // optional scores, then optional special tokens
    if (!need(4 * cnt)) return set_error(TM_E_INVALID, "truncated .tok scores");
    sc.resize(cnt);
    std::memcpy(sc.data(), raw.data() + pos, 4 * cnt);
    pos += 4 * cnt;
    if (pos < raw.size()) {
      if (!need(4)) return set_error(TM_E_INVALID, "truncated .tok special tokens");
      uint32_t ns = 0;
      for (int k = 0; k < 4; k++) ns |= (uint32_t)raw[pos + k] << (8 * k);
      pos += 4;
      for (uint32_t i = 0; i < ns; i++) {
        if (!need(1)) return set_error(TM_E_INVALID, "truncated .tok special tokens");
        const uint32_t l = raw[pos++];
        if (!need(l)) return set_error(TM_E_INVALID, "truncated .tok special tokens");
        sbl.insert(sbl.end(), raw.begin() + (ptrdiff_t)pos, raw.begin() + (ptrdiff_t)(pos + l));
        pos += l;
        soff.push_back((uint32_t)sbl.size());
      }
    }
  }
  auto give = [](const void* src, size_t bytes) { void* p = std::malloc(bytes ? bytes : 1); if (bytes) std::memcpy(p, src, bytes); return p; };
  *blob = (uint8_t*)give(bl.data(), bl.size());
  *offsets = (uint32_t*)give(off.data(), off.size() * 4);
  *count = (uint32_t)cnt;
  if (scores && !sc.empty()) *scores = (float*)give(sc.data(), sc.size() * 4);
  if (special_blob && special_offsets && n_special && soff.size() > 1) {
    *special_blob = (uint8_t*)give(sbl.data(), sbl.size());
    *special_offsets = (uint32_t*)give(soff.data(), soff.size() * 4);
    *n_special = (uint32_t)soff.size() - 1;
  }
  return TM_OK;
}

int tm_tok_write(const uint8_t header[5], const uint8_t* blob, const uint32_t* offsets, uint32_t count, const float* scores,
                 const uint8_t* special_blob, const uint32_t* special_offsets, uint32_t n_special, uint8_t** out, size_t* out_n) {
  if (!header || (count && (!blob || !offsets)) || !out || !out_n) return set_error(TM_E_INVALID, "null argument");
  if (n_special && !scores) return set_error(TM_E_INVALID, "special tokens can only follow a score section (trainvocab.go:436-447)");
  std::vector<uint8_t> raw(header, header + 5);
  raw.resize(8, 0);
  for (int k = 0; k < 8; k++) raw.push_back((uint8_t)((uint64_t)count >> (8 * k)));
  auto put = [&](const uint8_t* b, const uint32_t* o, uint32_t i) -> bool {
    const uint32_t l = o[i + 1] - o[i];
    if (l > 255) return false;
    raw.push_back((uint8_t)l);
    raw.insert(raw.end(), b + o[i], b + o[i + 1]);
    return true;
  };
  for (uint32_t i = 0; i < count; i++) if (!put(blob, offsets, i)) return set_error(TM_E_INVALID, "token %u longer than 255 bytes", i);
  if (scores) {
    const size_t at = raw.size();
    raw.resize(at + 4 * (size_t)count);
    std::memcpy(raw.data() + at, scores, 4 * (size_t)count);
    if (n_special) {
      for (int k = 0; k < 4; k++) raw.push_back((uint8_t)(n_special >> (8 * k)));
      for (uint32_t i = 0; i < n_special; i++) if (!put(special_blob, special_offsets, i)) return set_error(TM_E_INVALID, "special token %u longer than 255 bytes", i);
    }
  }
  uLongf cap = compressBound((uLong)raw.size());
  uint8_t* z = (uint8_t*)std::malloc(cap ? cap : 1);
  if (compress2(z, &cap, raw.data(), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK) { std::free(z); return set_error(TM_E_INVALID, "zlib: compress failed"); }
  *out = z;
  *out_n = cap;
  return TM_OK;
}

}  // extern "C"
