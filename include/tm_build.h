/* tm_build.h — host-side vocabulary construction for libtokenmonster_hip.so.
 *
 * tm_build_vocab restates the per-token metadata rules of the reference's vocabulary builder,
 * go/tokenmonster.go:3423-3793 (identical to training/trainvocab.go:548-907): "D "-duplicates,
 * flag bits, nWords, the two prioritised alternatives, beginByte, deleteToken; and writes the
 * result in the .vocab layout of go/tokenmonster.go:2602-2653.
 * It serves two purposes: (1) the per-candidate table build of the trainvocab loop,
 * (2) minting vocabularies for tests and benchmarks (include/tm_testsupport.h), because no pretrained
 * .vocab exists in the reference tree or in this image.
 *
 * tm_normalize is the host-side pre-step of Tokenize (go/tokenmonster.go:233-253: norm.Normalize
 * then capcode.Encode).
 */
#ifndef TM_BUILD_H
#define TM_BUILD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

void tm_free(void* p);

/* tokens: n_tokens byte strings, token k = blob[off[k] .. off[k+1]).  Duplicates and empty strings
 * are ignored; strings longer than 40 bytes are an error.  special[k] != 0 marks a special token
 * (flag 64, no alternatives, go :3504-3511).  single bytes must be included by the caller.
 * Returns TM_OK and a malloc'd .vocab image (free with tm_free). */
int tm_build_vocab(const uint8_t* blob, const uint32_t* off, uint32_t n_tokens, const uint8_t* special,
                   uint32_t capcode, uint32_t charset, uint32_t norm_flag, uint32_t level, int with_unk,
                   uint8_t** out, size_t* out_n);

/* The trainvocab worker's per-candidate step in ONE call (training/trainvocab.go:530-907 builds a candidate's tables in place): the same
 * token list -> the same rules -> records -> walk tables -> `device`, without the .vocab image in between that tm_build_vocab writes and
 * tm_vocab_load parses again (the search for a token's alternatives rides on the trie the tables need anyway).  The vocabulary is the one
 * tm_vocab_load(tm_build_vocab(...)) gives - tm_vocab_image() of it returns those very bytes (written on first request). */
struct tm_vocab;
int tm_vocab_build(const uint8_t* blob, const uint32_t* off, uint32_t n_tokens, const uint8_t* special, uint32_t capcode, uint32_t charset,
                   uint32_t norm_flag, uint32_t level, int with_unk, int device, struct tm_vocab** out);
/* ... and for every member of a tm_devices handle (tokenmonster_hip.h): built once, the device block replicated GPU to GPU. */
struct tm_devices;
struct tm_vocab_set;
int tm_vocab_build_all(struct tm_devices* g, const uint8_t* blob, const uint32_t* off, uint32_t n_tokens, const uint8_t* special, uint32_t capcode,
                       uint32_t charset, uint32_t norm_flag, uint32_t level, int with_unk, struct tm_vocab_set** out);

/* Host-side normalize + capcode (go/tokenmonster.go:242-253).  Returns a malloc'd buffer. */
int tm_normalize(const uint8_t* data, size_t n, uint32_t capcode, uint32_t norm_flag, uint8_t** out,
                 size_t* out_n);
/* The way back for ONE byte string (a token, a decoded fragment): capcode decoding as Vocab.Denormalize does it (go/tokenmonster.go:445-462;
 * tokenmonster-cpp/src/tokenmonster.cpp:3248-3253) - level 2 by the decoder of javascript/tokenmonster.js:1007-1065, level 1 drops the 0x7F
 * marker and the character behind it, level 0 copies.  Host code (token lists are small); documents go through tm_decode_batch.  Returns a
 * malloc'd buffer (tm_free). */
int tm_denormalize(const uint8_t* data, size_t n, uint32_t capcode, uint8_t** out, size_t* out_n);
/* Batch form: normalizes each document independently on `threads` host threads and re-packs into a
 * malloc'd buffer (*out_text, free with tm_free); out_offsets has ndocs+1 entries (caller-allocated).
 * Every normalizer flag bit is implemented (tokenmonster-cpp/src/tokenmonster.cpp:428-475: 1 NFD, 2 lowercase, 4 accents,
 * 8 quotemarks, 16 collapse, 32 trim, 64 leadingspace, 128 unixlines); capcode in {0, 2} — level 1 has no statement in the
 * reference tree and fails with TM_E_INVALID rather than producing approximately-right bytes. */
int tm_normalize_batch(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode,
                       uint32_t norm_flag, uint32_t threads, uint8_t** out_text, uint64_t* out_offsets);

/* ---- on-disk formats either side of the path (host only) --------------------------------------------------------------------------
 * .vocab (go/tokenmonster.go:2602-2653 Save): this library never mutates a vocabulary, so saving one is writing back the image it was
 * loaded from.  *image stays valid until tm_vocab_free. */
int tm_vocab_image(const struct tm_vocab* v, const uint8_t** image, size_t* n);
int tm_vocab_save(const struct tm_vocab* v, const char* path);
/* .tok token dictionaries (training/trainvocab.go:412-480: what getalltokens writes and trainvocab reads and writes): a zlib stream of
 * header[5] = {capcode, charset, normalization flag, level, reserve} + 3 reserved bytes, u64 count, count x {u8 length, bytes}, optionally
 * count x f32 score, optionally u32 nSpecial + nSpecial x {u8 length, bytes}.  tm_tok_read inflates and parses a whole file: tokens as
 * blob + offsets[count+1] (the layout tm_build_vocab takes), *scores NULL when the file has none; every returned buffer is malloc'd
 * (tm_free).  tm_tok_write produces the file bytes (scores / special tokens optional, NULL / 0). */
int tm_tok_read(const uint8_t* file, size_t n, uint8_t header[5], uint8_t** blob, uint32_t** offsets, uint32_t* count, float** scores,
                uint8_t** special_blob, uint32_t** special_offsets, uint32_t* n_special);
int tm_tok_write(const uint8_t header[5], const uint8_t* blob, const uint32_t* offsets, uint32_t count, const float* scores,
                 const uint8_t* special_blob, const uint32_t* special_offsets, uint32_t n_special, uint8_t** out, size_t* out_n);

#ifdef __cplusplus
}
#endif
#endif
