/* tm_build.h — host-side vocabulary construction for libtokenmonster_hip.so.
 *
 * tm_build_vocab restates the per-token metadata rules of the reference's vocabulary builder,
 * go/tokenmonster.go:3423-3793 (identical to training/trainvocab.go:548-907): "D "-duplicates,
 * flag bits, nWords, the two prioritised alternatives, beginByte, deleteToken; and writes the
 * result in the .vocab layout of go/tokenmonster.go:2602-2653.
 * It serves two purposes: (1) the per-candidate table build of the trainvocab loop,
 * (2) minting vocabularies of the BASELINE.json shapes, because no pretrained .vocab exists in the
 * reference tree or in this image.
 *
 * tm_synth_* are deterministic generators of synthetic lexicons, raw corpora and vocabularies of
 * the named shapes (english / englishcode / code).  tm_normalize is the host-side pre-step of
 * Tokenize (go/tokenmonster.go:233-253: norm.Normalize then capcode.Encode).
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

/* kinds of synthetic text */
#define TM_KIND_ENGLISH 0      /* prose only                     */
#define TM_KIND_ENGLISHCODE 1  /* 60% prose, 30% code, 10% log/JSON */
#define TM_KIND_CODE 2         /* code only                      */

/* Raw (un-normalized) synthetic corpus of about `nbytes` bytes split into documents whose lengths are
 * log-normal with the given median, clipped to [64, 65536].  text_out must hold nbytes + 65536 bytes,
 * offsets_out max_docs+1 entries.  Deterministic in (kind, seed). */
int tm_synth_corpus(uint32_t kind, uint64_t seed, uint64_t nbytes, uint32_t median_doc, uint8_t* text_out,
                    uint64_t* offsets_out, uint32_t max_docs, uint32_t* ndocs_out, uint64_t* nbytes_out);

/* Synthetic vocabulary of `vocab_size` IDs for text of `kind`: candidate substrings are counted on a
 * normalized sample of the corresponding synthetic corpus and the best vocab_size-|singles| are kept,
 * then passed through tm_build_vocab.  capcode in {0,2}; norm_flag as in the .vocab header. */
int tm_synth_vocab(uint32_t kind, uint32_t vocab_size, uint32_t capcode, uint32_t norm_flag, uint32_t level,
                   uint64_t seed, int with_unk, uint8_t** out, size_t* out_n);

/* Host-side normalize + capcode (go/tokenmonster.go:242-253).  Returns a malloc'd buffer. */
int tm_normalize(const uint8_t* data, size_t n, uint32_t capcode, uint32_t norm_flag, uint8_t** out,
                 size_t* out_n);
/* Batch form: normalizes each document independently on `threads` host threads and re-packs into a
 * malloc'd buffer (*out_text, free with tm_free); out_offsets has ndocs+1 entries (caller-allocated).
 * Supported in this round: norm_flag in {0, 1 (NFD), 2|1 (lowercase)}, capcode in {0, 2}; anything
 * else fails with TM_E_INVALID rather than producing approximately-right bytes. */
int tm_normalize_batch(const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, uint32_t capcode,
                       uint32_t norm_flag, uint32_t threads, uint8_t** out_text, uint64_t* out_offsets);

#ifdef __cplusplus
}
#endif
#endif
