/* tm_testsupport.h — TEST / BENCHMARK SUPPORT, not part of the product library.
 *
 * libtm_testsupport.so (tokenmonster_amd/testsupport/tm_synth.cpp) holds the deterministic generators of synthetic
 * lexicons, raw corpora and vocabularies of the BASELINE.json shapes (english / englishcode / code).  No pretrained
 * .vocab and no dataset exists in the reference tree or in this image, so tests, tools and bench.py mint their
 * inputs here.  It links against libtokenmonster_hip.so for the vocabulary builder and the host normalizer.
 */
#ifndef TM_TESTSUPPORT_H
#define TM_TESTSUPPORT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

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

#ifdef __cplusplus
}
#endif
#endif
