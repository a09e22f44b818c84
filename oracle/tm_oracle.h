/* oracle/tm_oracle.h — TEST INFRASTRUCTURE ONLY.  CPU restatement of TokenMonster's ungreedy
 * tokenization path in plain C.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product library never does.  See tm_oracle.c for the reference
 * file:line each function follows and for how the restatement is pinned. */
#ifndef TM_ORACLE_H
#define TM_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TMO_NONE 0xFFFFFFu /* go/tokenmonster.go:32 DOES_NOT_EXIST */

typedef struct tmo_vocab tmo_vocab;

/* go/tokenmonster.go:2656-2736 (Load), layout SURVEY.md Appendix A */
tmo_vocab* tmo_load(const uint8_t* file, size_t n);
void tmo_free(tmo_vocab* v);
const char* tmo_last_error(void);

uint32_t tmo_vocab_size(const tmo_vocab* v);
uint32_t tmo_n_info(const tmo_vocab* v);
uint32_t tmo_max_token_length(const tmo_vocab* v);
uint32_t tmo_n_reverse(const tmo_vocab* v);
uint32_t tmo_capcode(const tmo_vocab* v);

/* pansearch.Fast.LongestSubstring (call sites go/tokenmonster.go:1049...; semantics
 * tokenmonster-cpp/src/tokenmonster.cpp:786-877): longest prefix of key[0..n) that is a key.
 * returns 1 if found */
int tmo_longest(const tmo_vocab* v, const uint8_t* key, size_t n, uint32_t* index, uint32_t* length);

/* go/tokenmonster.go:1017-1279 (Vocab.tokenize) on ALREADY NORMALIZED bytes, pad byte 0 (Q1).
 * returns number of tokens (written up to cap), *missing as go/tokenmonster.go:1274 */
long long tmo_tokenize(const tmo_vocab* v, const uint8_t* data, size_t n, uint32_t* out, size_t cap,
                       long long* missing);
/* go/tokenmonster.go:1281-1543 (tokenizeCount): b-branches add 1, not 2 (quirk Q2) */
long long tmo_count(const tmo_vocab* v, const uint8_t* data, size_t n, long long* missing);

/* training/trainvocab.go:925-1176: scores[id] += bytes covered; scores[deleteToken]++ on b-branches;
 * tokens_in_text; missing_set = 256-bit set of bytes that had no token.  scores has n_reverse entries
 * and is ACCUMULATED into (caller zeroes). */
void tmo_score(const tmo_vocab* v, const uint8_t* data, size_t n, uint32_t* scores, uint64_t* tokens_in_text,
               uint8_t missing_set[32]);

/* The same accumulation over the byte range [start, stop) of ONE walk over a text of n bytes (training/trainvocab.go:909-922: after
 * "midway" the worker walks the whole dataset as one strip): entered at `start` with forwardDelete = fd0, left at the first token
 * boundary >= stop; *exit_state = 2 * (boundary - stop) + forwardDelete there.  A token that begins before `stop` is counted in
 * full.  Chaining ranges through their exit states reproduces tmo_score of the whole text (tests/test_dist_gloo.py). */
void tmo_score_range(const tmo_vocab* v, const uint8_t* data, size_t n, size_t start, int fd0, size_t stop, uint32_t* scores,
                     uint64_t* tokens_in_text, uint8_t missing_set[32], uint32_t* exit_state);

/* tmo_score of the whole text (ONE strip, trainvocab.go:909-922) computed on `threads` threads, exactly: strips of `strip` bytes entered in
 * states guessed from a walk that starts `warm` bytes before them (0: guess state 0), then chained from the front and redone where a guess
 * was wrong (tm_oracle.c).  Accumulates like tmo_score; returns the number of strips that had to be redone. */
long long tmo_score_strips_mt(const tmo_vocab* v, const uint8_t* data, size_t n, size_t strip, size_t warm, uint32_t threads, uint32_t* scores,
                              uint64_t* tokens_in_text, uint8_t missing_set[32]);

/* go/tokenmonster.go:445-...(decode of raw token bytes, no capcode decoding): concatenates reverse[id] */
long long tmo_decode_raw(const tmo_vocab* v, const uint32_t* toks, size_t n, uint8_t* out, size_t cap);

/* per-thread counters of which exit of the walk was taken (see tm_oracle.c) */
void tmo_stats(uint64_t out[9], int reset);

#ifdef __cplusplus
}
#endif
#endif
