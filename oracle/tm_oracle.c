/* oracle/tm_oracle.c — TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * Plain-C CPU restatement of TokenMonster's 6-branch ungreedy longest-match tokenization:
 *   walk + scoring + select : go/tokenmonster.go:1017-1279  (== tokenmonster-cpp/src/tokenmonster.cpp:1723-1991)
 *   count variant           : go/tokenmonster.go:1281-1543
 *   trainvocab accumulation : training/trainvocab.go:925-1176
 *   .vocab loader           : go/tokenmonster.go:2656-2736  (== tokenmonster.cpp:1287-1359)
 *   longest-prefix index    : pansearch.Fast (third-party github.com/alasdairforsythe/pansearch, NOT in
 *                             /root/reference and un-pinned: no go.mod).  Its published behaviour, as used at
 *                             go/tokenmonster.go:1049 and translated at tokenmonster.cpp:491-1280, is
 *                             "longest prefix that is a key; index = ordinal in (length, bytewise) order".
 *                             Restated here in the most boring way possible: one binary search per length,
 *                             longest first, over the records in file order.
 *
 * PINNING: (1) the reference's only known-answer vector, tokenmonster-cpp/tests/unit.cpp:87-112
 * ("ab a z" -> {3,0,1,0}, missing 1, count {4,1}) — tests/test_oracle.py; (2) differential runs against
 * oracle/_ref/libtmref.so, i.e. the reference's own C++ runtime compiled unmodified, on every fixture
 * vocabulary and corpus in tests/ (IDs, missing, count).  The Go implementation itself cannot be run in
 * this image (no Go toolchain, un-vendored deps); pad byte is 0 as in tokenmonster.cpp:1724-1726.
 */
#include "tm_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint8_t flag, n_words;     /* tokenInner, go/tokenmonster.go:79-83 */
  uint8_t len, len1, len2;   /* own key length; alt lengths (0 = none) */
  uint32_t index1, index2;   /* record ordinals of alternatives or TMO_NONE */
  uint32_t id, id1, id2;
  uint32_t key_off;          /* offset of key bytes in keys blob */
} tmo_row;

struct tmo_vocab {
  uint8_t capcode, charset, norm_flag, level, reserve;
  uint32_t unk, vocab_size, n_reverse, n_info, delete_id, max_len;
  tmo_row* rows;
  uint8_t* keys;             /* concatenated key bytes */
  uint32_t len_start[42];    /* records of length L are [len_start[L], len_start[L+1]) */
  uint8_t begin_byte[256];
  uint32_t* rev_off;         /* reverse[id] -> (offset,len) of the LAST record with that id (go :2715) */
  uint8_t* rev_len;
};

static _Thread_local char g_err[256];
const char* tmo_last_error(void) { return g_err; }

static uint32_t rd24(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16); }

tmo_vocab* tmo_load(const uint8_t* f, size_t n) {
  tmo_vocab* v = (tmo_vocab*)calloc(1, sizeof(*v));
  size_t pos = 0;
#define NEED(k) do { if (pos + (size_t)(k) > n) { snprintf(g_err, sizeof g_err, "truncated .vocab at %zu", pos); goto fail; } } while (0)
  NEED(24);
  v->capcode = f[0]; v->charset = f[1]; v->norm_flag = f[2]; v->level = f[3]; v->reserve = f[4];
  if (v->charset > 2 || v->capcode > 2) { snprintf(g_err, sizeof g_err, "not a TokenMonster vocabulary"); goto fail; }
  v->unk = rd24(f + 8); v->vocab_size = rd24(f + 11); v->n_reverse = rd24(f + 14); v->n_info = rd24(f + 17);
  v->delete_id = rd24(f + 20); v->max_len = f[23];
  pos = 24;
  v->rows = (tmo_row*)calloc(v->n_info ? v->n_info : 1, sizeof(tmo_row));
  v->keys = (uint8_t*)malloc((size_t)v->n_info * 40 + 1);
  v->rev_off = (uint32_t*)calloc(v->n_reverse ? v->n_reverse : 1, 4);
  v->rev_len = (uint8_t*)calloc(v->n_reverse ? v->n_reverse : 1, 1);
  uint32_t koff = 0;
  for (int L = 0; L < 42; L++) v->len_start[L] = 0;
  uint32_t prev_len = 0;
  for (uint32_t i = 0; i < v->n_info; i++) {
    NEED(1);
    uint32_t kl = f[pos++];
    if (kl > 40 || kl == 0) { snprintf(g_err, sizeof g_err, "bad key length %u", kl); goto fail; }
    NEED(kl + 15);
    tmo_row* r = &v->rows[i];
    r->len = (uint8_t)kl; r->key_off = koff;
    memcpy(v->keys + koff, f + pos, kl); koff += kl; pos += kl;
    if (kl < prev_len) { snprintf(g_err, sizeof g_err, "records not in length order at %u", i); goto fail; }
    for (uint32_t L = prev_len + 1; L <= kl; L++) v->len_start[L] = i;
    prev_len = kl;
    r->flag = f[pos]; r->n_words = f[pos + 1];
    r->index1 = rd24(f + pos + 2); r->index2 = rd24(f + pos + 5); r->id = rd24(f + pos + 8);
    pos += 15; /* flag, nWords, index, index2, id, f32 score */
    if (r->index1 != TMO_NONE) {            /* go/tokenmonster.go:2703-2706 */
      if (r->index1 >= i) { snprintf(g_err, sizeof g_err, "alt index not earlier at %u", i); goto fail; }
      r->len1 = v->rows[r->index1].len; r->id1 = v->rows[r->index1].id;
    }
    if (r->index2 != TMO_NONE) {            /* :2708-2711 */
      if (r->index2 >= i) { snprintf(g_err, sizeof g_err, "alt index2 not earlier at %u", i); goto fail; }
      r->len2 = v->rows[r->index2].len; r->id2 = v->rows[r->index2].id;
    }
    if (r->id >= v->n_reverse) { snprintf(g_err, sizeof g_err, "id out of range at %u", i); goto fail; }
    v->rev_off[r->id] = r->key_off; v->rev_len[r->id] = r->len;   /* :2715 last writer wins */
  }
  for (uint32_t L = prev_len + 1; L < 42; L++) v->len_start[L] = v->n_info;
  NEED(256);
  memcpy(v->begin_byte, f + pos, 256); pos += 256;
  NEED(3);
  { uint32_t nd = rd24(f + pos); pos += 3;
    for (uint32_t i = 0; i < nd; i++) { NEED(1); uint32_t l = f[pos++]; NEED(l + 7); pos += l + 7; } }
  if (pos != n) { snprintf(g_err, sizeof g_err, "trailing bytes in .vocab"); goto fail; }  /* :2731 */
  return v;
fail:
  tmo_free(v);
  return NULL;
#undef NEED
}

void tmo_free(tmo_vocab* v) {
  if (!v) return;
  free(v->rows); free(v->keys); free(v->rev_off); free(v->rev_len); free(v);
}

uint32_t tmo_vocab_size(const tmo_vocab* v) { return v->vocab_size; }
uint32_t tmo_n_info(const tmo_vocab* v) { return v->n_info; }
uint32_t tmo_max_token_length(const tmo_vocab* v) { return v->max_len; }
uint32_t tmo_n_reverse(const tmo_vocab* v) { return v->n_reverse; }
uint32_t tmo_capcode(const tmo_vocab* v) { return v->capcode; }

int tmo_longest(const tmo_vocab* v, const uint8_t* key, size_t n, uint32_t* index, uint32_t* length) {
  size_t L = n > 40 ? 40 : n;
  for (; L >= 1; L--) {
    uint32_t lo = v->len_start[L], hi = v->len_start[L + 1];
    while (lo < hi) {
      uint32_t mid = lo + (hi - lo) / 2;
      int c = memcmp(v->keys + v->rows[mid].key_off, key, L);
      if (c == 0) { *index = mid; *length = (uint32_t)L; return 1; }
      if (c < 0) lo = mid + 1; else hi = mid;
    }
  }
  *index = 0; *length = 0;
  return 0;
}

/* which exit of the walk was taken how often, so tests can prove every branch was exercised:
 * 0 s1, 1 s2, 2 s3, 3 s1b, 4 s2b, 5 s3b, 6 fast-exit/end-of-text default emit, 7 all-NOSCORE default emit, 8 not found */
static _Thread_local uint64_t g_stats[9];
void tmo_stats(uint64_t out[9], int reset) {
  for (int k = 0; k < 9; k++) { out[k] = g_stats[k]; if (reset) g_stats[k] = 0; }
}

#define NOSCORE (-1000000)
static int max0(int x) { return x > 0 ? x : 0; }

/* one walk, three sinks: mode 0 = ids (go :1017), 1 = count (go :1281), 2 = trainvocab histogram */
typedef struct {
  int mode;
  uint32_t* out; size_t cap; long long ntok;      /* mode 0/1 */
  uint32_t* scores; uint64_t tokens_in_text; uint8_t* missing_set; /* mode 2 */
  int negate;                /* mode 2: take the contributions back instead of adding them (tmo_score_strips_mt re-walks a strip it had entered in the wrong state) */
  int64_t* missing_cnt;      /* mode 2: per-byte counters instead of the bit set (bits cannot be taken back) */
} sink_t;

static void emit(const tmo_vocab* v, sink_t* s, uint32_t id, int adv, int with_delete) {
  if (s->mode == 0) {
    if ((size_t)s->ntok < s->cap) s->out[s->ntok] = id;
    s->ntok++;
    if (with_delete) { if ((size_t)s->ntok < s->cap) s->out[s->ntok] = v->delete_id; s->ntok++; }
  } else if (s->mode == 1) {
    s->ntok++;                        /* go/tokenmonster.go:1505-1521: +1 even for b-branches */
  } else {
    if (s->negate) {
      s->scores[id] -= (uint32_t)adv;
      if (with_delete) s->scores[v->delete_id]--;
      s->tokens_in_text -= with_delete ? 2 : 1;
      return;
    }
    s->scores[id] += (uint32_t)adv;   /* trainvocab.go:1109..1162 */
    if (with_delete) s->scores[v->delete_id]++;   /* :1134,1143,1152 (Q4: we use the ID) */
    s->tokens_in_text += with_delete ? 2 : 1;
  }
}

/* The walk over data[start .. stop) of a text of n bytes (all of it can be looked at), entered with forwardDelete = fd0: what one
 * goroutine of the reference does for start = 0, fd0 = 0, stop = n.  The general form restates the one property the multi-GPU scoring
 * pass rests on: the walk's whole state at a token boundary is (i, forwardDelete) — index/length are recomputed from the text — so a
 * range of the whole-buffer walk (training/trainvocab.go:909-922) can be entered at start with fd0 and left where the first token
 * begins at or behind `stop`.  *exit_state = 2 * (i - stop) + forwardDelete at that point. */
/* `data` holds n bytes of text followed by the pad byte 0 (go :1038-1046; tokenmonster.cpp:1726) */
static long long walk_range_padded(const tmo_vocab* v, const uint8_t* data, size_t n, size_t start, int fd0, size_t stop, sink_t* s, uint32_t* exit_state) {
  long long missing = 0;
  if (exit_state) *exit_state = 0;
  if (v->max_len == 0) return 0;                  /* go :960 */
  const int lenData = (int)n, maxlen = (int)v->max_len;
  const int off = v->charset == 2 ? 2 : 1;        /* go :1031-1034 */
  const int maxlen_sp = maxlen - off;             /* go :1036 */
  const int has_delete = v->delete_id != TMO_NONE;
  const uint8_t* bb = v->begin_byte;
  uint8_t lil[48];
  memset(lil, 0, sizeof lil);
  lil[0] = 32;                                    /* go :1030 */

  int i = (int)start, fd = (fd0 && start >= stop) ? 1 : 0;     /* (a range that is skipped whole hands its entry state on unchanged) */
  const int stopi = (int)stop;
  uint32_t index = 0, length = 0;
  if (fd0 && i < stopi) {
    /* entered in a forward-delete state: index/length are the longest match of ' ' + data[i:] (go :1088-1095) */
    int rem0 = lenData - i;
    int m = rem0 < maxlen_sp ? rem0 : maxlen_sp;
    if (m < 0) m = 0;
    memcpy(lil + off, data + i, (size_t)m);
    uint32_t lb = 0;
    tmo_longest(v, lil, (size_t)(m + off), &index, &lb);
    length = lb - (uint32_t)off;
    fd = 1;
    goto checkpoint;
  }
  while (i < stopi) {
    int rem = lenData - i;
    if (!tmo_longest(v, data + i, (size_t)(rem < maxlen ? rem : maxlen), &index, &length)) {
      /* go :1269-1276 */
      if (s->mode == 2) {
        if (s->negate) s->tokens_in_text--; else s->tokens_in_text++;
        if (s->missing_cnt) s->missing_cnt[data[i]] += s->negate ? -1 : 1;
        else if (s->missing_set) s->missing_set[data[i] >> 3] |= (uint8_t)(1u << (data[i] & 7));
      }
      else if (v->unk != TMO_NONE) { if (s->mode == 0) { if ((size_t)s->ntok < s->cap) s->out[s->ntok] = v->unk; } s->ntok++; }
      i++; missing++; fd = 0; g_stats[8]++;
      continue;
    }
  checkpoint:;
    if (i >= stopi) break;                        /* (never taken when stop == n: a second token is only found inside the text) */
    const tmo_row* O = &v->rows[index];
    int len = (int)length;
    int i1 = i + len;
    int looked = 0;
    if (i1 < lenData && ((O->flag & 32) == 0 || bb[data[i1]] != 12)) {       /* go :1057 */
      looked = 1;
      int s1 = NOSCORE, s2 = NOSCORE, s3 = NOSCORE, s1b = NOSCORE, s2b = NOSCORE, s3b = NOSCORE, best = NOSCORE;
      uint32_t x1 = 0, l1 = 0, x2 = 0, l2 = 0, x3 = 0, l3 = 0, x1b = 0, l1b = 0, x2b = 0, l2b = 0, x3b = 0, l3b = 0;
      /* candidate first tokens: k=0 greedy (go :1068), k=1 alt1 (:1111), k=2 alt2 (:1163, nested in alt1's test) */
      for (int k = 0; k < 3; k++) {
        int flen; const tmo_row* F;
        if (k == 0) { flen = len; F = O; }
        else if (k == 1) { if (O->index1 == TMO_NONE) break; flen = (int)O->len1 - fd; F = &v->rows[O->index1]; }
        else { if (O->index2 == TMO_NONE) break; flen = (int)O->len2 - fd; F = &v->rows[O->index2]; }
        int ik = i + flen;
        int remk = lenData - ik;
        uint32_t xk, lk;
        if (!tmo_longest(v, data + ik, (size_t)(remk < maxlen ? remk : maxlen), &xk, &lk)) continue;
        const tmo_row* S = &v->rows[xk];
        int nw = (int)F->n_words - fd;                                      /* go :1071,1117,1169 */
        int nb = bb[data[ik + (int)lk]];
        int BL = flen + (int)lk;                                            /* :1075, :1120, :1172 */
        int sc = (BL + (F->flag >> 7) + (S->flag >> 7) + max0(nw - 1) + max0((int)S->n_words - 1) +
                  ((S->flag >> 2) & 1) + ((nb >> 2) & 1) + (nw + (int)S->n_words + (nb >> 3)) * 100) -
                 ((F->flag & 1 & (S->flag >> 1)) * 103 + ((F->flag >> 3) & 1 & (S->flag >> 4)) * 100 +
                  (S->flag & 1 & nb) * 3);
        if (k > 0) sc -= (BL < len ? 100 : 0) + (BL == len ? 10000 : 0);   /* :1132-1133 */
        if (sc > best) best = sc;
        if (k == 0) { s1 = sc; x1 = xk; l1 = lk; } else if (k == 1) { s2 = sc; x2 = xk; l2 = lk; } else { s3 = sc; x3 = xk; l3 = lk; }
        /* forward-delete variant, go :1088-1108 / :1137-1160 / :1189-1212 */
        if (has_delete && (S->flag & 2) != 0 && nb == 1 && S->n_words == 0) {
          int m = remk < maxlen_sp ? remk : maxlen_sp;
          if (m < 0) m = 0;
          memcpy(lil + off, data + ik, (size_t)m);
          uint32_t xb, lb;
          tmo_longest(v, lil, (size_t)(m + off), &xb, &lb);                 /* found ignored, Q9 */
          if ((int)lb > (int)lk + 1) {
            int lbb = (int)lb - off;
            const tmo_row* Sb = &v->rows[xb];
            int nbb = bb[data[ik + lbb]];
            int BLb = flen + lbb;
            int scb = (BLb + (F->flag >> 7) + (Sb->flag >> 7) + max0(nw - 1) + max0((int)Sb->n_words - 1) +
                       ((nbb >> 2) & 1) + (nw + (int)Sb->n_words + (nbb >> 3)) * 100) -
                      ((F->flag & 1) * 103 + ((F->flag >> 3) & 1 & (Sb->flag >> 4)) * 100 +
                       (Sb->flag & 1 & nbb) * 3 + 1);
            if (k > 0) scb -= (BLb < len ? 100 : 0) + (BLb == len ? 10000 : 0);
            if (scb > best) best = scb;
            if (k == 0) { s1b = scb; x1b = xb; l1b = (uint32_t)lbb; } else if (k == 1) { s2b = scb; x2b = xb; l2b = (uint32_t)lbb; } else { s3b = scb; x3b = xb; l3b = (uint32_t)lbb; }
          }
        }
      }
      /* go :1217-1262 — first equal wins in this order */
      if (best != NOSCORE) {
        if (best == s1)  { g_stats[0]++; emit(v, s, O->id,  len, 0);               i += len;               index = x1;  length = l1;  fd = 0; goto checkpoint; }
        if (best == s2)  { g_stats[1]++; emit(v, s, O->id1, (int)O->len1 - fd, 0); i += (int)O->len1 - fd; index = x2;  length = l2;  fd = 0; goto checkpoint; }
        if (best == s3)  { g_stats[2]++; emit(v, s, O->id2, (int)O->len2 - fd, 0); i += (int)O->len2 - fd; index = x3;  length = l3;  fd = 0; goto checkpoint; }
        if (best == s1b) { g_stats[3]++; emit(v, s, O->id,  len, 1);               i += len;               index = x1b; length = l1b; fd = 1; goto checkpoint; }
        if (best == s2b) { g_stats[4]++; emit(v, s, O->id1, (int)O->len1 - fd, 1); i += (int)O->len1 - fd; index = x2b; length = l2b; fd = 1; goto checkpoint; }
        if (best == s3b) { g_stats[5]++; emit(v, s, O->id2, (int)O->len2 - fd, 1); i += (int)O->len2 - fd; index = x3b; length = l3b; fd = 1; goto checkpoint; }
      }
    }
    g_stats[looked ? 7 : 6]++;
    emit(v, s, O->id, len, 0);        /* go :1265-1267 */
    i += len; fd = 0;
  }
  if (exit_state) *exit_state = (uint32_t)(2 * (i - stopi) + fd);
  return missing;
}

static long long walk_range(const tmo_vocab* v, const uint8_t* src, size_t n, size_t start, int fd0, size_t stop, sink_t* s, uint32_t* exit_state) {
  uint8_t* data = (uint8_t*)malloc(n + 1);
  memcpy(data, src, n);
  data[n] = 0;                                    /* go :1038-1046, pad 0 (tokenmonster.cpp:1726) */
  const long long missing = walk_range_padded(v, data, n, start, fd0, stop, s, exit_state);
  free(data);
  return missing;
}

static long long walk(const tmo_vocab* v, const uint8_t* src, size_t n, sink_t* s) { return walk_range(v, src, n, 0, 0, n, s, NULL); }

long long tmo_tokenize(const tmo_vocab* v, const uint8_t* data, size_t n, uint32_t* out, size_t cap, long long* missing) {
  sink_t s; memset(&s, 0, sizeof s); s.mode = 0; s.out = out; s.cap = cap;
  long long m = walk(v, data, n, &s);
  if (missing) *missing = m;
  return s.ntok;
}

long long tmo_count(const tmo_vocab* v, const uint8_t* data, size_t n, long long* missing) {
  sink_t s; memset(&s, 0, sizeof s); s.mode = 1;
  long long m = walk(v, data, n, &s);
  if (missing) *missing = m;
  return s.ntok;
}

void tmo_score(const tmo_vocab* v, const uint8_t* data, size_t n, uint32_t* scores, uint64_t* tokens_in_text,
               uint8_t missing_set[32]) {
  sink_t s; memset(&s, 0, sizeof s); s.mode = 2; s.scores = scores; s.missing_set = missing_set;
  walk(v, data, n, &s);
  if (tokens_in_text) *tokens_in_text += s.tokens_in_text;
}

/* scoring mode over the byte range [start, stop) of a text of n bytes, entered in state (start offset already applied by the caller,
 * fd0); see walk_range */
void tmo_score_range(const tmo_vocab* v, const uint8_t* data, size_t n, size_t start, int fd0, size_t stop, uint32_t* scores,
                     uint64_t* tokens_in_text, uint8_t missing_set[32], uint32_t* exit_state) {
  sink_t s; memset(&s, 0, sizeof s); s.mode = 2; s.scores = scores; s.missing_set = missing_set;
  walk_range(v, data, n, start, fd0, stop, &s, exit_state);
  if (tokens_in_text) *tokens_in_text += s.tokens_in_text;
}

long long tmo_decode_raw(const tmo_vocab* v, const uint32_t* toks, size_t n, uint8_t* out, size_t cap) {
  size_t pos = 0;
  for (size_t i = 0; i < n; i++) {
    if (toks[i] >= v->n_reverse) continue;       /* go decode skips ids out of range */
    uint32_t l = v->rev_len[toks[i]];
    if (pos + l <= cap) memcpy(out + pos, v->keys + v->rev_off[toks[i]], l);
    pos += l;
  }
  return (long long)pos;
}

/* ---- the whole-buffer scoring walk on many threads, EXACT (bench.py's verification of the scoring pass at full size) -------------------
 * training/trainvocab.go:909-922 walks the dataset as ONE strip: serial.  The walk's whole state at a token boundary is (i, forwardDelete)
 * (walk_range above), so the text is cut into strips; the entry state of a strip is first GUESSED from a short walk that starts `warm`
 * bytes before it (walks from different states run into each other within a few tokens), every strip is walked from its guess on the
 * pool, and then the chain exit(k) == entry(k + 1) is checked strip by strip from the front: a strip whose guess was wrong is taken back
 * (the same walk with negated contributions) and walked again from the true state.  What is left satisfies entry(0) = (0, 0),
 * entry(k + 1) = exit(k) for every k - the serial walk, whatever the guesses were.  Returns the number of strips that had to be redone. */
#include <pthread.h>
typedef struct {
  const tmo_vocab* v; const uint8_t* data; size_t n, strip, nstrips, warm;
  uint32_t* entry; uint32_t* exit_; size_t next; pthread_mutex_t mu;
  uint32_t n_rev;
} strips_job;
typedef struct { strips_job* job; uint32_t* scores; uint64_t tokens; int64_t missing[256]; } strips_worker;

static void strip_walk(const strips_job* J, size_t k, uint32_t entry, sink_t* s, uint32_t* exit_state) {
  const size_t a = k * J->strip, b = a + J->strip < J->n ? a + J->strip : J->n;
  walk_range_padded(J->v, J->data, J->n, a + (entry >> 1), (int)(entry & 1u), b, s, exit_state);
}
static void* strips_thread(void* arg) {
  strips_worker* W = (strips_worker*)arg;
  strips_job* J = W->job;
  for (;;) {
    pthread_mutex_lock(&J->mu);
    const size_t k = J->next++;
    pthread_mutex_unlock(&J->mu);
    if (k >= J->nstrips) break;
    uint32_t e = 0;
    if (k > 0 && J->warm > 0) {
      const size_t a = k * J->strip, from = a > J->warm ? a - J->warm : 0;
      sink_t nul; memset(&nul, 0, sizeof nul); nul.mode = 1;
      walk_range_padded(J->v, J->data, J->n, from, 0, a, &nul, &e);
    }
    J->entry[k] = e;
    sink_t s; memset(&s, 0, sizeof s); s.mode = 2; s.scores = W->scores; s.missing_cnt = W->missing;
    strip_walk(J, k, e, &s, &J->exit_[k]);
    W->tokens += s.tokens_in_text;
  }
  return NULL;
}
long long tmo_score_strips_mt(const tmo_vocab* v, const uint8_t* src, size_t n, size_t strip, size_t warm, uint32_t threads, uint32_t* scores,
                              uint64_t* tokens_in_text, uint8_t missing_set[32]) {
  if (strip < 4096) strip = 4096;
  if (threads == 0) threads = 1;
  strips_job J; memset(&J, 0, sizeof J);
  uint8_t* data = (uint8_t*)malloc(n + 1);
  memcpy(data, src, n); data[n] = 0;
  J.v = v; J.data = data; J.n = n; J.strip = strip; J.nstrips = (n + strip - 1) / strip; J.warm = warm; J.n_rev = v->n_reverse;
  if (J.nstrips == 0) J.nstrips = 1;
  J.entry = (uint32_t*)calloc(J.nstrips, 4); J.exit_ = (uint32_t*)calloc(J.nstrips, 4);
  pthread_mutex_init(&J.mu, NULL);
  if (threads > J.nstrips) threads = (uint32_t)J.nstrips;
  strips_worker* W = (strips_worker*)calloc(threads, sizeof *W);
  pthread_t* th = (pthread_t*)calloc(threads, sizeof *th);
  for (uint32_t t = 0; t < threads; t++) { W[t].job = &J; W[t].scores = (uint32_t*)calloc(v->n_reverse ? v->n_reverse : 1, 4); }
  for (uint32_t t = 1; t < threads; t++) pthread_create(&th[t], NULL, strips_thread, &W[t]);
  strips_thread(&W[0]);
  for (uint32_t t = 1; t < threads; t++) pthread_join(th[t], NULL);
  /* the chain, from the front: a wrong guess is taken back and walked again */
  long long redone = 0;
  uint32_t e = 0;
  for (size_t k = 0; k < J.nstrips; k++) {
    if (J.entry[k] != e) {
      sink_t s; memset(&s, 0, sizeof s); s.mode = 2; s.scores = W[0].scores; s.missing_cnt = W[0].missing;
      s.negate = 1; strip_walk(&J, k, J.entry[k], &s, NULL);
      s.negate = 0; strip_walk(&J, k, e, &s, &J.exit_[k]);
      W[0].tokens += s.tokens_in_text;           /* (wraps through zero and back: unsigned arithmetic) */
      J.entry[k] = e;
      redone++;
    }
    e = J.exit_[k];
  }
  uint64_t tok = 0;
  int64_t miss[256]; memset(miss, 0, sizeof miss);
  for (uint32_t t = 0; t < threads; t++) {
    for (uint32_t i = 0; i < v->n_reverse; i++) scores[i] += W[t].scores[i];
    tok += W[t].tokens;
    for (int b = 0; b < 256; b++) miss[b] += W[t].missing[b];
    free(W[t].scores);
  }
  if (tokens_in_text) *tokens_in_text += tok;
  if (missing_set) for (int b = 0; b < 256; b++) if (miss[b] > 0) missing_set[b >> 3] |= (uint8_t)(1u << (b & 7));
  pthread_mutex_destroy(&J.mu);
  free(W); free(th); free(J.entry); free(J.exit_); free(data);
  return redone;
}
