/* tokenize_file.c — the C ABI of libtokenmonster_hip.so from plain C (what the cgo stub of INTEGRATION.md binds).
 *
 *   tokenize_file <file.vocab> <text file> [--lines]
 *
 * Tokenizes the file (already normalized bytes, go/tokenmonster.go:963) as one document — or, with --lines, every
 * line as its own document, the batch shape of tokenmonsterserver job 1 — and prints the ids of each document on one
 * line.  Needs an MI355X: without a usable device every call fails with TM_E_NODEVICE, there is no CPU path. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tokenmonster_hip.h"

static uint8_t* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* buf = (uint8_t*)malloc((size_t)sz + 1);
  if (!buf || fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "%s: read error\n", path); exit(2); }
  fclose(f);
  *n = (size_t)sz;
  return buf;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <file.vocab> <text file> [--lines]\n", argv[0]); return 2; }
  const int by_line = argc > 3 && strcmp(argv[3], "--lines") == 0;
  size_t nv, nt;
  uint8_t* vfile = slurp(argv[1], &nv);
  uint8_t* text = slurp(argv[2], &nt);

  tm_vocab* vocab = NULL;
  if (tm_vocab_load(vfile, nv, &vocab) != TM_OK) { fprintf(stderr, "tm_vocab_load: %s\n", tm_last_error()); return 1; }

  /* documents = byte ranges of one packed buffer; here the newline stays with its line */
  uint32_t ndocs = 0, cap = 16;
  uint64_t* offsets = (uint64_t*)malloc((cap + 1) * sizeof *offsets);
  offsets[0] = 0;
  if (by_line) {
    for (size_t i = 0; i < nt; i++)
      if (text[i] == '\n' || i + 1 == nt) {
        if (ndocs == cap) { cap *= 2; offsets = (uint64_t*)realloc(offsets, (cap + 1) * sizeof *offsets); }
        offsets[++ndocs] = i + 1;
      }
  } else {
    offsets[++ndocs] = nt;
  }

  /* worst case is two ids per byte (a forward delete before every token); TM_E_NOSPACE would report the real need */
  uint64_t tokens_cap = 2 * (uint64_t)nt + 16, total = 0;
  uint32_t* tokens = (uint32_t*)malloc(tokens_cap * sizeof *tokens);
  uint64_t* tok_offsets = (uint64_t*)malloc((ndocs + 1) * sizeof *tok_offsets);
  uint32_t* missing = (uint32_t*)malloc(ndocs * sizeof *missing);
  int rc = tm_tokenize_batch(vocab, text, offsets, ndocs, tokens, tokens_cap, tok_offsets, missing);
  if (rc != TM_OK) { fprintf(stderr, "tm_tokenize_batch: %d %s\n", rc, tm_last_error()); return 1; }
  for (uint32_t d = 0; d < ndocs; d++) {
    for (uint64_t i = tok_offsets[d]; i < tok_offsets[d + 1]; i++) printf(i == tok_offsets[d] ? "%u" : " %u", tokens[i]);
    printf("\n");
    total += tok_offsets[d + 1] - tok_offsets[d];
  }
  fprintf(stderr, "%u documents, %zu bytes, %llu tokens\n", ndocs, nt, (unsigned long long)total);
  tm_vocab_free(vocab);
  free(tokens); free(tok_offsets); free(missing); free(offsets); free(text); free(vfile);
  return 0;
}
