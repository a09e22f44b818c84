/* server_jobs.c — the tokenmonsterserver wire protocol (training/tokenmonsterserver.go:184-209, :300-800) for the jobs that sit on
 * the tokenize path, answered through the C ABI of libtokenmonster_hip.so.  Plain C, no Python, no torch.
 *
 * The reference's Python client (python/tokenmonster.py:1036-1089) talks to a Go subprocess over stdin/stdout: a request is a
 * 12-byte header { u8 job, u32 id, u56 payload length } + payload, a response starts with { u8 status, u64 length-or-id }.  This
 * harness reads those requests from stdin and writes byte-identical responses to stdout, so that the framing a maintainer keeps
 * when job 1 / job 20 call the GPU path can be tested without a Go toolchain:
 *   job 0   version            -> { 1, u32 VERSION }
 *   job 1   tokenize           payload { u32 n, n x { u64 len, raw text } } -> { 0, u64 total, u32 n } + n x { u64 len, ids }
 *           ids are 2 bytes each, or 4 once vocab.Len() > 65536 (:350-353 — NOT the 2/3 rule of TokenizeToSerialized's auto mode)
 *           the goroutine fan-out of :363-378 is ONE tm_tokenize_pipeline call
 *   job 20  count              same payload -> { 0, u64 4 + 8n, u32 n } + n x u64     (:753-800; one tm_count_batch_raw call)
 *   job 2/3/4 decode           payload { u32 n, n x { u64 len, ids of job bytes each } } -> like job 1 with decoded text (:399-446)
 *   job 5   new decoder        id = vocabulary -> { 1, u32 decoder id }                 (:449-471; the streaming Decoder, go/tokenmonster.go:552)
 *   job 6   unload decoder     id = decoder -> { 2, 0 }                                 (:473-482)
 *   job 7/8/9 decoder: decode  id = decoder, payload = ids of job - 5 bytes each -> { 0, u64 len } + the text that is complete so far (:484-504)
 *   job 12  save vocabulary    payload { u8 len, filename } -> { 2, 0 } or { 12, 0 }     (:537-554)
 *   job 10  load vocabulary    payload { u8 len, filename } -> { 1, u32 id } or { 12, 0 }
 *   job 11  unload             -> { 2, 0 } or { 10, 0 }
 *   anything else (14-19: vocabulary editing and YAML, not on the path)  -> { 15, 0 }   (:802-804)
 * Errors: unknown id 10, unloaded id 11 (header only, like sendError :104-114 without draining stdin, which a test harness must not).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tm_build.h"
#include "tokenmonster_hip.h"

enum { HEADER_IS_LENGTH = 0, HEADER_IS_ID = 1, HEADER_IS_EMPTY = 2, ERROR_ID_DOES_NOT_EXIST = 10, ERROR_ID_IS_UNLOADED = 11,
       ERROR_FILE_CANNOT_OPEN = 12, ERROR_NORMALIZATION_FAILED = 13, ERROR_INVALID_JOB = 15, VERSION = 5 };

static uint64_t rd(const uint8_t* p, int n) { uint64_t v = 0; for (int i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i); return v; }
static void wr(uint8_t* p, uint64_t v, int n) { for (int i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static int read_all(uint8_t* buf, size_t n) { return fread(buf, 1, n, stdin) == n; }
static void send9(uint8_t status, uint64_t v) { uint8_t h[9]; h[0] = status; wr(h + 1, v, 8); fwrite(h, 1, 9, stdout); }

#define MAX_VOCABS 64
static tm_vocab* g_vocabs[MAX_VOCABS];
static int g_used[MAX_VOCABS];   /* 0 never used, 1 loaded, 2 unloaded */
static uint32_t g_nvocabs;
#define MAX_DECODERS 256
static tm_decoder* g_decoders[MAX_DECODERS];
static int g_dec_used[MAX_DECODERS];   /* 0 never used, 1 live, 2 unloaded */
static uint32_t g_dec_vocab[MAX_DECODERS];   /* the vocabulary slot a decoder was made from */
static uint32_t g_ndecoders;

/* payload { u32 n, n x { u64 len, bytes } } -> packed text + offsets; returns 0 on a malformed payload */
static int unpack(const uint8_t* data, uint64_t len, uint32_t* n_out, uint8_t** text, uint64_t** offs) {
  if (len < 4) return 0;
  uint32_t n = (uint32_t)rd(data, 4);
  uint64_t pos = 4, total = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (pos + 8 > len) return 0;
    uint64_t l = rd(data + pos, 8);
    pos += 8;
    if (l > len - pos) return 0;
    pos += l; total += l;
  }
  *text = (uint8_t*)malloc(total ? total : 1);
  *offs = (uint64_t*)malloc(((size_t)n + 1) * 8);
  pos = 4; total = 0;
  (*offs)[0] = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint64_t l = rd(data + pos, 8);
    pos += 8;
    memcpy(*text + total, data + pos, l);
    pos += l; total += l;
    (*offs)[i + 1] = total;
  }
  *n_out = n;
  return 1;
}

static void send_batches(uint8_t status, uint32_t n, const uint8_t* bytes, const uint64_t* boff) {
  uint8_t h[13];
  h[0] = status;
  wr(h + 1, 4 + 8ull * n + (n ? boff[n] : 0), 8);
  wr(h + 9, n, 4);
  fwrite(h, 1, 13, stdout);
  for (uint32_t i = 0; i < n; i++) {
    uint8_t l[8];
    wr(l, boff[i + 1] - boff[i], 8);
    fwrite(l, 1, 8, stdout);
    fwrite(bytes + boff[i], 1, boff[i + 1] - boff[i], stdout);
  }
}

int main(void) {
  uint8_t h[12];
  if (tm_device_count() < 1) { fprintf(stderr, "server_jobs: no HIP device (%s)\n", tm_last_error()); return 2; }
  while (read_all(h, 12)) {
    const uint8_t job = h[0];
    const uint32_t id = (uint32_t)rd(h + 1, 4);
    const uint64_t len = rd(h + 5, 7);
    uint8_t* data = (uint8_t*)malloc(len ? len : 1);
    if (len && !read_all(data, len)) { free(data); return 1; }
    if (job == 0) {
      send9(HEADER_IS_ID, VERSION);
    } else if (job == 10) {                                        /* load (:505-523) */
      uint8_t status = ERROR_FILE_CANNOT_OPEN;
      uint32_t slot = 0;
      if (len >= 1 && (uint64_t)data[0] + 1 <= len) {
        char name[257];
        memcpy(name, data + 1, data[0]);
        name[data[0]] = 0;
        FILE* f = fopen(name, "rb");
        if (f) {
          fseek(f, 0, SEEK_END);
          long sz = ftell(f);
          fseek(f, 0, SEEK_SET);
          uint8_t* img = (uint8_t*)malloc(sz > 0 ? (size_t)sz : 1);
          tm_vocab* v = NULL;
          if (sz > 0 && fread(img, 1, (size_t)sz, f) == (size_t)sz && tm_vocab_load(img, (size_t)sz, &v) == TM_OK) {
            for (slot = 0; slot < g_nvocabs && g_used[slot] != 2; slot++) {}       /* reuse an unloaded slot like deletedVocabs */
            if (slot < MAX_VOCABS) { g_vocabs[slot] = v; g_used[slot] = 1; if (slot == g_nvocabs) g_nvocabs++; status = HEADER_IS_ID; }
            else tm_vocab_free(v);
          }
          free(img);
          fclose(f);
        }
      }
      send9(status, status == HEADER_IS_ID ? slot : 0);
    } else if (job == 11) {                                        /* unload (:525-535) */
      if (id < g_nvocabs && g_used[id] == 1) {
        /* a decoder keeps a pointer to its vocabulary (in the reference the Decoder keeps the Vocab alive through the GC): the decoders of
         * an unloaded vocabulary are unloaded with it, later jobs 7-9 on them answer ERROR_ID_IS_UNLOADED */
        for (uint32_t k = 0; k < g_ndecoders; k++)
          if (g_dec_used[k] == 1 && g_dec_vocab[k] == id) { tm_decoder_free(g_decoders[k]); g_decoders[k] = NULL; g_dec_used[k] = 2; }
        tm_vocab_free(g_vocabs[id]); g_vocabs[id] = NULL; g_used[id] = 2; send9(HEADER_IS_EMPTY, 0);
      }
      else send9(ERROR_ID_DOES_NOT_EXIST, 0);
    } else if (job == 12) {                                        /* save (:537-554) */
      if (id >= g_nvocabs) send9(ERROR_ID_DOES_NOT_EXIST, 0);
      else if (g_used[id] != 1) send9(ERROR_ID_IS_UNLOADED, 0);
      else {
        uint8_t status = ERROR_FILE_CANNOT_OPEN;
        if (len >= 1 && (uint64_t)data[0] + 1 <= len) {
          char name[257];
          memcpy(name, data + 1, data[0]);
          name[data[0]] = 0;
          if (tm_vocab_save(g_vocabs[id], name) == TM_OK) status = HEADER_IS_EMPTY;
        }
        send9(status, 0);
      }
    } else if (job == 5) {                                         /* new decoder (:449-471) */
      if (id >= g_nvocabs) send9(ERROR_ID_DOES_NOT_EXIST, 0);
      else if (g_used[id] != 1) send9(ERROR_ID_IS_UNLOADED, 0);
      else {
        uint32_t slot;
        tm_decoder* d = NULL;
        for (slot = 0; slot < g_ndecoders && g_dec_used[slot] != 2; slot++) {}     /* reuse an unloaded slot like deletedDecoders */
        if (slot >= MAX_DECODERS || tm_decoder_new(g_vocabs[id], &d) != TM_OK) send9(ERROR_INVALID_JOB, 0);
        else { g_decoders[slot] = d; g_dec_used[slot] = 1; g_dec_vocab[slot] = (uint32_t)id; if (slot == g_ndecoders) g_ndecoders++; send9(HEADER_IS_ID, slot); }
      }
    } else if (job == 6) {                                         /* unload decoder (:473-482; the reference answers 4 for an unknown id) */
      if (id >= g_ndecoders) send9(4, 0);
      else { if (g_dec_used[id] == 1) { tm_decoder_free(g_decoders[id]); g_decoders[id] = NULL; g_dec_used[id] = 2; } send9(HEADER_IS_EMPTY, 0); }
    } else if (job >= 7 && job <= 9) {                             /* decoder: decode (:484-504) */
      if (id >= g_ndecoders) send9(ERROR_ID_DOES_NOT_EXIST, 0);
      else if (g_dec_used[id] != 1) send9(ERROR_ID_IS_UNLOADED, 0);
      else {
        uint64_t cap = len * 24 + 256, got = 0;
        uint8_t* out = (uint8_t*)malloc(cap);
        int rc = tm_decoder_decode_serialized(g_decoders[id], data, len, (uint32_t)job - 5u, out, cap, &got);
        if (rc == TM_E_NOSPACE) {                                  /* the ids have been consumed, the text is kept: fetch it */
          cap = got + 1;
          out = (uint8_t*)realloc(out, cap);
          rc = tm_decoder_decode_serialized(g_decoders[id], NULL, 0, (uint32_t)job - 5u, out, cap, &got);
        }
        if (rc != TM_OK) { fprintf(stderr, "server_jobs: %s\n", tm_last_error()); send9(ERROR_INVALID_JOB, 0); }
        else { send9(HEADER_IS_LENGTH, got); fwrite(out, 1, got, stdout); }
        free(out);
      }
    } else if (job == 1 || job == 20 || (job >= 2 && job <= 4)) {
      if (id >= g_nvocabs) send9(ERROR_ID_DOES_NOT_EXIST, 0);
      else if (g_used[id] != 1) send9(ERROR_ID_IS_UNLOADED, 0);
      else {
        tm_vocab* v = g_vocabs[id];
        uint32_t n = 0;
        uint8_t* text = NULL;
        uint64_t* offs = NULL;
        if (!unpack(data, len, &n, &text, &offs)) send9(ERROR_INVALID_JOB, 0);
        else if (job == 1) {                                       /* :339-394 */
          const uint32_t enc = tm_vocab_size(v) > 65536 ? 4 : 2;   /* quirk: 2 or 4, by Len(), not by len(reverse) */
          uint64_t cap = (n ? offs[n] : 0) * enc + 16;
          uint8_t* out = (uint8_t*)malloc(cap);
          uint64_t* boff = (uint64_t*)malloc(((size_t)n + 1) * 8);
          uint32_t used = 0;
          int rc = tm_tokenize_pipeline(v, text, offs, n, 1, enc, 0, n > 1 ? 0 : 1, out, cap, boff, NULL, &used, NULL);
          if (rc == TM_E_NOSPACE) {
            cap = boff[n];
            out = (uint8_t*)realloc(out, cap ? cap : 1);
            rc = tm_tokenize_pipeline(v, text, offs, n, 1, enc, 0, n > 1 ? 0 : 1, out, cap, boff, NULL, &used, NULL);
          }
          if (rc != TM_OK) { fprintf(stderr, "server_jobs: %s\n", tm_last_error()); send9(ERROR_NORMALIZATION_FAILED, 0); }
          else send_batches(HEADER_IS_LENGTH, n, out, boff);
          free(out); free(boff);
        } else if (job == 20) {                                    /* :753-800 */
          uint64_t* counts = (uint64_t*)calloc((size_t)n + 1, 8);
          int rc = tm_count_batch_raw(v, text, offs, n, counts, NULL);
          if (rc != TM_OK) { fprintf(stderr, "server_jobs: %s\n", tm_last_error()); send9(ERROR_NORMALIZATION_FAILED, 0); }
          else {
            uint8_t hh[13];
            hh[0] = HEADER_IS_LENGTH;
            wr(hh + 1, 4 + 8ull * n, 8);
            wr(hh + 9, n, 4);
            fwrite(hh, 1, 13, stdout);
            for (uint32_t i = 0; i < n; i++) { uint8_t l[8]; wr(l, counts[i], 8); fwrite(l, 1, 8, stdout); }
          }
          free(counts);
        } else {                                                   /* decode, job = bytes per id (:399-446) */
          const uint32_t enc = job;
          uint64_t nt = 0;
          uint64_t* toff = (uint64_t*)malloc(((size_t)n + 1) * 8);
          toff[0] = 0;
          for (uint32_t i = 0; i < n; i++) { nt += (offs[i + 1] - offs[i]) / enc; toff[i + 1] = nt; }
          uint32_t* ids = (uint32_t*)malloc((nt ? nt : 1) * 4);
          for (uint32_t i = 0; i < n; i++)
            for (uint64_t k = 0; k < toff[i + 1] - toff[i]; k++) ids[toff[i] + k] = (uint32_t)rd(text + offs[i] + k * enc, enc > 3 ? 4 : (int)enc);
          uint64_t cap = nt * 8 + 64;
          uint8_t* out = (uint8_t*)malloc(cap);
          uint64_t* ooff = (uint64_t*)malloc(((size_t)n + 1) * 8);
          int rc = tm_decode_batch(v, ids, toff, n, 0, out, cap, ooff);
          if (rc == TM_E_NOSPACE) { cap = ooff[n]; out = (uint8_t*)realloc(out, cap ? cap : 1); rc = tm_decode_batch(v, ids, toff, n, 0, out, cap, ooff); }
          if (rc != TM_OK) { fprintf(stderr, "server_jobs: %s\n", tm_last_error()); send9(ERROR_INVALID_JOB, 0); }
          else send_batches(HEADER_IS_LENGTH, n, out, ooff);
          free(toff); free(ids); free(out); free(ooff);
        }
        free(text); free(offs);
      }
    } else {
      send9(ERROR_INVALID_JOB, 0);
    }
    fflush(stdout);
    free(data);
  }
  for (uint32_t i = 0; i < g_ndecoders; i++) if (g_dec_used[i] == 1) tm_decoder_free(g_decoders[i]);
  for (uint32_t i = 0; i < g_nvocabs; i++) if (g_used[i] == 1) tm_vocab_free(g_vocabs[i]);
  return 0;
}
