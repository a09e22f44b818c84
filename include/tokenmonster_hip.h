/* tokenmonster_hip.h — C ABI of libtokenmonster_hip.so: MI355X (gfx950) batch tokenizer for
 * TokenMonster vocabularies.  This is the drop-in boundary for ONE path of the reference:
 * the 6-branch ungreedy longest-match loop, go/tokenmonster.go:1017-1279 (Vocab.tokenize) and its
 * copies (:1281 count, :1545/:1817/:2089 serialized, training/trainvocab.go:925-1176 scoring).
 *
 * The reference has no FFI of its own; the seams this library replaces are
 *   (*Vocab).Tokenize / Count / TokenizeToSerialized      go/tokenmonster.go:959, :971, :986
 *   tokenmonsterserver job 1 / job 20 goroutine fan-out   training/tokenmonsterserver.go:363-378, :773-787
 *   trainvocab worker inner loop                          training/trainvocab.go:925-1176
 *   Load (table upload; .vocab layout)                    go/tokenmonster.go:2656-2736
 * INTEGRATION.md shows the cgo stub a maintainer adds on the Go side for each entry point.
 *
 * Conventions: extern "C", plain pointers and sizes.  Every host pointer is borrowed for the
 * duration of the call only (cgo pointer-passing rule).  Every function returns TM_OK (0) or a
 * negative TM_E_* code; tm_last_error() gives a thread-local message.  There is NO CPU fallback
 * inside this library: if no gfx950 device is usable the call fails (TM_E_NODEVICE) and the Go
 * side keeps using its own vocab.tokenize.
 *
 * Input text is ALREADY NORMALIZED bytes (what go/tokenmonster.go:963 `normalize` returns), unless
 * stated otherwise.  The look-ahead pad byte is 0 (tokenmonster-cpp/src/tokenmonster.cpp:1724-1726).
 */
#ifndef TOKENMONSTER_HIP_H
#define TOKENMONSTER_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TM_OK 0
#define TM_E_INVALID (-1)   /* bad argument / malformed .vocab */
#define TM_E_NODEVICE (-2)  /* no usable gfx950 device */
#define TM_E_HIP (-3)       /* a HIP runtime call failed */
#define TM_E_NOSPACE (-4)   /* caller buffer too small; required size reported via out-params */
#define TM_E_LIMIT (-5)     /* input exceeds a documented limit (batch bytes, trie nodes) */
#define TM_E_INPUT (-6)     /* the walk cannot advance on this text with this vocabulary (the reference loops forever on it: a UTF-16
                               vocabulary with one-byte keys beside the delete token); nothing is wrong with the device */

#define TM_E_INTERNAL (-7)  /* the pipeline's stages disagree with each other (the emit stage met a transition the match stage never wrote): a fault
                               of this library, never of the input.  Callers that fall back to the CPU path on errors may do so here; on TM_E_INPUT they
                               must NOT (the reference's own walk does not terminate on that input) */

#define TM_NONE 0xFFFFFFu   /* go/tokenmonster.go:32 DOES_NOT_EXIST */

typedef struct tm_vocab tm_vocab;     /* immutable device-resident vocabulary tables */
typedef struct tm_batch tm_batch;     /* reusable device workspace for one batch of documents */
typedef struct tm_dataset tm_dataset; /* device-resident normalized dataset for the scoring pass */
typedef struct tm_decoder tm_decoder; /* streaming Decoder: per-connection host state */

const char* tm_last_error(void);
int tm_device_count(void);
/* Select the HIP device used by the calling thread's subsequent calls. */
int tm_set_device(int device);

/* ---- vocabulary: replaces Load's table construction (go/tokenmonster.go:2656-2736) ------------ */
/* Parses the bytes of a .vocab file (layout: SURVEY.md Appendix A), builds the longest-match
 * index and per-record rows, uploads them to the current device's HBM. */
int tm_vocab_load(const uint8_t* vocab_file, size_t n, tm_vocab** out);
/* The same on a named device: tm_set_device's "current device" belongs to the calling OS thread, which a goroutine under cgo does not own. */
int tm_vocab_load_on(const uint8_t* vocab_file, size_t n, int device, tm_vocab** out);
/* Batches, decoders and lanes of the vocabulary must not be used afterwards.  Kernels that an asynchronous entry point (tm_batch_run on a
 * caller's stream, tm_score_device, tm_score_finish ...) has already launched may still be in flight: tm_vocab_free waits for them (an
 * event per stream the tables were used on - this vocabulary's work only, not the device as hipFree would) and then parks the device
 * memory for the next tm_vocab_load.  A caller's stream on which this vocabulary's kernels were launched must still exist at this point. */
void tm_vocab_free(tm_vocab* v);
/* OPTIONAL, for large vocabularies (tables of 10 MB and more against the 4 MB of L2 an XCD has): lays the tables out by USE.  The match
 * kernel's table walk is replayed on the host over `normalized_sample` (text as tm_normalize writes it; a few MiB of what the vocabulary
 * will be used on), the trie's nodes are renumbered - most used first, so that the rows, space-prefix links and suffix links that are
 * gathered most share cache lines - and the tables are written again into the device block they lie in.  Token ids and every result are
 * unchanged (only internal node numbers move); nothing else of this vocabulary may run meanwhile, and the call waits for what already does.
 * Not for vocabularies made by tm_vocab_block_import (no records).  Cost: about as long as loading the vocabulary + 0.2 s per MiB of sample. */
int tm_vocab_tune(tm_vocab* v, const uint8_t* normalized_sample, uint64_t n);
/* tm_vocab_load + tm_vocab_tune in one: the caller hands a sample of normalized text with the file and the tables are built for the device once,
 * laid out by use from the start (current device; sample_n == 0: plain tm_vocab_load).  Measured on one MI355X, 1 GiB (profiles/r06_tuned_tables.txt):
 * match kernel -1 % with 32 000 ids (6 MB of tables), -5 % with 100 256 ids (16 MB) and on the 65 536-id scoring pass. */
int tm_vocab_load_sample(const uint8_t* vocab_file, size_t n, const uint8_t* normalized_sample, uint64_t sample_n, tm_vocab** out);
/* The device block of a vocabulary from process to process (the data-parallel scoring mode: ONE rank builds a candidate's tables, the others
 * take the finished block - e.g. as the destination of an RCCL broadcast - instead of repeating tm_build_vocab + tm_vocab_load).
 * tm_vocab_block_export describes the block of `v` (plain data: send it as bytes) and returns its device pointer; tm_vocab_block_import makes
 * an empty vocabulary of that shape on `device` and returns the device pointer the caller has to fill with the exporter's `bytes` bytes
 * before the first use.  An imported vocabulary tokenizes, counts, scores and decodes on the device; it has no host tables (tm_vocab_image /
 * tm_vocab_save and the streaming decoder fail). */
typedef struct tm_vocab_block {
  uint64_t bytes;            /* bytes the tables occupy from the block's start (what has to be sent) */
  uint64_t part_bytes[8];    /* root, walk tables, rows, space-prefix links, node values, reverse offsets, reverse bytes, begin_byte */
  uint32_t idle_off, n_da, n_info, max_len, off, bstart, spl_hint, link_off, direct_off, delete_id, unk_id;
  uint32_t n_ids, vocab_size, capcode, charset, norm_flag, level, reserve, n_nodes, pad;   /* pad: TM_VOCAB_BLOCK_FORMAT of the exporting build */
} tm_vocab_block;
#define TM_VOCAB_BLOCK_FORMAT 6u   /* layout of the device tables inside a block (tm_tables.h); an importer refuses any other */
int tm_vocab_block_export(const tm_vocab* v, tm_vocab_block* meta, void** device_ptr);
int tm_vocab_block_import(const tm_vocab_block* meta, int device, tm_vocab** out, void** device_ptr);
/* Synchronous device-to-device copy (also between two devices of the node with peer access), for callers that have no HIP binding of
 * their own: e.g. to fill an imported block from the exporter's pointer inside one process. */
int tm_device_copy(void* dst_device, const void* src_device, uint64_t bytes);
uint32_t tm_vocab_size(const tm_vocab* v);             /* go :2477 Len()              */
uint32_t tm_vocab_n_info(const tm_vocab* v);           /* index records incl. "D " duplicates */
uint32_t tm_vocab_n_ids(const tm_vocab* v);            /* len(reverse) = highest ID + 1 */
uint32_t tm_vocab_max_token_length(const tm_vocab* v); /* go :2498                    */
uint32_t tm_vocab_capcode(const tm_vocab* v);          /* go :2390                    */
uint32_t tm_vocab_charset(const tm_vocab* v);
uint32_t tm_vocab_normalization(const tm_vocab* v);    /* normalizer flag byte        */
uint32_t tm_vocab_unk(const tm_vocab* v);              /* unk id or TM_NONE           */
uint32_t tm_vocab_delete_token(const tm_vocab* v);     /* deleteToken id or TM_NONE   */
uint64_t tm_vocab_device_bytes(const tm_vocab* v);     /* HBM held by the tables      */

/* ---- batch tokenize, host buffers: drop-in for the goroutine fan-out ------------------------- */
/* Tokenizes `ndocs` independent documents.  Document d is text[offsets[d] .. offsets[d+1]).
 * tokens_out receives all IDs, document after document; tok_offsets[ndocs+1] the prefix sums;
 * missing[d] the count go/tokenmonster.go:1274 returns.  If tokens_cap is too small nothing is
 * written to tokens_out, tok_offsets IS filled (so tok_offsets[ndocs] is the required capacity)
 * and TM_E_NOSPACE is returned.  Equivalent of calling Vocab.tokenize on each document. */
int tm_tokenize_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs,
                      uint32_t* tokens_out, uint64_t tokens_cap, uint64_t* tok_offsets, uint32_t* missing);

/* Same walk, counts only: go/tokenmonster.go:1281 tokenizeCount semantics (b-branches count 1,
 * quirk Q2).  counts[ndocs], missing[ndocs]. */
int tm_count_batch(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs,
                   uint64_t* counts, uint32_t* missing);

/* Vocab.Count on RAW text (go :971: normalize, then tokenizeCount): what tokenmonsterserver job 20 calls per document
 * (training/tokenmonsterserver.go:753-800).  The normalization runs on the device. */
int tm_count_batch_raw(const tm_vocab* v, const uint8_t* raw, const uint64_t* offsets, uint32_t ndocs,
                       uint64_t* counts, uint32_t* missing);

/* TokenizeToSerialized (go/tokenmonster.go:986): encoding_length 2, 3 or 4 bytes little-endian per
 * ID (0 = auto: 2 if n_ids <= 65536 else 3, go :990-996).  bytes_out/byte_offsets as above. */
int tm_tokenize_batch_serialized(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs,
                                 uint32_t encoding_length, uint8_t* bytes_out, uint64_t bytes_cap,
                                 uint64_t* byte_offsets, uint32_t* missing, uint32_t* encoding_length_used);

/* ---- large batches, host to host --------------------------------------------------------------------------------- */
/* The three calls above borrow one LANE of the vocabulary (a HIP stream + a grow-only device workspace): steady state does
 * no allocation and never touches the NULL stream, and concurrent callers (cgo calls run on distinct OS threads) take
 * different lanes and overlap on the device; at most TM_LANES (environment, default 8) calls run at once, further callers wait.
 *
 * tm_tokenize_pipeline is the large-batch form of Vocab.TokenizeToSerialized over many documents: the corpus is cut into
 * chunks of whole documents (about chunk_bytes, 0 = 32 MiB; the first two are smaller) that run H2D | normalize + tokenize + serialize | D2H on `lanes`
 * lanes at once (0 = 4): a lane uploads its next chunk while the current one computes, so that PCIe moves text in and ids out behind the kernels.  raw != 0: text is
 * RAW UTF-8 and is normalized on the device (go/tokenmonster.go:242-253); raw == 0: already normalized.  Output: ids of all
 * documents back to back, encoding_length bytes each, little-endian (0 = automatic, go :990-996); byte_offsets[ndocs+1] is
 * always filled, so on TM_E_NOSPACE byte_offsets[ndocs] is the capacity required.  Buffers from tm_host_alloc (or registered
 * with tm_host_register) are DMA'd directly; pageable buffers go through pinned staging at memcpy speed. */
typedef struct tm_pipeline_stats {
  uint32_t chunks, lanes;
  int input_pinned, output_pinned;
  uint64_t normalized_bytes;
  uint32_t host_fallback_docs;
  uint32_t ring;              /* 1: the call ran on the ring (raw text, page-locked buffers on both sides: no host round trip inside a chunk) */
  uint32_t ring_exact_chunks; /* chunks the ring handed to the exact path (documents for the host normalizer, a long document ...) */
} tm_pipeline_stats;
int tm_tokenize_pipeline(const tm_vocab* v, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, int raw,
                         uint32_t encoding_length, uint64_t chunk_bytes, uint32_t lanes, uint8_t* bytes_out, uint64_t bytes_cap,
                         uint64_t* byte_offsets, uint32_t* missing, uint32_t* encoding_length_used, tm_pipeline_stats* stats);
/* Page-locked host memory for the buffers of tm_tokenize_pipeline (hipHostMalloc / hipHostRegister).  hipHostMalloc places it on the NUMA
 * node nearest to the current device; the pipeline's worker threads (the calling thread included, for the length of the call) run on that
 * node's CPUs (TM_NUMA=0 in the environment: leave the threads where they are).  tm_device_numa_node: the node of a device's PCIe root as
 * /sys/bus/pci/devices/<bdf>/numa_node gives it, -1 if the host does not say - what a caller pins its own feeding threads to. */
int tm_device_numa_node(int device);
void* tm_host_alloc(size_t bytes);
void tm_host_free(void* p);
int tm_host_register(void* p, size_t bytes);
int tm_host_unregister(void* p);

/* ---- batch tokenize, device-resident: what bench.py times ------------------------------------ */
/* A tm_batch owns device buffers sized for up to max_bytes of text in up to max_docs documents. */
int tm_batch_create(const tm_vocab* v, uint64_t max_bytes, uint32_t max_docs, tm_batch** out);
void tm_batch_free(tm_batch* b);
/* H2D of packed text + offsets (synchronous). */
int tm_batch_upload(tm_batch* b, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs);
/* Raw (un-normalized) input: H2D of packed raw UTF-8 documents, then tm_batch_normalize runs the pre-step of
 * Tokenize (go/tokenmonster.go:242-253: norm.Normalize + capcode.Encode) ON THE DEVICE into the batch's text buffer
 * (max_bytes of tm_batch_create must cover the normalized size, about 1.1x raw with capcode 2).  The device pass handles ASCII, every
 * two-byte script (U+0080..U+07FF: accented Latin, Greek, Cyrillic, Armenian, Hebrew, Arabic ...: NFD, case and capcode from a table the
 * host normalizer fills), Latin Extended Additional under NFD (U+1E00..U+1EFF, what Vietnamese adds: a letter and one or two marks each), the
 * three-byte characters the normalizer leaves alone (General Punctuation, CJK ideographs, kana, symbols), the voiced kana under NFD (a kana and
 * U+3099 / U+309A each: Japanese text), Hangul syllables (decomposed by arithmetic) and the four-byte characters of caseless, NFD-stable blocks
 * (emoji, symbols, plane-2 ideographs), the three-byte combining marks of U+0800..U+1FFF where they stand in canonical order (virama, nukta, the
 * Thai tone marks ...: Hindi, Thai) and the three-byte decimal digits and lower-case letters (Georgian); documents with anything else (cased scripts beyond the BMP, a capital
 * without a lower-case form - U+03D2..U+03D4 -, a character of three bytes that NFD splits in three, marks out of canonical order or behind a character that
 * ends in one of its own, malformed UTF-8) are normalized by the host normalizer inside the same call, from their original
 * bytes; tm_batch_host_fallback_docs reports how many.  (TM_NORM_WG_PER_CU in the environment: the grid of the pass, workgroups per compute
 * unit, default 64 - a tuning knob, profiles/r05_issue_model.txt.)
 * Supported: capcode 0 and 2 (level 1 has no statement in the reference tree and is refused) and every normalization flag
 * (training/README.md:110-123) ON THE DEVICE: NFD and lowercase in the pass itself; quotemarks, collapse, trim, leadingspace and unixlines -
 * and what accents does to the two-byte characters - in a filter pass in front of it that states the reference's in-place loops
 * (tokenmonster.cpp:245-425) per byte, their quirks included (tm_norm.hip: k_pf_*: a wavefront per document, one sweep; one more trip to the
 * host, for the count of the filtered documents' pieces; ~1.7 ms per GiB on top of the pass).  tm_batch_normalize synchronizes `stream`; afterwards tm_batch_run tokenizes the normalized
 * documents.  (Where the normalized text lies between the two calls is the library's business: when every document was normalized on the
 * device it stays in the normalizer's per-piece slabs and the match kernel reads it from there - no packing pass; tm_batch_download_text
 * packs it on request and returns it in document order either way.) */
int tm_batch_upload_raw(tm_batch* b, const uint8_t* raw, const uint64_t* raw_offsets, uint32_t ndocs);
int tm_batch_normalize(tm_batch* b, void* stream);
uint64_t tm_batch_normalized_bytes(const tm_batch* b);
uint32_t tm_batch_host_fallback_docs(const tm_batch* b);
int tm_batch_download_text(tm_batch* b, uint8_t* text_out, uint64_t text_cap, uint64_t* offsets_out);
/* Runs the whole device pipeline (segments, match+branch+link, resolve, scan, emit) on `stream` (a hipStream_t, NULL =
 * default stream).  Inputs and outputs stay in HBM.  Asynchronous with respect to the host. */
int tm_batch_run(tm_batch* b, void* stream);
/* As tm_batch_run but brackets every kernel with HIP events on `stream` and returns per-kernel
 * milliseconds in ms[TM_NUM_KERNELS] (synchronizes). */
#define TM_NUM_KERNELS 5
int tm_batch_run_timed(tm_batch* b, void* stream, float* ms);
const char* tm_kernel_name(int k);
/* Test hooks: sets the switches and returns the previous value; flags < 0 only queries.  Every bit forces a rarely taken path of
 * the product with the SAME results, so that the tests can cover it: 6 = dense T(p,1) array for every segment, 8 = per-lane
 * normalizer kernel instead of k_norm_emit2, 10 = K4 tile walk that stores every id directly (its overflow path), 11 = the device normalizer packs its text
 * (the path of a batch with host-normalized documents) instead of leaving it in its slabs for the match kernel, 12 = group tree of
 * long documents with fan-out 4 from 9 segments on, 13 = 64 KiB mailbox for the small host <-> device transfers, 14 = the last member of tm_score_multi gives up
 * after the members' first meeting (an error path: the call must return that member's error), 15 = the id-staging form of the K4 walk (what vocabularies
 * of more than 65 536 ids use) for the two-plane rows too, instead of the position-staging form.  Other bits are
 * ignored (a -DTM_DEVEL build, tools/ only, adds profiling bits that switch phases of the match kernel off).  The switches are process-wide,
 * so they are armed only in a process started with TM_TEST_HOOKS in its environment (the test suite, bench.py --also-flags): anywhere
 * else the call changes nothing and returns 0 - one caller of a server cannot change the code path under the others. */
int tm_debug_flags(int flags);
/* Totals of the last run (synchronizes the stream used by the last run). */
int tm_batch_totals(tm_batch* b, uint64_t* total_tokens, uint64_t* total_missing);
/* D2H of results of the last run. */
int tm_batch_download(tm_batch* b, uint32_t* tokens_out, uint64_t tokens_cap, uint64_t* tok_offsets,
                      uint32_t* missing);
/* Raw device pointers of the last run's results (valid until the next run/free).  The id buffer starts at max_bytes / 2 + 2 * max_docs
 * + 1024 ids and grows on demand: call tm_batch_totals (or tm_batch_download) after tm_batch_run and BEFORE fetching these pointers —
 * it synchronizes, and if the run produced more ids than the buffer held it reallocates the buffer and repeats the emit stage, so a
 * pointer fetched earlier may dangle and ids beyond the old capacity were not written. */
const uint32_t* tm_batch_device_tokens(const tm_batch* b);
const uint64_t* tm_batch_device_tok_offsets(const tm_batch* b);
uint64_t tm_batch_device_bytes(const tm_batch* b);

/* ---- decode: Decode / decode_raw (go/tokenmonster.go:445-550; tokenmonster.cpp:1404-1425) ------------------------ */
/* ids of document d = tokens[tok_offsets[d] .. tok_offsets[d+1]).  The gather of reverse[id] (byte counts of tiles of 2048 ids -> scan -> gather
 * through LDS) runs on the device; ids >= tm_vocab_n_ids are skipped.  raw != 0: the concatenated token bytes as they are
 * (decode_raw); raw == 0: capcode decoding (javascript/tokenmonster.js:1007-1065) follows — on the device for the documents of a capcode-2
 * UTF-8 vocabulary made of ASCII, the two-byte scripts (U+0080..U+07FF), the three-byte characters without case (punctuation, CJK, kana,
 * Hangul, symbols) and the four-byte characters of blocks that are caseless throughout (emoji, symbols, the ideographs of plane 2); on the
 * host for documents with a letter whose upper-case form has another length, a three- or four-byte letter with case or malformed UTF-8, and
 * for capcode 1.  tm_decode_host_docs(): how many documents of the calling thread's last tm_decode_batch went to the host decoder.
 * out_offsets[ndocs+1] is always filled; TM_E_NOSPACE if out_cap is too small (required size in out_offsets[ndocs]).
 * Like the tokenize entry points the call borrows a lane of the vocabulary (its stream, grow-only device arenas and pinned
 * staging): callable concurrently, no allocation in steady state, nothing on the NULL stream. */
int tm_decode_batch(const tm_vocab* v, const uint32_t* tokens, const uint64_t* tok_offsets, uint32_t ndocs, int raw,
                    uint8_t* out, uint64_t out_cap, uint64_t* out_offsets);
uint32_t tm_decode_host_docs(void);
/* The same on the ids a batch HOLDS after tm_batch_run, device-resident: ids in HBM -> text in HBM, in grow-only buffers of the batch (what
 * bench.py --workload decode times; a service that detokenizes what it has just tokenized never moves the ids over the host link).
 * Synchronizes `stream`.  *decoded_bytes: bytes of the documents the device decoded; *host_docs: documents it left to the host decoder
 * (those are decoded by tm_batch_decoded_download, which returns every document's text in document order - out_offsets[ndocs+1] always
 * filled, TM_E_NOSPACE with the size required in out_offsets[ndocs] if out_cap is too small). */
int tm_batch_decode(tm_batch* b, int raw, void* stream, uint64_t* decoded_bytes, uint32_t* host_docs);
/* ... with HIP events on `stream` around its stages: ms[0] byte counts of the tiles + scan, ms[1] the gather of the tokens' bytes and the documents'
 * offsets (k_dec_gather), ms[2] capcode decoding (k_dec_capcode; 0 when that is not the device's) - what bench.py's decode line prices its roofline on */
int tm_batch_decode_timed(tm_batch* b, int raw, void* stream, uint64_t* decoded_bytes, uint32_t* host_docs, float* ms);
int tm_batch_decoded_download(tm_batch* b, uint8_t* out, uint64_t out_cap, uint64_t* out_offsets);

/* Streaming Decoder (go/tokenmonster.go:552-700 NewDecoder / Decode / DecodeSerialized / Flush; server jobs 5-9): ids arrive a few
 * at a time, a call returns the text that is COMPLETE so far; the bytes of a character that is not (a token may end in the middle of a
 * UTF-8 sequence) and the state of the capcode decoder are carried to the next call.  Per-connection host state: the gather of a
 * handful of ids runs on the host copy of the reverse table.  out_len receives the number of bytes decoded; on TM_E_NOSPACE
 * (out_cap too small) the ids HAVE been consumed and the text is kept: call again with n = 0 and a buffer of *out_len bytes.
 * tm_decoder_flush returns (and forgets) the held-back bytes.
 * Quirk kept from the reference: the number of bytes held back is incompleteUTF8Bytes' return value, which for a character that
 * is cut is the number of bytes still MISSING (go/tokenmonster.go:183-185), not the number present; where that exceeds the
 * buffer (Go panics there) the whole buffer is held back. */
int tm_decoder_new(const tm_vocab* v, tm_decoder** out);
void tm_decoder_free(tm_decoder* d);
int tm_decoder_decode(tm_decoder* d, const uint32_t* tokens, uint64_t n, uint8_t* out, uint64_t out_cap, uint64_t* out_len);
int tm_decoder_decode_serialized(tm_decoder* d, const uint8_t* data, uint64_t nbytes, uint32_t encoding_length, uint8_t* out,
                                 uint64_t out_cap, uint64_t* out_len);
int tm_decoder_flush(tm_decoder* d, uint8_t* out, uint64_t out_cap, uint64_t* out_len);

/* ---- trainvocab scoring pass: replaces training/trainvocab.go:925-1176 ------------------------ */
/* Upload the normalized dataset once (trainvocab.go:1660-1665 keeps it for the whole run). */
int tm_dataset_upload(const uint8_t* normalized, uint64_t n, tm_dataset** out);
int tm_dataset_upload_on(const uint8_t* normalized, uint64_t n, int device, tm_dataset** out);   /* on a named device */
int tm_dataset_device(const tm_dataset* d);                                                       /* the device it lives on */
void tm_dataset_free(tm_dataset* d);
/* Walks each strip [strip_off[k], strip_off[k]+strip_len[k]) of the dataset exactly as the worker
 * does and accumulates scores[id] += bytes covered, scores[deleteToken] += 1 per forward-delete,
 * *tokens_in_text, and the 256-bit set of bytes that had no token.  scores has tm_vocab_n_ids()
 * entries and is OVERWRITTEN.  n_strips == 0 means one strip = the whole dataset. */
int tm_score(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len,
             uint32_t n_strips, uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]);
/* Device-resident variant for multi-GPU reduction: leaves the histogram in HBM and returns its
 * device pointer, *n_words = n_ids + 4 + 256 uint32: scores[n_ids] | tokens_in_text as four 16-bit limbs |
 * missing[256] per-byte counters - every word is a plain sum over ranks, so ONE RCCL all-reduce(sum, uint32)
 * merges the partial results of data-parallel ranks. */
int tm_score_device(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len,
                    uint32_t n_strips, void* stream, uint32_t** dev_hist, uint64_t* n_words);

/* As tm_score_device, but also copies the histogram (device to device, on `stream`) into a caller-owned device
 * buffer of dst_words >= n_ids + 4 + 256 uint32 — e.g. the storage of a torch tensor that is then all-reduced. */
int tm_score_device_into(const tm_vocab* v, tm_dataset* d, const uint64_t* strip_off, const uint64_t* strip_len,
                         uint32_t n_strips, void* stream, uint32_t* dst_device, uint64_t dst_words);

/* ---- one whole-buffer walk over several GPUs (training/trainvocab.go:909-922: after "midway" the worker walks the dataset as ONE
 * strip) --------------------------------------------------------------------------------------------------------------------
 * Rank r owns the bytes [off, off+len) of the dataset and has uploaded them followed by a halo of >= 128 bytes of the text that
 * comes next (continues != 0; the last rank has none).  The walk's state at a byte is (offset of the next token start, pending
 * forward-delete flag) = one of 80 ENTRY STATES (2 * offset + flag, offset < 40), and what a range does to it is a map of 80 entries:
 *   tm_score_begin   runs the match kernel over the range and returns exits[80]: exits[e] = the entry state of the NEXT range if this
 *                    one is entered in state e (0xFF: e cannot occur).  Ranks all-gather their 80 bytes; rank r's true entry state is
 *                    exits[r-1][ exits[r-2][ ... exits[0][0] ] ].
 *   tm_score_finish  completes the pass from that entry state: histogram exactly as tm_score_device[_into] leaves it (dst_device may be
 *                    NULL).  A token that begins inside the range is counted by this rank even if it ends in the halo.
 * Summed over the ranks (one all-reduce) the histograms equal tm_score of the whole dataset as one strip, bit for bit.
 * tm_score_read copies the histogram of the last pass to the host in tm_score's form.
 * continues: 0 = the text ends with the range; 1 = more text follows and the dataset holds >= 128 bytes of it behind the range (fewer:
 * TM_E_INVALID - the exit states would silently differ from the whole-buffer walk's); 2 = text follows and ALL of it is in the dataset
 * (it ends inside the halo).  One caller per dataset between the two halves of a pass. */
int tm_score_begin(const tm_vocab* v, tm_dataset* d, uint64_t off, uint64_t len, int continues, void* stream, uint8_t* exits);
int tm_score_finish(const tm_vocab* v, tm_dataset* d, uint32_t entry_state, void* stream, uint32_t* dst_device, uint64_t dst_words);
int tm_score_read(const tm_vocab* v, tm_dataset* d, uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]);

/* ---- several devices of one node behind one handle: the multi-GPU half of the path for a host that stays ONE process ------------------------
 * The reference's parallelism is in-process: trainvocab starts `workers` goroutines over one dataset (training/trainvocab.go:1827-1829; after
 * "midway" each walks it as ONE strip, :909-922), the server fans a job's documents out over goroutines (training/tokenmonsterserver.go:363-378,
 * :773-787).  tm_devices names the GPUs; every call below drives all of them from inside the library, one host thread per device, and the ONE
 * collective of the path - the sum of the scoring pass's histograms - is an ncclAllReduce(sum, uint32) over xGMI inside tm_score_multi.
 * RCCL is loaded at the first collective (librccl.so.1 is 570 MB: the single-GPU entry points never map it).
 *
 * tm_devices_open: the first max_devices visible devices (<= 0: all).  In a test process (TM_TEST_HOOKS set, like tm_debug_flags) TM_VIRTUAL_DEVICES=N
 * gives the handle N members that all sit on device 0 instead - the multi-device code paths on a one-GPU box; tm_devices_open_list names the devices itself (a device may appear
 * more than once).  Members that share a device cannot form an RCCL communicator (RCCL refuses two ranks on one device): there, with TM_RCCL=0, and
 * where librccl cannot be loaded, member 0 sums the histograms by peer copies and an add kernel - same result.  tm_devices_rccl_ranks: the
 * number of ranks of the communicator the handle uses (it is made on the first call of this or of tm_score_multi), 0 and *why_not = the
 * reason if there is none.  One-member handles skip the collective unless TM_RCCL=1.  Every call of this section leaves the calling thread's
 * current device as it found it. */
typedef struct tm_devices tm_devices;
typedef struct tm_vocab_set tm_vocab_set;       /* one replica of a vocabulary per member */
typedef struct tm_dataset_set tm_dataset_set;   /* one byte range of a normalized dataset per member */
int tm_devices_open(int max_devices, tm_devices** out);
int tm_devices_open_list(const int* devices, int n, tm_devices** out);
int tm_devices_count(const tm_devices* g);
int tm_devices_device(const tm_devices* g, int member);
int tm_devices_rccl_ranks(tm_devices* g, const char** why_not);
void tm_devices_close(tm_devices* g);            /* after every vocabulary set and dataset set made from it has been freed */
/* Load for all members: the tables are built once (tm_vocab_load on member 0) and the finished device block is copied device to device into
 * every other member (tm_vocab_block_export / _import).  tm_vocab_set_member(s, 0) is a full vocabulary (decoder, image); the other members
 * have device tables only.  The handles stay owned by the set. */
int tm_vocab_load_all(tm_devices* g, const uint8_t* vocab_file, size_t n, tm_vocab_set** out);
int tm_vocab_set_count(const tm_vocab_set* s);
const tm_vocab* tm_vocab_set_member(const tm_vocab_set* s, int member);
void tm_vocab_set_free(tm_vocab_set* s);
/* tm_vocab_tune for every member: the tables are laid out again once (member 0) and the block goes to the others as at tm_vocab_load_all. */
int tm_vocab_set_tune(tm_vocab_set* s, const uint8_t* normalized_sample, uint64_t n);
/* tm_tokenize_pipeline over every device (the server's fan-out, tokenmonsterserver.go:363-378): the chunks of whole documents are handed to
 * lanes_per_device lanes (0 = 4) of EVERY member from one queue, so a faster or less loaded device takes more of them; ids land in document
 * order whichever device computed them.  No collective.  Arguments and results exactly as tm_tokenize_pipeline. */
int tm_tokenize_pipeline_multi(const tm_vocab_set* s, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, int raw,
                               uint32_t encoding_length, uint64_t chunk_bytes, uint32_t lanes_per_device, uint8_t* bytes_out, uint64_t bytes_cap,
                               uint64_t* byte_offsets, uint32_t* missing, uint32_t* encoding_length_used, tm_pipeline_stats* stats);
/* The dataset of a training run cut into one contiguous byte range per member (multiples of 4 bytes, trainvocab.go:1674), each uploaded with
 * the 128 bytes that follow it (tokens straddle the cuts); a dataset of less than 4 KiB per member uses fewer members.
 * tm_dataset_set_range: bytes of a member's range, *halo_bytes (may be NULL) the bytes of following text it holds as well. */
int tm_dataset_upload_sharded(tm_devices* g, const uint8_t* normalized, uint64_t n, tm_dataset_set** out);
uint64_t tm_dataset_set_range(const tm_dataset_set* s, int member, uint64_t* halo_bytes);
void tm_dataset_set_free(tm_dataset_set* s);
/* One scoring pass over the WHOLE dataset as one strip (trainvocab.go:909-922), bit-identical to tm_score(v, whole dataset, n_strips = 0) on one
 * device: every member runs the match kernel on its range, the 80-entry exit maps of all ranges are gathered on every device (ncclAllGather
 * of 80 bytes per member; peer copies where there is no communicator) and chained there into the member's entry state, the histogram walk
 * follows on the same stream, and one all-reduce(sum) of the n_ids + 4 + 256 uint32 histogram words merges them: a member's host thread
 * enqueues its whole half of the pass and waits once.  Results as tm_score. */
int tm_score_multi(const tm_vocab_set* vs, tm_dataset_set* ds, uint32_t* scores, uint64_t* tokens_in_text, uint8_t missing_set[32]);

#ifdef __cplusplus
}
#endif
#endif
