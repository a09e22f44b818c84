// tm_pipeline.h — shared by the HIP translation units of libtokenmonster_hip.so: constants of the segment pipeline,
// the tm_batch workspace behind the opaque handle, and the host-side entry points one unit offers the others.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <vector>

#include "tm_device.h"
#include "tm_norm_masks.h"

namespace tmh {

constexpr int SEG = 256;                 // bytes of one document segment (one wavefront); 256 -> 28 wavefronts per CU (LDS)
constexpr int NPOS = SEG + 40;           // positions whose descriptors a segment needs (look-ahead <= 40)
constexpr int NPOS_PAD = (NPOS + 63) / 64 * 64;
constexpr int TEXT_LEN = SEG + 96;       // staged text: position i may read up to i + 40
constexpr int SLAB_BYTES = 2048;         // the device normalizer's slab for one 1 KiB piece of raw text (tm_norm.hip); K1 can stage the text from there
constexpr int ENT = 80;                  // entry states of a segment: 40 offsets x fd{0,1}
#ifndef TM_K1_WAVES
#define TM_K1_WAVES 4
#endif
constexpr int WAVES = TM_K1_WAVES;                 // wavefronts per workgroup in K1 (8: +14 %, the halo barrier couples more wavefronts)
constexpr uint32_t R_INVALID = 0xFFFFFFFFu;
constexpr uint32_t J_EXIT = 1024;           // jump targets >= J_EXIT: left the segment; J_EXIT + next entry state
constexpr uint32_t J_INVALID = 2047;        // state is not reachable (no forward-delete match there)
constexpr uint32_t ID_NONE = 0xFFFFFFu;
constexpr int NOSCORE = -1000000;
constexpr int SIDE_STRIDE = 16;            // uint2 entries of a segment's side list: [0] = {count or SIDE_DENSE, 0}, then {position, T(p,1)}
constexpr uint32_t SIDE_DENSE = 0xFFFFFFFFu;  // more (p,1) states than the list holds: they are in the dense R1 array instead
constexpr uint32_t LONG_SEGS = 512;        // documents with more segments than this are resolved hierarchically

constexpr int SCAN_T = 256, SCAN_PER = 16, SCAN_CH = SCAN_T * SCAN_PER;   // exclusive scan u32 -> u64: elements per workgroup

// segment groups of long documents (hierarchical resolve, tm_kernels.hip)
constexpr uint32_t GROUP_FAN = 64;         // children per group of the tree over a long document's segments
struct Group { uint32_t first_child, nchildren, doc, level; };   // children: segments (level 0 = leaf groups) or groups of the level below
struct LongDoc { uint32_t doc, first_group, ngroups, pad; };

// per-lane select on a wave-uniform lane mask: bit set -> a, else b (one v_cndmask, the mask stays in scalar registers)
__device__ __forceinline__ uint32_t sel_mask(unsigned long long mask, uint32_t a, uint32_t b) {
#ifdef TM_EMU
  return ((mask >> emu::cur->lane) & 1ull) ? a : b;
#else
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
  return r;
#endif
}
// bit (n & 31) of x: v_bfe_u32 takes the low five bits of its offset operand by itself, the compiler does not drop the `& 31` of the C form
__device__ __forceinline__ uint32_t bit_of(uint32_t x, uint32_t n) {
#ifdef TM_EMU
  return (x >> (n & 31u)) & 1u;
#else
  return __builtin_amdgcn_ubfe(x, n, 1u);
#endif
}
// a + (bit of the lane in a wave-uniform mask): the mask goes in as the carry of ONE add (v_addc_co), instead of an add and a select
__device__ __forceinline__ uint32_t add_mask_bit(uint32_t a, unsigned long long mask) {
#ifdef TM_EMU
  return a + (uint32_t)((mask >> emu::cur->lane) & 1ull);
#else
  uint32_t r;
  unsigned long long carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(a), "s"(mask));
  return r;
#endif
}
// the value a wave-uniform lane holds, as a scalar (v_readlane)
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int l) {
#ifdef TM_EMU
  return __shfl(v, l);
#else
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
#endif
}
// The value another lane holds as an operand of this lane's next instruction (a DPP move: no LDS round trip).  ctrl: 0x110 + n = row_shr:n
// (rows of 16 lanes), 0x138 = wave_shr:1, 0x130 = wave_shl:1, 0x142 / 0x143 = row_bcast:15 / row_bcast:31; rows: which rows of 16 lanes take part.
// A lane without a source - shifted in from outside its row / the wavefront, or in a row that does not take part - keeps `old`.
#ifdef TM_EMU
static inline uint32_t tm_dpp_emu(uint32_t old, uint32_t src, int ctrl, int rows) {
  const int lane = (int)emu::cur->lane, row = lane >> 4;
  int from = lane; bool have = false;
  if (ctrl > 0x110 && ctrl <= 0x11F) { from = lane - (ctrl - 0x110); have = (lane & 15) >= ctrl - 0x110; }
  else if (ctrl == 0x138) { from = lane - 1; have = lane >= 1; }
  else if (ctrl == 0x130) { from = lane + 1; have = lane < 63; }
  else if (ctrl == 0x142) { from = row * 16 - 1; have = row >= 1; }
  else if (ctrl == 0x143) { from = 31; have = row >= 2; }
  const uint32_t v = __shfl(src, from & 63);
  return (have && ((rows >> row) & 1)) ? v : old;
}
#define TM_DPP(old, src, ctrl, rows) tm_dpp_emu(old, src, ctrl, rows)
#define TM_DPP0(src, ctrl) tm_dpp_emu(0u, src, ctrl, 0xF)
#else
#define TM_DPP(old, src, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, rows, 0xF, false))
#define TM_DPP0(src, ctrl) ((uint32_t)__builtin_amdgcn_mov_dpp((int)(src), ctrl, 0xF, 0xF, true))      // all rows; a lane without a source gets 0 (bound_ctrl: no `old` to set up)
#endif
// A wavefront's window on global memory: a buffer resource (base and size wave-uniform), so that the range check is the hardware's - a dword
// load outside [base, base + bytes) returns 0 and costs no branch and no 64-bit compare (offsets are unsigned: "before the window" is
// outside too).  `bytes` a multiple of 4 and offsets multiples of 4, so that no dword is partly inside.
#ifdef TM_EMU
struct TmWindow { const uint8_t* base; uint32_t bytes; };
static inline TmWindow tm_window(const void* base, uint32_t bytes) { return TmWindow{(const uint8_t*)base, bytes}; }
static inline uint32_t tm_window_u32(const TmWindow& w, uint32_t off) { uint32_t v = 0; if (off < w.bytes && w.bytes - off >= 4u) __builtin_memcpy(&v, w.base + off, 4); return v; }
static inline uint4 tm_window_u128(const TmWindow& w, uint32_t off) { return uint4{tm_window_u32(w, off), tm_window_u32(w, off + 4), tm_window_u32(w, off + 8), tm_window_u32(w, off + 12)}; }
#else
typedef __amdgpu_buffer_rsrc_t TmWindow;
__device__ __forceinline__ TmWindow tm_window(const void* base, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ uint32_t tm_window_u32(TmWindow w, uint32_t off) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(w, (int)off, 0, 0); }
__device__ __forceinline__ uint4 tm_window_u128(TmWindow w, uint32_t off) {      // four dwords, each range-checked by itself
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(w, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
#endif
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long mask, uint32_t acc) {      // acc + #set bits of mask below the lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, acc));
}

}  // namespace tmh

struct tm_batch {
  const tm_vocab* vocab = nullptr;
  uint64_t max_bytes = 0;
  uint32_t max_docs = 0;
  uint64_t max_segs = 0;
  uint64_t nbytes = 0, nseg = 0;
  uint32_t ndocs = 0;
  uint64_t device_bytes = 0;
  hipStream_t last_stream = nullptr;
  hipStream_t aux_stream = nullptr;    // fetch of the host-fallback documents, beside the normalizer pass
  // device buffers
  uint8_t* d_text = nullptr;
  bool text_borrowed = false;          // scoring pass: text belongs to a tm_dataset
  uint64_t* d_offsets = nullptr;       // packed batches: doc_begin = d_offsets, doc_end = d_offsets + 1
  const uint64_t* d_doc_begin = nullptr;
  const uint64_t* d_doc_end = nullptr;
  const uint64_t* d_doc_vis = nullptr;   // how far a document may look at the text (null: its end); see k_match_branch
  const uint8_t* d_doc_entry = nullptr;  // entry state of a document's first segment (null: 0); see k_resolve
  uint32_t* d_doc_nseg = nullptr;
  uint64_t* d_doc_seg_start = nullptr;
  uint32_t* d_seg_doc = nullptr;
  uint32_t* d_R0 = nullptr;          // T(p,0): SEG words per segment (position p of segment g at g * SEG + p)
  uint2* d_side = nullptr;           // per segment: its few T(p,1) words (SIDE_STRIDE entries: header + {position, word})
  uint32_t* d_R1 = nullptr;          // T(p,1) per position, written only for segments whose side list overflows
  uint32_t* d_exitmap = nullptr;       // per segment: 80 entries {next entry state | #ids << 8}: written only for a segment with a count that does not fit the 16-bit form
  uint16_t* d_exit16 = nullptr;        // per segment: 80 entries {next entry state | #ids << 7}, 0xFFFF = unreachable (exit_entry, tm_kernels.hip)
  uint8_t* d_seg_entry = nullptr;
  uint32_t* d_seg_tokbase = nullptr;
  uint4* d_seg_par = nullptr;          // per segment: begin | length | entry state | first output index (k_seg_params)
  uint32_t* d_doc_ntok = nullptr;
  uint32_t* d_doc_events = nullptr;
  uint32_t* d_doc_missing = nullptr;
  uint32_t* d_doc_fd = nullptr;        // delete tokens emitted per document (K4)
  uint64_t* d_tok_offsets = nullptr;
  uint64_t* d_scan_tmp = nullptr;   // block sums
  uint64_t* d_totals = nullptr;     // [0] nseg total (device-computed), [1] token total, [2] missing total
  uint32_t* d_error = nullptr;
  // long documents (hierarchical resolve)
  uint32_t ngroups = 0, nlong = 0, cap_groups = 0, cap_long = 0;
  std::vector<uint32_t> level_first;   // groups of level k+1 are d_groups[level_first[k] .. level_first[k+1])
  tmh::Group* d_groups = nullptr;
  tmh::LongDoc* d_longs = nullptr;
  uint2* d_gmap = nullptr;
  uint8_t* d_group_entry = nullptr;
  uint32_t* d_group_base = nullptr;
  // raw (un-normalized) input of tm_batch_upload_raw / tm_batch_normalize
  uint8_t* d_raw = nullptr;
  // the filter pass of a vocabulary with byte-level normalization flags (tm_norm.hip: k_pf_*): the filtered text and its documents
  uint8_t* d_rawf = nullptr; uint64_t rawf_cap = 0;
  uint64_t* d_rawf_off = nullptr; uint64_t rawf_docs_cap = 0;
  void* d_pf_piece = nullptr; uint64_t pf_piece_cap = 0;
  void* d_pf_doc = nullptr;
  uint32_t* d_acc = nullptr;             // what `accents` leaves of the two-byte characters
  uint8_t* d_slab = nullptr;            // normalizer: one 2 KiB slab per 1 KiB piece
  uint64_t slab_pieces = 0;             // pieces of the batch that was normalized last (raw_pieces may already count the next upload)
  bool text_in_slabs = false;           // the normalized text of this batch has not been packed into d_text: K1 stages it from the slabs (k_seg_src)
  uint64_t slab_cap = 0;
  uint64_t* d_raw_off = nullptr;
  uint64_t raw_cap = 0, raw_bytes = 0, raw_pieces = 0, piece_cap = 0;
  uint32_t raw_docs = 0, raw_docs_cap = 0;
  std::vector<uint64_t> h_raw_off;
  uint32_t host_fallback_docs = 0;
  uint32_t* d_doc_npiece = nullptr;     // pieces per raw document, then (scan) first piece of each document
  uint64_t* d_doc_piece_start = nullptr;
  uint32_t* d_piece_doc = nullptr;
  uint32_t* d_piece_sum = nullptr;      // per-piece run summary (pass 1)
  uint8_t* d_piece_carry = nullptr;     // per-piece carries (pass 2)
  uint32_t* d_piece_len = nullptr;      // normalized bytes per piece (pass 3)
  uint64_t* d_piece_off = nullptr;      // their exclusive scan
  tmh::NmTwo* d_two = nullptr;          // the device normalizer's table of the two-byte characters (tm_norm_masks.h)
  uint8_t* d_need_host = nullptr;       // per document: needs the host normalizer
  uint64_t* d_nbegin = nullptr;         // normalized document ranges (GPU documents packed first, fallback documents after)
  uint64_t* d_nend = nullptr;
  uint64_t* d_ninfo = nullptr;          // [0] #fallback docs [1] #long docs [2] #segments (device-computed)
  // grow-only staging for the host fallback
  uint64_t* d_fb_roff = nullptr; uint64_t* d_fb_noff = nullptr; uint32_t* d_fb_ids = nullptr;   // lists of the host-fallback documents
  // small transfers (counters, lists of a few hundred KB) go through a pinned mailbox and a copy kernel on the caller's stream
  // instead of the copy engines, where they would queue behind the bulk transfers of other lanes (small_d2h / small_h2d)
  uint8_t* h_mail = nullptr; uint64_t mail_pos = 0;
  uint8_t* h_groups = nullptr; uint64_t h_groups_cap = 0;   // pinned staging of the group tree of long documents (build_groups)
  struct MailPending { void* dst; const uint8_t* slot; uint64_t n; hipStream_t st; };
  std::vector<MailPending> mail_pending;
  std::vector<hipStream_t> mail_streams;   // streams that have used mailbox slots since it last wrapped
  uint64_t last_totals[3] = {0, 0, 0};   // d_totals as read by ensure_output
  uint8_t* h_fb_raw = nullptr; uint8_t* h_fb_norm = nullptr;     // pinned host staging of the fallback documents (raw in, normalized out)
  uint64_t h_fb_raw_cap = 0, h_fb_norm_cap = 0;
  uint32_t* d_out = nullptr;
  uint64_t out_cap = 0;
  uint16_t* d_out16 = nullptr;          // set for the length of a launch: K4 writes two-byte ids here instead (a chunk of the ring, launch_emit)
  uint64_t out16_cap = 0;
  // a chunk of the host-to-host ring (tm_host.hip): what the host would have read back between the stages - the number of segments the normalizer
  // pass left, whether the chunk can be taken on this path at all, the number of ids - stays on the device in these control words, and the
  // kernels behind the pass are launched over a bound and look the counts up (d_ctl != null; k_chunk_ctl / k_chunk_done, tm_kernels.hip)
  uint64_t* d_ctl_store = nullptr;      // 8 words, allocated on first use
  const uint64_t* d_ctl = nullptr;      // = d_ctl_store while the workspace belongs to a slot of the ring
  // tm_batch_decode (tm_decode.hip): the ids of this batch decoded where they lie - two grow-only arenas and what the last call left in them
  uint8_t* d_dec_a = nullptr; uint64_t dec_a_cap = 0;      // lengths, byte offset of every id, scan sums, the documents' offsets and decoded lengths
  uint8_t* d_dec_b = nullptr; uint64_t dec_b_cap = 0;      // the gathered bytes | the same after capcode decoding
  uint64_t dec_total = 0, dec_o_doff = 0, dec_o_declen = 0, dec_o_dec = 0;
  uint32_t dec_ndocs = 0;
  bool dec_raw = false;                                     // decode_raw: the gathered bytes are the result
  bool dec_capcode = false;                                 // the capcode decoder ran on the device (else the gathered bytes are the result, or all the host's)
  hipEvent_t ev[TM_NUM_KERNELS + 1] = {};
  bool have_events = false;
};

// device-resident normalized dataset of the scoring pass (tm_score.hip)
struct tm_dataset {
  uint8_t* d_text = nullptr;
  uint64_t n = 0;
  tm_batch* ws = nullptr;          // workspace, created on first use and reused by every scoring pass
  uint32_t ws_docs = 0;
  uint32_t* d_hist = nullptr;      // scores | 4 token limbs | 256 missing counters
  uint64_t hist_words = 0, hist_cap = 0;
  unsigned long long* d_tokens = nullptr;
  uint32_t* d_missing_bits = nullptr;
  int n_cu = 256;
  int device = 0;
  // byte ranges of a whole-buffer walk (tm_score_begin / tm_score_finish)
  uint64_t* d_vis = nullptr;       // per strip: how far it may look at the text
  uint8_t* d_entry = nullptr;      // per strip: entry state
  uint8_t* d_exits = nullptr;      // per strip: exit state for each of the ENT entry states
  uint32_t strip_cap = 0;
  bool prepared = false;
  // the strips of the last pass, as the workspace holds them on the device (score_prepare: a pass over the same strips does not upload them again)
  std::vector<uint64_t> strips_key; const tm_batch* ws_strips_owner = nullptr; bool strips_valid = false;
  // one scoring pass at a time per dataset (it owns ONE workspace); host threads that build and load the next candidates
  // (tm_build_vocab, tm_vocab_load) run beside the pass of the current one
  std::mutex mu;
  hipStream_t stream = nullptr;    // tm_score's own stream
};

namespace tmh {

// tm_kernels.hip
hipError_t batch_alloc_bytes(tm_batch* b, void** p, uint64_t bytes);
int make_workspace(const tm_vocab* v, uint64_t max_bytes, uint32_t max_docs, bool own_text, bool with_output, tm_batch** out);
void scan_u32(const uint32_t* in, uint64_t n, uint64_t* block_sums, uint64_t* total, uint64_t* out, hipStream_t st);
void launch_doc_units(const uint64_t* doc_begin, const uint64_t* doc_end, uint32_t ndocs, uint32_t unit, uint32_t* doc_nunits, hipStream_t st);
void launch_unit_owner(const uint64_t* doc_unit_start, uint32_t ndocs, uint64_t nunits, uint32_t* unit_doc, hipStream_t st, const uint64_t* count_dev = nullptr);      // count_dev: nunits is a bound, this the count
// (TM_TRACE: every buffer that is replaced by a larger one after a workspace exists - each is a hipFree, which waits for the whole device)
inline void trace_grow(const char* what, uint64_t bytes) { static const bool on = getenv("TM_TRACE") != nullptr; if (on) fprintf(stderr, "[grow] %s -> %.2f MB\n", what, bytes / 1048576.0); }
int reserve_groups(tm_batch* b, uint32_t ngroups, uint32_t nlong);
int debug_flags();        // the test hooks in force (tm_debug_flags)
bool hooks_armed();       // the process was started with TM_TEST_HOOKS in its environment: only then do tm_debug_flags / TM_VIRTUAL_DEVICES change anything
uint32_t long_segs();    // documents with more segments than this hang under the group tree (LONG_SEGS; 8 under test hook bit 12)
int build_groups(tm_batch* b, const uint64_t* begin, const uint64_t* end, uint32_t ndocs, hipStream_t st);
int run_pipeline(tm_batch* b, hipStream_t st, bool timed, float* ms, bool emit);
int pipeline_match(tm_batch* b, hipStream_t st, hipEvent_t* ev, bool for_score = false);      // for_score: the rows are for the scoring walk (tm_score.hip)
int pipeline_resolve(tm_batch* b, hipStream_t st, hipEvent_t* ev, int mode);
void launch_doc_exits(tm_batch* b, uint8_t* d_exits, hipStream_t st);
// scoring variant of the chain kernel: histogram in HBM (scores | 4 limbs | 256 counters), see tm_score.hip
bool raw_upload_replaces_buffers(const tm_batch* b, const uint64_t* raw_offsets, uint32_t ndocs);
void pack_text(tm_batch* b, hipStream_t st);     // tm_norm.hip: the normalizer's slabs packed into d_text (no-op unless text_in_slabs)
void launch_chain_hist(tm_batch* b, uint32_t delete_id, int n_cu, uint32_t* d_hist, unsigned long long* d_tokens, uint32_t* d_missing_bits,
                       uint32_t n_ids, hipStream_t st);
int ensure_output(tm_batch* b);
// small device <-> host transfers that bypass the copy engines (tm_kernels.hip).  small_d2h's destination is filled by small_sync
// (which synchronizes the stream); small_h2d's source may be reused as soon as the call returns (pageable memory, or at most MAIL_MAX bytes).
constexpr uint64_t MAIL_BYTES = 4ull << 20, MAIL_MAX = 1ull << 20;
int small_d2h(tm_batch* b, void* host_dst, const void* dev_src, uint64_t bytes, hipStream_t st);
int small_h2d(tm_batch* b, void* dev_dst, const void* host_src, uint64_t bytes, hipStream_t st);
int small_sync(tm_batch* b, hipStream_t st);
int batch_upload_on(tm_batch* b, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, hipStream_t st);
void launch_serialize(const uint32_t* ids, uint64_t n, uint32_t enc, uint8_t* out, hipStream_t st);
// the host-to-host ring: a chunk's stages behind the upload enqueued on `st` without a host round trip (tm_norm.hip / tm_kernels.hip)
constexpr uint32_t RING_HOST_DOCS = 1, RING_LONG_DOCS = 2, RING_UNDECIDED = 4, RING_SLAB = 8, RING_SHORT_PIECE = 16, RING_BYTES = 32, RING_SEGS = 64,
                   RING_ERROR = 128, RING_OUT_CAP = 256;      // status bits of a chunk the ring hands to the exact path instead
bool ring_supported(const tm_vocab* v);
int raw_prepare(tm_batch* b, uint64_t nbytes, uint32_t ndocs, uint64_t npieces, hipStream_t st);      // the buffers tm_batch_upload_raw fills, grown to size
int ring_enqueue_normalize(tm_batch* b, hipStream_t st, uint64_t seg_bound, const uint64_t* h_raw_off);      // h_raw_off: the chunk's document offsets on the host (ndocs + 1)
void launch_chunk_ctl(tm_batch* b, uint64_t seg_bound, hipStream_t st);
// K0 .. K4, the ids packed to `enc` bytes into d_bytes (16-byte aligned), and the chunk's verdict written to `h_status` (page-locked host memory, 8 words:
// status bits, ids, normalized bytes, segments, device error word)
// *ids_at: where the packed ids lie when the stream gets there - d_bytes, or the workspace's own id buffer (four-byte ids need no packing)
int ring_enqueue_tokenize(tm_batch* b, hipStream_t st, uint32_t enc, uint8_t* d_bytes, uint64_t d_bytes_cap, uint64_t* h_status, const uint8_t** ids_at);
// tm_decode.hip: the stages of a decode on a stream, in buffers of the caller
constexpr uint64_t DEC_HOST = ~0ull;      // k_dec_capcode's length of a document it leaves to the host decoder (scripts beyond Latin, malformed UTF-8)
// the arena of a decode behind `base` bytes of the caller's own: byte counts of the tiles of ids (tm_decode.hip) | first document of a tile | byte offset of a
// tile | scan block sums | total | byte offset of every document | decoded length of every document
struct DecArena { uint64_t ntiles, o_len, o_first, o_off, o_sums, o_total, o_doff, o_declen, bytes; };
constexpr uint32_t DEC_TILE_IDS = 2048;
inline DecArena dec_arena(uint64_t base, uint64_t n, uint32_t ndocs) {
  auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
  DecArena a;
  a.ntiles = (n + DEC_TILE_IDS - 1) / DEC_TILE_IDS;
  const uint64_t sblocks = (a.ntiles + 1 + SCAN_CH - 1) / SCAN_CH + 2;
  a.o_len = up(base); a.o_first = a.o_len + up((a.ntiles + 1) * 4); a.o_off = a.o_first + up((a.ntiles + 1) * 4); a.o_sums = a.o_off + up((a.ntiles + 2) * 8);
  a.o_total = a.o_sums + up(sblocks * 8); a.o_doff = a.o_total + 256; a.o_declen = a.o_doff + up(((uint64_t)ndocs + 1) * 8); a.bytes = a.o_declen + up((uint64_t)ndocs * 8 + 8);
  return a;
}
// lengths: the bytes of every tile and their scan (the total at A + a.o_total); copy: the gather itself and the documents' byte offsets (A + a.o_doff)
void launch_decode_lengths(const tm_vocab* v, const uint32_t* d_tok, uint64_t n, const uint64_t* d_toff, uint32_t ndocs, const DecArena& a, uint8_t* A, hipStream_t st);
void launch_decode_copy(const tm_vocab* v, const uint32_t* d_tok, uint64_t n, const uint64_t* d_toff, uint32_t ndocs, const DecArena& a, uint8_t* A, uint8_t* d_out, hipStream_t st);
// d_sums: two words the kernel adds up - bytes the device decoded, documents it left to the host decoder
int launch_decode_capcode(const tm_vocab* v, const uint8_t* d_out, const uint64_t* d_doff, uint32_t ndocs, uint8_t* d_dec, uint64_t* d_declen, uint64_t* d_sums, hipStream_t st);
// tm_host.hip: tm_tokenize_pipeline over the lanes of one vocabulary or of its replicas on several devices
int tokenize_pipeline_on(const tm_vocab* const* vs, uint32_t nv, const uint8_t* text, const uint64_t* offsets, uint32_t ndocs, int raw, uint32_t encoding_length,
                         uint64_t chunk_bytes, uint32_t lanes, uint8_t* bytes_out, uint64_t bytes_cap, uint64_t* byte_offsets, uint32_t* missing,
                         uint32_t* encoding_length_used, tm_pipeline_stats* stats);
// what the device's error word (tm_batch::d_error) means for the caller: bit 0 = an entry state from which the walk never leaves its segment
// (k_resolve and the group kernels: the text / vocabulary pair does not advance, TM_E_INPUT), bit 1 = K4 met a row that K1 never wrote or a
// chain longer than its segment (an inconsistency of this library, TM_E_INTERNAL).  0 -> TM_OK.
int error_from_flag(uint32_t err);
// tm_score.hip: the error word of the dataset's last pass, once its stream has been synchronized
int score_check(tm_dataset* d);
// ... and the two halves of a byte range's pass with the entry state staying on the device (tm_multi.hip)
int score_begin_device(const tm_vocab* v, tm_dataset* d, uint64_t off, uint64_t len, int continues, hipStream_t st);
const uint8_t* score_exits_device(const tm_dataset* d);
uint8_t* score_entry_device(tm_dataset* d);
uint32_t* score_error_device(tm_dataset* d);
int score_finish_device(const tm_vocab* v, tm_dataset* d, hipStream_t st);
// tm_norm.hip
int batch_upload_raw_on(tm_batch* b, const uint8_t* raw, const uint64_t* raw_offsets, uint32_t ndocs, hipStream_t st);
// tm_normalize.cpp
bool normalize_supported(uint32_t capcode, uint32_t norm_flag);
bool normalize_on_device(uint32_t capcode, uint32_t norm_flag);

}  // namespace tmh
