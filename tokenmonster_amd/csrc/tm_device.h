// tm_device.h — host-side structures behind the opaque handles of tokenmonster_hip.h (HIP TUs only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "tokenmonster_hip.h"
#include "tm_internal.h"
#include "tm_tables.h"

// Kernel launches, raw LDS addresses and the two register-pinning asm statements go through these macros so that tools/emu (test
// infrastructure: the kernel sources compiled for the host, work-items as fibers) can build the same files.  TM_EMU is never defined
// in the product build, where every macro expands to exactly the tokens it replaced.
#ifdef TM_EMU
#define TM_LAUNCH(kern, grid, block, shmem, stream, ...) emu::launch(grid, block, [=]() { kern(__VA_ARGS__); })
#define TM_LDS_SPACE
#define TM_LDS_SPACE_UNALIGNED __attribute__((aligned(1)))
#define TM_LDS_ADDR(p) emu::lds_addr(p)
#define TM_LDS_PTR(T, a) ((T*)emu::lds_ptr(a))
#define TM_LDS_OBJECTS(a, b) emu::lds_objects(&(a), sizeof(a), &(b), sizeof(b))
#define TM_KEEP_IN_VGPRS2(a, b) ((void)0)
#define TM_KEEP_IN_VGPRS4(a, b, c, d) ((void)0)
#else
#define TM_LAUNCH(kern, grid, block, shmem, stream, ...) kern<<<grid, block, shmem, stream>>>(__VA_ARGS__)
#define TM_LDS_SPACE __attribute__((address_space(3)))
#define TM_LDS_SPACE_UNALIGNED __attribute__((address_space(3), aligned(1)))
#define TM_LDS_ADDR(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)(p))     // 32-bit LDS byte address of an object
#define TM_LDS_PTR(T, a) ((T*)(uintptr_t)(a))                                                        // and back (T: a TM_LDS_SPACE type)
#define TM_LDS_OBJECTS(a, b) ((void)0)
#define TM_KEEP_IN_VGPRS2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define TM_KEEP_IN_VGPRS4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#endif

// Streams: the data a kernel touches exactly once — the text K1 reads, the T(p,0) rows / side lists / exit maps it writes (6 GB per GiB
// of text), the rows K4 reads and the ids it writes — carry the non-temporal hint, so that they do not push the vocabulary tables out
// of the 4 MB L2 of an XCD (K1's time follows the tables' cache footprint; measured -2.5 % on K1, profiles/r03_k1_variants_ab.txt).
#ifndef TM_EMU
#define TM_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define TM_STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define TM_STREAM_STORE(ptr, val) (*(ptr) = (val))
#define TM_STREAM_LOAD(ptr) (*(ptr))
#endif

namespace tmh {

int hip_fail(hipError_t e, const char* what);

struct HostVocab {
  uint8_t capcode = 0, charset = 0, norm_flag = 0, level = 0, reserve = 0;
  uint32_t unk = TM_NONE, vocab_size = 0, n_ids = 0, n_info = 0, delete_id = TM_NONE, max_len = 0;
  mutable std::vector<uint8_t> image;   // the .vocab bytes (tm_vocab_image / tm_vocab_save): what the vocabulary was loaded from, or - for one built straight from a
                                        // token list (tm_vocab_build) - written from the records below the first time somebody asks (vocab_image())
  std::vector<uint8_t> keys;        // concatenated key bytes
  std::vector<uint32_t> key_off;    // n_info + 1
  // the records as the .vocab file holds them, in file order = pansearch order
  std::vector<uint8_t> rec_flag, rec_nwords;
  std::vector<uint32_t> rec_id, rec_index1, rec_index2;
  std::vector<float> rec_score;
  std::vector<Row> rows;
  uint8_t begin_byte[256];
  std::vector<uint32_t> root;
  std::vector<uint2> tab;           // direct depth-2 map, edge hash, suffix links (tm_tables.h)
  std::vector<uint32_t> vals;       // node value per record ordinal
  std::vector<uint4> spl;           // space-prefix links
  std::vector<uint32_t> rev_off;    // n_ids + 1: reverse[id] = rev_bytes[rev_off[id] .. rev_off[id+1])  (last record wins, go :2715)
  std::vector<uint8_t> rev_bytes;
  std::vector<uint32_t> rev_pack;   // the device's form of rev_off (d_rev_off): place | length << kRevPlaceBits per id (tm_tables.h)
  uint32_t idle_off = 0, n_da = 0, n_nodes = 0, off = 1, bstart = kNone, spl_hint = 0, link_off = 0, direct_off = 0;
};

// .vocab bytes -> records -> tables (what tm_vocab_load does on the host)
int parse_vocab(const uint8_t* f, size_t n, HostVocab& hv);
int parse_records(const uint8_t* f, size_t n, HostVocab& hv);        // the first half: header, keys, records, begin_byte (checked)

// The byte trie of the keys, built WITHOUT a hash of its edges: the keys are visited in plain lexicographic order (the file keeps them by
// length, bytewise within a length: a merge of the length groups), where every key continues the path of the one before it behind their
// common prefix, and a key's prefixes have all been visited before it.  Accepting node id == record ordinal, internal nodes are numbered
// from n_info in order of creation (depth-first: chains of one-child nodes get consecutive ids).
struct Trie {
  static constexpr uint32_t kRoot = kNodeMask;
  uint32_t n_info = 0, n_nodes = 0;
  std::vector<uint8_t> depth_of, byte_of;          // per node
  std::vector<uint32_t> parent_of;                  // per node (kRoot for depth 1)
  std::vector<uint32_t> kid_start, kid;             // children of node n: kid[kid_start[n] .. kid_start[n + 1]) = byte << 24 | child, bytes ascending
  uint32_t root_child[256];                         // children of the root (kNone: none)
  uint32_t find(uint32_t node, uint32_t byte) const {
    if (node == kRoot) return root_child[byte];
    uint32_t lo = kid_start[node], hi = kid_start[node + 1];
    while (hi - lo > 8) { const uint32_t mid = (lo + hi) / 2; if ((kid[mid] >> 24) <= byte) lo = mid; else hi = mid; }
    for (; lo < hi; lo++) if ((kid[lo] >> 24) == byte) return kid[lo] & 0xFFFFFFu;
    return kNone;
  }
};
// on_key(ordinal, path_ord, lex_rank): called once per key in lexicographic order, path_ord[d] = ordinal of the key that is this key's prefix of
// length d + 1 (kNone: that prefix is not a key), for d + 1 < length of the key: what the builder's search for alternatives (go/tokenmonster.go:3597)
// needs, for free.  May be null.
struct TrieVisitor { virtual void key(uint32_t ordinal, const uint32_t* path_ord) = 0; virtual ~TrieVisitor() = default; };
int build_trie(const HostVocab& hv, Trie& t, TrieVisitor* on_key);
// rows, double array, links, direct map, space-prefix links, reverse table: everything the device block is made of (tm_tables.h)
int build_tables(HostVocab& hv, const Trie& t, const std::vector<uint32_t>* perm = nullptr);     // perm: node ids by use (tm_vocab_tune), or none
// tm_build.cpp: token list -> records + trie (the rules of go/tokenmonster.go:3423-3793 == training/trainvocab.go:548-907)
int build_vocab_records(const std::vector<std::string>& tokens, const std::vector<uint8_t>& special, uint32_t capcode, uint32_t charset, uint32_t norm_flag,
                        uint32_t level, bool with_unk, HostVocab& hv, Trie& trie, const std::vector<float>* token_scores = nullptr);
// the .vocab bytes of the vocabulary (written from the records when it was built from a token list)
const std::vector<uint8_t>& vocab_image(const HostVocab& hv);

}  // namespace tmh

struct tm_vocab;
namespace tmh {
struct LanePool;                       // tm_host.hip: streams + workspaces the host-buffer entry points borrow
void pool_destroy(LanePool* p);
// Every entry point that takes a vocabulary (or a batch / dataset bound to one) runs on the device the vocabulary's tables live
// on, whatever device is current for the calling OS thread (cgo moves goroutines between threads): makes it current.
int enter_device(const tm_vocab* v);
}  // namespace tmh

struct tm_vocab {
  tmh::HostVocab host;
  tmh::Tables tables{};
  int device = 0;
  uint64_t device_bytes = 0;
  void* d_block = nullptr;           // the one device allocation the table pointers below point into (tm_vocab.hip: block cache)
  size_t block_bytes = 0;
  bool tuned = false;                // the tables are laid out by use (tm_vocab_tune)
  uint64_t part_bytes[8] = {};       // root, tab, rows, spl, vals, rev_off, rev_bytes, begin_byte: laid out in this order, each on a 256-byte boundary
  uint32_t* d_root = nullptr;
  uint2* d_tab = nullptr;
  uint4* d_spl = nullptr;
  uint32_t* d_vals = nullptr;
  uint32_t* d_rev_off = nullptr;
  uint8_t* d_rev_bytes = nullptr;
  tmh::Row* d_rows = nullptr;
  uint8_t* d_begin_byte = nullptr;
  mutable tmh::LanePool* pool = nullptr;   // created on first use; the tables themselves are immutable
  // one event per stream that kernels reading the tables have been launched on, re-recorded BEHIND every such launch (note_table_use):
  // tm_vocab_free parks the device block for the next load, and the block must not be refilled while such a kernel is still in flight.
  // (Recorded at launch time, not at free time: by then the stream may have been destroyed by its owner.)
  mutable std::mutex use_mu;
  mutable std::vector<std::pair<hipStream_t, hipEvent_t>> last_use;
};
namespace tmh { void note_table_use(const tm_vocab* v, hipStream_t st); }    // call AFTER the launch
