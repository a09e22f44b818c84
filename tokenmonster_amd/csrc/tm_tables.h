// tm_tables.h — device-resident vocabulary tables shared by host flattener and HIP kernels.
//
// The reference's longest-match index is pansearch.Fast (10 Bloom filters + hash maps + sorted arrays,
// ~22-30 MB, tokenmonster-cpp/src/tokenmonster.cpp:491-1280): built for a CPU cache hierarchy and for
// one lookup at a time.  Any structure is admissible as long as LongestSubstring returns the same
// (index, length, found) with index = record ordinal in the .vocab file.
// Here it is a byte trie whose accepting nodes ARE the record ordinals:
//   depth 1   root[256]            (staged in LDS by every workgroup)
//   depth 2   tab[..]              direct map on the first two bytes (1 MiB, L2-resident)
//   depth >=3 tab[0..n_da)         a DOUBLE-ARRAY trie: the child of node n over byte b is the 16-byte entry base(n) + b, valid if
//             its check word is n.  One 16-byte gather per byte of the walk, a hit or a miss and never "occupied by another key,
//             try the next bucket"; the address of the next probe is ONE add (base + byte) on what the gather returned, the test
//             ONE compare — against two multiplies, a shift and two masked compares for the open-addressing hash of (parent, byte)
//             this replaces (rounds 1-3: 16-byte buckets of two 8-byte slots at load 0.3) — and the array is full (no empty
//             slots to keep the probe sequences short): 16 bytes per edge instead of 27.
// Every entry a walk can stand on carries the 32-bit CHILD FILTER of its node (bit b & 31: the node has a child over some byte
// congruent to b): the next probe is only issued if the bit of the next text byte is set.  Half of all positions end on a probe
// that cannot hit.
// A 32-bit node value carries everything a look-ahead needs about the token it accepts, so scoring a
// branch never touches the row table:
//   bits  0..20  node id; id < n_info  <=>  the prefix is a vocabulary key and id is its record ordinal
//   bit   21     node has children (lets a walk stop without a failing probe)
//   bits 22..26  nWords of the accepted token (go/tokenmonster.go:81)
//   bits 27..31  the five flag bits a *second* token is asked for (go :1075-1084):
//                b0 ends-with-letter(1) b1 begins-with-letter(2) b2 begins-with-space(4)
//                b3 begins-on-capcode(16) b4 all-letters-or-all-punct(128)
//                When Tables::spl_hint is set, b2 of a token that begins with a letter (never also with a space) says
//                instead whether its forward-delete probe can succeed at all (see Tables::spl).
#pragma once
#include <cstdint>

namespace tmh {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kNodeBits = 21;
constexpr uint32_t kNodeMask = (1u << kNodeBits) - 1;
constexpr uint32_t kHasChildren = 1u << 21;
constexpr uint32_t kRevPlaceBits = 26;                // reverse table on the device: a key's place in rev_bytes (26 bits: 64 MB) | its length << 26 (keys of at most 40 bytes)
constexpr uint32_t kMaxNodes = (1u << 20) - 2;       // 20 bits: a link-format entry keeps two depths beside the node id
constexpr uint32_t kLinkNodeMask = (1u << 20) - 1;
constexpr uint32_t kL2Size = 65536;
constexpr uint32_t kDirectSlots = 2 * kL2Size;   // the direct map in uint2 units (16-byte entries)

// ---- one-child chains ("tails") ----------------------------------------------------------------------------------------------------------
// The deep end of the trie is made of chains: below a node of a multi-word token there is, for twenty or thirty bytes, exactly one child per
// node and no key until the chain's end.  Walked a byte per gather such a chain sets the number of dependent rounds of a whole wavefront
// (the deepest walk of a 256-byte segment: 21 rounds on the englishcode shapes; capping the walks at 12 bytes - wrong results - makes
// k_match_branch 16 % faster, profiles/r05_k1_tails.txt).  So a node c below which the trie is a chain of kTailMin..kTailMax non-accepting
// one-child nodes down to a node e (a key, or the 32nd node of a longer chain) says so in its BASE WORD - the w of every entry a walk can
// stand on c through: its double-array entry, link-format entries that lead to it, space-prefix links, other chain records:
//     kTailFlag | index of c's chain record in 16-byte entries of tab
// (a walk never probes for c's only child, so the base is not needed; the shift of "(base + byte) << 4" drops the flag for code that does not
// look).  The record is three entries: a header in link format - x = e | len << 20 (bytes from c down to e), y = value of e if it is a key,
// else 0, z = child filter of e (0: the walk cannot go on), w = base word of e, possibly a chain word again - and 32 bytes of string.  A
// walk that stands on c with `go` set compares the next len bytes of text with the string in ONE round: equal, and it stands on e; not
// equal (or the text ends first), and it ends on c - whose suffix link is a valid, if shallower, start for the next position, and nothing
// between c and e is a key.
constexpr uint32_t kTailFlag = 0x80000000u, kTailMin = 5u, kTailMax = 32u;
__host__ __device__ inline bool is_tail_word(uint32_t w) { return (int32_t)w < 0; }
__host__ __device__ inline uint32_t tail_record(uint32_t w) { return w & 0x7FFFFFFFu; }
__host__ __device__ inline uint32_t tail_len(uint32_t hx) { return (hx >> 20) & 63u; }

__host__ __device__ inline uint32_t node_id(uint32_t v) { return v & kNodeMask; }
// link-format word x: node | depth of the node << 20 | depth of the deepest accepting node on the way << 26
__host__ __device__ inline uint32_t link_node(uint32_t x) { return x & kLinkNodeMask; }
__host__ __device__ inline uint32_t link_depth(uint32_t x) { return (x >> 20) & 63u; }
__host__ __device__ inline uint32_t link_bestlen(uint32_t x) { return x >> 26; }
__host__ __device__ inline uint32_t node_nwords(uint32_t v) { return (v >> 22) & 31u; }
__host__ __device__ inline uint32_t node_flag5(uint32_t v) { return v >> 27; }
__host__ __device__ inline uint32_t flag8_to_flag5(uint32_t f) {
  return (f & 1u) | (((f >> 1) & 1u) << 1) | (((f >> 2) & 1u) << 2) | (((f >> 4) & 1u) << 3) | (((f >> 7) & 1u) << 4);
}

// Row of record `index` (16 B): what the walk needs when the record is the FIRST token of a branch (tokenOuter,
// go/tokenmonster.go:63-77, with id1/id2/length/length2 resolved as Load does, :2703-2712) — with everything about the three
// candidate first tokens (the record itself, alternative 1, alternative 2) that does not depend on the text folded into constants
// at load, so that scoring a branch (go :1075-1084) is a handful of adds:
//   x = id  | c0     << 20      c0     = allLetters + max0(nWords-1) + nWords*100 of the record (its length comes with the match)
//   y = id1 | fpart1 << 20      fpartK = lenK + allLettersK + max0(nWordsK-1) + nWordsK*100 of alternative K
//   z = id2 | fpart2 << 20
//   w = len1 | len2 << 6 | endsWithLetter{0,1,2} << 12 | endsOnCapcode{0,1,2} << 15 | (nWords >= 2){0,1,2} << 18 | flag&32 << 21
//       len1 == 0 <=> index == DOES_NOT_EXIST, len2 == 0 <=> index2 == DOES_NOT_EXIST
// ids take 20 bits: a vocabulary with more than 2^20 ids is refused at load (the reference's format allows 2^24 - 1; its trained
// vocabularies stop at 100 256).  In a forward-delete state (go :1088-1105: nWords - 1, length - 1) a candidate's constant is
// c - 100 - (nWords >= 2) resp. fpart - 101 - (nWords >= 2).
constexpr uint32_t kRowIdBits = 20, kRowIdMask = (1u << kRowIdBits) - 1;
struct alignas(16) Row { uint32_t x, y, z, w; };

struct Tables {
  const uint32_t* root;    // [256]
  const uint2* tab;        // one table for everything a walk gathers, in 16-byte entries (uint4 on the device):
                           //   [0, n_da]                double-array trie of the edges at depth >= 3.  Entry base(n) + b of the child c of n
                           //                           over byte b: x = n (kNone: empty), y = node value of c, z = child filter of c,
                           //                           w = base(c) (entry index; 0 if c has no children).  base(n) + 255 <= n_da for
                           //                           every n; entry n_da stays empty (idle walks gather it: idle_off)
                           //   [direct_off/16, +65536) direct map on the first two bytes, index b0 | b1<<8 (the little-endian u16 at
                           //                           the position), link format: the whole answer for depth <= 2 and, if the
                           //                           node b0b1 has children, where to go on
                           //   [link_off/16, +nodes)   suffix links, one per trie node n (string s): where the walk of s[1:] ends up,
                           //                           so the walk at text position p+1 CONTINUES from the walk at p instead of
                           //                           starting over (Aho-Corasick failure links turned into longest-prefix state)
                           //   link format:            x = node m reached | depth(m) << 20 | depth of the deepest accepting node on
                           //                               the path to m << 26
                           //                           y = value of that accepting node (0: none)
                           //                           z = child filter of m if all of s[1:] is in the trie (the walk may probe on), else 0
                           //                           w = base(m), or m's chain word (kTailFlag, above)
                           //   [behind the links]      chain records, three entries each
  const uint4* spl;        // [n_info] "space-prefix link" of record s: where the walk of ' '+s (the forward-delete probe of
                           //   go/tokenmonster.go:1088-1095; ' ' 0x00 + s for UTF-16) ends up, so that probe only has to CONTINUE:
                           //   x = node id reached | continue-flag << 21 | best accepting depth << 22 ; y = value of that node;
                           //   z = 32-bit child filter of the node reached (0 unless the continue flag is set): most probes end here;
                           //   w = base of the node reached
  const uint32_t* vals;    // [n_info] node value of every record (the split pipeline hands positions over as record ordinals)
  const Row* rows;         // [n_info]
  const uint8_t* begin_byte;  // [256]  go/tokenmonster.go:43
  uint32_t idle_off, n_da;            // byte offset of the always-empty entry behind the double array / its number of entries
  uint32_t n_info, max_len;
  uint32_t off;            // 1, or 2 for UTF-16 (lilbufOffset, go :1031-1034)
  uint32_t bstart;         // node value after consuming ' ' (and 0x00 for UTF-16), kNone if absent
  uint32_t has_delete, delete_id, unk_id;
  uint32_t spl_hint;       // b2 of letter-initial tokens is the forward-delete hint
  uint32_t link_off, direct_off;   // byte offsets of the suffix links / the direct map inside tab
                                   // (the chain records lie behind the suffix links; their entries name them by index: nothing here has to know where)
  uint32_t last_rec;               // ... except where tab ENDS: the last entry index a chain record may begin at (three entries before the end).  A chain
                                   // word is an index a walk dereferences as it finds it; the walker clamps it to this, so that a block whose contents
                                   // do not belong to its description (a stale copy adopted by an importer) reads wrong bytes, not beyond the table.
};

}  // namespace tmh
