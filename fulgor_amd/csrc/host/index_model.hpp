// Host-side model of a Fulgor index as the MI355X engine keeps it (and uploads it to HBM).
//
// Mirrors `template <typename ColorSets> struct index` (include/index.hpp:16-110 in the reference):
//   m_k2u  (sshash::dictionary)  -> Dict          (own GPU layout, see common/kmer_common.h)
//   m_u2c + rank9                -> Dict::unitig_csid dense table (u2c(), index.hpp:37) folded into records
//   m_color_sets (hybrid)        -> HybridSets    (same bit stream as hybrid.hpp:37-95 writes; the
//                                   Elias-Fano offsets are held decoded as plain u64)
//   m_filenames                  -> filenames
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../common/kmer_common.h"

namespace fg {

// same numbering as the reference's enum index_t (include/util.hpp:18)
enum IndexType : int { IDX_HYBRID = 0, IDX_DIFF = 1, IDX_META = 2, IDX_META_DIFF = 3 };
// same numbering as enum encoding_t (include/util.hpp:19)
enum Encoding : int { ENC_DELTA_GAPS = 0, ENC_BITMAP = 1, ENC_COMPLEMENT = 2, ENC_SYMDIFF = 3 };

struct HybridSets {  // fulgor::hybrid, include/color_sets/hybrid.hpp:338-352
    uint32_t num_colors = 0;
    uint32_t sparse_thr = 0;  // m_sparse_set_threshold_size      = u32(0.25 * n)
    uint32_t dense_thr = 0;   // m_very_dense_set_threshold_size  = u32(0.75 * n)
    std::vector<uint64_t> offsets;  // num_sets + 1 bit offsets into `bits`
    std::vector<uint64_t> bits;     // the bit vector, padded with 4 zero words
    uint64_t nbits = 0;
    // Device-side form of the gap-coded lists, built at load (hybrid_build_blocks): every list is cut into
    // blocks of 64 values, each stored as 64 fixed-width offsets from the block's first candidate value, so
    // that one wavefront decodes a block in one step (lane i -> value i) instead of walking a universal code.
    std::vector<uint32_t> set_size;    // num_sets: list sizes (the delta(size) header, decoded once)
    std::vector<uint64_t> blk_first;   // num_sets + 1: first block header of every set (bitmap sets own none)
    std::vector<uint64_t> blk_wbase;   // num_sets: first 32-bit word of the set's block data in blk_words
    std::vector<uint64_t> blk_hdr;     // see blk_pack() in common/kmer_common.h
    std::vector<uint32_t> blk_words;   // packed offsets, every block starts on a 32-bit word; 64 padding words
    uint64_t num_sets() const { return offsets.empty() ? 0 : offsets.size() - 1; }
};

struct Dict {
    uint32_t k = 0, m = 0;
    uint64_t num_kmers = 0;
    uint64_t total_bases = 0;
    std::vector<uint64_t> strings;  // bit-plane words of the unitig concatenation, padded with 2 zero words (host only:
                                    // export / dump / self check; the records carry their own context)
    uint32_t seed = 0;
    std::vector<uint32_t> records;  // REC_WORDS words per super-k-mer record, in unitig order (saved; spill bits clear)
    uint32_t num_buckets = 0;       // hashed buckets; behind them DICT_TAIL_BUCKETS more (slots carried past the last one), then the overflow region
    std::vector<uint32_t> table;    // BUCKET_WORDS words per bucket: the records placed by linear probing (rebuilt at load)
    // unitig table (export / u2c)
    std::vector<uint64_t> unitig_off;    // num_unitigs + 1 base offsets into the concatenation
    std::vector<uint32_t> unitig_csid;   // u2c
    uint64_t num_unitigs() const { return unitig_csid.size(); }
    uint64_t num_records() const { return records.size() / REC_WORDS; }
};
constexpr uint32_t DICT_TAIL_BUCKETS = 1024;

// ---- meta / differential / meta-differential colour sets (see codecs_build.hpp) ----------------------
enum OpKind : uint32_t { OP_OR_GAPS = 0, OP_OR_BITMAP = 1, OP_OR_COMP = 2, OP_XOR_GAPS = 3 };

struct SetOp {         // 32 bytes, read as two 16-byte loads on the device
    uint64_t body;     // bit position of the first gap code / of the bitmap
    uint64_t soff;     // unused (kept for the 32-byte layout)
    uint32_t ncodes;   // gap codes (0 for bitmaps)
    uint32_t kind;     // OpKind
    uint32_t base;     // first colour of the partition (0 for differential)
    uint32_t np;       // colours in the partition
};

// Device form of the ops, built at load / convert (codecs_build.hpp: build_generic_device). Every op of every
// codec contributes a set of colours that is XORed into the set under construction (members of disjoint
// partitions, a representative, a symmetric difference), so the device needs no universal-code decoder:
//   span op   the op's universe covers at most GOP_SPAN_WORDS 32-bit words of the colour space: one 32-byte
//             record {first word | count << 24, the words}, fetched with one request and XORed by one lane
//   block op  larger universes: the same packed blocks / bitmap chunks as the hybrid gap lists (GenOpDev)
// dev_set_ops lists, per colour set, references to them: span record index, or GOP_BLOCK_REF | block-op index.
constexpr uint32_t GOP_SPAN_WORDS = 7;
constexpr uint32_t GOP_BLOCK_REF = 0x80000000u;
struct GenOpDev {      // 32 bytes, same layout as the device's ListDesc
    uint64_t begin;    // first data word in dev_blk_words
    uint64_t soff;     // first block header
    uint32_t ncodes;   // number of blocks
    uint32_t pad0, pad1, pad2;
};

struct GenericSets {
    int type = IDX_META;
    uint32_t num_colors = 0;
    uint32_t partition_size = 0, cluster_size = 0, num_partitions = 0;
    uint64_t num_partial_sets = 0, num_clusters = 0;
    std::vector<uint64_t> bits;  // arena, padded with 2 words
    uint64_t nbits = 0;
    std::vector<SetOp> ops;
    std::vector<uint64_t> set_ops_off;  // num_sets + 1
    std::vector<uint32_t> set_ops;      // op indices
    // device form (acceleration structure, rebuilt at load): see GenOpDev
    std::vector<uint32_t> dev_set_ops;   // parallel to set_ops
    std::vector<uint32_t> dev_span;      // 8 words per span op
    std::vector<GenOpDev> dev_ops;       // block ops
    std::vector<uint64_t> dev_blk_hdr;
    std::vector<uint32_t> dev_blk_words; // 64 padding words
    std::vector<uint32_t> set_bytes;    // algorithmic bytes per colour set (accounting only)
    uint64_t num_sets() const { return set_ops_off.empty() ? 0 : set_ops_off.size() - 1; }
};

struct HostIndex {
    int type = IDX_HYBRID;
    Dict dict;
    HybridSets hybrid;    // always present (ingestion format, export)
    GenericSets generic;  // present when type != IDX_HYBRID: the codec the queries run on
    std::vector<std::string> filenames;
};

}  // namespace fg
