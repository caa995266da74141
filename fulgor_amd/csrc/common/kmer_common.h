// Shared host/device definitions of the k-mer dictionary ("k2u") layout used by the MI355X engine.
//
// This replaces the role of sshash::dictionary / streaming_query in the reference
// (call sites: src/ps_full_intersection.cpp:341-352, src/ps_threshold_union.cpp:330-346).
// The reference's SSHash sources are not vendored (external/sshash is empty), and per SURVEY F7 the
// per-read result does not depend on the dictionary's internals; this layout is therefore designed
// for the GPU: one table of 64-byte buckets keyed by the minimizer, both strands of every unitig, whose
// 16-byte super-k-mer records carry their own unitig context (one line fetch per lookup).
//
// Everything in this header compiles for both host (g++) and device (hipcc).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FG_HD __host__ __device__ __forceinline__
#else
#define FG_HD inline
#endif

namespace fg {

// ---- base encoding -------------------------------------------------------------------------
// A=0 C=1 G=2 T=3 (complement = 3-x = flip both bits). 0xFF = not a nucleotide.
FG_HD uint32_t base_code(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0xFF;
    }
}

// branch-free form of base_code for the device hot loops: (b >> 1) & 3 maps A,C,T,G -> 0,1,2,3; x ^ (x >> 1)
// turns that into A0 C1 G2 T3; validity = letter range (0x40..0x7F) and bit (b & 31) of the set {A,C,G,T}
FG_HD uint32_t base_code_fast(uint32_t b) {
    const uint32_t x = (b >> 1) & 3u;
    const uint32_t ok = (uint32_t)((b & 0xC0u) == 0x40u) & ((0x0010008Au >> (b & 31u)) & 1u);
    return ok ? (x ^ (x >> 1)) : 0xFFu;
}

// ---- bit-plane L-mers (L <= 32) --------------------------------------------------------------
// An L-mer is two 32-bit planes: lo = bit0 of every base, hi = bit1; base i (i=0 is the first /
// leftmost base) lives at bit i of each plane.
FG_HD uint32_t low_mask32(uint32_t L) { return L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1u); }

FG_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

// reverse complement of one plane of an L-mer: reverse the L bits, then flip them
FG_HD uint32_t rc_plane(uint32_t p, uint32_t L) { return (~brev32(p) >> (32 - L)) & low_mask32(L); }

// 64-bit key of an L-mer
FG_HD uint64_t lmer_key(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// murmur3 finalizer: a bijection on u64
FG_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

FG_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

// Minimizer order: a cheap 22-bit multiplicative hash of the m-mer AS READ (not of its canonical form), smaller
// first, ties to the left (22 bits so that order << 10 | position packs into one u32 min for units of up to 1024
// m-mers). The dictionary holds every unitig on BOTH strands, so a read is always compared in its own
// orientation: no canonical forms, no reverse complements and no strand cases in the lookup kernel; memory
// (twice the records) is what the 288 GB of HBM are for. The order only has to be the same on host and device
// and reasonably random (any function of the m-mer is CORRECT: ties go to the left on both sides); buckets are
// addressed by dict_hash of the m-mer. Three instructions on the device: shift-add, multiply, and-or with the
// position. `lo` may carry bits above the m-mer (they are shifted out), `hi` must be masked to m bits.
// (round 4; the two-multiply, xor-shift hash before it cost seven and gave the same 15.9 runs per 150-base read.)
constexpr uint32_t ORDER_POS_BITS = 10;
FG_HD uint32_t minimizer_order(uint32_t lo, uint32_t hi, uint32_t m) {
    return (((lo << (32u - m)) + hi) * 0x2C1B3C6Du) >> ORDER_POS_BITS;
}

// ---- bucket hash of a minimizer (the m-mer as read: planes lo, hi) ---------------------------------------
// Minimizers are the m-mers with the SMALLEST order hash, so the bucket hash must not be correlated
// with minimizer_order: other multipliers, other mixing. bucket = mulhi32(dict_hash(lo, hi, seed), num_buckets).
FG_HD uint32_t dict_hash(uint32_t lo, uint32_t hi, uint32_t seed) {
    uint32_t x = (lo ^ seed) * 0xCC9E2D51u ^ hi * 0x1B873593u;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    return x;
}

// ---- 16-byte super-k-mer record, four per 64-byte bucket ------------------------------------------
// The dictionary is ONE table of 64-byte buckets: a hashed region addressed by the minimizer and, behind it, an
// overflow region. A lookup is one line fetch, nothing else (no pilot table, no string fetch), plus one more for
// the few keys that do not fit their bucket. A record is self-contained: it carries the CONTEXT of its minimizer
// occurrence, the CL = 2k - m bases [pos - (k - m), pos + k) of the unitig strand it was cut from (bit planes; bases
// outside the unitig are 0 and outside every valid window), so a k-mer is verified against the record alone.
// Window s (0 <= s <= k - m) of the context is the k-mer starting at context base s; the record is valid for windows
// smin..smax: the k-mers of that strand whose leftmost smallest-order m-mer is this occurrence (a super-k-mer).
//   w0  context, lo plane, bases 0..31
//   w1  context, hi plane, bases 0..31
//   w2  lo plane bases 32..44 | hi plane bases 32..44 << 13 | smin << 26 | redirect flag << 31
//   w3  colour-set id (27 bits; u2c folded in, index.hpp:37 of the reference) | smax << 27 | spill << 31
// An empty slot has smin > smax.
// When the records of the keys living in a bucket do not all fit, the bucket's LAST slot is a REDIRECT: w1 = first
// overflow bucket, w2 = empty | redirect flag, w3 = number of overflow buckets; the keys that did not fit (whole keys)
// share those consecutive overflow buckets, and every query that meets the redirect reads the first REDIRECT_DIRECT
// of them at once (a record is verified by its context: records of other keys simply do not match).
// spill (bit 31 of w3 of a bucket's last slot): the query goes on with the next bucket. Set on the overflow
// buckets of a run from the REDIRECT_DIRECT-th on (but the last); never on a hashed bucket (when more than four keys
// live in one — rare — the surplus keys go to its overflow run too). A bucket thus hands a query on to at most three
// others if it is a hashed one and to at most one if not, which bounds the lookup kernel's ring of waiting buckets.
// Needs 2k - m <= 45 and k - m <= 15 (k = 31, m = 17: exactly 45 bases, 15 windows: 16 runs per 150-base read, so that four
// reads fill a pass of the lookup kernel; round 2 had 43 bases for m = 19 and a 31-bit colour-set id).
constexpr uint32_t REC_WORDS = 4;
constexpr uint32_t BUCKET_RECS = 4;
constexpr uint32_t BUCKET_WORDS = REC_WORDS * BUCKET_RECS;
constexpr uint32_t REC_CTX_MAX = 45;
constexpr uint32_t REC_HI_BITS = REC_CTX_MAX - 32;            // context bases held in w2, per plane
constexpr uint32_t REC_HI_MASK = (1u << REC_HI_BITS) - 1u;
constexpr uint32_t REC_MAX_CSID = 0x07FFFFFFu;                // (also the width of a redirect's bucket count)
constexpr uint32_t REC_SPILL = 0x80000000u;
constexpr uint32_t REC_W2_EMPTY = 15u << 26;  // smin = 15 > smax = 0 (w3 = 0)
constexpr uint32_t REC_W2_REDIRECT = REC_W2_EMPTY | 0x80000000u;
// Up to DICT_NARROW_BUCKETS buckets (a 4 GB table, about 120 M distinct 31-mers) a (bucket, source lane) pair of the lookup kernel's ring
// packs into one 32-bit word; larger tables — up to DICT_MAX_BUCKETS: 128 GB of the 288 — run the kernel's WIDE instantiations, whose ring
// keeps the source lane in a word of its own (round 6: before, such a collection was refused at load).
constexpr uint32_t DICT_NARROW_BUCKETS = 1u << 26;
constexpr uint32_t DICT_MAX_BUCKETS = 1u << 31;
constexpr uint32_t REDIRECT_DIRECT = 3;
FG_HD uint32_t rec_w2(uint64_t ctx_lo, uint64_t ctx_hi, uint32_t smin) {
    return (uint32_t)(ctx_lo >> 32) | ((uint32_t)(ctx_hi >> 32) << REC_HI_BITS) | (smin << 26);
}
FG_HD uint32_t rec_w3(uint32_t csid, uint32_t smax) { return csid | (smax << 27); }
FG_HD uint32_t rec_smin(uint32_t w2) { return (w2 >> 26) & 15u; }
FG_HD uint32_t rec_smax(uint32_t w3) { return (w3 >> 27) & 15u; }
FG_HD uint64_t rec_ctx_lo(uint32_t w0, uint32_t w2) { return (uint64_t)w0 | ((uint64_t)(w2 & REC_HI_MASK) << 32); }
FG_HD uint64_t rec_ctx_hi(uint32_t w1, uint32_t w2) { return (uint64_t)w1 | ((uint64_t)((w2 >> REC_HI_BITS) & REC_HI_MASK) << 32); }
// ---- unitig strings ----------------------------------------------------------------------------
// word w holds bases [32w, 32w+32): low 32 bits = lo plane, high 32 bits = hi plane.
// extract an L-mer (L<=32) starting at base s from two consecutive words
FG_HD void string_lmer(uint64_t w0, uint64_t w1, uint32_t sh, uint32_t L, uint32_t& lo, uint32_t& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    // sh <= 31: one v_alignbit (funnel shift) per plane
    lo = __builtin_amdgcn_alignbit((uint32_t)w1, (uint32_t)w0, sh) & low_mask32(L);
    hi = __builtin_amdgcn_alignbit((uint32_t)(w1 >> 32), (uint32_t)(w0 >> 32), sh) & low_mask32(L);
#else
    uint64_t l = ((uint64_t)(uint32_t)w1 << 32) | (uint32_t)w0;
    uint64_t h = (w1 & 0xFFFFFFFF00000000ULL) | (w0 >> 32);
    lo = (uint32_t)(l >> sh) & low_mask32(L);
    hi = (uint32_t)(h >> sh) & low_mask32(L);
#endif
}

// ---- packed blocks of the gap-coded lists (all codecs) --------------------------------------------------
// A block holds up to 64 consecutive values v_0 < v_1 < ... of one list as `width`-bit offsets from
// `start` (= previous value + 1, or 0 for the first block): v_i = start + field_i. Header word:
//   start:27 | width:5 | count-1:6 | first data word, relative to the start of the list's region (its headers, then its data):26
// (so num_colors <= 2^27; a list owns fewer than n/4 codes, hence fewer than 2^26 data words).
// Where a list is dense — its next 64 values fall within BLK_CHUNK_SPAN colours — the block is instead a
// plain bitmap chunk: width = BLK_CHUNK_WIDTH, start = first colour rounded down to a multiple of 32,
// count = number of 32-bit words (<= 64), holding ALL values of the list below start + BLK_CHUNK_SPAN.
// A chunk is ORed into the result word-wise, one word per lane: no per-value work at all.
constexpr uint32_t BLK_VALUES = 64;
constexpr uint32_t BLK_CHUNK_WIDTH = 31;
constexpr uint32_t BLK_CHUNK_SPAN = 64 * 32;
constexpr uint32_t BLK_MAX_COLORS = 1u << 27;
FG_HD uint64_t blk_pack(uint32_t start, uint32_t width, uint32_t count, uint32_t rel_word) {
    return (uint64_t)start | ((uint64_t)width << 27) | ((uint64_t)(count - 1) << 32) | ((uint64_t)rel_word << 38);
}
FG_HD uint32_t blk_start(uint64_t h) { return (uint32_t)h & 0x7FFFFFFu; }
FG_HD uint32_t blk_width(uint64_t h) { return ((uint32_t)h >> 27) & 31u; }
FG_HD uint32_t blk_count(uint64_t h) { return ((uint32_t)(h >> 32) & 63u) + 1u; }
FG_HD uint32_t blk_rel_word(uint64_t h) { return (uint32_t)(h >> 38); }
// bits of the Elias-delta code of x (gamma(len+1) then len bits, len = floor(log2(x+1)))
FG_HD uint32_t delta_code_bits(uint32_t x) {
    const uint64_t y = (uint64_t)x + 1;
    uint32_t len = 0;
    while ((y >> (len + 1)) != 0) ++len;
    uint32_t z = 0;
    while (((len + 1) >> (z + 1)) != 0) ++z;
    return 2 * z + 1 + len;
}

}  // namespace fg
