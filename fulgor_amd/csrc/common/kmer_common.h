// Shared host/device definitions of the k-mer dictionary ("k2u") layout used by the MI355X engine.
//
// This replaces the role of sshash::dictionary / streaming_query in the reference
// (call sites: src/ps_full_intersection.cpp:341-352, src/ps_threshold_union.cpp:330-346).
// The reference's SSHash sources are not vendored (external/sshash is empty), and per SURVEY F7 the
// per-read result does not depend on the dictionary's internals; this layout is therefore designed
// for the GPU: bit-plane packed unitig strings, one 8-byte record per super-k-mer, a
// pilot-displaced perfect hash over canonical minimizers.
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

// 64-bit key of an L-mer and its canonical (strand independent) form
FG_HD uint64_t lmer_key(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
FG_HD uint64_t canonical_key(uint32_t lo, uint32_t hi, uint32_t L) {
    uint64_t f = lmer_key(lo, hi);
    uint64_t r = lmer_key(rc_plane(lo, L), rc_plane(hi, L));
    return f < r ? f : r;
}

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

// Minimizer order: a cheap 24-bit multiplicative hash of the canonical m-mer key, smaller first; ties are
// broken by position (leftmost in the orientation in which the window is read), so that reading the same
// window on the other strand selects the rightmost tie. Packed as (order << 4 | position) in a u32 min.
// (The order only has to be the same on host and device and reasonably random; the perfect hash below is
// keyed by the canonical key itself through the 64-bit mixer.)
FG_HD uint32_t order24(uint64_t canonical) {
    uint32_t x = (uint32_t)canonical * 0x9E3779B1u ^ (uint32_t)(canonical >> 32) * 0x85EBCA77u;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    return x >> 8;
}

// ---- perfect hash over canonical minimizer keys ------------------------------------------------
// Minimizers are the m-mers with the SMALLEST order hash, so mix64(key) itself is far from uniform
// over the selected keys; the perfect hash therefore uses a second, seeded mix of it:
//   h = phf_hash(canonical key, seed); bucket = fastrange(high32(h), num_buckets);
//   slot = fastrange(mix32(low32(h) ^ pilot * PHI32), num_slots)
FG_HD uint64_t phf_hash(uint64_t key, uint64_t seed) { return mix64(key ^ (seed * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL)); }
constexpr uint32_t PHI32 = 0x9E3779B1u;
FG_HD uint32_t phf_bucket(uint64_t h, uint32_t num_buckets) { return mulhi32((uint32_t)(h >> 32), num_buckets); }
FG_HD uint32_t phf_slot(uint64_t h, uint32_t pilot, uint32_t num_slots) {
    uint32_t v = (uint32_t)h ^ (pilot * PHI32);
    v ^= v >> 15;
    v *= 0x2c1b3c6du;
    v ^= v >> 12;
    return mulhi32(v, num_slots);
}

// ---- 8-byte super-k-mer record -----------------------------------------------------------------
// bits  0..30  pos   : absolute base offset of the minimizer occurrence in the concatenated unitigs
// bit   31     fwd   : the m-mer at pos, read in unitig orientation, is its own canonical form (key(fwd) <=
//                      key(rc)). A query whose minimizer m-mer has the same flag lies on the unitig's strand,
//                      otherwise on the opposite one (m-mers of odd length are never palindromes), so only
//                      ONE orientation has to be compared against the string.
// bits 32..35  jmin  : smallest offset (minimizer start - k-mer start) of a k-mer of this super-k-mer
// bits 36..39  jmax  : largest such offset                       (requires k - m <= 15)
// bits 40..62  csid  : colour-set id of the unitig (u2c folded in; index.hpp:37 in the reference)
// bit  63      tag   : 0 = record, 1 = reference into the overflow array {offset:32, count:31}
constexpr uint64_t REC_TAG = 1ULL << 63;
constexpr uint32_t REC_MAX_CSID = (1u << 23) - 1;
// empty slot: jmin=15 > jmax=0 never matches
constexpr uint64_t REC_EMPTY = (15ULL << 32);

FG_HD uint64_t rec_pack(uint32_t pos, bool fwd_canonical, uint32_t jmin, uint32_t jmax, uint32_t csid) {
    return (uint64_t)(pos & 0x7FFFFFFFu) | ((uint64_t)fwd_canonical << 31) | ((uint64_t)jmin << 32) | ((uint64_t)jmax << 36) |
           ((uint64_t)csid << 40);
}
FG_HD uint32_t rec_pos(uint64_t r) { return (uint32_t)r & 0x7FFFFFFFu; }
FG_HD bool rec_fwd(uint64_t r) { return (r >> 31) & 1u; }
// is the L-mer its own canonical form?
FG_HD bool is_fwd_canonical(uint32_t lo, uint32_t hi, uint32_t L) {
    return lmer_key(lo, hi) <= lmer_key(rc_plane(lo, L), rc_plane(hi, L));
}
FG_HD uint32_t rec_jmin(uint64_t r) { return (uint32_t)(r >> 32) & 15u; }
FG_HD uint32_t rec_jmax(uint64_t r) { return (uint32_t)(r >> 36) & 15u; }
FG_HD uint32_t rec_csid(uint64_t r) { return (uint32_t)(r >> 40) & REC_MAX_CSID; }
FG_HD uint64_t ovf_pack(uint32_t off, uint32_t cnt) { return REC_TAG | (uint64_t)off | ((uint64_t)cnt << 32); }
FG_HD uint32_t ovf_off(uint64_t r) { return (uint32_t)r; }
FG_HD uint32_t ovf_cnt(uint64_t r) { return (uint32_t)(r >> 32) & 0x7FFFFFFFu; }

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
