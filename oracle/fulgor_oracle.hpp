// ORACLE — test infrastructure only. Never linked into, imported by or called from the product path
// (fulgor_amd/, libfulgor_gpu.so). Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// PARITY UNPINNED: the reference (jermp/fulgor v4.2.0) ships no golden vectors or unit tests
//  and cannot be compiled here: its hot path #includes external/sshash (k-mer dictionary,
// streaming_query), external/sshash/external/pthash/external/bits (bit_vector, rank9, elias_fano,
// Elias-delta codes) and essentials, all un-vendored git submodules whose pinned commits are unknown
// (.gitmodules: sshash `branch = v4`). This file restates, in plain C++17, the algorithms that ARE in
// the reference tree, each function citing the file:line it follows, and the published semantics of
// the missing `bits` primitives as fixed by their call sites. It is pinned by (a) the reference's own
// executable specification util::check_intersection / util::check_union (include/util.hpp:106-208),
// restated below as brute_force_*, and (b) an independent k-mer-level oracle computed straight from
// the genomes (oracle/kmer_oracle.py -> tests/golden/).
//
// What is restated:
//   bits::bit_vector iterator / Elias gamma+delta   -> BitStream, BitCursor          
//   hybrid::builder::encode_color_set               -> HybridColors::encode          hybrid.hpp:37-95
//   hybrid::forward_iterator                        -> HybridCursor                  hybrid.hpp:151-305
//   intersect (HYBRID)                              -> hybrid_intersect              ps_full_intersection.cpp:32-127
//   merge (HYBRID)                                  -> hybrid_merge                  ps_threshold_union.cpp:16-40
//   index::fetch_color_set_ids                      -> Index::fetch_color_set_ids    ps_full_intersection.cpp:334-374
//   index::pseudoalign_full_intersection            -> Index::full_intersection      ps_full_intersection.cpp:376-400
//   index::pseudoalign_threshold_union              -> Index::threshold_union        ps_threshold_union.cpp:320-402
//   index::load (dump text format)                  -> Index::load_dump              src/index.cpp:122-305
//   util::vec_to_tsv / ascii formatter              -> format_ascii                  util.hpp:245-261, ps_utils.cpp:55-71
// The k-mer dictionary (sshash, absent) is replaced by an exact map canonical k-mer -> unitig id;
// lookup_advanced's contract as used by the call sites is: valid ACGT window present in the dBG in
// either orientation -> {kmer found, contig_id}; otherwise invalid.
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------------------------
// bit streams (LSB-first inside 64-bit words) and integer codes
// ---------------------------------------------------------------------------------------------
struct BitStream {
    std::vector<uint64_t> w;
    uint64_t n = 0;

    void push_bits(uint64_t v, unsigned len) {  // bits::bit_vector::builder::append_bits
        for (unsigned i = 0; i < len; ++i) push_bit((v >> i) & 1);
    }
    void push_bit(bool b) {
        if ((n & 63) == 0) w.push_back(0);
        if (b) w.back() |= 1ULL << (n & 63);
        ++n;
    }
    bool get(uint64_t i) const { return (w[i >> 6] >> (i & 63)) & 1; }
    void seal() { w.resize((n + 63) / 64 + 2, 0); }  // padding for cursors
};

struct BitCursor {  // bits::bit_vector::iterator as used by the reference
    const BitStream* s = nullptr;
    uint64_t p = 0;
    BitCursor() {}
    BitCursor(const BitStream* bs, uint64_t pos) : s(bs), p(pos) {}
    uint64_t position() const { return p; }
    void skip_to(uint64_t pos) { p = pos; }
    uint64_t window() const {  // the 64 bits starting at the cursor (streams are sealed with padding)
        const unsigned sh = p & 63;
        uint64_t v = s->w[p >> 6] >> sh;
        if (sh) v |= s->w[(p >> 6) + 1] << (64 - sh);
        return v;
    }
    uint64_t take(unsigned len) {
        if (len == 0) return 0;
        uint64_t v = window();
        if (len < 64) v &= (1ULL << len) - 1;
        p += len;
        return v;
    }
    uint64_t next() {  // position of the next set bit at or after the cursor; cursor moves past it
        for (;;) {
            uint64_t v = window();
            if (v) { p += (unsigned)__builtin_ctzll(v); return p++; }
            p += 64;
        }
    }
};

static inline unsigned msb_u64(uint64_t x) { return 63u - (unsigned)__builtin_clzll(x); }

static inline void write_gamma(BitStream& b, uint64_t x) {
    uint64_t y = x + 1;
    unsigned c = msb_u64(y);
    for (unsigned i = 0; i < c; ++i) b.push_bit(0);
    b.push_bit(1);
    b.push_bits(y & ((1ULL << c) - 1), c);
}
static inline void write_delta(BitStream& b, uint64_t x) {
    uint64_t y = x + 1;
    unsigned len = msb_u64(y);
    write_gamma(b, len);
    b.push_bits(len ? (y & ((1ULL << len) - 1)) : 0, len);
}
static inline uint64_t read_gamma(BitCursor& c) {
    const uint64_t start = c.position();
    const unsigned z = (unsigned)(c.next() - start);  // zeros before the terminating one
    return (c.take(z) | (1ULL << z)) - 1;
}
static inline uint64_t read_delta(BitCursor& c) {
    uint64_t len = read_gamma(c);
    return (c.take((unsigned)len) | (1ULL << len)) - 1;
}

enum { ENC_DELTA_GAPS = 0, ENC_BITMAP = 1, ENC_COMPLEMENT = 2 };  // util.hpp:19

// ---------------------------------------------------------------------------------------------
// hybrid colour sets
// ---------------------------------------------------------------------------------------------
struct HybridColors {
    uint32_t num_colors = 0, sparse_thr = 0, dense_thr = 0;
    std::vector<uint64_t> offsets;  // plain copy of the Elias-Fano m_offsets
    BitStream bits;

    void init(uint32_t n) {  // hybrid.hpp:12-35
        num_colors = n;
        sparse_thr = (uint32_t)(0.25 * n);
        dense_thr = (uint32_t)(0.75 * n);
        offsets.assign(1, 0);
    }
    uint64_t num_sets() const { return offsets.size() - 1; }

    void encode(const uint32_t* set, uint64_t size) {  // hybrid.hpp:37-95
        write_delta(bits, size);
        if (size < sparse_thr) {
            write_delta(bits, set[0]);
            for (uint64_t i = 1; i < size; ++i) write_delta(bits, set[i] - set[i - 1] - 1);
        } else if (size < dense_thr) {
            std::vector<bool> bm(num_colors, false);
            for (uint64_t i = 0; i < size; ++i) bm[set[i]] = true;
            for (uint32_t c = 0; c < num_colors; ++c) bits.push_bit(bm[c]);
        } else {
            std::vector<bool> in(num_colors, false);
            for (uint64_t i = 0; i < size; ++i) in[set[i]] = true;
            int64_t prev = -1;
            for (uint32_t c = 0; c < num_colors; ++c) {
                if (in[c]) continue;
                write_delta(bits, (uint64_t)(c - (prev + 1)));  // first: c itself (prev = -1)
                prev = c;
            }
        }
        offsets.push_back(bits.n);
    }
    void seal() { bits.seal(); }
};

struct HybridCursor {  // hybrid::forward_iterator, hybrid.hpp:151-305
    const HybridColors* h = nullptr;
    uint64_t begin = 0, bitmap_begin = 0;
    int enc = 0;
    BitCursor it;
    uint32_t pos_in_set = 0, set_size = 0, pos_in_comp = 0, comp_size = 0;
    uint32_t comp_val = 0, prev_val = 0, curr_val = 0;

    HybridCursor() {}
    HybridCursor(const HybridColors* hc, uint64_t b) : h(hc), begin(b) { rewind(); }

    uint32_t num_colors() const { return h->num_colors; }
    uint32_t size() const { return set_size; }
    int encoding_type() const { return enc; }
    uint32_t value() const { return curr_val; }
    uint32_t comp_value() const { return comp_val; }

    void rewind() {  // :162-189
        pos_in_set = pos_in_comp = comp_size = 0;
        comp_val = (uint32_t)-1;
        prev_val = (uint32_t)-1;
        curr_val = 0;
        it = BitCursor(&h->bits, begin);
        set_size = (uint32_t)read_delta(it);
        if (set_size < h->sparse_thr) {
            enc = ENC_DELTA_GAPS;
            curr_val = (uint32_t)read_delta(it);
        } else if (set_size < h->dense_thr) {
            enc = ENC_BITMAP;
            bitmap_begin = it.position();
            curr_val = (uint32_t)(it.next() - bitmap_begin);
        } else {
            enc = ENC_COMPLEMENT;
            comp_size = num_colors() - set_size;
            if (comp_size > 0) comp_val = (uint32_t)read_delta(it);
            advance_past_complemented();
        }
    }
    void reinit_for_complemented_set_iteration() {  // :193-205
        pos_in_comp = 0;
        prev_val = (uint32_t)-1;
        curr_val = 0;
        it = BitCursor(&h->bits, begin);
        read_delta(it);
        comp_val = comp_size > 0 ? (uint32_t)read_delta(it) : num_colors();
    }
    void next() {  // :211-238
        if (enc == ENC_COMPLEMENT) {
            ++curr_val;
            if (curr_val >= num_colors()) { curr_val = num_colors(); return; }
            advance_past_complemented();
        } else {
            ++pos_in_set;
            if (pos_in_set >= set_size) { curr_val = num_colors(); return; }
            if (enc == ENC_DELTA_GAPS) {
                prev_val = curr_val;
                curr_val = (uint32_t)read_delta(it) + prev_val + 1;
            } else {
                curr_val = (uint32_t)(it.next() - bitmap_begin);
            }
        }
    }
    void next_comp() {  // :240-248
        ++pos_in_comp;
        if (pos_in_comp >= comp_size) { comp_val = num_colors(); return; }
        prev_val = comp_val;
        comp_val = (uint32_t)read_delta(it) + prev_val + 1;
    }
    void next_geq(uint32_t lower) {  // :254-264 (only reached for non-complement lists on the query path)
        if (enc == ENC_COMPLEMENT) {
            if (value() > lower) return;
            while (comp_val < lower) {
                ++pos_in_comp;
                if (pos_in_comp >= comp_size) break;
                prev_val = comp_val;
                comp_val = (uint32_t)read_delta(it) + prev_val + 1;
            }
            curr_val = lower + (comp_val == lower);
        } else {
            while (value() < lower) next();
        }
    }

private:
    void advance_past_complemented() {  // next_comp_val, :286-294
        while (curr_val == comp_val) {
            ++curr_val;
            ++pos_in_comp;
            if (pos_in_comp >= comp_size) break;
            prev_val = comp_val;
            comp_val = (uint32_t)read_delta(it) + prev_val + 1;
        }
    }
};

// ps_full_intersection.cpp:32-127
static inline void hybrid_intersect(std::vector<HybridCursor>& its, std::vector<uint32_t>& colors,
                                    std::vector<uint32_t>& complement_set) {
    if (its.empty()) return;
    std::sort(its.begin(), its.end(), [](const HybridCursor& a, const HybridCursor& b) { return a.size() < b.size(); });
    const uint32_t n = its[0].num_colors();
    size_t num_sparse = 0;
    while (num_sparse != its.size() && its[num_sparse].encoding_type() != ENC_COMPLEMENT) ++num_sparse;

    if (num_sparse == 0) {  // all lists are complemented: union of the complements, then invert (:51-91)
        for (auto& c : its) c.reinit_for_complemented_set_iteration();
        uint32_t cand = n;
        for (auto& c : its) cand = std::min(cand, c.comp_value());
        while (cand < n) {
            uint32_t nxt = n;
            for (auto& c : its) {
                if (c.comp_value() == cand) c.next_comp();
                nxt = std::min(nxt, c.comp_value());
            }
            complement_set.push_back(cand);
            cand = nxt;
        }
        uint32_t c = 0;
        for (uint32_t x : complement_set) {
            for (; c < x; ++c) colors.push_back(c);
            c = x + 1;
        }
        for (; c < n; ++c) colors.push_back(c);
        return;
    }

    std::vector<bool> keep(n, true);  // :93-101
    for (size_t i = num_sparse; i < its.size(); ++i) {
        HybridCursor c = its[i];
        c.reinit_for_complemented_set_iteration();
        for (; c.comp_value() < n; c.next_comp()) keep[c.comp_value()] = false;
    }
    uint32_t cand = its[0].value();  // leapfrog over the non-complemented lists (:105-126)
    size_t i = 1;
    while (cand < n) {
        for (; i != num_sparse; ++i) {
            its[i].next_geq(cand);
            uint32_t v = its[i].value();
            if (v != cand) { cand = v; i = 0; break; }
        }
        if (i == num_sparse) {
            if (keep[cand]) colors.push_back(cand);
            its[0].next();
            cand = its[0].value();
            i = 1;
        }
    }
}

struct ScoredCursor {
    HybridCursor item;
    uint32_t score;
};

// ps_threshold_union.cpp:16-40
static inline void hybrid_merge(std::vector<ScoredCursor>& its, std::vector<uint32_t>& colors, int64_t min_score) {
    if (its.empty()) return;
    const uint32_t n = its[0].item.num_colors();
    std::vector<int32_t> scores(n, 0);
    for (auto& sc : its) {
        if (sc.item.encoding_type() == ENC_COMPLEMENT) {
            sc.item.reinit_for_complemented_set_iteration();
            min_score -= sc.score;
            for (; sc.item.comp_value() < n; sc.item.next_comp()) scores[sc.item.comp_value()] -= (int32_t)sc.score;
        } else {
            const uint32_t sz = sc.item.size();
            for (uint32_t i = 0; i < sz; ++i, sc.item.next()) scores[sc.item.value()] += (int32_t)sc.score;
        }
    }
    for (uint32_t c = 0; c < n; ++c)
        if (scores[c] >= min_score) colors.push_back(c);
}

// util.hpp:106-158 — executable specification of the intersection
static inline std::vector<uint32_t> brute_force_intersection(std::vector<HybridCursor> its) {
    std::vector<uint32_t> acc;
    for (size_t i = 0; i < its.size(); ++i) {
        its[i].rewind();
        std::vector<uint32_t> s;
        for (uint32_t v = its[i].value(); v < its[i].num_colors(); its[i].next(), v = its[i].value()) s.push_back(v);
        if (i == 0) { acc.swap(s); continue; }
        std::vector<uint32_t> t;
        std::set_intersection(acc.begin(), acc.end(), s.begin(), s.end(), std::back_inserter(t));
        acc.swap(t);
    }
    return acc;
}
// util.hpp:160-208 — executable specification of the threshold union
static inline std::vector<uint32_t> brute_force_union(std::vector<ScoredCursor> its, uint64_t min_score) {
    std::vector<uint32_t> out;
    if (its.empty()) return out;
    const uint32_t n = its[0].item.num_colors();
    std::vector<uint32_t> scores(n, 0);
    for (auto& sc : its) {
        sc.item.rewind();
        for (uint32_t v = sc.item.value(); v < n; sc.item.next(), v = sc.item.value()) scores[v] += sc.score;
    }
    for (uint32_t c = 0; c < n; ++c)
        if (scores[c] >= min_score) out.push_back(c);
    return out;
}

// ---------------------------------------------------------------------------------------------
// exact k-mer map (stands in for sshash::dictionary) + a streaming query (stands in for
// sshash::streaming_query: extend the previous hit along its unitig before falling back to a lookup)
// ---------------------------------------------------------------------------------------------
static inline int nuc(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

struct KmerMap {
    static constexpr uint64_t FWD_IS_CANON = 1ULL << 63;
    uint64_t mask = 0;
    std::vector<uint64_t> keys;  // (canonical k-mer + 1) | FWD_IS_CANON flag; 0 = empty
    std::vector<uint64_t> vals;  // unitig id << 32 | start position in the unitig concatenation
    static uint64_t h(uint64_t x) {
        x ^= x >> 31; x *= 0x7fb5d329728ea185ULL; x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL; x ^= x >> 33;
        return x;
    }
    void reserve(uint64_t nk) {
        uint64_t cap = 16;
        while (cap < nk * 2) cap <<= 1;
        keys.assign(cap, 0);
        vals.assign(cap, 0);
        mask = cap - 1;
    }
    void put(uint64_t canon, bool fwd_is_canon, uint32_t unitig, uint32_t pos) {
        for (uint64_t i = h(canon) & mask;; i = (i + 1) & mask) {
            if (keys[i] == 0) {
                keys[i] = (canon + 1) | (fwd_is_canon ? FWD_IS_CANON : 0);
                vals[i] = ((uint64_t)unitig << 32) | pos;
                return;
            }
            if ((keys[i] & ~FWD_IS_CANON) == canon + 1) throw std::runtime_error("k-mer occurs in two unitigs");
        }
    }
    int64_t slot(uint64_t canon) const {
        for (uint64_t i = h(canon) & mask;; i = (i + 1) & mask) {
            if (keys[i] == 0) return -1;
            if ((keys[i] & ~FWD_IS_CANON) == canon + 1) return (int64_t)i;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// the index (hybrid colour sets)
// ---------------------------------------------------------------------------------------------
struct Index {
    uint32_t k = 0;
    KmerMap k2u;
    std::string ubases;               // unitig concatenation
    std::vector<uint64_t> uoff;       // unitig boundaries
    std::vector<uint32_t> u2c_table;  // u2c(unitig) = rank1(m_u2c, unitig), index.hpp:37
    HybridColors colors;
    std::vector<std::string> filenames;

    uint64_t num_colors() const { return colors.num_colors; }
    uint32_t u2c(uint64_t unitig) const { return u2c_table[unitig]; }

    // Streams the k-mers of `s`: calls hit(unitig id) for every positive k-mer, in order. Windows with
    // a non-ACGT base are negative. Same answers as one exact lookup per k-mer.
    template <typename F>
    void stream_kmers(const char* s, uint64_t len, F hit) const {
        const uint64_t km = (1ULL << (2 * k)) - 1;
        uint64_t fw = 0, rv = 0;
        uint32_t run = 0;
        bool have = false, same = false;
        uint64_t unitig = 0, pos = 0;
        for (uint64_t i = 0; i < len; ++i) {
            const int c = nuc(s[i]);
            if (c < 0) { run = 0; have = false; continue; }
            fw = ((fw << 2) | (uint64_t)c) & km;
            rv = (rv >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
            if (++run < k) continue;
            if (have) {  // try to extend along the unitig of the previous hit
                if (same && pos + k < uoff[unitig + 1] && nuc(ubases[pos + k]) == c) { ++pos; hit(unitig); continue; }
                if (!same && pos > uoff[unitig] && 3 - nuc(ubases[pos - 1]) == c) { --pos; hit(unitig); continue; }
            }
            const uint64_t canon = fw < rv ? fw : rv;
            const int64_t sl = k2u.slot(canon);
            if (sl < 0) { have = false; continue; }
            unitig = k2u.vals[sl] >> 32;
            pos = (uint32_t)k2u.vals[sl];
            same = ((k2u.keys[sl] & KmerMap::FWD_IS_CANON) != 0) == (fw == canon);
            have = true;
            hit(unitig);
        }
    }

    // one answer per k-mer window, in order: f(position, unitig id or -1). Same lookups as stream_kmers.
    template <typename F>
    void lookup_every_kmer(const char* s, uint64_t len, F f) const {
        if (len < k) return;
        std::vector<int64_t> ans(len - k + 1, -1);
        const uint64_t km = (1ULL << (2 * k)) - 1;
        uint64_t fw = 0, rv = 0;
        uint32_t run = 0;
        for (uint64_t i = 0; i < len; ++i) {
            const int c = nuc(s[i]);
            if (c < 0) { run = 0; continue; }
            fw = ((fw << 2) | (uint64_t)c) & km;
            rv = (rv >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
            if (++run < k) continue;
            const int64_t sl = k2u.slot(fw < rv ? fw : rv);
            if (sl >= 0) ans[i + 1 - k] = (int64_t)(k2u.vals[sl] >> 32);
        }
        for (uint64_t p = 0; p < ans.size(); ++p) f(p, ans[p]);
    }

    struct Triple { uint32_t start_pos_in_query, num_kmers, color_set_id; };  // kmer_conservation_triple, util.hpp:74-78

    // src/kmer_conservation.cpp:7-54
    void kmer_conservation(const char* seq, uint64_t len, std::vector<Triple>& out) const {
        if (len < k) return;
        out.clear();
        const uint64_t invalid = (uint64_t)-1;
        Triple kct = {0, 0, 0};
        uint64_t prev = invalid;
        auto push = [&]() {
            if (prev != invalid) { kct.color_set_id = (uint32_t)prev; out.push_back(kct); }
        };
        lookup_every_kmer(seq, len, [&](uint64_t i, int64_t unitig) {
            if (unitig >= 0) {
                const uint64_t cs = u2c((uint64_t)unitig);
                if (prev != cs) { push(); kct.num_kmers = 0; kct.start_pos_in_query = (uint32_t)i; }
                kct.num_kmers += 1;
                prev = cs;
            } else {
                push();
                prev = invalid;
            }
        });
        push();
    }

    // src/kmer_matches.cpp:7-30 (hybrid colour sets)
    void kmer_matches(const char* seq, uint64_t len, std::vector<uint8_t>& positive, std::vector<uint32_t>& counts) const {
        std::fill(counts.begin(), counts.end(), 0);
        positive.clear();
        if (len < k) return;
        positive.assign(len - k + 1, 0);
        lookup_every_kmer(seq, len, [&](uint64_t i, int64_t unitig) {
            if (unitig < 0) return;
            positive[i] = 1;
            HybridCursor it = color_set(u2c((uint64_t)unitig));
            const uint32_t sz = it.size();
            for (uint32_t j = 0; j != sz; ++j, it.next()) counts[it.value()] += 1;
        });
    }

    void add_unitigs(const char* bases, const uint64_t* off, const uint32_t* csid, uint64_t nu) {
        ubases.assign(bases, off[nu]);
        uoff.assign(off, off + nu + 1);
        u2c_table.assign(csid, csid + nu);
        uint64_t nk = 0;
        for (uint64_t u = 0; u < nu; ++u) nk += off[u + 1] - off[u] - k + 1;
        k2u.reserve(nk);
        const uint64_t km = (1ULL << (2 * k)) - 1;
        for (uint64_t u = 0; u < nu; ++u) {
            uint64_t fw = 0, rv = 0;
            for (uint64_t i = off[u]; i < off[u + 1]; ++i) {
                const int c = nuc(bases[i]);
                if (c < 0) throw std::runtime_error("unitig contains a non-ACGT character");
                fw = ((fw << 2) | (uint64_t)c) & km;
                rv = (rv >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
                if (i + 1 >= off[u] + k) k2u.put(fw < rv ? fw : rv, fw < rv, (uint32_t)u, (uint32_t)(i + 1 - k));
            }
        }
    }

    // adopt an already encoded hybrid stream (same layout as hybrid.hpp:338-345 after EF decoding)
    void set_colors(uint32_t n, uint32_t sparse_thr, uint32_t dense_thr, const uint64_t* words, uint64_t nbits,
                    const uint64_t* offsets, uint64_t num_sets) {
        colors.num_colors = n;
        colors.sparse_thr = sparse_thr;
        colors.dense_thr = dense_thr;
        colors.offsets.assign(offsets, offsets + num_sets + 1);
        colors.bits.w.assign(words, words + (nbits + 63) / 64);
        colors.bits.n = nbits;
        colors.seal();
    }

    // src/index.cpp:122-305 (dump text format)
    void load_dump(const std::string& base) {
        uint64_t nc = 0, nu = 0, ns = 0;
        {
            std::ifstream in(base + ".metadata.txt");
            if (!in.is_open()) throw std::runtime_error("cannot open metadata file");
            std::string line;
            while (std::getline(in, line)) {
                size_t eq = line.find('=');
                std::string key = line.substr(0, eq);
                uint64_t v = std::strtoull(line.c_str() + eq + 1, nullptr, 10);
                if (key == "k") k = (uint32_t)v;
                if (key == "num_colors") nc = v;
                if (key == "num_unitigs") nu = v;
                if (key == "num_color_sets") ns = v;
            }
        }
        {
            std::ifstream in(base + ".filenames.txt");
            std::string f;
            while (in >> f) filenames.push_back(f);
        }
        {
            std::ifstream in(base + ".color_sets.txt");
            if (!in.is_open()) throw std::runtime_error("cannot open color sets file");
            colors.init((uint32_t)nc);
            std::string line;
            std::vector<uint32_t> v;
            for (uint64_t i = 0; i < ns; ++i) {
                std::getline(in, line);
                const char* p = line.c_str() + line.find("size=") + 5;
                char* e;
                uint64_t sz = std::strtoull(p, &e, 10);
                v.clear();
                for (uint64_t j = 0; j < sz; ++j) { p = e; v.push_back((uint32_t)std::strtoul(p, &e, 10)); }
                colors.encode(v.data(), v.size());
            }
            colors.seal();
        }
        {
            std::ifstream in(base + ".unitigs.fa");
            if (!in.is_open()) throw std::runtime_error("cannot open unitigs file");
            std::string bases, header, seq;
            std::vector<uint64_t> off(1, 0);
            std::vector<uint32_t> csid;
            for (uint64_t u = 0; u < nu; ++u) {
                std::getline(in, header);
                std::getline(in, seq);
                csid.push_back((uint32_t)std::strtoull(header.c_str() + header.find("color_set_id=") + 13, nullptr, 10));
                bases += seq;
                off.push_back(bases.size());
            }
            add_unitigs(bases.data(), off.data(), csid.data(), nu);
        }
    }

    HybridCursor color_set(uint64_t id) const { return HybridCursor(&colors, colors.offsets[id]); }  // hybrid.hpp:309-313

    // ps_full_intersection.cpp:334-374
    void fetch_color_set_ids(const char* seq, uint64_t len, std::vector<uint32_t>& ids) const {
        if (len < k) return;  // NB: returns before clearing, as the reference does
        std::vector<uint64_t> unitigs;
        uint64_t prev = (uint64_t)-1;
        stream_kmers(seq, len, [&](uint64_t u) {
            if (u != prev) { unitigs.push_back(u); prev = u; }
        });
        ids.clear();
        std::sort(unitigs.begin(), unitigs.end());
        unitigs.erase(std::unique(unitigs.begin(), unitigs.end()), unitigs.end());
        for (uint64_t u : unitigs) ids.push_back(u2c(u));
        std::sort(ids.begin(), ids.end());
        ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    }

    // ps_full_intersection.cpp:376-400
    void full_intersection(const std::vector<uint32_t>& ids, std::vector<uint32_t>& colors_out, std::vector<uint32_t>& tmp,
                           bool self_check = false) const {
        std::vector<HybridCursor> its;
        for (uint32_t id : ids) its.push_back(color_set(id));
        colors_out.clear();
        tmp.clear();
        std::vector<HybridCursor> copy;
        if (self_check) copy = its;
        hybrid_intersect(its, colors_out, tmp);
        if (self_check && !ids.empty() && brute_force_intersection(copy) != colors_out)
            throw std::runtime_error("oracle: intersect disagrees with check_intersection");
    }

    // ps_threshold_union.cpp:320-402
    void threshold_union(const char* seq, uint64_t len, double threshold, std::vector<uint32_t>& colors_out,
                         bool self_check = false) const {
        if (len < k) return;
        colors_out.clear();
        struct Scored { uint64_t item; uint32_t score; };
        std::vector<Scored> unitigs;
        uint64_t num_positive = 0, prev = (uint64_t)-1;
        // a negative k-mer does NOT reset prev_unitig_id (the reference only updates it on a hit)
        stream_kmers(seq, len, [&](uint64_t u) {
            ++num_positive;
            if (u != prev) { unitigs.push_back({u, 1}); prev = u; }
            else ++unitigs.back().score;
        });
        std::sort(unitigs.begin(), unitigs.end(), [](const Scored& a, const Scored& b) { return a.item < b.item; });
        std::vector<Scored> sets;
        uint64_t pu = (uint64_t)-1;
        for (auto& u : unitigs) {
            if (u.item != pu) { sets.push_back({u2c(u.item), u.score}); pu = u.item; }
            else sets.back().score += u.score;
        }
        std::sort(sets.begin(), sets.end(), [](const Scored& a, const Scored& b) { return a.item < b.item; });
        std::vector<ScoredCursor> its;
        uint64_t ps = (uint64_t)-1;
        for (auto& s : sets) {
            if (s.item != ps) { its.push_back({color_set(s.item), s.score}); ps = s.item; }
            else its.back().score += s.score;
        }
        const uint64_t min_score = (uint64_t)((double)num_positive * threshold);  // :389
        std::vector<ScoredCursor> copy;
        if (self_check) copy = its;
        hybrid_merge(its, colors_out, (int64_t)min_score);
        if (self_check && !its.empty() && brute_force_union(copy, min_score) != colors_out)
            throw std::runtime_error("oracle: merge disagrees with check_union");
    }
};

}  // namespace oracle
#include "fulgor_oracle_codecs.hpp"
namespace oracle {

// index<ColorSets> for the four codecs (include/index_types.hpp): the k-mer side is shared, the colour
// side is selected by `type` (index_t numbering of include/util.hpp:18: 0 hybrid, 1 diff, 2 meta, 3 meta-diff)
struct AnyIndex : Index {
    int type = 0;
    MetaColors meta;
    DiffColors diff;
    MetaDiffColors mdiff;

    std::vector<std::vector<uint32_t>> decode_all() const {
        std::vector<std::vector<uint32_t>> sets(colors.num_sets());
        for (uint64_t id = 0; id < colors.num_sets(); ++id) {
            HybridCursor c = color_set(id);
            for (uint32_t v = c.value(); v < colors.num_colors; c.next(), v = c.value()) sets[id].push_back(v);
        }
        return sets;
    }
    // re-encode the colour sets with another codec (same colour numbering, same colour-set ids)
    void convert(int new_type, uint32_t psize, uint32_t csize) {
        const auto sets = decode_all();
        const uint32_t n = colors.num_colors;
        if (new_type == 1) build_diff(diff, sets, n, csize);
        else if (new_type == 2) build_meta(meta, sets, n, psize);
        else if (new_type == 3) build_metadiff(mdiff, sets, n, psize, csize);
        else if (new_type != 0) throw std::runtime_error("unknown index type");
        type = new_type;
    }

    void any_full_intersection(const std::vector<uint32_t>& ids, std::vector<uint32_t>& out, std::vector<uint32_t>& tmp) const {
        out.clear();
        tmp.clear();
        if (type == 0) { full_intersection(ids, out, tmp); return; }
        if (type == 1) {
            std::vector<DiffCursor> its;
            for (uint32_t id : ids) its.push_back(diff_color_set(diff, id));
            diff_intersect(its, out);
        } else if (type == 2) {
            std::vector<MetaCursor> its;
            for (uint32_t id : ids) its.push_back(meta_color_set(meta, id));
            meta_intersect<MetaCursor, false>(its, out, tmp);
        } else {
            std::vector<MetaDiffCursor> its;
            for (uint32_t id : ids) its.push_back(metadiff_color_set(mdiff, id));
            meta_intersect<MetaDiffCursor, true>(its, out, tmp);
        }
    }

    // ps_threshold_union.cpp:320-402 for the non-hybrid codecs (the k-mer side is identical)
    void any_threshold_union(const char* seq, uint64_t len, double threshold, std::vector<uint32_t>& out) const {
        if (type == 0) { threshold_union(seq, len, threshold, out); return; }
        if (len < k) return;
        out.clear();
        struct S { uint64_t item; uint32_t score; };
        std::vector<S> unitigs;
        uint64_t num_positive = 0, prev = (uint64_t)-1;
        stream_kmers(seq, len, [&](uint64_t u) {
            ++num_positive;
            if (u != prev) { unitigs.push_back({u, 1}); prev = u; }
            else ++unitigs.back().score;
        });
        std::sort(unitigs.begin(), unitigs.end(), [](const S& a, const S& b) { return a.item < b.item; });
        std::vector<S> sets;
        uint64_t pu = (uint64_t)-1;
        for (auto& u : unitigs) {
            if (u.item != pu) { sets.push_back({u2c(u.item), u.score}); pu = u.item; }
            else sets.back().score += u.score;
        }
        std::sort(sets.begin(), sets.end(), [](const S& a, const S& b) { return a.item < b.item; });
        std::vector<S> merged;
        for (auto& s : sets) {
            if (merged.empty() || merged.back().item != s.item) merged.push_back(s);
            else merged.back().score += s.score;
        }
        const uint64_t min_score = (uint64_t)((double)num_positive * threshold);
        if (type == 1) {
            std::vector<Scored<DiffCursor>> its;
            for (auto& s : merged) its.push_back({diff_color_set(diff, s.item), s.score});
            merge_diff(its, out, min_score);
        } else if (type == 2) {
            std::vector<Scored<MetaCursor>> its;
            for (auto& s : merged) its.push_back({meta_color_set(meta, s.item), s.score});
            merge_meta(its, out, min_score);
        } else {
            std::vector<Scored<MetaDiffCursor>> its;
            for (auto& s : merged) its.push_back({metadiff_color_set(mdiff, s.item), s.score});
            merge_metadiff(its, out, min_score);
        }
    }
};

// util.hpp:245-261 + ps_utils.cpp:55-71: "<id>\t<count>[\t<c>...]\n"
static inline void format_ascii(uint32_t query_id, const std::vector<uint32_t>& colors, std::string& out) {
    out += std::to_string(query_id);
    out += '\t';
    out += std::to_string(colors.size());
    for (uint32_t c : colors) { out += '\t'; out += std::to_string(c); }
    out += '\n';
}

// psa_compressed_formatter + formatter_buffer of ONE worker (src/ps_utils.cpp:27-46, 138-243): file =
// u64 num_colors, then blocks {u64 num_bits, data}; a block is closed when the bytes added since the last
// flush exceed 2^14 (checked after every record) and once more at the end (~formatter_buffer).
static inline void format_compressed(uint32_t first_id, const uint64_t* off, const uint32_t* colors, uint64_t n,
                                     uint32_t num_colors, std::string& out) {
    const uint64_t hdr = num_colors;  // writes 8 bytes
    out.append(reinterpret_cast<const char*>(&hdr), 8);
    const uint32_t sparse_thr = (uint32_t)(0.25 * num_colors), dense_thr = (uint32_t)(0.75 * num_colors);
    BitStream bv;
    uint32_t num_bytes = 0;
    auto flush = [&]() {
        const uint64_t nb = bv.n;
        out.append(reinterpret_cast<const char*>(&nb), 8);
        out.append(reinterpret_cast<const char*>(bv.w.data()), num_bytes);
        bv = BitStream();
        num_bytes = 0;
    };
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t* c = colors + off[r];
        const uint32_t size = (uint32_t)(off[r + 1] - off[r]);
        const size_t before = bv.w.size() * 8;
        write_delta(bv, first_id + (uint32_t)r);
        write_delta(bv, size);
        if (size == 0) {
        } else if (size < sparse_thr) {
            write_delta(bv, c[0]);
            for (uint32_t i = 1; i < size; ++i) write_delta(bv, c[i] - (c[i - 1] + 1));
        } else if (size < dense_thr) {
            std::vector<bool> bm(num_colors, false);
            for (uint32_t i = 0; i < size; ++i) bm[c[i]] = true;
            for (uint32_t v = 0; v < num_colors; ++v) bv.push_bit(bm[v]);
        } else {
            std::vector<bool> in(num_colors, false);
            for (uint32_t i = 0; i < size; ++i) in[c[i]] = true;
            int64_t prev = -1;
            for (uint32_t v = 0; v < num_colors; ++v) {
                if (in[v]) continue;
                write_delta(bv, (uint64_t)(v - (prev + 1)));
                prev = v;
            }
        }
        num_bytes += (uint32_t)(bv.w.size() * 8 - before);
        if (num_bytes > (1u << 14)) flush();
    }
    flush();
}

// inverse of format_compressed (test helper): (ids, CSR)
static inline void parse_compressed(const std::string& file, std::vector<uint32_t>& ids, std::vector<uint64_t>& off,
                                    std::vector<uint32_t>& colors) {
    uint64_t num_colors;
    memcpy(&num_colors, file.data(), 8);
    const uint32_t n = (uint32_t)num_colors, sparse_thr = (uint32_t)(0.25 * n), dense_thr = (uint32_t)(0.75 * n);
    size_t p = 8;
    off.assign(1, 0);
    while (p < file.size()) {
        uint64_t nbits;
        memcpy(&nbits, file.data() + p, 8);
        p += 8;
        BitStream bv;
        bv.w.assign((nbits + 63) / 64, 0);
        memcpy(bv.w.data(), file.data() + p, bv.w.size() * 8);
        bv.n = nbits;
        bv.seal();
        p += (nbits + 63) / 64 * 8;
        BitCursor c(&bv, 0);
        while (c.position() < nbits) {
            ids.push_back((uint32_t)read_delta(c));
            const uint32_t size = (uint32_t)read_delta(c);
            if (size == 0) {
            } else if (size < sparse_thr) {
                uint32_t v = (uint32_t)read_delta(c);
                colors.push_back(v);
                for (uint32_t i = 1; i < size; ++i) { v += (uint32_t)read_delta(c) + 1; colors.push_back(v); }
            } else if (size < dense_thr) {
                for (uint32_t v = 0; v < n; ++v)
                    if (c.take(1)) colors.push_back(v);
            } else {
                std::vector<bool> missing(n, false);
                int64_t prev = -1;
                for (uint32_t i = 0; i < n - size; ++i) { prev = prev + 1 + (int64_t)read_delta(c); missing[prev] = true; }
                for (uint32_t v = 0; v < n; ++v)
                    if (!missing[v]) colors.push_back(v);
            }
            off.push_back(colors.size());
        }
    }
}

}  // namespace oracle
