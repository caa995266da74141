// Encoder for the hybrid colour-set codec + the packed-block form the device decodes.
//
// Produces exactly the bit stream hybrid::builder::encode_color_set does in the reference
// (include/color_sets/hybrid.hpp:37-95): delta(size), then
//   size <  sparse_thr : delta(first), delta(gap-1)...
//   size <  dense_thr  : raw bitmap of num_colors bits
//   else               : delta(first missing), delta(gap-1)... over the num_colors-size missing colours
#pragma once
#include <stdexcept>
#include <thread>
#include "bits.hpp"
#include "index_model.hpp"

namespace fg {

struct HybridEncoder {
    uint32_t n = 0, sparse_thr = 0, dense_thr = 0;
    BitWriter bw;
    std::vector<uint64_t> offsets;

    void init(uint64_t num_colors) {
        n = (uint32_t)num_colors;
        sparse_thr = (uint32_t)(0.25 * n);  // hybrid.hpp:20
        dense_thr = (uint32_t)(0.75 * n);   // hybrid.hpp:21
        offsets.assign(1, 0);
    }

    void encode(const uint32_t* set, uint64_t size) {
        bw.delta(size);
        if (size < sparse_thr) {
            uint32_t prev = set[0];
            bw.delta(prev);
            for (uint64_t i = 1; i < size; ++i) {
                bw.delta(set[i] - (prev + 1));
                prev = set[i];
            }
        } else if (size < dense_thr) {
            std::vector<uint64_t> bm((n + 63) / 64, 0);
            for (uint64_t i = 0; i < size; ++i) bm[set[i] >> 6] |= 1ULL << (set[i] & 63);
            bw.append_stream(bm, n);
        } else {
            // walk the missing colours in increasing order
            uint32_t prev = 0;
            bool first = true;
            uint64_t i = 0;
            for (uint32_t c = 0; c < n; ++c) {
                if (i < size && set[i] == c) { ++i; continue; }
                if (first) { bw.delta(c); first = false; }
                else bw.delta(c - (prev + 1));
                prev = c;
            }
        }
        offsets.push_back(bw.nbits);
    }

    void finish(HybridSets& h) {
        h.num_colors = n;
        h.sparse_thr = sparse_thr;
        h.dense_thr = dense_thr;
        h.offsets.swap(offsets);
        h.nbits = bw.nbits;
        h.bits.swap(bw.words);
        h.bits.resize((h.nbits + 63) / 64 + 4, 0);
    }
};

// classify a list from its header; returns the encoding, fills size / number of gap codes / bit
// position right after the header
inline int hybrid_header(const HybridSets& h, uint64_t id, uint32_t& size, uint32_t& ncodes, uint64_t& body) {
    BitReader r(h.bits.data(), h.offsets[id]);
    size = (uint32_t)r.delta();
    body = r.pos;
    if (size < h.sparse_thr) { ncodes = size; return ENC_DELTA_GAPS; }
    if (size < h.dense_thr) { ncodes = 0; return ENC_BITMAP; }
    ncodes = h.num_colors - size;
    return ENC_COMPLEMENT;
}

// decode a whole set (host utility)
inline void hybrid_decode(const HybridSets& h, uint64_t id, std::vector<uint32_t>& out) {
    out.clear();
    uint32_t size, ncodes; uint64_t body;
    int enc = hybrid_header(h, id, size, ncodes, body);
    BitReader r(h.bits.data(), body);
    if (enc == ENC_BITMAP) {
        for (uint32_t c = 0; c < h.num_colors; ++c) {
            uint64_t p = body + c;
            if ((h.bits[p >> 6] >> (p & 63)) & 1) out.push_back(c);
        }
        return;
    }
    uint32_t prev = 0xFFFFFFFFu;
    if (enc == ENC_DELTA_GAPS) {
        for (uint32_t i = 0; i < ncodes; ++i) { prev = prev + 1 + (uint32_t)r.delta(); out.push_back(prev); }
    } else {
        uint32_t c = 0;
        for (uint32_t i = 0; i < ncodes; ++i) {
            prev = prev + 1 + (uint32_t)r.delta();
            for (; c < prev; ++c) out.push_back(c);
            c = prev + 1;
        }
        for (; c < h.num_colors; ++c) out.push_back(c);
    }
}

// cut a sorted list of distinct values into device blocks (common/kmer_common.h):
// emit(start, width, count field, data words, first value of the block, values in the block)
template <typename Emit>
inline void cut_blocks(const uint32_t* vals, uint32_t n, Emit emit) {
    for (uint32_t i = 0; i < n;) {
        const uint32_t cnt = std::min(BLK_VALUES, n - i);
        const uint32_t origin = vals[i] & ~31u;
        if (vals[i + cnt - 1] - origin < BLK_CHUNK_SPAN) {  // dense here: bitmap chunk
            uint32_t j = i + cnt;
            while (j < n && vals[j] - origin < BLK_CHUNK_SPAN) ++j;
            const uint32_t nw = ((vals[j - 1] - origin) >> 5) + 1;
            emit(origin, BLK_CHUNK_WIDTH, nw, nw, vals + i, j - i);
            i = j;
        } else {
            const uint32_t start = i ? vals[i - 1] + 1 : 0u, span = vals[i + cnt - 1] - start;
            uint32_t width = 0;
            while ((span >> width) != 0) ++width;
            emit(start, width, cnt, (uint32_t)(((uint64_t)cnt * width + 31) / 32), vals + i, cnt);
            i += cnt;
        }
    }
}
inline void write_block_words(uint32_t* w, uint32_t start, uint32_t width, const uint32_t* v, uint32_t nv) {
    if (width == BLK_CHUNK_WIDTH) {
        for (uint32_t i = 0; i < nv; ++i) w[(v[i] - start) >> 5] |= 1u << ((v[i] - start) & 31);
    } else {
        for (uint32_t i = 0; i < nv && width; ++i) {
            const uint64_t f = (uint64_t)(v[i] - start) << ((i * width) & 31);
            w[(i * width) >> 5] |= (uint32_t)f;
            if (f >> 32) w[((i * width) >> 5) + 1] |= (uint32_t)(f >> 32);
        }
    }
}

// packed 64-value blocks for every gap-coded list (multi-threaded over lists); see common/kmer_common.h
inline void hybrid_build_blocks(HybridSets& h, unsigned nthreads = 0) {
    const uint64_t ns = h.num_sets();
    if (h.num_colors > BLK_MAX_COLORS) throw std::runtime_error("more than 2^27 colours are not supported");
    h.set_size.assign(ns, 0);
    h.blk_first.assign(ns + 1, 0);
    h.blk_wbase.assign(ns, 0);
    std::vector<uint64_t> nwords(ns + 1, 0);
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    auto run = [&](auto fn) {
        std::vector<std::thread> th;
        uint64_t chunk = (ns + nthreads - 1) / nthreads;
        for (unsigned t = 0; t < nthreads; ++t) {
            uint64_t a = std::min<uint64_t>(ns, t * chunk), b = std::min<uint64_t>(ns, a + chunk);
            if (a < b) th.emplace_back(fn, a, b);
        }
        for (auto& x : th) x.join();
    };
    // walks the blocks of one list: emit(start, width, count, nwords, values of the block, number of values)
    auto walk = [&](uint64_t id, std::vector<uint32_t>& vals, auto emit) {
        uint32_t size, ncodes; uint64_t body;
        hybrid_header(h, id, size, ncodes, body);
        h.set_size[id] = size;
        if (!ncodes) return;
        BitReader r(h.bits.data(), body);
        vals.resize(ncodes);
        uint32_t prev = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < ncodes; ++i) { prev = prev + 1 + (uint32_t)r.delta(); vals[i] = prev; }
        cut_blocks(vals.data(), ncodes, emit);
    };
    run([&](uint64_t a, uint64_t b) {
        std::vector<uint32_t> vals;
        for (uint64_t id = a; id < b; ++id)
            walk(id, vals, [&](uint32_t, uint32_t, uint32_t, uint32_t nw, const uint32_t*, uint32_t) {
                ++h.blk_first[id + 1];
                nwords[id + 1] += nw;
            });
    });
    // A list of two or more blocks keeps its headers (two words each) in front of its block data, so that the header
    // fetch and the data fetch of a short list touch the same 128-byte line; a single block's header travels in the
    // per-set descriptor instead. Every list starts on an even word (headers are read as u64), rel_word counts from
    // the start of the list's region, headers included (a gap-coded list owns fewer than n/4 <= 2^25 codes of at most 27
    // bits and one header per 64 codes at most: always below 2^26 words).
    uint64_t begin_of_next = 0;
    for (uint64_t i = 0; i < ns; ++i) {
        const uint64_t nb = h.blk_first[i + 1];
        uint64_t tot = nwords[i + 1] + (nb >= 2 ? 2 * nb : 0);
        tot += tot & 1;
        h.blk_first[i + 1] += h.blk_first[i];
        // The memory system moves 128-byte lines (32 words) and the intersection kernel is bound by the number of lines it
        // requests: a region of at most one line does not straddle a line boundary, a longer one starts on one.
        uint64_t at = begin_of_next;
        if (tot > 32 ? (at & 31) != 0 : (at >> 5) != ((at + tot - 1) >> 5)) at = (at + 31) & ~(uint64_t)31;
        h.blk_wbase[i] = at;
        begin_of_next = at + tot;
        nwords[i + 1] = begin_of_next;
    }
    h.blk_hdr.assign(h.blk_first[ns], 0);
    h.blk_words.assign(nwords[ns] + 64, 0);  // a wave reads up to 64 * 27 bits + 1 word past a block's start
    run([&](uint64_t a, uint64_t b) {
        std::vector<uint32_t> vals;
        for (uint64_t id = a; id < b; ++id) {
            const uint64_t nb = h.blk_first[id + 1] - h.blk_first[id];
            uint64_t* hdr = h.blk_hdr.data() + h.blk_first[id];
            uint32_t* base = h.blk_words.data() + h.blk_wbase[id];
            uint64_t rel = nb >= 2 ? 2 * nb : 0;
            uint64_t j = 0;
            walk(id, vals, [&](uint32_t start, uint32_t width, uint32_t cnt, uint32_t nw, const uint32_t* v, uint32_t nv) {
                const uint64_t hd = blk_pack(start, width, cnt, (uint32_t)rel);
                hdr[j] = hd;
                if (nb >= 2) {
                    base[2 * j] = (uint32_t)hd;
                    base[2 * j + 1] = (uint32_t)(hd >> 32);
                }
                ++j;
                write_block_words(base + rel, start, width, v, nv);
                rel += nw;
            });
        }
    });
}

}  // namespace fg
