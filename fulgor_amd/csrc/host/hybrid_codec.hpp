// Encoder for the hybrid colour-set codec + decoder restart samples.
//
// Produces exactly the bit stream hybrid::builder::encode_color_set does in the reference
// (include/color_sets/hybrid.hpp:37-95): delta(size), then
//   size <  sparse_thr : delta(first), delta(gap-1)...
//   size <  dense_thr  : raw bitmap of num_colors bits
//   else               : delta(first missing), delta(gap-1)... over the num_colors-size missing colours
#pragma once
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
        h.bits.resize((h.nbits + 63) / 64 + 2, 0);
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

// decode a whole set (host utility, used for export and samples)
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

// restart samples for every gap-coded list (multi-threaded over lists)
inline void hybrid_build_samples(HybridSets& h, unsigned nthreads = 0) {
    const uint64_t ns = h.num_sets();
    h.sample_off.assign(ns + 1, 0);
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    // pass 1: counts
    auto count_range = [&](uint64_t a, uint64_t b) {
        for (uint64_t id = a; id < b; ++id) {
            uint32_t size, ncodes; uint64_t body;
            hybrid_header(h, id, size, ncodes, body);
            h.sample_off[id + 1] = ncodes ? (ncodes - 1) / SAMPLE_STRIDE : 0;
        }
    };
    auto run = [&](auto fn) {
        std::vector<std::thread> th;
        uint64_t chunk = (ns + nthreads - 1) / nthreads;
        for (unsigned t = 0; t < nthreads; ++t) {
            uint64_t a = std::min<uint64_t>(ns, t * chunk), b = std::min<uint64_t>(ns, a + chunk);
            if (a < b) th.emplace_back(fn, a, b);
        }
        for (auto& x : th) x.join();
    };
    run(count_range);
    for (uint64_t i = 0; i < ns; ++i) h.sample_off[i + 1] += h.sample_off[i];
    h.samples.assign(h.sample_off[ns], 0);
    auto fill_range = [&](uint64_t a, uint64_t b) {
        for (uint64_t id = a; id < b; ++id) {
            uint32_t size, ncodes; uint64_t body;
            hybrid_header(h, id, size, ncodes, body);
            if (ncodes <= SAMPLE_STRIDE) continue;
            BitReader r(h.bits.data(), body);
            uint32_t prev = 0xFFFFFFFFu;
            uint64_t* dst = h.samples.data() + h.sample_off[id];
            for (uint32_t i = 0; i < ncodes; ++i) {
                prev = prev + 1 + (uint32_t)r.delta();
                if ((i + 1) % SAMPLE_STRIDE == 0 && i + 1 < ncodes)
                    *dst++ = ((uint64_t)prev << 32) | (uint32_t)(r.pos - h.offsets[id]);
            }
        }
    };
    run(fill_range);
}

}  // namespace fg
