// Meta, differential and meta-differential colour sets for the MI355X engine.
//
// The encoded streams follow the reference builders bit for bit at the list level:
//   meta             partial sets = hybrid lists over the partition's colours        (meta.hpp:27-67, hybrid.hpp:37-95)
//   differential     representative: delta(size) + delta-gaps; set: delta(diff size) delta(set size) +
//                    delta-gaps of the symmetric difference with the representative   (differential.hpp:21-98)
//   meta-differential partitions whose partial sets are differential                 (meta_differential.hpp:35-73)
// All lists live in ONE bit arena (the reference keeps one bit_vector per partition; the concatenation
// is a layout choice for HBM). On top of the streams the engine keeps, per colour set, a short list of
// *ops* — (kind, colour base, universe, stream position, code count) — so that a wavefront can rebuild
// the set as an n-bit bitmap from its ops (device form of an op: build_generic_device below):
//   OR_GAPS    sparse hybrid list            -> set bits
//   OR_BITMAP  hybrid bitmap                 -> OR shifted words
//   OR_COMP    complemented hybrid list      -> fill [base, base+np), then clear the listed colours
//   XOR_GAPS   representative / differential -> toggle bits (set = representative XOR difference)
// Which colours form a partition and which sets form a cluster is decided by construction heuristics in
// the reference (sketching + k-means, out of scope); any assignment is a valid index. convert_sets() uses
// fixed-width colour ranges and runs of consecutive colour sets, and keeps the colour numbering and the
// colour-set ids of the hybrid index, so results are identical across the four codecs.
#pragma once
#include <algorithm>
#include <iterator>
#include <stdexcept>
#include <unordered_map>
#include "hybrid_codec.hpp"

namespace fg {

namespace detail {

// hybrid list of `rel` (relative colours) over a universe of np colours, appended to the arena
inline SetOp encode_partial_hybrid(BitWriter& bw, const std::vector<uint32_t>& rel, uint32_t np, uint32_t base,
                                   std::vector<uint32_t>& op_bytes) {
    const uint64_t start_bits = bw.nbits;
    struct Done { BitWriter& b; uint64_t s; std::vector<uint32_t>& o; ~Done() { o.push_back((uint32_t)((b.nbits - s + 7) / 8)); } } done{bw, start_bits, op_bytes};
    const uint32_t sparse_thr = (uint32_t)(0.25 * np), dense_thr = (uint32_t)(0.75 * np);  // hybrid.hpp:20-21 with n = np
    SetOp op{};
    op.base = base;
    op.np = np;
    const uint64_t size = rel.size();
    bw.delta(size);
    op.body = bw.nbits;
    if (size < sparse_thr) {
        op.kind = OP_OR_GAPS;
        op.ncodes = (uint32_t)size;
        bw.delta(rel[0]);
        for (uint64_t i = 1; i < size; ++i) bw.delta(rel[i] - rel[i - 1] - 1);
    } else if (size < dense_thr) {
        op.kind = OP_OR_BITMAP;
        std::vector<uint64_t> bm((np + 63) / 64, 0);
        for (uint32_t c : rel) bm[c >> 6] |= 1ULL << (c & 63);
        bw.append_stream(bm, np);
    } else {
        op.kind = OP_OR_COMP;
        op.ncodes = np - (uint32_t)size;
        uint32_t prev = 0;
        bool first = true;
        uint64_t i = 0;
        for (uint32_t c = 0; c < np; ++c) {
            if (i < size && rel[i] == c) { ++i; continue; }
            if (first) { bw.delta(c); first = false; }
            else bw.delta(c - (prev + 1));
            prev = c;
        }
    }
    return op;
}

inline void write_gaps(BitWriter& bw, const std::vector<uint32_t>& v) {
    if (v.empty()) return;
    bw.delta(v[0]);
    for (size_t i = 1; i < v.size(); ++i) bw.delta(v[i] - v[i - 1] - 1);
}

// differential container over `sets` (each a sorted list over [0, np)): clusters of `csize` consecutive
// sets; returns per set the pair (representative op, difference op)
inline void encode_differential(BitWriter& bw, std::vector<SetOp>& ops, const std::vector<std::vector<uint32_t>>& sets,
                                uint32_t np, uint32_t base, uint32_t csize, std::vector<std::pair<uint32_t, uint32_t>>& per_set,
                                uint64_t& num_clusters, std::vector<uint32_t>& op_bytes) {
    std::vector<uint32_t> cnt(np), rep, diff;
    for (size_t a = 0; a < sets.size(); a += csize) {
        const size_t b = std::min(sets.size(), a + csize);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t i = a; i < b; ++i)
            for (uint32_t c : sets[i]) ++cnt[c];
        rep.clear();
        for (uint32_t c = 0; c < np; ++c)
            if (2 * (uint64_t)cnt[c] > b - a) rep.push_back(c);
        // process_partition: delta(size) + gaps (differential.hpp:21-43)
        uint64_t start_bits = bw.nbits;
        bw.delta(rep.size());
        SetOp rop{};
        rop.kind = OP_XOR_GAPS; rop.base = base; rop.np = np; rop.body = bw.nbits; rop.ncodes = (uint32_t)rep.size();
        write_gaps(bw, rep);
        const uint32_t rep_op = (uint32_t)ops.size();
        ops.push_back(rop);
        op_bytes.push_back((uint32_t)((bw.nbits - start_bits + 7) / 8));
        ++num_clusters;
        for (size_t i = a; i < b; ++i) {  // process_color_set (differential.hpp:45-98)
            diff.clear();
            std::set_symmetric_difference(sets[i].begin(), sets[i].end(), rep.begin(), rep.end(), std::back_inserter(diff));
            start_bits = bw.nbits;
            bw.delta(diff.size());
            bw.delta(sets[i].size());
            SetOp dop{};
            dop.kind = OP_XOR_GAPS; dop.base = base; dop.np = np; dop.body = bw.nbits; dop.ncodes = (uint32_t)diff.size();
            write_gaps(bw, diff);
            per_set.push_back({rep_op, (uint32_t)ops.size()});
            ops.push_back(dop);
            op_bytes.push_back((uint32_t)((bw.nbits - start_bits + 7) / 8));
        }
    }
}

struct VecHash {
    size_t operator()(const std::vector<uint32_t>& v) const {
        uint64_t h = 1469598103934665603ULL;
        for (uint32_t c : v) h = (h ^ c) * 1099511628211ULL;
        return (size_t)h;
    }
};

}  // namespace detail

// colours an op contributes (absolute, ascending)
inline void op_members(const GenericSets& g, const SetOp& op, std::vector<uint32_t>& out) {
    out.clear();
    if (op.kind == OP_OR_BITMAP) {
        for (uint32_t c = 0; c < op.np; ++c) {
            const uint64_t p = op.body + c;
            if ((g.bits[p >> 6] >> (p & 63)) & 1) out.push_back(op.base + c);
        }
        return;
    }
    BitReader r(g.bits.data(), op.body);
    uint32_t prev = 0xFFFFFFFFu;
    if (op.kind == OP_OR_COMP) {  // the universe minus the listed colours
        uint32_t c = 0;
        for (uint32_t i = 0; i < op.ncodes; ++i) {
            prev = prev + 1 + (uint32_t)r.delta();
            for (; c < prev; ++c) out.push_back(op.base + c);
            c = prev + 1;
        }
        for (; c < op.np; ++c) out.push_back(op.base + c);
        return;
    }
    for (uint32_t i = 0; i < op.ncodes; ++i) {
        prev = prev + 1 + (uint32_t)r.delta();
        out.push_back(op.base + prev);
    }
}

// device form of every op (two passes over the ops, multi-threaded)
inline void build_generic_device(GenericSets& g, unsigned nthreads = 0) {
    if (g.num_colors > BLK_MAX_COLORS) throw std::runtime_error("more than 2^27 colours are not supported");
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    const size_t n = g.ops.size();
    std::vector<uint64_t> span_off(n + 1, 0), bop_off(n + 1, 0), blk_off(n + 1, 0), word_off(n + 1, 0);
    auto run = [&](auto fn) {
        std::vector<std::thread> th;
        const size_t per = (n + nthreads - 1) / nthreads;
        for (unsigned t = 0; t < nthreads; ++t) {
            size_t a = std::min(n, t * per), b = std::min(n, a + per);
            if (a < b) th.emplace_back(fn, a, b);
        }
        for (auto& x : th) x.join();
    };
    auto span_words = [](const SetOp& op) -> uint32_t {
        return op.np ? ((op.base + op.np - 1) >> 5) - (op.base >> 5) + 1 : 0u;
    };
    run([&](size_t a, size_t b) {
        std::vector<uint32_t> vals;
        for (size_t i = a; i < b; ++i) {
            const SetOp& op = g.ops[i];
            if (span_words(op) <= GOP_SPAN_WORDS) { span_off[i + 1] = 1; continue; }
            bop_off[i + 1] = 1;
            op_members(g, op, vals);
            cut_blocks(vals.data(), (uint32_t)vals.size(), [&](uint32_t, uint32_t, uint32_t, uint32_t nw, const uint32_t*, uint32_t) {
                ++blk_off[i + 1];
                word_off[i + 1] += nw;
            });
        }
    });
    for (size_t i = 0; i < n; ++i) {
        span_off[i + 1] += span_off[i]; bop_off[i + 1] += bop_off[i];
        blk_off[i + 1] += blk_off[i]; word_off[i + 1] += word_off[i];
    }
    if (span_off[n] >= GOP_BLOCK_REF || bop_off[n] >= GOP_BLOCK_REF) throw std::runtime_error("too many ops for 31-bit references");
    g.dev_span.assign(span_off[n] * 8 + 8, 0);
    g.dev_ops.assign(bop_off[n], GenOpDev{});
    g.dev_blk_hdr.assign(blk_off[n], 0);
    g.dev_blk_words.assign(word_off[n] + 64, 0);
    std::vector<uint32_t> ref(n, 0);
    run([&](size_t a, size_t b) {
        std::vector<uint32_t> vals;
        for (size_t i = a; i < b; ++i) {
            const SetOp& op = g.ops[i];
            op_members(g, op, vals);
            const uint32_t sw = span_words(op);
            if (sw <= GOP_SPAN_WORDS) {
                uint32_t* rec = g.dev_span.data() + span_off[i] * 8;
                const uint32_t w0 = op.base >> 5;
                rec[0] = w0 | (sw << 24);
                for (uint32_t c : vals) rec[1 + (c >> 5) - w0] |= 1u << (c & 31);
                ref[i] = (uint32_t)span_off[i];
                continue;
            }
            GenOpDev& d = g.dev_ops[bop_off[i]];
            d.begin = word_off[i];
            d.soff = blk_off[i];
            d.ncodes = (uint32_t)(blk_off[i + 1] - blk_off[i]);
            ref[i] = GOP_BLOCK_REF | (uint32_t)bop_off[i];
            uint64_t* hdr = g.dev_blk_hdr.data() + blk_off[i];
            uint32_t* base = g.dev_blk_words.data() + word_off[i];
            uint64_t rel = 0;
            cut_blocks(vals.data(), (uint32_t)vals.size(), [&](uint32_t start, uint32_t width, uint32_t cnt, uint32_t nw, const uint32_t* v, uint32_t nv) {
                *hdr++ = blk_pack(start, width, cnt, (uint32_t)rel);
                write_block_words(base + rel, start, width, v, nv);
                rel += nw;
            });
        }
    });
    g.dev_set_ops.resize(g.set_ops.size());
    for (size_t i = 0; i < g.set_ops.size(); ++i) g.dev_set_ops[i] = ref[g.set_ops[i]];
}

// re-encode the colour sets of a hybrid index with another codec
inline void convert_sets(const HybridSets& h, int type, uint32_t psize, uint32_t csize, GenericSets& g) {
    if (type != IDX_DIFF && type != IDX_META && type != IDX_META_DIFF) throw std::runtime_error("convert: unknown index type");
    if (psize == 0 || csize == 0) throw std::runtime_error("convert: partition and cluster sizes must be positive");
    const uint32_t n = h.num_colors;
    const uint64_t ns = h.num_sets();
    g = GenericSets();
    g.type = type;
    g.num_colors = n;
    g.partition_size = type == IDX_DIFF ? n : psize;
    g.cluster_size = type == IDX_META ? 0 : csize;
    BitWriter bw;
    g.set_ops_off.assign(1, 0);
    std::vector<uint32_t> set;
    std::vector<uint32_t> op_bytes;  // encoded bytes of every op's list (header included)

    if (type == IDX_DIFF) {
        g.num_partitions = 1;
        std::vector<std::vector<uint32_t>> chunk;
        std::vector<std::pair<uint32_t, uint32_t>> per_set;
        // clusters are runs of consecutive ids: stream them through in multiples of csize
        const uint64_t step = (uint64_t)csize * 4096;
        for (uint64_t a = 0; a < ns; a += step) {
            const uint64_t b = std::min(ns, a + step);
            chunk.assign(b - a, {});
            for (uint64_t id = a; id < b; ++id) hybrid_decode(h, id, chunk[id - a]);
            per_set.clear();
            detail::encode_differential(bw, g.ops, chunk, n, 0, csize, per_set, g.num_clusters, op_bytes);
            for (auto& pr : per_set) {
                g.set_ops.push_back(pr.first);
                g.set_ops.push_back(pr.second);
                g.set_ops_off.push_back(g.set_ops.size());
            }
        }
    } else {
        const uint32_t P = (n + psize - 1) / psize;
        g.num_partitions = P;
        // pass 1: split every set by partition; distinct restrictions numbered by first appearance
        std::vector<std::unordered_map<std::vector<uint32_t>, uint32_t, detail::VecHash>> seen(P);
        std::vector<std::vector<std::vector<uint32_t>>> partial(P);
        std::vector<uint64_t> lists_off(1, 0);
        std::vector<std::pair<uint32_t, uint32_t>> lists;  // (partition, local id)
        std::vector<uint32_t> rel;
        for (uint64_t id = 0; id < ns; ++id) {
            hybrid_decode(h, id, set);
            size_t i = 0;
            while (i < set.size()) {
                const uint32_t p = set[i] / psize;
                rel.clear();
                for (; i < set.size() && set[i] / psize == p; ++i) rel.push_back(set[i] - p * psize);
                auto it = seen[p].find(rel);
                uint32_t lid;
                if (it == seen[p].end()) {
                    lid = (uint32_t)partial[p].size();
                    seen[p].emplace(rel, lid);
                    partial[p].push_back(rel);
                } else {
                    lid = it->second;
                }
                lists.push_back({p, lid});
            }
            lists_off.push_back(lists.size());
        }
        seen.clear();
        // pass 2: encode the partial sets of every partition, remember their ops
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> pops(P);  // local id -> (op a, op b or ~0)
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t np = std::min(psize, n - p * psize), base = p * psize;
            g.num_partial_sets += partial[p].size();
            if (type == IDX_META) {
                for (auto& r : partial[p]) {
                    pops[p].push_back({(uint32_t)g.ops.size(), 0xFFFFFFFFu});
                    g.ops.push_back(detail::encode_partial_hybrid(bw, r, np, base, op_bytes));
                }
            } else {
                detail::encode_differential(bw, g.ops, partial[p], np, base, csize, pops[p], g.num_clusters, op_bytes);
            }
            std::vector<std::vector<uint32_t>>().swap(partial[p]);
        }
        // pass 3: per colour set, the ops of its partial sets in partition order
        for (uint64_t id = 0; id < ns; ++id) {
            for (uint64_t j = lists_off[id]; j < lists_off[id + 1]; ++j) {
                const auto& pr = pops[lists[j].first][lists[j].second];
                g.set_ops.push_back(pr.first);
                if (pr.second != 0xFFFFFFFFu) g.set_ops.push_back(pr.second);
            }
            g.set_ops_off.push_back(g.set_ops.size());
        }
    }
    g.nbits = bw.nbits;
    g.bits.swap(bw.words);
    g.bits.resize((g.nbits + 63) / 64 + 2, 0);
    // algorithmic bytes of a colour set (meta form): the bytes of every list it touches plus
    // 16 bytes of offsets per list
    g.set_bytes.assign(ns, 0);
    for (uint64_t id = 0; id < ns; ++id)
        for (uint64_t o = g.set_ops_off[id]; o < g.set_ops_off[id + 1]; ++o) g.set_bytes[id] += op_bytes[g.set_ops[o]] + 16;
    build_generic_device(g);
}

// host decode of one colour set through its ops (self check / export)
inline void generic_decode(const GenericSets& g, uint64_t id, std::vector<uint32_t>& out) {
    std::vector<uint64_t> T((g.num_colors + 63) / 64 + 1, 0);
    for (uint64_t o = g.set_ops_off[id]; o < g.set_ops_off[id + 1]; ++o) {
        const SetOp& op = g.ops[g.set_ops[o]];
        if (op.kind == OP_OR_BITMAP) {
            for (uint32_t c = 0; c < op.np; ++c) {
                const uint64_t p = op.body + c;
                if ((g.bits[p >> 6] >> (p & 63)) & 1) T[(op.base + c) >> 6] |= 1ULL << ((op.base + c) & 63);
            }
            continue;
        }
        if (op.kind == OP_OR_COMP)
            for (uint32_t c = 0; c < op.np; ++c) T[(op.base + c) >> 6] |= 1ULL << ((op.base + c) & 63);
        BitReader r(g.bits.data(), op.body);
        uint32_t prev = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < op.ncodes; ++i) {
            prev = prev + 1 + (uint32_t)r.delta();
            const uint32_t c = op.base + prev;
            if (op.kind == OP_OR_GAPS) T[c >> 6] |= 1ULL << (c & 63);
            else if (op.kind == OP_OR_COMP) T[c >> 6] &= ~(1ULL << (c & 63));
            else T[c >> 6] ^= 1ULL << (c & 63);
        }
    }
    out.clear();
    for (uint32_t w = 0; w < T.size(); ++w)
        for (uint64_t x = T[w]; x; x &= x - 1) out.push_back(w * 64 + (uint32_t)__builtin_ctzll(x));
}

// host decode of one colour set through the DEVICE form of its ops (self check of build_generic_device):
// what k_generic computes, word for word
inline void generic_decode_device(const GenericSets& g, uint64_t id, std::vector<uint32_t>& out) {
    std::vector<uint32_t> T((g.num_colors + 31) / 32 + 64, 0);
    for (uint64_t o = g.set_ops_off[id]; o < g.set_ops_off[id + 1]; ++o) {
        const uint32_t ref = g.dev_set_ops[o];
        if (!(ref & GOP_BLOCK_REF)) {
            const uint32_t* rec = g.dev_span.data() + (uint64_t)ref * 8;
            for (uint32_t j = 0; j < GOP_SPAN_WORDS; ++j) T[(rec[0] & 0xFFFFFFu) + j] ^= rec[1 + j];
            continue;
        }
        const GenOpDev& d = g.dev_ops[ref & ~GOP_BLOCK_REF];
        for (uint32_t b = 0; b < d.ncodes; ++b) {
            const uint64_t h = g.dev_blk_hdr[d.soff + b];
            const uint32_t* w = g.dev_blk_words.data() + d.begin + blk_rel_word(h);
            const uint32_t width = blk_width(h), cnt = blk_count(h), start = blk_start(h);
            if (width == BLK_CHUNK_WIDTH) {
                for (uint32_t j = 0; j < cnt; ++j) T[(start >> 5) + j] ^= w[j];
            } else {
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint64_t bit = (uint64_t)i * width;
                    const uint64_t two = (uint64_t)w[bit >> 5] | ((uint64_t)w[(bit >> 5) + 1] << 32);
                    const uint32_t v = start + (uint32_t)((two >> (bit & 31)) & ((1ULL << width) - 1ULL));
                    T[v >> 5] ^= 1u << (v & 31);
                }
            }
        }
    }
    out.clear();
    for (uint32_t w = 0; w < T.size(); ++w)
        for (uint32_t x = T[w]; x; x &= x - 1) out.push_back(w * 32 + (uint32_t)__builtin_ctz(x));
}

}  // namespace fg
