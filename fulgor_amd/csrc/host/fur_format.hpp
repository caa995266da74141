// Binary index files in the reference's own layout: .fur / .mfur / .dfur / .mdfur.
//
// What is here: the Fulgor-OWNED sections, in the order `index<ColorSets>::visit_impl` writes them
// (include/index.hpp:93-102): version number, [k2u], u2c bit vector + rank9, the colour-set container
// (hybrid.hpp:338-345, meta.hpp:274-281, differential.hpp:326-334, meta_differential.hpp:308-320), filenames
// (filenames.hpp:37-41), with writers, parsers and the conversions to and from plain colour sets.
//
// What is NOT here, and why this is not ".fur support": the k2u section is an sshash::dictionary, whose layout lives
// in a submodule that is not vendored, and the layouts of the `bits` / `essentials` primitives below (bit_vector,
// rank9, darray1, compact_vector, elias_fano<false,false>, the vector / POD visitors) are written FROM MEMORY of
// upstream — every struct marked [UNVERIFIED]. None of it has seen a file written by the reference. The k2u section is
// therefore pluggable (K2uCodec): the only codec shipped writes and reads the engine's own unitig block behind a tag;
// a real SSHash section is refused with a message that says so. The round trip write -> read of our own files is tested
// (tests/test_host_cpu.py); validation against a real index remains to be done before any claim of compatibility.
#pragma once
#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include "codecs_build.hpp"
#include "hybrid_codec.hpp"

namespace fg {

inline bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

namespace fur {

// ---- essentials-style visitors: POD = raw bytes, vector = u64 count then the elements  [UNVERIFIED] ----------------
struct Out {
    std::ostream& s;
    template <typename T> void pod(const T& v) { s.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
    template <typename T> void vec(const std::vector<T>& v) {
        pod<uint64_t>(v.size());
        if (!v.empty()) s.write(reinterpret_cast<const char*>(v.data()), v.size() * sizeof(T));
    }
};
struct In {
    std::istream& s;
    template <typename T> void pod(T& v) {
        s.read(reinterpret_cast<char*>(&v), sizeof(T));
        if (!s) throw std::runtime_error("truncated index file");
    }
    template <typename T> void vec(std::vector<T>& v) {
        uint64_t n;
        pod(n);
        if (n > (1ULL << 40)) throw std::runtime_error("corrupt index file (vector length)");
        v.resize(n);
        if (n) s.read(reinterpret_cast<char*>(v.data()), n * sizeof(T));
        if (!s) throw std::runtime_error("truncated index file");
    }
};

// ---- bits primitives  [UNVERIFIED layouts] ---------------------------------------------------------------------------
struct BV {  // bits::bit_vector: u64 num_bits, vector<u64> data (bit i = bit i % 64 of word i / 64)
    uint64_t num_bits = 0;
    std::vector<uint64_t> data;
    void write(Out& o) const { o.pod(num_bits); o.vec(data); }
    void read(In& i) { i.pod(num_bits); i.vec(data); if (data.size() * 64 < num_bits) throw std::runtime_error("corrupt bit vector"); }
    bool get(uint64_t p) const { return (data[p >> 6] >> (p & 63)) & 1; }
    void resize(uint64_t n) { num_bits = n; data.assign((n + 63) / 64, 0); }
    void set(uint64_t p) { data[p >> 6] |= 1ULL << (p & 63); }
    static BV from_writer(const BitWriter& w) { BV b; b.num_bits = w.nbits; b.data = w.words; b.data.resize((w.nbits + 63) / 64); return b; }
};

struct Rank9 {  // bits::rank9: per 512-bit block {ones before the block, 7 x 9-bit counts of its first 1..7 words}, + a final pair
    std::vector<uint64_t> pairs;
    void build(const BV& b) {
        const uint64_t nw = b.data.size(), nblocks = (nw + 7) / 8;
        pairs.assign(2 * (nblocks + 1), 0);
        uint64_t total = 0;
        for (uint64_t blk = 0; blk < nblocks; ++blk) {
            pairs[2 * blk] = total;
            uint64_t sub = 0, packed = 0;
            for (uint64_t w = 0; w < 8; ++w) {
                if (w) packed |= sub << (9 * (w - 1));
                if (blk * 8 + w < nw) sub += (uint64_t)__builtin_popcountll(b.data[blk * 8 + w]);
            }
            pairs[2 * blk + 1] = packed;
            total += sub;
        }
        pairs[2 * nblocks] = total;
    }
    void write(Out& o) const { o.vec(pairs); }
    void read(In& i) { i.vec(pairs); }
};

struct DArray1 {  // bits::darray1 (select on ones): block = 1024 ones, subblock = 32, overflow beyond 2^16 positions
    uint64_t positions = 0;
    std::vector<int64_t> block_inventory;
    std::vector<uint16_t> subblock_inventory;
    std::vector<uint64_t> overflow_positions;
    void build(const BV& b) {
        constexpr uint64_t BLOCK = 1024, SUB = 32, MAXD = 1 << 16;
        std::vector<uint64_t> cur;
        positions = 0;
        block_inventory.clear(); subblock_inventory.clear(); overflow_positions.clear();
        auto flush = [&] {
            if (cur.empty()) return;
            if (cur.back() - cur.front() < MAXD) {
                block_inventory.push_back((int64_t)cur.front());
                for (size_t i = 0; i < cur.size(); i += SUB) subblock_inventory.push_back((uint16_t)(cur[i] - cur.front()));
            } else {
                block_inventory.push_back(-(int64_t)overflow_positions.size() - 1);
                for (uint64_t p : cur) overflow_positions.push_back(p);
                for (size_t i = 0; i < cur.size(); i += SUB) subblock_inventory.push_back((uint16_t)-1);
            }
            cur.clear();
        };
        for (uint64_t p = 0; p < b.num_bits; ++p)
            if (b.get(p)) {
                cur.push_back(p);
                ++positions;
                if (cur.size() == BLOCK) flush();
            }
        flush();
    }
    void write(Out& o) const { o.pod(positions); o.vec(block_inventory); o.vec(subblock_inventory); o.vec(overflow_positions); }
    void read(In& i) { i.pod(positions); i.vec(block_inventory); i.vec(subblock_inventory); i.vec(overflow_positions); }
};

struct CV {  // bits::compact_vector: u64 size, u64 width, u64 mask, vector<u64> data (one spare word)
    uint64_t size = 0, width = 0, mask = 0;
    std::vector<uint64_t> data;
    void init(uint64_t n, uint64_t w) {
        size = n; width = w; mask = w >= 64 ? ~0ULL : (1ULL << w) - 1;
        data.assign((n * w + 63) / 64 + 1, 0);
    }
    void set(uint64_t i, uint64_t v) {
        if (!width) return;
        const uint64_t p = i * width, sh = p & 63;
        data[p >> 6] |= v << sh;
        if (sh + width > 64) data[(p >> 6) + 1] |= v >> (64 - sh);
    }
    uint64_t get(uint64_t i) const {
        if (!width) return 0;
        const uint64_t p = i * width, sh = p & 63;
        uint64_t v = data[p >> 6] >> sh;
        if (sh + width > 64) v |= data[(p >> 6) + 1] << (64 - sh);
        return v & mask;
    }
    void write(Out& o) const { o.pod(size); o.pod(width); o.pod(mask); o.vec(data); }
    void read(In& i) {
        i.pod(size); i.pod(width); i.pod(mask); i.vec(data);
        if (width > 64 || data.size() * 64 < size * width) throw std::runtime_error("corrupt compact vector");
    }
};

struct EF {  // bits::elias_fano<false, false>: u64 back, bit_vector high_bits, darray1 high_bits_d1, compact_vector low_bits
    uint64_t back = 0;
    BV high;
    DArray1 d1;
    CV low;
    void encode(const std::vector<uint64_t>& v) {  // non-decreasing; universe = v.back()
        const uint64_t n = v.size(), u = n ? v.back() : 0;
        const uint64_t l = (n && u / n) ? msb64(u / n) : 0;
        low.init(n, l);
        high.resize(n + (u >> l) + 1);
        for (uint64_t i = 0; i < n; ++i) {
            low.set(i, l ? v[i] & ((1ULL << l) - 1) : 0);
            high.set((v[i] >> l) + i);
        }
        d1.build(high);
        back = u;
    }
    std::vector<uint64_t> decode() const {
        std::vector<uint64_t> v;
        v.reserve(low.size);
        uint64_t i = 0;
        for (uint64_t p = 0; p < high.num_bits && i < low.size; ++p)
            if (high.get(p)) { v.push_back(((p - i) << low.width) | low.get(i)); ++i; }
        if (i != low.size) throw std::runtime_error("corrupt Elias-Fano sequence");
        return v;
    }
    void write(Out& o) const { o.pod(back); high.write(o); d1.write(o); low.write(o); }
    void read(In& i) { i.pod(back); high.read(i); d1.read(i); low.read(i); }
};

typedef std::vector<std::vector<uint32_t>> Sets;  // plain colour sets, ascending colours

// ---- hybrid (hybrid.hpp:338-352) ----------------------------------------------------------------------------------------
struct HybridSec {
    uint32_t num_colors = 0, sparse_thr = 0, dense_thr = 0;
    EF offsets;      // num_sets + 1 bit offsets
    BV color_sets;
    void write(Out& o) const { o.pod(num_colors); o.pod(sparse_thr); o.pod(dense_thr); offsets.write(o); color_sets.write(o); }
    void read(In& i) { i.pod(num_colors); i.pod(sparse_thr); i.pod(dense_thr); offsets.read(i); color_sets.read(i); }
    static HybridSec from(const HybridSets& h) {
        HybridSec s;
        s.num_colors = h.num_colors; s.sparse_thr = h.sparse_thr; s.dense_thr = h.dense_thr;
        s.offsets.encode(h.offsets);
        s.color_sets.num_bits = h.nbits;
        s.color_sets.data.assign(h.bits.begin(), h.bits.begin() + (h.nbits + 63) / 64);
        return s;
    }
    static HybridSec from_sets(const Sets& sets, uint32_t n) {
        HybridEncoder e;
        e.init(n);
        for (auto& v : sets) e.encode(v.data(), v.size());
        HybridSets h;
        e.finish(h);
        return from(h);
    }
    void to(HybridSets& h) const {  // thresholds are READ, not recomputed
        h = HybridSets();
        h.num_colors = num_colors; h.sparse_thr = sparse_thr; h.dense_thr = dense_thr;
        h.offsets = offsets.decode();
        h.nbits = color_sets.num_bits;
        h.bits = color_sets.data;
        h.bits.resize((h.nbits + 63) / 64 + 4, 0);
        if (h.offsets.empty() || h.offsets.back() != h.nbits) throw std::runtime_error("corrupt hybrid colour sets (offsets)");
    }
    Sets to_sets() const {
        HybridSets h;
        to(h);
        Sets out(h.num_sets());
        for (uint64_t i = 0; i < out.size(); ++i) hybrid_decode(h, i, out[i]);
        return out;
    }
};

// ---- meta<hybrid> (meta.hpp:9-17, 20-90, 274-287) ---------------------------------------------------------------------------
struct MetaSec {
    struct Endpoint { uint32_t min_color, num_color_sets_before; };
    uint32_t num_colors = 0;
    CV meta_color_sets;      // per set: size, then `size` global partial-set ids
    EF meta_offsets;         // element offsets, num_sets + 1
    std::vector<HybridSec> partial;
    std::vector<Endpoint> endpoints;  // num_partitions + 1
    void write(Out& o) const {
        o.pod(num_colors); meta_color_sets.write(o); meta_offsets.write(o);
        o.pod<uint64_t>(partial.size());
        for (auto& p : partial) p.write(o);
        o.vec(endpoints);
    }
    void read(In& i) {
        i.pod(num_colors); meta_color_sets.read(i); meta_offsets.read(i);
        uint64_t np; i.pod(np);
        if (np > (1u << 24)) throw std::runtime_error("corrupt meta colour sets");
        partial.resize(np);
        for (auto& p : partial) p.read(i);
        i.vec(endpoints);
        if (endpoints.size() != np + 1) throw std::runtime_error("corrupt meta colour sets (endpoints)");
    }
    // partitions = colour ranges of psize colours (the reference clusters colours by sketches: construction, out of scope)
    static MetaSec from_sets(const Sets& sets, uint32_t n, uint32_t psize) {
        MetaSec m;
        m.num_colors = n;
        const uint32_t P = (n + psize - 1) / psize;
        std::vector<std::map<std::vector<uint32_t>, uint32_t>> ids(P);
        std::vector<Sets> distinct(P);
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> lists(sets.size());  // (partition, local id)
        std::vector<uint32_t> rel;
        for (size_t s = 0; s < sets.size(); ++s) {
            size_t i = 0;
            while (i < sets[s].size()) {
                const uint32_t p = sets[s][i] / psize;
                rel.clear();
                for (; i < sets[s].size() && sets[s][i] / psize == p; ++i) rel.push_back(sets[s][i] - p * psize);
                auto it = ids[p].find(rel);
                if (it == ids[p].end()) { it = ids[p].emplace(rel, (uint32_t)distinct[p].size()).first; distinct[p].push_back(rel); }
                lists[s].push_back({p, it->second});
            }
        }
        uint32_t before = 0;
        for (uint32_t p = 0; p < P; ++p) {
            m.endpoints.push_back({p * psize, before});
            m.partial.push_back(HybridSec::from_sets(distinct[p], std::min(psize, n - p * psize)));
            before += (uint32_t)distinct[p].size();
        }
        m.endpoints.push_back({n, before});
        uint64_t ints = 0;
        for (auto& l : lists) ints += l.size() + 1;
        m.meta_color_sets.init(ints, (uint64_t)std::ceil(std::log2((double)std::max<uint32_t>(before, 2))));  // meta.hpp:27-28
        std::vector<uint64_t> offs(1, 0);
        uint64_t at = 0;
        for (auto& l : lists) {
            m.meta_color_sets.set(at++, l.size());
            for (auto& pr : l) m.meta_color_sets.set(at++, m.endpoints[pr.first].num_color_sets_before + pr.second);
            offs.push_back(at);
        }
        m.meta_offsets.encode(offs);
        return m;
    }
    Sets to_sets() const {
        const std::vector<uint64_t> offs = meta_offsets.decode();
        std::vector<Sets> part(partial.size());
        for (size_t p = 0; p < partial.size(); ++p) part[p] = partial[p].to_sets();
        Sets out(offs.empty() ? 0 : offs.size() - 1);
        for (size_t s = 0; s < out.size(); ++s) {
            const uint64_t size = meta_color_sets.get(offs[s]);
            for (uint64_t j = 0; j < size; ++j) {
                const uint64_t gid = meta_color_sets.get(offs[s] + 1 + j);
                size_t p = 0;  // partition by scanning num_color_sets_before (meta.hpp:227-235)
                while (p + 1 < endpoints.size() && endpoints[p + 1].num_color_sets_before <= gid) ++p;
                if (p >= part.size() || gid - endpoints[p].num_color_sets_before >= part[p].size()) throw std::runtime_error("corrupt meta colour set");
                for (uint32_t c : part[p][gid - endpoints[p].num_color_sets_before]) out[s].push_back(c + endpoints[p].min_color);
            }
        }
        return out;
    }
};

// ---- differential (differential.hpp:21-157, 326-340) ---------------------------------------------------------------------------
struct DiffSec {
    uint32_t num_colors = 0;
    EF representative_offsets;  // one bit offset per cluster
    EF color_set_offsets;       // one per set (no sentinel)
    BV color_sets, clusters;    // clusters: 1 at the last set of each cluster
    Rank9 clusters_rank;
    void write(Out& o) const {
        o.pod(num_colors); representative_offsets.write(o); color_set_offsets.write(o); color_sets.write(o); clusters.write(o);
        clusters_rank.write(o);
    }
    void read(In& i) {
        i.pod(num_colors); representative_offsets.read(i); color_set_offsets.read(i); color_sets.read(i); clusters.read(i);
        clusters_rank.read(i);
    }
    uint64_t num_sets() const { return clusters.num_bits; }
    // clusters = runs of csize consecutive sets, representative = colours of more than half of the run's sets
    static DiffSec from_sets(const Sets& sets, uint32_t n, uint32_t csize) {
        DiffSec d;
        d.num_colors = n;
        BitWriter bw;
        std::vector<uint64_t> rep_off, set_off;
        d.clusters.resize(sets.size());
        std::vector<uint32_t> cnt(n), rep, diff;
        for (size_t a = 0; a < sets.size(); a += csize) {
            const size_t b = std::min(sets.size(), a + csize);
            std::fill(cnt.begin(), cnt.end(), 0);
            for (size_t i = a; i < b; ++i)
                for (uint32_t c : sets[i]) ++cnt[c];
            rep.clear();
            for (uint32_t c = 0; c < n; ++c)
                if (2 * (uint64_t)cnt[c] > b - a) rep.push_back(c);
            rep_off.push_back(bw.nbits);
            bw.delta(rep.size());
            detail::write_gaps(bw, rep);
            for (size_t i = a; i < b; ++i) {
                diff.clear();
                std::set_symmetric_difference(sets[i].begin(), sets[i].end(), rep.begin(), rep.end(), std::back_inserter(diff));
                set_off.push_back(bw.nbits);
                bw.delta(diff.size());
                bw.delta(sets[i].size());
                detail::write_gaps(bw, diff);
            }
            d.clusters.set(b - 1);
        }
        d.color_sets = BV::from_writer(bw);
        d.clusters_rank.build(d.clusters);
        d.representative_offsets.encode(rep_off);
        d.color_set_offsets.encode(set_off);
        return d;
    }
    Sets to_sets() const {
        const std::vector<uint64_t> ro = representative_offsets.decode(), so = color_set_offsets.decode();
        if (so.size() != clusters.num_bits) throw std::runtime_error("corrupt differential colour sets");
        std::vector<uint64_t> words = color_sets.data;
        words.resize(words.size() + 2, 0);
        auto gaps = [&](BitReader& br, uint64_t size, std::vector<uint32_t>& v) {
            v.clear();
            uint64_t prev = 0;
            for (uint64_t i = 0; i < size; ++i) { prev = i ? prev + br.delta() + 1 : br.delta(); v.push_back((uint32_t)prev); }
        };
        Sets out(so.size());
        std::vector<uint32_t> rep, diff;
        uint64_t cluster = 0;
        for (size_t s = 0; s < so.size(); ++s) {
            if (s == 0 || clusters.get(s - 1)) {
                if (cluster >= ro.size()) throw std::runtime_error("corrupt differential colour sets (clusters)");
                BitReader br(words.data(), ro[cluster++]);
                gaps(br, br.delta(), rep);
            }
            BitReader br(words.data(), so[s]);
            const uint64_t dsize = br.delta();
            br.delta();  // size of the set itself
            gaps(br, dsize, diff);
            std::set_symmetric_difference(rep.begin(), rep.end(), diff.begin(), diff.end(), std::back_inserter(out[s]));
        }
        return out;
    }
};

// ---- meta-differential (meta_differential.hpp:8-16, 19-110, 308-330) -------------------------------------------------------------
struct MetaDiffSec {
    struct Endpoint { uint64_t min_color, num_color_sets; };
    uint32_t num_colors = 0, num_partition_sets = 0;
    EF partition_sets_offsets, relative_colors_offsets;
    std::vector<Endpoint> endpoints;  // one per partition
    std::vector<DiffSec> partial;
    BV relative_colors, partition_sets, partition_sets_partitions;  // ..._partitions: 1 at the last set of each group sharing a partition set
    Rank9 psp_rank;
    void write(Out& o) const {
        o.pod(num_colors); o.pod(num_partition_sets); partition_sets_offsets.write(o); relative_colors_offsets.write(o);
        o.vec(endpoints);
        o.pod<uint64_t>(partial.size());
        for (auto& p : partial) p.write(o);
        relative_colors.write(o); partition_sets.write(o); partition_sets_partitions.write(o); psp_rank.write(o);
    }
    void read(In& i) {
        i.pod(num_colors); i.pod(num_partition_sets); partition_sets_offsets.read(i); relative_colors_offsets.read(i);
        i.vec(endpoints);
        uint64_t np; i.pod(np);
        if (np != endpoints.size()) throw std::runtime_error("corrupt meta-differential colour sets");
        partial.resize(np);
        for (auto& p : partial) p.read(i);
        relative_colors.read(i); partition_sets.read(i); partition_sets_partitions.read(i); psp_rank.read(i);
    }
    static MetaDiffSec from_sets(const Sets& sets, uint32_t n, uint32_t psize, uint32_t csize) {
        MetaDiffSec m;
        m.num_colors = n;
        const uint32_t P = (n + psize - 1) / psize;
        std::vector<std::map<std::vector<uint32_t>, uint32_t>> ids(P);
        std::vector<Sets> distinct(P);
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> lists(sets.size());
        std::vector<uint32_t> rel;
        for (size_t s = 0; s < sets.size(); ++s) {
            size_t i = 0;
            while (i < sets[s].size()) {
                const uint32_t p = sets[s][i] / psize;
                rel.clear();
                for (; i < sets[s].size() && sets[s][i] / psize == p; ++i) rel.push_back(sets[s][i] - p * psize);
                auto it = ids[p].find(rel);
                if (it == ids[p].end()) { it = ids[p].emplace(rel, (uint32_t)distinct[p].size()).first; distinct[p].push_back(rel); }
                lists[s].push_back({p, it->second});
            }
        }
        for (uint32_t p = 0; p < P; ++p) {
            m.partial.push_back(DiffSec::from_sets(distinct[p], std::min(psize, n - p * psize), csize));
            m.endpoints.push_back({(uint64_t)p * psize, (uint64_t)distinct[p].size()});
        }
        BitWriter ps, rc;
        std::vector<uint64_t> pso(1, 0), rco(1, 0);
        m.partition_sets_partitions.resize(sets.size());
        for (size_t s = 0; s < sets.size(); ++s) {
            bool fresh = s == 0 || lists[s].size() != lists[s - 1].size();
            for (size_t j = 0; !fresh && j < lists[s].size(); ++j) fresh = lists[s][j].first != lists[s - 1][j].first;
            if (fresh) {  // a new partition set: delta(size), delta(first), then PLAIN gaps (meta_differential.hpp:46-49)
                if (s) m.partition_sets_partitions.set(s - 1);
                ps.delta(lists[s].size());
                for (size_t j = 0; j < lists[s].size(); ++j) ps.delta(j ? lists[s][j].first - lists[s][j - 1].first : lists[s][j].first);
                pso.push_back(ps.nbits);
                ++m.num_partition_sets;
            }
            for (auto& pr : lists[s]) rc.append(pr.second, msb64(std::max<uint64_t>(1, m.endpoints[pr.first].num_color_sets)) + 1);
            rco.push_back(rc.nbits);
        }
        // (no bit behind the last group: the reference sets one only where the NEXT partition set begins, meta_differential.hpp:40-41)
        m.psp_rank.build(m.partition_sets_partitions);
        m.partition_sets = BV::from_writer(ps);
        m.relative_colors = BV::from_writer(rc);
        m.partition_sets_offsets.encode(pso);
        m.relative_colors_offsets.encode(rco);
        return m;
    }
    Sets to_sets() const {
        const std::vector<uint64_t> pso = partition_sets_offsets.decode(), rco = relative_colors_offsets.decode();
        std::vector<Sets> part(partial.size());
        for (size_t p = 0; p < partial.size(); ++p) part[p] = partial[p].to_sets();
        std::vector<uint64_t> psw = partition_sets.data, rcw = relative_colors.data;
        psw.resize(psw.size() + 2, 0);
        rcw.resize(rcw.size() + 2, 0);
        Sets out(rco.empty() ? 0 : rco.size() - 1);
        std::vector<uint32_t> pset;
        uint64_t group = 0;
        for (size_t s = 0; s < out.size(); ++s) {
            if (s == 0 || partition_sets_partitions.get(s - 1)) {
                if (group + 1 >= pso.size()) throw std::runtime_error("corrupt meta-differential colour sets (partition sets)");
                BitReader br(psw.data(), pso[group++]);
                const uint64_t size = br.delta();
                pset.clear();
                uint64_t prev = 0;
                for (uint64_t j = 0; j < size; ++j) { prev = j ? prev + br.delta() : br.delta(); pset.push_back((uint32_t)prev); }
            }
            BitReader br(rcw.data(), rco[s]);
            for (uint32_t p : pset) {
                if (p >= part.size()) throw std::runtime_error("corrupt meta-differential colour set");
                const uint64_t id = br.take(msb64(std::max<uint64_t>(1, endpoints[p].num_color_sets)) + 1);
                if (id >= part[p].size()) throw std::runtime_error("corrupt meta-differential colour set");
                for (uint32_t c : part[p][id]) out[s].push_back(c + (uint32_t)endpoints[p].min_color);
            }
        }
        return out;
    }
};

// ---- filenames (filenames.hpp:37-44) ----------------------------------------------------------------------------------------
struct FilenamesSec {
    std::vector<uint32_t> offsets;
    std::vector<char> chars;
    void write(Out& o) const { o.vec(offsets); o.vec(chars); }
    void read(In& i) { i.vec(offsets); i.vec(chars); }
    static FilenamesSec from(const std::vector<std::string>& names) {
        FilenamesSec f;
        f.offsets.push_back(0);
        for (auto& n : names) { f.chars.insert(f.chars.end(), n.begin(), n.end()); f.offsets.push_back((uint32_t)f.chars.size()); }
        return f;
    }
    std::vector<std::string> to() const {
        std::vector<std::string> out;
        for (size_t i = 0; i + 1 < offsets.size(); ++i) {
            if (offsets[i + 1] < offsets[i] || offsets[i + 1] > chars.size()) throw std::runtime_error("corrupt filenames section");
            out.emplace_back(chars.begin() + offsets[i], chars.begin() + offsets[i + 1]);
        }
        return out;
    }
};

// ---- the k2u section: pluggable ----------------------------------------------------------------------------------------------
// The reference stores an sshash::dictionary here. Its layout is not available to this build; a codec for it can be
// registered once it is (and validated against a real file). Shipped: the engine's own unitig block behind a tag.
struct K2uCodec {
    std::function<void(Out&, const Dict&)> write;
    std::function<bool(In&, Dict&, std::string&)> read;  // false + message when the section is not this codec's
};
static const char K2U_TAG[8] = {'F', 'G', 'K', '2', 'U', '0', '0', '1'};
inline K2uCodec own_k2u_codec() {
    K2uCodec c;
    c.write = [](Out& o, const Dict& d) {
        o.s.write(K2U_TAG, 8);
        o.pod(d.k); o.pod(d.m); o.pod(d.num_kmers); o.pod(d.total_bases);
        o.vec(d.strings); o.vec(d.unitig_off);
    };
    c.read = [](In& i, Dict& d, std::string& why) {
        char tag[8];
        i.s.read(tag, 8);
        if (!i.s || memcmp(tag, K2U_TAG, 8) != 0) {
            why = "the k2u section is not the engine's own block: it is presumably an SSHash dictionary, whose on-disk layout is not "
                  "available to this build. Run `fulgor dump` with the reference and open the dump basename instead";
            return false;
        }
        i.pod(d.k); i.pod(d.m); i.pod(d.num_kmers); i.pod(d.total_bases);
        i.vec(d.strings); i.vec(d.unitig_off);
        return true;
    };
    return c;
}

inline int type_of_suffix(const std::string& path) {  // tools/pseudoalign.cpp:294-306: mdfur, mfur, dfur, fur in this order
    if (ends_with(path, "mdfur")) return IDX_META_DIFF;
    if (ends_with(path, "mfur")) return IDX_META;
    if (ends_with(path, "dfur")) return IDX_DIFF;
    if (ends_with(path, "fur")) return IDX_HYBRID;
    return -1;
}

inline Sets all_sets(const HybridSets& h) {
    Sets out(h.num_sets());
    for (uint64_t i = 0; i < out.size(); ++i) hybrid_decode(h, i, out[i]);
    return out;
}

// index<ColorSets>::visit_impl order (include/index.hpp:93-102)
inline void write_fur(const HostIndex& idx, const std::string& path, uint32_t psize, uint32_t csize, const K2uCodec& k2u = own_k2u_codec()) {
    const int type = type_of_suffix(path);
    if (type < 0) throw std::runtime_error("output name must end in .fur, .mfur, .dfur or .mdfur");
    std::ofstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open output index file");
    Out o{f};
    const uint8_t version[3] = {4, 2, 0};  // util.hpp:31-35 (essentials::version_number as three bytes: [UNVERIFIED])
    o.s.write((const char*)version, 3);
    k2u.write(o, idx.dict);
    {   // u2c: 1 at the last unitig of each colour set (builder.hpp:116,131,171), + rank9
        const auto& cs = idx.dict.unitig_csid;
        BV u2c;
        u2c.resize(cs.size());
        for (size_t u = 0; u < cs.size(); ++u)
            if (u + 1 == cs.size() || cs[u + 1] != cs[u]) u2c.set(u);
        Rank9 r;
        r.build(u2c);
        u2c.write(o);
        r.write(o);
    }
    const uint32_t n = idx.hybrid.num_colors;
    if (type == IDX_HYBRID) HybridSec::from(idx.hybrid).write(o);
    else if (type == IDX_META) MetaSec::from_sets(all_sets(idx.hybrid), n, psize).write(o);
    else if (type == IDX_DIFF) DiffSec::from_sets(all_sets(idx.hybrid), n, csize).write(o);
    else MetaDiffSec::from_sets(all_sets(idx.hybrid), n, psize, csize).write(o);
    FilenamesSec::from(idx.filenames).write(o);
    if (!f) throw std::runtime_error("write error on index file");
}

// -> the unitig block, u2c and the colour sets (as the hybrid stream every other structure of the engine is built from);
// psize / csize = the partition and cluster shape found in the file (0 where the codec has none)
inline void read_fur(const std::string& path, HostIndex& idx, uint32_t& psize, uint32_t& csize, const K2uCodec& k2u = own_k2u_codec()) {
    const int type = type_of_suffix(path);
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open index file");
    In in{f};
    uint8_t version[3];
    f.read((char*)version, 3);
    if (!f || version[0] != 4)  // util.hpp:91-95: only the MAJOR number must agree
        throw std::runtime_error("MAJOR index version mismatch: Fulgor index needs rebuilding");
    std::string why;
    if (!k2u.read(in, idx.dict, why)) throw std::runtime_error(why);
    BV u2c;
    Rank9 r;
    u2c.read(in);
    r.read(in);
    if (idx.dict.unitig_off.size() != u2c.num_bits + 1) throw std::runtime_error("corrupt index file (u2c)");
    {  // the unitig table as load_binary checks it: the dictionary builder trusts it
        const auto& uo = idx.dict.unitig_off;
        if (uo.empty() || uo[0] != 0 || uo.back() != idx.dict.total_bases) throw std::runtime_error("corrupt index file (unitig table)");
        for (size_t u = 0; u + 1 < uo.size(); ++u)
            if (uo[u + 1] < uo[u] + idx.dict.k) throw std::runtime_error("corrupt index file (unitig offsets)");
    }
    idx.dict.unitig_csid.resize(u2c.num_bits);
    uint32_t id = 0;
    for (uint64_t u = 0; u < u2c.num_bits; ++u) { idx.dict.unitig_csid[u] = id; id += u2c.get(u); }  // index.hpp:37
    psize = csize = 0;
    Sets sets;
    uint32_t n = 0;
    if (type == IDX_HYBRID) {
        HybridSec s;
        s.read(in);
        s.to(idx.hybrid);
        n = s.num_colors;
    } else {
        if (type == IDX_META) {
            MetaSec s; s.read(in); sets = s.to_sets(); n = s.num_colors;
            psize = s.endpoints.size() > 1 ? s.endpoints[1].min_color - s.endpoints[0].min_color : n;
        } else if (type == IDX_DIFF) {
            DiffSec s; s.read(in); sets = s.to_sets(); n = s.num_colors;
            for (uint64_t b = 0; b < s.clusters.num_bits && !csize; ++b) if (s.clusters.get(b)) csize = (uint32_t)b + 1;
        } else {
            MetaDiffSec s; s.read(in); sets = s.to_sets(); n = s.num_colors;
            psize = s.endpoints.size() > 1 ? (uint32_t)(s.endpoints[1].min_color - s.endpoints[0].min_color) : n;
            if (!s.partial.empty())
                for (uint64_t b = 0; b < s.partial[0].clusters.num_bits && !csize; ++b) if (s.partial[0].clusters.get(b)) csize = (uint32_t)b + 1;
        }
        HybridEncoder e;
        e.init(n);
        for (auto& v : sets) {
            if (v.empty() || v.back() >= n) throw std::runtime_error("corrupt colour set");
            e.encode(v.data(), v.size());
        }
        e.finish(idx.hybrid);
    }
    FilenamesSec fn;
    fn.read(in);
    idx.filenames = fn.to();
    if (idx.filenames.size() != n) throw std::runtime_error("corrupt index file (filenames)");
    if (id != idx.hybrid.num_sets()) throw std::runtime_error("corrupt index file (u2c / colour sets)");
    idx.type = type;
}

}  // namespace fur
}  // namespace fg
