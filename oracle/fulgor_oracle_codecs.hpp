// ORACLE — test infrastructure only (see fulgor_oracle.hpp; PARITY UNPINNED).
// CPU restatement of the reference's three other colour-set codecs and of the algorithms that run on them:
//   meta<hybrid>            include/color_sets/meta.hpp:93-244          -> MetaColors, MetaCursor
//   differential            include/color_sets/differential.hpp:170-300 -> DiffColors, DiffCursor
//   meta_differential       include/color_sets/meta_differential.hpp:112-276 -> MetaDiffColors, MetaDiffCursor
//   next_geq_intersect      src/ps_full_intersection.cpp:6-30
//   diff_intersect          src/ps_full_intersection.cpp:129-240
//   meta_intersect<It,diff> src/ps_full_intersection.cpp:242-332
//   merge_meta / merge_diff / merge_metadiff   src/ps_threshold_union.cpp:42-318
// plus encoders (meta.hpp:59-67, differential.hpp:21-98, meta_differential.hpp:35-73) fed by a simple
// deterministic partition/cluster assignment (construction heuristics of the reference — sketching,
// k-means — are out of scope; any assignment yields a valid index).
#pragma once
#include "fulgor_oracle.hpp"

namespace oracle {

// bits::rank9-style rank over a plain bit vector: number of ones in [0, pos)
struct RankedBits {
    std::vector<bool> b;
    std::vector<uint64_t> pre;  // ones before block i (blocks of 64)
    void push(bool x) { b.push_back(x); }
    void set_last() { if (!b.empty()) b.back() = true; }
    void build() {
        pre.assign(b.size() / 64 + 2, 0);
        uint64_t c = 0;
        for (size_t i = 0; i < b.size(); ++i) {
            if (i % 64 == 0) pre[i / 64] = c;
            c += b[i];
        }
        for (size_t i = (b.size() + 63) / 64; i < pre.size(); ++i) pre[i] = c;
    }
    uint64_t rank1(uint64_t pos) const {
        uint64_t c = pre[pos / 64];
        for (uint64_t i = pos / 64 * 64; i < pos; ++i) c += b[i];
        return c;
    }
};

// =================================================================================================
// meta<hybrid>
// =================================================================================================
struct MetaColors {
    uint32_t num_colors = 0;
    std::vector<uint64_t> lists;    // m_meta_color_sets (bits::compact_vector): per set [size, id0, id1, ...]
    std::vector<uint64_t> offsets;  // m_meta_color_sets_offsets: element offsets, num_sets + 1
    std::vector<HybridColors> partial;
    struct Endpoint { uint32_t min_color, num_color_sets_before; };
    std::vector<Endpoint> endpoints;  // num_partitions + 1

    uint64_t num_sets() const { return offsets.size() - 1; }
    uint32_t num_partitions() const { return (uint32_t)endpoints.size() - 1; }
};

struct MetaCursor {  // meta::forward_iterator, meta.hpp:93-236
    const MetaColors* m = nullptr;
    HybridCursor part_it;
    uint64_t begin = 0;
    uint32_t curr_meta_color = 0, curr_val = 0;
    uint32_t meta_size = 0, pos_in_meta = 0;
    uint32_t curr_part_size = 0, pos_in_part = 0;
    uint32_t part_id = 0, part_min = 0, part_max = 0;

    MetaCursor() {}
    MetaCursor(const MetaColors* mc, uint64_t b) : m(mc), begin(b), meta_size((uint32_t)mc->lists[b]) { rewind(); }

    void rewind() { init(); change_partition(); }
    void init() { pos_in_meta = 0; part_id = 0; part_min = 0; }
    uint32_t value() const { return curr_val; }
    bool has_next() const { return pos_in_part != curr_part_size; }
    void next_in_partition() { ++pos_in_part; part_it.next(); update_curr_val(); }
    void next() {  // :118-129
        if (pos_in_part == curr_part_size - 1) {
            if (pos_in_meta == meta_size - 1) { curr_val = num_colors(); return; }
            ++pos_in_meta;
            change_partition();
        } else {
            next_in_partition();
        }
    }
    void next_geq(uint32_t lower) { while (value() < lower) next(); }
    uint32_t partial_set_size() const { return part_it.size(); }
    uint32_t meta_color() const { return curr_meta_color; }
    void read_partition_id() {  // :161-164
        curr_meta_color = (uint32_t)m->lists[begin + 1 + pos_in_meta];
        while (part_id + 1 < m->endpoints.size() && curr_meta_color >= m->endpoints[part_id + 1].num_color_sets_before) ++part_id;
    }
    void next_partition_id() {  // :166-173
        ++pos_in_meta;
        if (pos_in_meta == meta_size) { part_id = num_partitions(); return; }
        read_partition_id();
    }
    void next_geq_partition_id(uint32_t lower) { while (partition_id() < lower) next_partition_id(); }
    void update_partition() {  // :181-196
        part_min = m->endpoints[part_id].min_color;
        part_max = m->endpoints[part_id + 1].min_color;
        const uint32_t before = m->endpoints[part_id].num_color_sets_before;
        const HybridColors& h = m->partial[part_id];
        part_it = HybridCursor(&h, h.offsets[curr_meta_color - before]);
        curr_part_size = part_it.size();
        pos_in_part = 0;
        update_curr_val();
    }
    void change_partition() { read_partition_id(); update_partition(); }
    uint32_t partition_id() const { return part_id; }
    uint32_t meta_color_set_size() const { return meta_size; }
    uint32_t num_colors() const { return m->num_colors; }
    uint32_t num_partitions() const { return m->num_partitions(); }
    uint32_t partition_min_color() const { return part_min; }
    uint32_t partition_max_color() const { return part_max; }

private:
    void update_curr_val() { curr_val = part_it.value() + part_min; }
};

// =================================================================================================
// differential
// =================================================================================================
struct DiffColors {
    uint32_t num_colors = 0;
    std::vector<uint64_t> rep_offsets;  // m_representative_offsets (one per cluster)
    std::vector<uint64_t> set_offsets;  // m_color_set_offsets (one per set; NOT +1, differential.hpp:302)
    BitStream bits;
    RankedBits clusters;  // 1 at the last set of each cluster

    uint64_t num_sets() const { return set_offsets.size(); }

    // builder::process_partition, differential.hpp:21-43
    void begin_cluster(const std::vector<uint32_t>& rep) {
        rep_offsets.push_back(bits.n);
        curr_rep = rep;
        clusters.set_last();
        write_delta(bits, rep.size());
        if (!rep.empty()) {
            write_delta(bits, rep[0]);
            for (size_t i = 1; i < rep.size(); ++i) write_delta(bits, rep[i] - rep[i - 1] - 1);
        }
    }
    // builder::process_color_set, differential.hpp:45-98: symmetric difference against the representative
    void add_set(const uint32_t* set, uint64_t size) {
        set_offsets.push_back(bits.n);
        clusters.push(false);
        std::vector<uint32_t> d;
        size_t i = 0, j = 0;
        while (i < size && j < curr_rep.size()) {
            if (set[i] == curr_rep[j]) { ++i; ++j; }
            else if (set[i] < curr_rep[j]) d.push_back(set[i++]);
            else d.push_back(curr_rep[j++]);
        }
        for (; i < size; ++i) d.push_back(set[i]);
        for (; j < curr_rep.size(); ++j) d.push_back(curr_rep[j]);
        write_delta(bits, d.size());
        write_delta(bits, size);
        if (!d.empty()) {
            write_delta(bits, d[0]);
            for (size_t p = 1; p < d.size(); ++p) write_delta(bits, d[p] - d[p - 1] - 1);
        }
    }
    void finish() { clusters.set_last(); clusters.build(); bits.seal(); }  // builder::build, :124-134

private:
    std::vector<uint32_t> curr_rep;
};

struct DiffCursor {  // differential::forward_iterator, differential.hpp:170-290
    const DiffColors* d = nullptr;
    uint64_t diff_begin = 0, rep_begin = 0;
    uint64_t rep_size = 0, diff_size = 0, pos_diff = 0, pos_rep = 0;
    uint32_t cur_rep = 0, cur_diff = 0, prev_rep = 0, prev_diff = 0, cur_val = 0, set_size = 0;
    BitCursor rep_it, diff_it;

    DiffCursor() {}
    DiffCursor(const DiffColors* dc, uint64_t set_begin, uint64_t representative_begin)
        : d(dc), diff_begin(set_begin), rep_begin(representative_begin) { rewind(); }

    void rewind() { init(); update_curr_val(); }
    void full_rewind() { init(); }
    uint32_t size() const { return set_size; }
    uint32_t value() const { return cur_val; }
    uint32_t num_colors() const { return d->num_colors; }
    uint64_t representative_begin() const { return rep_begin; }
    uint32_t representative_val() const { return cur_rep; }
    uint32_t differential_val() const { return cur_diff; }

    void next() {  // :192-206
        if (pos_rep >= rep_size && pos_diff >= diff_size) { cur_val = num_colors(); return; }
        if (pos_rep >= rep_size || cur_diff < cur_rep) next_differential_val();
        else if (pos_diff >= diff_size || cur_rep < cur_diff) next_representative_val();
        update_curr_val();
    }
    void next_geq(uint32_t lower) { while (value() < lower) next(); }
    void next_representative_val() {  // :221-230
        ++pos_rep;
        prev_rep = cur_rep;
        cur_rep = pos_rep < rep_size ? prev_rep + (uint32_t)read_delta(rep_it) + 1 : num_colors();
    }
    void next_differential_val() {  // :234-243
        ++pos_diff;
        prev_diff = cur_diff;
        cur_diff = pos_diff < diff_size ? prev_diff + (uint32_t)read_delta(diff_it) + 1 : num_colors();
    }

private:
    void init() {  // :257-278
        diff_it = BitCursor(&d->bits, diff_begin);
        rep_it = BitCursor(&d->bits, rep_begin);
        diff_size = read_delta(diff_it);
        rep_size = read_delta(rep_it);
        set_size = (uint32_t)read_delta(diff_it);
        cur_diff = diff_size == 0 ? num_colors() : (uint32_t)read_delta(diff_it);
        prev_diff = 0;
        cur_rep = rep_size == 0 ? num_colors() : (uint32_t)read_delta(rep_it);
        prev_rep = 0;
        pos_diff = pos_rep = 0;
    }
    void update_curr_val() {  // :280-288
        while (cur_rep == cur_diff && pos_rep <= rep_size && pos_diff <= diff_size) {
            next_differential_val();
            next_representative_val();
        }
        cur_val = std::min(cur_diff, cur_rep);
    }
};

static inline DiffCursor diff_color_set(const DiffColors& d, uint64_t id) {  // differential.hpp:294-300
    return DiffCursor(&d, d.set_offsets[id], d.rep_offsets[d.clusters.rank1(id)]);
}

// =================================================================================================
// meta_differential
// =================================================================================================
struct MetaDiffColors {
    uint32_t num_colors = 0, num_partition_sets = 0;
    std::vector<uint64_t> pset_offsets;   // m_partition_sets_offsets (bit offsets into psets)
    std::vector<uint64_t> rel_offsets;    // m_relative_colors_offsets (num_sets + 1 bit offsets into rel)
    struct Endpoint { uint64_t min_color, num_color_sets; };
    std::vector<Endpoint> endpoints;      // one per partition
    std::vector<DiffColors> partial;
    BitStream rel, psets;
    RankedBits pset_groups;               // 1 at the last colour set that shares a partition set

    uint64_t num_sets() const { return rel_offsets.size() - 1; }
    uint32_t num_partitions() const { return (uint32_t)endpoints.size(); }
};

struct MetaDiffCursor {  // meta_differential::forward_iterator, meta_differential.hpp:112-265
    const MetaDiffColors* m = nullptr;
    DiffCursor part_it;
    BitCursor pset_it, rel_it;
    uint64_t meta_size = 0, begin_pset = 0, begin_rel = 0, pos_in_meta = 0, pos_in_part = 0;
    uint64_t cur_rel = 0, cur_part = 0, cur_part_size = 0, cur_val = 0, part_min = 0, sets_before = 0;

    MetaDiffCursor() {}
    MetaDiffCursor(const MetaDiffColors* mc, uint64_t bp, uint64_t br) : m(mc), begin_pset(bp), begin_rel(br) { rewind(); }

    void rewind() { init(); change_partition(); }
    void init() {  // :127-135
        sets_before = 0;
        pos_in_meta = pos_in_part = 0;
        cur_part = 0;
        pset_it = BitCursor(&m->psets, begin_pset);
        rel_it = BitCursor(&m->rel, begin_rel);
        meta_size = read_delta(pset_it);
    }
    uint64_t value() const { return cur_val; }
    bool has_next() const { return pos_in_part != cur_part_size; }
    void next() {  // :142-153
        if (pos_in_part == cur_part_size - 1) {
            if (pos_in_meta == meta_size - 1) { cur_val = num_colors(); return; }
            ++pos_in_meta;
            change_partition();
        } else {
            next_in_partition();
        }
    }
    void next_geq(uint64_t lower) { while (value() < lower) next(); }
    void next_in_partition() { ++pos_in_part; part_it.next(); update_curr_val(); }
    void change_partition() { read_partition_id(); update_partition(); }
    void next_partition_id() {  // :176-183
        ++pos_in_meta;
        if (pos_in_meta == meta_size) { cur_part = num_partitions(); return; }
        read_partition_id();
    }
    void read_partition_id() {  // :185-197 (plain gaps after the first id)
        const uint64_t delta = read_delta(pset_it);
        for (uint64_t i = 0; i < delta; ++i) sets_before += m->endpoints[cur_part + i].num_color_sets;
        cur_part += delta;
        const unsigned w = msb_u64(m->endpoints[cur_part].num_color_sets) + 1;
        cur_rel = rel_it.take(w);
    }
    void next_geq_partition_id(uint32_t lower) { while (partition_id() < lower) next_partition_id(); }
    void update_partition() {  // :205-214
        part_min = m->endpoints[cur_part].min_color;
        pos_in_part = 0;
        part_it = diff_color_set(m->partial[cur_part], cur_rel);
        cur_part_size = part_it.size();
        update_curr_val();
    }
    uint32_t partial_set_size() const { return part_it.size(); }
    uint32_t partition_id() const { return (uint32_t)cur_part; }
    uint32_t partition_min_color() const { return (uint32_t)part_min; }
    uint32_t partition_max_color() const { return (uint32_t)part_min + part_it.num_colors(); }
    uint32_t meta_color() const { return (uint32_t)(sets_before + cur_rel); }
    uint32_t num_colors() const { return m->num_colors; }
    uint32_t num_partitions() const { return m->num_partitions(); }
    uint64_t meta_color_set_size() const { return meta_size; }
    DiffCursor partition_it() const { return part_it; }

private:
    void update_curr_val() { cur_val = part_min + part_it.value(); }
};

// =================================================================================================
// algorithms
// =================================================================================================
// ps_full_intersection.cpp:6-30
template <typename It>
static inline void next_geq_intersect(It* begin, It* end, std::vector<uint32_t>& colors, uint32_t num_colors) {
    uint32_t cand = begin->value();
    const size_t size = end - begin;
    size_t i = 1;
    while (cand < num_colors) {
        for (; i != size; ++i) {
            begin[i].next_geq(cand);
            uint32_t v = (uint32_t)begin[i].value();
            if (v != cand) { cand = v; i = 0; break; }
        }
        if (i == size) {
            colors.push_back(cand);
            begin->next();
            cand = (uint32_t)begin->value();
            i = 1;
        }
    }
}

// ps_full_intersection.cpp:129-240
static inline void diff_intersect(std::vector<DiffCursor>& its, std::vector<uint32_t>& colors, uint32_t lower_bound = 0) {
    if (its.empty()) return;
    const uint32_t n = its[0].num_colors();
    std::sort(its.begin(), its.end(), [](const DiffCursor& a, const DiffCursor& b) { return a.representative_begin() < b.representative_begin(); });
    const uint32_t num_its = (uint32_t)its.size();
    uint32_t num_groups = 1;
    for (uint32_t i = 1; i < num_its; ++i) num_groups += its[i].representative_begin() != its[i - 1].representative_begin();
    std::vector<std::vector<uint32_t>> groups(num_groups);
    {
        std::vector<uint32_t> counts(n, 0);
        uint32_t gid = 0, gsize = 0;
        for (uint32_t k = 0; k < num_its; ++k) {
            DiffCursor it = its[k];
            ++gsize;
            const bool last = k + 1 == num_its || its[k + 1].representative_begin() != it.representative_begin();
            if (gsize == 1 && last) {  // singleton: decode the set (:177-185)
                for (uint32_t i = 0; i < it.size(); ++i, it.next()) groups[gid].push_back(it.value());
                ++gid;
                gsize = 0;
                continue;
            }
            it.full_rewind();
            for (uint32_t v = it.differential_val(); v != n; it.next_differential_val(), v = it.differential_val()) ++counts[v];
            if (last) {  // a colour survives iff all members agree with/against the representative (:190-203)
                it.full_rewind();
                uint32_t v = it.representative_val();
                for (uint32_t c = 0; c < n; ++c) {
                    if (v < c) { it.next_representative_val(); v = it.representative_val(); }
                    if ((counts[c] == gsize && v != c) || (counts[c] == 0 && v == c)) groups[gid].push_back(c);
                }
                ++gid;
                gsize = 0;
                std::fill(counts.begin(), counts.end(), 0);
            }
        }
    }
    std::sort(groups.begin(), groups.end(), [](const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) { return a.size() < b.size(); });
    std::vector<size_t> at(num_groups, 0);
    for (auto& g : groups)
        if (g.empty()) return;
    uint32_t cand = groups[0][0];
    size_t i = 1;
    while (cand < n) {
        for (; i != num_groups; ++i) {
            while (at[i] != groups[i].size() && groups[i][at[i]] < cand) ++at[i];
            if (at[i] == groups[i].size()) { cand = n; break; }
            uint32_t v = groups[i][at[i]];
            if (v != cand) { cand = v; i = 0; break; }
        }
        if (i == num_groups) {
            colors.push_back(cand + lower_bound);
            ++at[0];
            if (at[0] == groups[0].size()) break;
            cand = groups[0][at[0]];
            i = 1;
        }
    }
}

// ps_full_intersection.cpp:242-332
template <typename It, bool is_differential>
static inline void meta_intersect(std::vector<It>& its, std::vector<uint32_t>& colors, std::vector<uint32_t>& partition_ids) {
    if (its.empty()) return;
    std::sort(its.begin(), its.end(), [](const It& a, const It& b) { return a.meta_color_set_size() < b.meta_color_set_size(); });
    const uint32_t num_partitions = its[0].num_partitions();
    {  // step 1: partitions in common (:250-276)
        uint32_t cand = its[0].partition_id();
        size_t i = 1;
        while (cand < num_partitions) {
            for (; i != its.size(); ++i) {
                its[i].next_geq_partition_id(cand);
                uint32_t v = its[i].partition_id();
                if (v != cand) { cand = v; i = 0; break; }
            }
            if (i == its.size()) {
                partition_ids.push_back(cand);
                its[0].next_partition_id();
                cand = its[0].partition_id();
                i = 1;
            }
        }
    }
    for (auto& it : its) { it.init(); it.change_partition(); }  // step 2 (:279-331)
    for (uint32_t pid : partition_ids) {
        bool same = true;
        its.front().next_geq_partition_id(pid);
        its.front().update_partition();
        const uint32_t mc = its.front().meta_color();
        for (size_t i = 1; i != its.size(); ++i) {
            its[i].next_geq_partition_id(pid);
            its[i].update_partition();
            if (its[i].meta_color() != mc) same = false;
        }
        if (same) {
            It& f = its.front();
            while (f.has_next()) { colors.push_back((uint32_t)f.value()); f.next_in_partition(); }
        } else {
            std::sort(its.begin(), its.end(), [](const It& a, const It& b) {
                return a.partial_set_size() < b.partial_set_size() ||
                       (a.partial_set_size() == b.partial_set_size() && a.meta_color() < b.meta_color());
            });
            size_t back = 0;
            for (size_t cur = 1; cur < its.size(); ++cur)
                if (its[cur].meta_color() != its[back].meta_color()) std::swap(its[++back], its[cur]);
            if constexpr (is_differential) {
                std::vector<DiffCursor> ds;
                for (size_t i = 0; i <= back; ++i) ds.push_back(its[i].partition_it());
                const uint32_t lower = its[0].partition_max_color() - ds[0].num_colors();
                diff_intersect(ds, colors, lower);
            } else {
                next_geq_intersect(its.data(), its.data() + back + 1, colors, its[0].partition_max_color());
            }
        }
    }
}

template <typename It>
struct Scored { It item; uint32_t score; };

// shared step of merge_meta / merge_metadiff: partitions whose summed score reaches min_score
template <typename It>
static inline std::vector<uint32_t> scored_partitions(std::vector<Scored<It>>& its, uint64_t min_score) {
    std::vector<uint32_t> out;
    const uint32_t num_partitions = its[0].item.num_partitions();
    uint32_t cand = num_partitions;
    for (auto& s : its) cand = std::min(cand, s.item.partition_id());
    while (cand < num_partitions) {
        uint32_t nxt = num_partitions, score = 0;
        for (auto& s : its) {
            if (s.item.partition_id() == cand) { score += s.score; s.item.next_partition_id(); }
            nxt = std::min(nxt, s.item.partition_id());
        }
        if (score >= min_score) out.push_back(cand);
        cand = nxt;
    }
    return out;
}

// ps_threshold_union.cpp:42-120
static inline void merge_meta(std::vector<Scored<MetaCursor>>& its, std::vector<uint32_t>& colors, uint64_t min_score) {
    if (its.empty()) return;
    const uint32_t n = its[0].item.num_colors();
    std::vector<uint32_t> pids = scored_partitions(its, min_score);
    std::vector<uint32_t> scores(n, 0);
    for (auto& s : its) { s.item.init(); s.item.change_partition(); }
    for (uint32_t pid : pids) {
        uint32_t upper = 0;
        for (auto& s : its) {
            s.item.next_geq_partition_id(pid);
            if (s.item.partition_id() == pid) { s.item.update_partition(); upper = s.item.partition_max_color(); }
        }
        std::sort(its.begin(), its.end(), [](const Scored<MetaCursor>& a, const Scored<MetaCursor>& b) {
            return a.item.partition_id() < b.item.partition_id() ||
                   (a.item.partition_id() == b.item.partition_id() && a.item.meta_color() < b.item.meta_color());
        });
        uint64_t meta_score = its.front().score;
        auto flush = [&](Scored<MetaCursor>& s) {
            while (s.item.value() < upper) { scores[s.item.value()] += (uint32_t)meta_score; s.item.next(); }
        };
        size_t i = 1;
        for (; i < its.size(); ++i) {
            if (its[i].item.partition_id() != pid) break;
            if (its[i].item.meta_color() != its[i - 1].item.meta_color()) { flush(its[i - 1]); meta_score = 0; }
            meta_score += its[i].score;
        }
        flush(its[i - 1]);
    }
    for (uint32_t c = 0; c < n; ++c)
        if (scores[c] >= min_score) colors.push_back(c);
}

// ps_threshold_union.cpp:122-185
static inline void merge_diff(std::vector<Scored<DiffCursor>>& its, std::vector<uint32_t>& colors, uint64_t min_score) {
    if (its.empty()) return;
    const uint32_t n = its[0].item.num_colors();
    std::sort(its.begin(), its.end(), [](const Scored<DiffCursor>& a, const Scored<DiffCursor>& b) {
        return a.item.representative_begin() < b.item.representative_begin();
    });
    std::vector<uint32_t> pscores(n, 0), scores(n, 0);
    uint32_t score = 0, gsize = 0;
    for (size_t k = 0; k < its.size(); ++k) {
        Scored<DiffCursor> it = its[k];
        ++gsize;
        score += it.score;
        const bool last = k + 1 == its.size() || its[k + 1].item.representative_begin() != it.item.representative_begin();
        if (gsize == 1 && last) {
            for (uint32_t i = 0; i < it.item.size(); ++i, it.item.next()) scores[it.item.value()] += it.score;
            score = 0;
            gsize = 0;
            continue;
        }
        it.item.full_rewind();
        for (uint32_t v = it.item.differential_val(); v != n; it.item.next_differential_val(), v = it.item.differential_val())
            pscores[v] += it.score;
        if (last) {
            it.item.full_rewind();
            uint32_t v = it.item.representative_val();
            for (uint32_t c = 0; c < n; ++c) {
                if (v == c) {
                    scores[c] += score - pscores[c];
                    it.item.next_representative_val();
                    v = it.item.representative_val();
                } else {
                    scores[c] += pscores[c];
                }
            }
            score = 0;
            gsize = 0;
            std::fill(pscores.begin(), pscores.end(), 0);
        }
    }
    for (uint32_t c = 0; c < n; ++c)
        if (scores[c] >= min_score) colors.push_back(c);
}

// ps_threshold_union.cpp:187-318
static inline void merge_metadiff(std::vector<Scored<MetaDiffCursor>>& its, std::vector<uint32_t>& colors, uint64_t min_score) {
    if (its.empty()) return;
    const uint32_t n = its[0].item.num_colors();
    const uint32_t num_its = (uint32_t)its.size();
    std::vector<uint32_t> pids = scored_partitions(its, min_score);
    std::vector<uint32_t> scores(n, 0), pscores(n, 0);
    for (auto& s : its) { s.item.init(); s.item.change_partition(); }
    for (uint32_t pid : pids) {
        uint32_t num_sets = 0;
        for (auto& s : its) {
            s.item.next_geq_partition_id(pid);
            if (s.item.partition_id() == pid) { s.item.update_partition(); ++num_sets; }
        }
        std::sort(its.begin(), its.end(), [&](const Scored<MetaDiffCursor>& a, const Scored<MetaDiffCursor>& b) {
            const uint32_t ap = a.item.partition_id(), bp = b.item.partition_id();
            if (ap == pid && bp == pid) {
                const uint64_t ar = a.item.partition_it().representative_begin(), br = b.item.partition_it().representative_begin();
                const uint32_t am = a.item.meta_color(), bmc = b.item.meta_color();
                return am < bmc || (am == bmc && ar < br);
            }
            return ap < bp;
        });
        const uint32_t lower = its.front().item.partition_min_color();
        const uint32_t npc = its.front().item.partition_it().num_colors();
        uint32_t pscore = 0, gsize = 0, meta_score = 0;
        for (uint32_t k = 0; k < num_its; ++k) {
            Scored<MetaDiffCursor> it = its[k];
            if (it.item.partition_id() != pid) break;
            meta_score += it.score;
            --num_sets;
            ++gsize;
            if (num_sets != 0 && its[k + 1].item.meta_color() == it.item.meta_color()) continue;
            DiffCursor di = it.item.partition_it();
            pscore += meta_score;
            const bool last = num_sets == 0 || its[k + 1].item.partition_it().representative_begin() != di.representative_begin();
            if (last && gsize == 1) {
                for (uint32_t i = 0; i < di.size(); ++i, di.next()) scores[lower + di.value()] += meta_score;
                pscore = 0;
                gsize = 0;
                meta_score = 0;
                continue;
            }
            di.full_rewind();
            for (uint32_t v = di.differential_val(); v != npc; di.next_differential_val(), v = di.differential_val()) pscores[v] += meta_score;
            meta_score = 0;
            if (last) {
                di.full_rewind();
                uint32_t v = di.representative_val();
                for (uint32_t c = 0; c < npc; ++c) {
                    if (v == c) {
                        scores[lower + c] += pscore - pscores[c];
                        di.next_representative_val();
                        v = di.representative_val();
                    } else {
                        scores[lower + c] += pscores[c];
                    }
                }
                pscore = 0;
                gsize = 0;
                std::fill(pscores.begin(), pscores.begin() + npc, 0);
            }
        }
    }
    for (uint32_t c = 0; c < n; ++c)
        if (scores[c] >= min_score) colors.push_back(c);
}

// =================================================================================================
// builders from decoded colour sets + a deterministic assignment
//   partitions: colour ranges of `psize` colours (no colour permutation -> same numbering as hybrid)
//   clusters  : consecutive sets in groups of `csize`; representative = colours present in more than
//               half of the cluster's sets
// =================================================================================================
static inline std::vector<uint32_t> majority(const std::vector<std::vector<uint32_t>>& sets, size_t a, size_t b, uint32_t n) {
    std::vector<uint32_t> cnt(n, 0), rep;
    for (size_t i = a; i < b; ++i)
        for (uint32_t c : sets[i]) ++cnt[c];
    for (uint32_t c = 0; c < n; ++c)
        if (2 * (uint64_t)cnt[c] > b - a) rep.push_back(c);
    return rep;
}

static inline void build_diff(DiffColors& d, const std::vector<std::vector<uint32_t>>& sets, uint32_t n, uint32_t csize) {
    d.num_colors = n;
    for (size_t a = 0; a < sets.size(); a += csize) {
        const size_t b = std::min(sets.size(), a + csize);
        d.begin_cluster(majority(sets, a, b, n));
        for (size_t i = a; i < b; ++i) d.add_set(sets[i].data(), sets[i].size());
    }
    d.finish();
}

// per partition: distinct non-empty restrictions of the sets, numbered by first appearance
struct PartialTable {
    std::vector<std::vector<std::vector<uint32_t>>> partial;  // [partition][local id] -> relative colours
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> lists;  // per set: (partition, local id)
};
static inline PartialTable split_sets(const std::vector<std::vector<uint32_t>>& sets, uint32_t n, uint32_t psize) {
    const uint32_t P = (n + psize - 1) / psize;
    PartialTable t;
    t.partial.resize(P);
    t.lists.resize(sets.size());
    std::vector<std::vector<std::pair<std::vector<uint32_t>, uint32_t>>> seen(P);  // simple linear dedup per hash bucket
    std::vector<std::vector<std::vector<uint32_t>>> buckets(P, std::vector<std::vector<uint32_t>>(1 << 12));
    for (size_t s = 0; s < sets.size(); ++s) {
        size_t i = 0;
        while (i < sets[s].size()) {
            const uint32_t p = sets[s][i] / psize;
            std::vector<uint32_t> rel;
            for (; i < sets[s].size() && sets[s][i] / psize == p; ++i) rel.push_back(sets[s][i] - p * psize);
            uint64_t h = 1469598103934665603ULL;
            for (uint32_t c : rel) h = (h ^ c) * 1099511628211ULL;
            auto& bk = buckets[p][h & 4095];
            uint32_t id = (uint32_t)-1;
            for (uint32_t cand : bk)
                if (t.partial[p][cand] == rel) { id = cand; break; }
            if (id == (uint32_t)-1) {
                id = (uint32_t)t.partial[p].size();
                t.partial[p].push_back(rel);
                bk.push_back(id);
            }
            t.lists[s].push_back({p, id});
        }
    }
    return t;
}

static inline void build_meta(MetaColors& m, const std::vector<std::vector<uint32_t>>& sets, uint32_t n, uint32_t psize) {
    PartialTable t = split_sets(sets, n, psize);
    const uint32_t P = (uint32_t)t.partial.size();
    m.num_colors = n;
    m.partial.resize(P);
    m.endpoints.clear();
    uint32_t before = 0;
    for (uint32_t p = 0; p < P; ++p) {
        m.endpoints.push_back({p * psize, before});
        m.partial[p].init(std::min(psize, n - p * psize));
        for (auto& rel : t.partial[p]) m.partial[p].encode(rel.data(), rel.size());
        m.partial[p].seal();
        before += (uint32_t)t.partial[p].size();
    }
    m.endpoints.push_back({n, before});
    m.offsets.assign(1, 0);
    for (auto& l : t.lists) {  // encode_metacolor_set, meta.hpp:59-67
        m.lists.push_back(l.size());
        for (auto& pr : l) m.lists.push_back(m.endpoints[pr.first].num_color_sets_before + pr.second);
        m.offsets.push_back(m.lists.size());
    }
}

static inline void build_metadiff(MetaDiffColors& m, const std::vector<std::vector<uint32_t>>& sets, uint32_t n, uint32_t psize,
                                  uint32_t csize) {
    PartialTable t = split_sets(sets, n, psize);
    const uint32_t P = (uint32_t)t.partial.size();
    m.num_colors = n;
    m.partial.resize(P);
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t np = std::min(psize, n - p * psize);
        build_diff(m.partial[p], t.partial[p], np, csize);
        m.endpoints.push_back({(uint64_t)p * psize, (uint64_t)t.partial[p].size()});
    }
    m.rel_offsets.assign(1, 0);
    m.pset_offsets.assign(1, 0);
    m.num_partition_sets = 0;
    for (size_t s = 0; s < sets.size(); ++s) {
        bool same_as_prev = s > 0 && t.lists[s].size() == t.lists[s - 1].size();
        for (size_t i = 0; same_as_prev && i < t.lists[s].size(); ++i) same_as_prev = t.lists[s][i].first == t.lists[s - 1][i].first;
        if (!same_as_prev) {  // process_meta_color_partition_set, meta_differential.hpp:35-54
            m.pset_groups.set_last();
            write_delta(m.psets, t.lists[s].size());
            write_delta(m.psets, t.lists[s][0].first);
            for (size_t i = 1; i < t.lists[s].size(); ++i) write_delta(m.psets, t.lists[s][i].first - t.lists[s][i - 1].first);
            m.pset_offsets.push_back(m.psets.n);
            ++m.num_partition_sets;
        }
        m.pset_groups.push(false);  // process_metacolor_set, :62-73
        for (auto& pr : t.lists[s]) m.rel.push_bits(pr.second, msb_u64(m.endpoints[pr.first].num_color_sets) + 1);
        m.rel_offsets.push_back(m.rel.n);
    }
    m.pset_groups.set_last();
    m.pset_groups.build();
    m.rel.seal();
    m.psets.seal();
}

static inline MetaDiffCursor metadiff_color_set(const MetaDiffColors& m, uint64_t id) {  // meta_differential.hpp:269-276
    return MetaDiffCursor(&m, m.pset_offsets[m.pset_groups.rank1(id)], m.rel_offsets[id]);
}
static inline MetaCursor meta_color_set(const MetaColors& m, uint64_t id) { return MetaCursor(&m, m.offsets[id]); }  // meta.hpp:240-244

}  // namespace oracle
