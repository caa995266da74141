// Test-data tool: SYNTHETIC index with the shape of salmonella_4546 (README.md:312-316 of the reference:
// k=31, 4546 colours, ~43.8M k-mers, ~1.88M unitigs, ~0.97M colour sets). The real collection is a
// Zenodo download that is not available offline, so every number measured on this index is
// labelled synthetic. Not part of the query engine.
//
// Model (seeded, deterministic):
//   * a random binary phylogeny over 4546 strains; colour ids are a random permutation of the leaves
//   * the 10 real salmonella_10 genomes stand for 10 top-level clades C_0..C_9 of that tree; each keeps
//     one "type strain" t_g that carries the genome unchanged
//   * CORE: every salmonella_10 unitig (real sequence, real colour set S over the 10 genomes) is cut into
//     pieces of ~geometric(1/24) k-mers; a piece is carried by (union of C_g, g in S) minus 1-3 random
//     clades (the strains that lost / mutated that segment), type strains always kept
//   * ACCESSORY: random-sequence contigs of ~1100 k-mers, each carried by a random clade A (log-uniform
//     size); consecutive unitigs (~geometric(1/22) k-mers) drop a random sub-clade of A
//   * with probability 0.35 a unitig additionally flips 1-4 random strains (sporadic gain/loss)
//   * identical colour sets are merged; unitigs are sorted by colour-set id (builder.hpp:114-131)
// Reads for the benchmark are drawn from the 10 real genomes and from the accessory sequence, so reads
// cross several unitigs with nested colour sets, as real reads do.
//
// usage: synth_s4546 <s10_dump_base> <out.fgidx> <out_accessory.txt> [seed [core_stride [target_kmers [profile]]]]
//        core_stride > 1 keeps every core_stride-th salmonella_10 unitig only and target_kmers shrinks the accessory part: a small
//        index with the same 4546 colours and list shapes, for tests that move the whole index through text files
//        profile 1 = CORE-HEAVY (round 3: bounds the risk of the default model, whose reads have small results — median <= 10
//        colours — while real Salmonella reads are expected to be dominated by core-genome sets of >= 3409 colours):
//        nine salmonella_10 unitigs in ten are carried by the whole collection whatever their set over the ten genomes, a piece
//        loses at most one small clade (<= 64 strains) with probability 0.3. More than 70 % of the mapped reads then have
//        results of at least 3409 colours, and the dense (complemented) lists dominate the colour work.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <unordered_map>

#include "../host/index_io.hpp"

using namespace fg;

namespace {

constexpr uint32_t N = 4546;
constexpr uint32_t NW = (N + 63) / 64;
constexpr uint32_t K = 31;
uint64_t TARGET_KMERS = 43788757;
uint64_t CORE_STRIDE = 1;
int PROFILE = 0;  // 1: core-heavy

uint64_t rng_state;
inline uint64_t rnd() {
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
inline uint32_t geometric(double mean) {  // >= 1
    double u = (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
    uint32_t v = 1 + (uint32_t)(-std::log(1.0 - u) * (mean - 1.0));
    return v;
}

struct Node { uint32_t a, b; int left, right, parent; };  // leaves [a,b) in DFS order
std::vector<Node> tree;
std::vector<uint32_t> perm;  // DFS leaf position -> colour id

int build_tree(uint32_t a, uint32_t b, int parent) {
    int id = (int)tree.size();
    tree.push_back({a, b, -1, -1, parent});
    if (b - a > 1) {
        uint32_t sz = b - a;
        double f = 0.15 + 0.7 * (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
        uint32_t l = std::min(sz - 1, std::max<uint32_t>(1, (uint32_t)(sz * f)));
        int L = build_tree(a, a + l, id);
        int R = build_tree(a + l, b, id);
        tree[id].left = L;
        tree[id].right = R;
    }
    return id;
}

struct Bitmap { uint64_t w[NW]; };
std::vector<Bitmap> node_bm;

void clade_bitmap(int node, Bitmap& bm) {
    memset(&bm, 0, sizeof(bm));
    for (uint32_t i = tree[node].a; i < tree[node].b; ++i) bm.w[perm[i] >> 6] |= 1ULL << (perm[i] & 63);
}

// random strict descendant of `node` (or -1 if it is a leaf): walk down, stop with probability 0.35
int random_subclade(int node) {
    if (tree[node].left < 0) return -1;
    int cur = (rnd() & 1) ? tree[node].left : tree[node].right;
    while (tree[cur].left >= 0 && (rnd() % 100) >= 35) cur = (rnd() & 1) ? tree[cur].left : tree[cur].right;
    return cur;
}

// sporadic presence/absence: flip 1-4 random strains (horizontal transfer, assembly gaps), never a
// type strain
void sporadic(Bitmap& bm, const Bitmap& keep) {
    const uint32_t flips = 1 + (uint32_t)(rnd() % 4);
    for (uint32_t t = 0; t < flips; ++t) {
        const uint32_t c = (uint32_t)(rnd() % N);
        if ((keep.w[c >> 6] >> (c & 63)) & 1) continue;
        bm.w[c >> 6] ^= 1ULL << (c & 63);
    }
}

struct SetTable {
    std::unordered_map<uint64_t, std::vector<uint32_t>> by_hash;  // hash -> set ids
    std::vector<Bitmap> sets;
    uint32_t intern(const Bitmap& bm) {
        uint64_t h = 1469598103934665603ULL;
        for (uint32_t i = 0; i < NW; ++i) h = mix64(h ^ bm.w[i]);
        auto& v = by_hash[h];
        for (uint32_t id : v)
            if (memcmp(&sets[id], &bm, sizeof(Bitmap)) == 0) return id;
        v.push_back((uint32_t)sets.size());
        sets.push_back(bm);
        return (uint32_t)sets.size() - 1;
    }
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <s10_dump_base> <out.fgidx> <out_accessory.txt> [seed]\n", argv[0]); return 1; }
    const std::string base = argv[1];
    rng_state = argc > 4 ? strtoull(argv[4], nullptr, 10) : 4546;
    if (argc > 5) CORE_STRIDE = std::max<uint64_t>(1, strtoull(argv[5], nullptr, 10));
    if (argc > 6) TARGET_KMERS = strtoull(argv[6], nullptr, 10);
    if (argc > 7) PROFILE = atoi(argv[7]);

    // ---- phylogeny ----
    tree.reserve(2 * N);
    build_tree(0, N, -1);
    perm.resize(N);
    for (uint32_t i = 0; i < N; ++i) perm[i] = i;
    for (uint32_t i = N - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
    node_bm.resize(tree.size());
    for (size_t i = 0; i < tree.size(); ++i) clade_bitmap((int)i, node_bm[i]);
    // 10 top clades: repeatedly split the largest
    std::vector<int> top{0};
    while (top.size() < 10) {
        size_t bi = 0;
        for (size_t i = 1; i < top.size(); ++i)
            if (tree[top[i]].b - tree[top[i]].a > tree[top[bi]].b - tree[top[bi]].a) bi = i;
        int nd = top[bi];
        top[bi] = tree[nd].left;
        top.push_back(tree[nd].right);
    }
    Bitmap type_strains;
    memset(&type_strains, 0, sizeof(type_strains));
    for (int g = 0; g < 10; ++g) {
        uint32_t c = perm[tree[top[g]].a];
        type_strains.w[c >> 6] |= 1ULL << (c & 63);
    }
    // nodes bucketed by log2(size) for log-uniform clade sampling
    std::vector<std::vector<int>> by_log(14);
    for (size_t i = 0; i < tree.size(); ++i) by_log[msb64(tree[i].b - tree[i].a)].push_back((int)i);
    auto random_clade = [&]() {
        for (;;) {
            auto& v = by_log[rnd() % by_log.size()];
            if (!v.empty()) return v[rnd() % v.size()];
        }
    };

    auto random_small_clade = [&]() {  // at most 64 strains
        for (;;) {
            auto& v = by_log[rnd() % 7];
            if (!v.empty()) return v[rnd() % v.size()];
        }
    };

    // ---- salmonella_10 dump: unitigs + their real colour sets ----
    std::vector<uint16_t> real_set;  // mask over the 10 genomes, per S10 colour-set id
    {
        std::ifstream in(base + ".color_sets.txt");
        if (!in.is_open()) { fprintf(stderr, "cannot open %s.color_sets.txt\n", base.c_str()); return 1; }
        std::string line;
        while (std::getline(in, line)) {
            const char* p = line.c_str() + line.find("size=") + 5;
            char* e;
            uint64_t sz = strtoull(p, &e, 10);
            uint16_t m = 0;
            for (uint64_t j = 0; j < sz; ++j) { p = e; m |= (uint16_t)(1u << strtoul(p, &e, 10)); }
            real_set.push_back(m);
        }
    }
    SetTable st;
    struct Unitig { uint32_t set; uint64_t off; uint32_t len; };
    std::vector<Unitig> unitigs;
    std::string seqs;  // all unitig sequences, in creation order
    uint64_t nk = 0;
    {
        std::ifstream in(base + ".unitigs.fa");
        if (!in.is_open()) { fprintf(stderr, "cannot open %s.unitigs.fa\n", base.c_str()); return 1; }
        std::string header, seq;
        uint64_t unitig_no = 0;
        while (std::getline(in, header) && std::getline(in, seq)) {
            if (unitig_no++ % CORE_STRIDE) continue;
            const uint32_t sid = (uint32_t)strtoul(header.c_str() + header.find("color_set_id=") + 13, nullptr, 10);
            Bitmap carriers;
            memset(&carriers, 0, sizeof(carriers));
            for (int g = 0; g < 10; ++g)
                if ((real_set[sid] >> g) & 1)
                    for (uint32_t i = 0; i < NW; ++i) carriers.w[i] |= node_bm[top[g]].w[i];
            if (PROFILE == 1 && rnd() % 100 < 90) carriers = node_bm[0];  // a core locus of the whole collection
            const uint64_t nkm = seq.size() - K + 1;
            for (uint64_t s = 0; s < nkm;) {
                uint64_t piece = std::min<uint64_t>(nkm - s, geometric(24.0));
                Bitmap bm = carriers;
                if (rnd() % 100 < (PROFILE == 1 ? 30u : 85u)) {
                    const uint32_t drops = PROFILE == 1 ? 1u : 1 + (uint32_t)(rnd() % 3);  // 1-3 clades lost this segment
                    for (uint32_t t = 0; t < drops; ++t) {
                        const Bitmap& d = node_bm[PROFILE == 1 ? random_small_clade() : random_clade()];
                        for (uint32_t i = 0; i < NW; ++i) bm.w[i] &= ~d.w[i] | (type_strains.w[i] & carriers.w[i]);
                    }
                }
                if (rnd() % 100 < 35) sporadic(bm, type_strains);
                bool any = false;
                for (uint32_t i = 0; i < NW; ++i) any |= bm.w[i] != 0;
                if (!any) bm = carriers;
                unitigs.push_back({st.intern(bm), seqs.size(), (uint32_t)(piece + K - 1)});
                seqs.append(seq, s, piece + K - 1);
                nk += piece;
                s += piece;
            }
        }
    }
    fprintf(stderr, "core: %zu unitigs, %llu k-mers, %zu colour sets\n", unitigs.size(), (unsigned long long)nk, st.sets.size());

    // ---- accessory contigs ----
    std::string accessory;
    const char* ALPHA = "ACGT";
    while (nk < TARGET_KMERS) {
        const uint64_t contig_k = std::min<uint64_t>(TARGET_KMERS - nk, 600 + rnd() % 1000);
        const size_t c0 = accessory.size();
        for (uint64_t i = 0; i < contig_k + K - 1; ++i) accessory.push_back(ALPHA[rnd() & 3]);
        const int A = random_clade();
        for (uint64_t s = 0; s < contig_k;) {
            uint64_t piece = std::min<uint64_t>(contig_k - s, geometric(22.0));
            Bitmap bm = node_bm[A];
            if (rnd() % 100 < 70) {
                int d = random_subclade(A);
                if (d >= 0)
                    for (uint32_t i = 0; i < NW; ++i) bm.w[i] &= ~node_bm[d].w[i];
            }
            if (rnd() % 100 < 35) sporadic(bm, type_strains);
            {
                bool any = false;
                for (uint32_t i = 0; i < NW; ++i) any |= bm.w[i] != 0;
                if (!any) bm = node_bm[A];
            }
            unitigs.push_back({st.intern(bm), seqs.size(), (uint32_t)(piece + K - 1)});
            seqs.append(accessory, c0 + s, piece + K - 1);
            nk += piece;
            s += piece;
        }
        accessory.push_back('N');
    }
    fprintf(stderr, "total: %zu unitigs, %llu k-mers, %zu colour sets, accessory %zu bases\n", unitigs.size(),
            (unsigned long long)nk, st.sets.size(), accessory.size());

    // ---- assemble the index ----
    HostIndex idx;
    idx.type = IDX_HYBRID;
    {
        HybridEncoder enc;
        enc.init(N);
        std::vector<uint32_t> v;
        uint64_t ints = 0;
        for (auto& bm : st.sets) {
            v.clear();
            for (uint32_t w = 0; w < NW; ++w)
                for (uint64_t x = bm.w[w]; x; x &= x - 1) v.push_back(w * 64 + (uint32_t)__builtin_ctzll(x));
            ints += v.size();
            enc.encode(v.data(), v.size());
        }
        enc.finish(idx.hybrid);
        hybrid_build_blocks(idx.hybrid);
        fprintf(stderr, "colour stream: %.1f MB, %.1f M integers; packed gap blocks: %zu blocks, %.1f MB\n",
                idx.hybrid.nbits / 8e6, ints / 1e6, idx.hybrid.blk_hdr.size(),
                (idx.hybrid.blk_hdr.size() * 8.0 + idx.hybrid.blk_words.size() * 4.0) / 1e6);
    }
    {
        std::vector<uint32_t> order(unitigs.size());
        for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return unitigs[a].set < unitigs[b].set; });
        std::string bases;
        bases.reserve(seqs.size());
        std::vector<uint64_t> off(1, 0);
        std::vector<uint32_t> csid;
        for (uint32_t i : order) {
            bases.append(seqs, unitigs[i].off, unitigs[i].len);
            off.push_back(bases.size());
            csid.push_back(unitigs[i].set);
        }
        std::string().swap(seqs);
        build_dict(idx.dict, K, 17, bases.data(), bases.size(), off, csid);
        const DictStats ds = dict_stats(idx.dict);
        fprintf(stderr, "dictionary: %llu k-mers, %llu super-k-mer records in %llu hashed buckets of 64 bytes (%.0f MB with %llu overflow buckets), %llu redirects, %llu spill buckets\n",
                (unsigned long long)idx.dict.num_kmers, (unsigned long long)ds.records, (unsigned long long)ds.buckets,
                idx.dict.table.size() * 4e-6, (unsigned long long)ds.overflow_buckets, (unsigned long long)ds.redirects,
                (unsigned long long)ds.spill_buckets);
    }
    verify_dict(idx.dict, 997);
    for (uint32_t c = 0; c < N; ++c) idx.filenames.push_back("synthetic_strain_" + std::to_string(c) + ".fasta");
    save_binary(idx, argv[2]);
    {
        std::ofstream o(argv[3], std::ios::binary);
        o.write(accessory.data(), accessory.size());
    }
    return 0;
}
