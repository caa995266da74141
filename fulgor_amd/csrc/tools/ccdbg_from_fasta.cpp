// Test-data tool: builds a coloured compacted de Bruijn graph from a few hundred FASTA genomes at most and writes it
// in the reference's dump format (src/index.cpp:59-120). It stands in for GGCAT + `fulgor build`
// (Rust, not available here) for SMALL collections such as test_data/salmonella_10; it is not part
// of the query engine. Per SURVEY F7 any valid decomposition of the k-mer set into monochromatic
// paths gives the same pseudoalignment results, so unitigs need not equal GGCAT's.
//
// usage: ccdbg_from_fasta <k> <out_base> <genome1.fa[.gz]> [genome2 ...]   (colour id = argument order)
#include <zlib.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <string>
#include <vector>

static inline int code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

static std::vector<std::string> read_fasta(const char* path) {
    gzFile f = gzopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    std::vector<std::string> seqs;
    static char buf[1 << 16];
    while (gzgets(f, buf, sizeof(buf))) {
        if (buf[0] == '>') { seqs.emplace_back(); continue; }
        if (seqs.empty()) seqs.emplace_back();
        for (char* p = buf; *p; ++p)
            if (*p != '\n' && *p != '\r') seqs.back().push_back(*p);
    }
    gzclose(f);
    return seqs;
}

struct Table {  // open addressing: canonical k-mer -> colour mask of W words
    std::vector<uint64_t> key, val;
    std::vector<uint8_t> used, visited;
    uint64_t mask;
    uint32_t W;
    Table(unsigned log2cap, uint32_t words) : key(1ULL << log2cap), val((1ULL << log2cap) * words, 0), used(1ULL << log2cap, 0),
                                              visited(1ULL << log2cap, 0), mask((1ULL << log2cap) - 1), W(words) {}
    std::vector<uint64_t> colour(uint64_t i) const { return std::vector<uint64_t>(val.begin() + i * W, val.begin() + (i + 1) * W); }
    bool same_colour(uint64_t i, uint64_t j) const { return std::equal(val.begin() + i * W, val.begin() + (i + 1) * W, val.begin() + j * W); }
    static uint64_t h(uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    }
    int64_t find(uint64_t k) const {
        for (uint64_t i = h(k) & mask;; i = (i + 1) & mask) {
            if (!used[i]) return -1;
            if (key[i] == k) return (int64_t)i;
        }
    }
    uint64_t insert(uint64_t k) {
        for (uint64_t i = h(k) & mask;; i = (i + 1) & mask) {
            if (!used[i]) { used[i] = 1; key[i] = k; return i; }
            if (key[i] == k) return i;
        }
    }
};

static uint32_t K;
static uint64_t KMASK;
static inline uint64_t rc(uint64_t x) {  // first base most significant, 2 bits per base
    uint64_t r = 0;
    for (uint32_t i = 0; i < K; ++i) { r = (r << 2) | (3 - (x & 3)); x >>= 2; }
    return r;
}
static inline uint64_t canon(uint64_t x) { uint64_t r = rc(x); return x < r ? x : r; }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <k> <out_base> <genome.fa[.gz]>...\n", argv[0]); return 1; }
    K = (uint32_t)atoi(argv[1]);
    if (K < 3 || K > 31 || (K & 1) == 0) { fprintf(stderr, "k must be odd, 3..31\n"); return 1; }
    KMASK = (1ULL << (2 * K)) - 1;
    std::string base = argv[2];
    const int ng = argc - 3;
    const uint32_t W = (uint32_t)(ng + 63) / 64;

    uint64_t total = 0;
    std::vector<std::vector<std::string>> genomes;
    for (int g = 0; g < ng; ++g) {
        genomes.push_back(read_fasta(argv[3 + g]));
        for (auto& s : genomes.back()) total += s.size();
    }
    unsigned lg = 10;
    while ((1ULL << lg) < total * 2) ++lg;  // generous: distinct k-mers <= total bases
    if (lg > 30) lg = 30;
    Table T(lg, W);
    uint64_t distinct = 0;
    for (int g = 0; g < ng; ++g)
        for (auto& s : genomes[g]) {
            uint64_t f = 0, r = 0;
            uint32_t run = 0;
            for (char ch : s) {
                int c = code(ch);
                if (c < 0) { run = 0; continue; }
                f = ((f << 2) | (uint64_t)c) & KMASK;
                r = (r >> 2) | ((uint64_t)(3 - c) << (2 * (K - 1)));
                if (++run >= K) {
                    uint64_t i = T.insert(f < r ? f : r);
                    bool fresh = true;
                    for (uint32_t w = 0; w < W; ++w) fresh = fresh && T.val[i * W + w] == 0;
                    if (fresh) ++distinct;
                    T.val[i * W + (g >> 6)] |= 1ULL << (g & 63);
                }
            }
        }
    fprintf(stderr, "distinct canonical %u-mers: %llu\n", K, (unsigned long long)distinct);

    // colour sets: distinct masks, numbered by increasing mask value (word 0 first)
    std::map<std::vector<uint64_t>, uint32_t> set_id;
    for (uint64_t i = 0; i <= T.mask; ++i)
        if (T.used[i]) set_id.emplace(T.colour(i), 0);
    { uint32_t id = 0; for (auto& kv : set_id) kv.second = id++; }
    fprintf(stderr, "distinct colour sets: %zu\n", set_id.size());

    // unitigs: maximal monochromatic non-branching paths
    auto unique_succ = [&](uint64_t x, uint64_t& out) -> bool {  // oriented successor if exactly one exists
        int n = 0;
        for (uint64_t b = 0; b < 4; ++b) {
            uint64_t y = ((x << 2) | b) & KMASK;
            if (T.find(canon(y)) >= 0) { out = y; ++n; }
        }
        return n == 1;
    };
    auto extend = [&](uint64_t start, std::vector<uint64_t>& path) {
        uint64_t cur = start;
        const uint64_t colour_of = (uint64_t)T.find(canon(cur));
        for (;;) {
            uint64_t nxt, back;
            if (!unique_succ(cur, nxt)) break;
            if (!unique_succ(rc(nxt), back) || back != rc(cur)) break;
            int64_t ni = T.find(canon(nxt));
            if (T.visited[ni] || !T.same_colour((uint64_t)ni, colour_of)) break;
            T.visited[ni] = 1;
            path.push_back(nxt);
            cur = nxt;
        }
    };
    struct Unitig { uint32_t set; std::string seq; };
    std::vector<Unitig> unitigs;
    const char* ALPHA = "ACGT";
    auto kmer_str = [&](uint64_t x) {
        std::string s(K, 'A');
        for (uint32_t i = 0; i < K; ++i) s[i] = ALPHA[(x >> (2 * (K - 1 - i))) & 3];
        return s;
    };
    uint64_t nk_check = 0;
    for (uint64_t i = 0; i <= T.mask; ++i) {
        if (!T.used[i] || T.visited[i]) continue;
        T.visited[i] = 1;
        std::vector<uint64_t> fw, bw;
        extend(T.key[i], fw);
        extend(rc(T.key[i]), bw);
        // path = rc(reverse(bw)) + key + fw
        std::vector<uint64_t> path;
        for (auto it = bw.rbegin(); it != bw.rend(); ++it) path.push_back(rc(*it));
        path.push_back(T.key[i]);
        path.insert(path.end(), fw.begin(), fw.end());
        std::string seq = kmer_str(path[0]);
        for (size_t j = 1; j < path.size(); ++j) seq.push_back(ALPHA[path[j] & 3]);
        nk_check += path.size();
        unitigs.push_back({set_id[T.colour(i)], std::move(seq)});
    }
    if (nk_check != distinct) { fprintf(stderr, "internal error: %llu k-mers in unitigs\n", (unsigned long long)nk_check); return 1; }
    std::stable_sort(unitigs.begin(), unitigs.end(), [](const Unitig& a, const Unitig& b) { return a.set < b.set; });
    fprintf(stderr, "unitigs: %zu\n", unitigs.size());

    {
        std::ofstream o(base + ".metadata.txt");
        o << "k=" << K << '\n' << "num_kmers=" << distinct << '\n' << "num_colors=" << ng << '\n'
          << "num_unitigs=" << unitigs.size() << '\n' << "num_color_sets=" << set_id.size() << '\n';
    }
    {
        std::ofstream o(base + ".filenames.txt");
        for (int g = 0; g < ng; ++g) {
            std::string p = argv[3 + g];
            o << p.substr(p.find_last_of("/\\") + 1) << '\n';
        }
    }
    {
        std::ofstream o(base + ".unitigs.fa");
        for (auto& u : unitigs) o << "> color_set_id=" << u.set << '\n' << u.seq << '\n';
    }
    {
        std::ofstream o(base + ".color_sets.txt");
        for (auto& kv : set_id) {
            int size = 0;
            for (uint64_t w : kv.first) size += __builtin_popcountll(w);
            o << "size=" << size;
            for (int g = 0; g < ng; ++g)
                if ((kv.first[g >> 6] >> (g & 63)) & 1) o << ' ' << g;
            o << '\n';
        }
    }
    return 0;
}
