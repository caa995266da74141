// Synthetic read generator (bench/test tooling; not part of the query engine). SURVEY §8(d):
// PRNG = splitmix64; reads are generated in shards of 65536 reads, shard s uses seed (seed + s), so any
// slice of the global read set can be produced independently (one slice per GPU rank).
// Per read: 5% are uniform random ACGT (unmapped controls); otherwise pick a genome uniformly, a start
// uniformly among the windows of `read_len` ACGT-only bases, copy, reverse-complement with p=1/2, then
// substitute each base with probability 1/100 (uniform over the three other bases).
#include <zlib.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Run { uint64_t genome_pos; uint64_t len; };
struct Genome {
    std::string seq;                 // concatenation of all contigs (non-ACGT kept as 'N')
    std::vector<uint64_t> run_start; // maximal ACGT runs
    std::vector<uint64_t> run_len;
};
struct Genomes {
    std::vector<Genome> g;
};

inline uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

inline int code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

void add_sequence(Genome& G, const std::string& contig) {
    if (!G.seq.empty()) G.seq.push_back('N');  // contigs never join
    G.seq += contig;
}

void finish_genome(Genome& G) {
    uint64_t start = 0, len = 0;
    for (uint64_t i = 0; i <= G.seq.size(); ++i) {
        if (i < G.seq.size() && code(G.seq[i]) >= 0) {
            if (len == 0) start = i;
            ++len;
        } else {
            if (len) { G.run_start.push_back(start); G.run_len.push_back(len); }
            len = 0;
        }
    }
}

}  // namespace

extern "C" {

void* fgt_genomes_new() { return new Genomes(); }

// append one genome from a FASTA(.gz) file; returns its number of bases or -1
int64_t fgt_genomes_add_fasta(void* h, const char* path) {
    gzFile f = gzopen(path, "rb");
    if (!f) return -1;
    Genome G;
    std::string contig;
    static thread_local char buf[1 << 16];
    while (gzgets(f, buf, sizeof(buf))) {
        if (buf[0] == '>') {
            if (!contig.empty()) add_sequence(G, contig);
            contig.clear();
            continue;
        }
        for (char* p = buf; *p; ++p)
            if (*p != '\n' && *p != '\r') contig.push_back(*p);
    }
    if (!contig.empty()) add_sequence(G, contig);
    gzclose(f);
    finish_genome(G);
    int64_t n = (int64_t)G.seq.size();
    static_cast<Genomes*>(h)->g.push_back(std::move(G));
    return n;
}

// append one genome given as raw bases (used for synthetic accessory sequence)
int64_t fgt_genomes_add_raw(void* h, const char* bases, uint64_t n) {
    Genome G;
    G.seq.assign(bases, n);
    finish_genome(G);
    static_cast<Genomes*>(h)->g.push_back(std::move(G));
    return (int64_t)n;
}

void fgt_genomes_free(void* h) { delete static_cast<Genomes*>(h); }

// reads [first, first+count) of the global read set -> out[count * read_len]; returns 0 or -1
int fgt_generate_reads(void* h, uint64_t first, uint64_t count, uint32_t read_len, uint64_t seed, char* out, int nthreads) {
    const Genomes& GS = *static_cast<Genomes*>(h);
    const uint64_t ng = GS.g.size();
    if (ng == 0) return -1;
    // cumulative numbers of valid starts per genome
    std::vector<std::vector<uint64_t>> cum(ng);
    for (uint64_t g = 0; g < ng; ++g) {
        cum[g].push_back(0);
        for (uint64_t L : GS.g[g].run_len) cum[g].push_back(cum[g].back() + (L >= read_len ? L - read_len + 1 : 0));
        if (cum[g].back() == 0) return -1;
    }
    const uint64_t SHARD = 65536;
    const uint64_t s0 = first / SHARD, s1 = (first + count + SHARD - 1) / SHARD;
    if (nthreads < 1) nthreads = 1;
    auto work = [&](uint64_t sa, uint64_t sb) {
        const char* ALPHA = "ACGT";
        std::vector<char> tmp(read_len);
        for (uint64_t s = sa; s < sb; ++s) {
            uint64_t st = seed + s;
            for (uint64_t i = s * SHARD; i < (s + 1) * SHARD && i < first + count; ++i) {
                // every read consumes its random numbers whether or not it is inside the slice
                const bool keep = i >= first;
                char* dst = keep ? out + (i - first) * (uint64_t)read_len : tmp.data();
                if (splitmix64(st) % 100 < 5) {
                    for (uint32_t b = 0; b < read_len; ++b) dst[b] = ALPHA[splitmix64(st) & 3];
                    continue;
                }
                const uint64_t g = splitmix64(st) % ng;
                const uint64_t t = splitmix64(st) % cum[g].back();
                const uint64_t ri = std::upper_bound(cum[g].begin(), cum[g].end(), t) - cum[g].begin() - 1;
                const uint64_t pos = GS.g[g].run_start[ri] + (t - cum[g][ri]);
                const char* src = GS.g[g].seq.data() + pos;
                const bool rc = splitmix64(st) & 1;
                for (uint32_t b = 0; b < read_len; ++b) {
                    int c = rc ? 3 - code(src[read_len - 1 - b]) : code(src[b]);
                    if (splitmix64(st) % 100 == 0) c = (c + 1 + (int)(splitmix64(st) % 3)) & 3;
                    dst[b] = ALPHA[c];
                }
            }
        }
    };
    std::vector<std::thread> th;
    const uint64_t ns = s1 - s0, per = (ns + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        uint64_t a = s0 + std::min<uint64_t>(ns, t * per), b = s0 + std::min<uint64_t>(ns, (t + 1) * per);
        if (a < b) th.emplace_back(work, a, b);
    }
    for (auto& x : th) x.join();
    return 0;
}

}  // extern "C"
