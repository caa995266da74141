// ORACLE — test infrastructure only (see the header of fulgor_oracle.hpp; PARITY UNPINNED).
// C entry points over the CPU restatement so that tests/ and bench.py's cpu_baseline leg can drive it
// through ctypes. Threading follows the reference driver: `nthreads` workers pull chunks of reads
// (tools/pseudoalign.cpp:66-74); results are returned as CSR in read order.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include "fulgor_oracle.hpp"

using namespace oracle;

static thread_local std::string g_err;

extern "C" {

const char* fo_last_error() { return g_err.c_str(); }

void* fo_index_load_dump(const char* base) {
    try {
        auto* ix = new AnyIndex();
        ix->load_dump(base);
        return ix;
    } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}

void* fo_index_from_arrays(uint32_t k, const char* unitig_bases, const uint64_t* unitig_off, const uint32_t* unitig_csid,
                           uint64_t num_unitigs, uint32_t num_colors, uint32_t sparse_thr, uint32_t dense_thr,
                           const uint64_t* words, uint64_t nbits, const uint64_t* offsets, uint64_t num_sets) {
    try {
        auto* ix = new AnyIndex();
        ix->k = k;
        ix->add_unitigs(unitig_bases, unitig_off, unitig_csid, num_unitigs);
        ix->set_colors(num_colors, sparse_thr, dense_thr, words, nbits, offsets, num_sets);
        return ix;
    } catch (std::exception& e) { g_err = e.what(); return nullptr; }
}

void fo_index_free(void* h) { delete static_cast<AnyIndex*>(h); }

void fo_index_info(void* h, uint64_t* k, uint64_t* num_colors, uint64_t* num_sets, uint64_t* num_unitigs, uint64_t* nbits) {
    auto* ix = static_cast<AnyIndex*>(h);
    *k = ix->k;
    *num_colors = ix->colors.num_colors;
    *num_sets = ix->colors.num_sets();
    *num_unitigs = ix->u2c_table.size();
    *nbits = ix->colors.bits.n;
}
// encoded stream export (encoder parity against the product's encoder)
const uint64_t* fo_colors_words(void* h) { return static_cast<AnyIndex*>(h)->colors.bits.w.data(); }
const uint64_t* fo_colors_offsets(void* h) { return static_cast<AnyIndex*>(h)->colors.offsets.data(); }

void fo_free(void* p) { free(p); }

// re-encode the colour sets with another codec: 1 differential, 2 meta, 3 meta-differential (0 = back to hybrid)
int fo_index_convert(void* h, int type, uint32_t partition_size, uint32_t cluster_size) {
    try {
        static_cast<AnyIndex*>(h)->convert(type, partition_size, cluster_size);
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"

namespace {

// runs fn(read index, output vector) over all reads with a pool of workers; gathers CSR
template <typename Fn>
int run_batch(uint64_t n, int nthreads, uint64_t** out_off, uint32_t** out_vals, Fn fn) {
    try {
        if (nthreads < 1) nthreads = 1;
        std::vector<std::vector<uint32_t>> res(n);
        std::atomic<uint64_t> next{0};
        std::atomic<bool> failed{false};
        std::string err;
        auto work = [&]() {
            try {
                for (;;) {
                    uint64_t b = next.fetch_add(256);
                    if (b >= n) break;
                    uint64_t e = std::min(n, b + 256);
                    for (uint64_t r = b; r < e; ++r) fn(r, res[r]);
                }
            } catch (std::exception& ex) { if (!failed.exchange(true)) err = ex.what(); }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work);
        for (auto& t : th) t.join();
        if (failed) { g_err = err; return -1; }
        uint64_t* off = (uint64_t*)malloc((n + 1) * sizeof(uint64_t));
        off[0] = 0;
        for (uint64_t r = 0; r < n; ++r) off[r + 1] = off[r] + res[r].size();
        uint32_t* vals = (uint32_t*)malloc(std::max<uint64_t>(1, off[n]) * sizeof(uint32_t));
        for (uint64_t r = 0; r < n; ++r)
            if (!res[r].empty()) memcpy(vals + off[r], res[r].data(), res[r].size() * sizeof(uint32_t));
        *out_off = off;
        *out_vals = vals;
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}

}  // namespace

extern "C" {

int fo_fetch_color_set_ids(void* h, const char* bases, const uint64_t* offs, uint64_t n, uint64_t** out_off,
                           uint32_t** out_ids, int nthreads) {
    auto* ix = static_cast<AnyIndex*>(h);
    return run_batch(n, nthreads, out_off, out_ids, [&](uint64_t r, std::vector<uint32_t>& out) {
        out.clear();  // callers clear first (ps_utils.cpp:277)
        ix->fetch_color_set_ids(bases + offs[r], offs[r + 1] - offs[r], out);
    });
}

int fo_full_intersection(void* h, const char* bases, const uint64_t* offs, uint64_t n, uint64_t** out_off,
                         uint32_t** out_colors, int nthreads, int self_check) {
    auto* ix = static_cast<AnyIndex*>(h);
    return run_batch(n, nthreads, out_off, out_colors, [&](uint64_t r, std::vector<uint32_t>& out) {
        std::vector<uint32_t> ids, tmp;
        ix->fetch_color_set_ids(bases + offs[r], offs[r + 1] - offs[r], ids);
        if (ix->type == 0) ix->full_intersection(ids, out, tmp, self_check != 0);
        else ix->any_full_intersection(ids, out, tmp);
    });
}

int fo_intersect_ids(void* h, const uint32_t* ids, const uint64_t* id_offs, uint64_t n, uint64_t** out_off,
                     uint32_t** out_colors, int nthreads, int self_check) {
    auto* ix = static_cast<AnyIndex*>(h);
    return run_batch(n, nthreads, out_off, out_colors, [&](uint64_t r, std::vector<uint32_t>& out) {
        std::vector<uint32_t> v(ids + id_offs[r], ids + id_offs[r + 1]), tmp;
        if (ix->type == 0) ix->full_intersection(v, out, tmp, self_check != 0);
        else ix->any_full_intersection(v, out, tmp);
    });
}

int fo_threshold_union(void* h, const char* bases, const uint64_t* offs, uint64_t n, double tau, uint64_t** out_off,
                       uint32_t** out_colors, int nthreads, int self_check) {
    auto* ix = static_cast<AnyIndex*>(h);
    return run_batch(n, nthreads, out_off, out_colors, [&](uint64_t r, std::vector<uint32_t>& out) {
        out.clear();
        if (ix->type == 0) ix->threshold_union(bases + offs[r], offs[r + 1] - offs[r], tau, out, self_check != 0);
        else ix->any_threshold_union(bases + offs[r], offs[r + 1] - offs[r], tau, out);
    });
}

// CPU baseline: the reference's worker loop without output (tools/pseudoalign.cpp:12-54 with -o /dev/null):
// every worker runs fetch_color_set_ids + the chosen algorithm and only counts mapped reads.
// algo 0 = full-intersection, 1 = threshold-union. Returns wall seconds.
double fo_time_pseudoalign(void* h, const char* bases, const uint64_t* offs, uint64_t n, int algo, double tau,
                           int nthreads, uint64_t* num_mapped, uint64_t* total_colors) {
    auto* ix = static_cast<AnyIndex*>(h);
    if (nthreads < 1) nthreads = 1;
    std::atomic<uint64_t> next{0}, mapped{0}, total{0};
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        std::vector<uint32_t> ids, colors, tmp;
        uint64_t m = 0, tc = 0;
        for (;;) {
            uint64_t b = next.fetch_add(1024);
            if (b >= n) break;
            uint64_t e = std::min(n, b + 1024);
            for (uint64_t r = b; r < e; ++r) {
                ids.clear();
                colors.clear();
                ix->fetch_color_set_ids(bases + offs[r], offs[r + 1] - offs[r], ids);  // done for both algorithms (ps_utils.cpp:275-280)
                if (algo == 0) ix->any_full_intersection(ids, colors, tmp);
                else ix->any_threshold_union(bases + offs[r], offs[r + 1] - offs[r], tau, colors);
                if (!colors.empty()) ++m;
                tc += colors.size();
            }
        }
        mapped += m;
        total += tc;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (num_mapped) *num_mapped = mapped;
    if (total_colors) *total_colors = total;
    return sec;
}

// ascii formatter (ps_utils.cpp:55-71): returns a malloc'd buffer
char* fo_format_ascii(const uint64_t* off, const uint32_t* colors, uint64_t n, uint32_t first_id, uint64_t* out_len) {
    std::string s;
    std::vector<uint32_t> v;
    for (uint64_t r = 0; r < n; ++r) {
        v.assign(colors + off[r], colors + off[r + 1]);
        format_ascii(first_id + (uint32_t)r, v, s);
    }
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size() + 1);
    *out_len = s.size();
    return p;
}

char* fo_format_compressed(const uint64_t* off, const uint32_t* colors, uint64_t n, uint32_t first_id, uint32_t num_colors,
                           uint64_t* out_len) {
    std::string s;
    format_compressed(first_id, off, colors, n, num_colors, s);
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    *out_len = s.size();
    return p;
}

// parses a compressed output file back into (ids, CSR); buffers are malloc'd
int fo_parse_compressed(const char* file, uint64_t len, uint64_t* n_out, uint32_t** ids, uint64_t** off, uint32_t** colors) {
    try {
        std::vector<uint32_t> vi, vc;
        std::vector<uint64_t> vo;
        parse_compressed(std::string(file, len), vi, vo, vc);
        *n_out = vi.size();
        *ids = (uint32_t*)malloc(std::max<size_t>(1, vi.size()) * 4);
        *off = (uint64_t*)malloc(vo.size() * 8);
        *colors = (uint32_t*)malloc(std::max<size_t>(1, vc.size()) * 4);
        memcpy(*ids, vi.data(), vi.size() * 4);
        memcpy(*off, vo.data(), vo.size() * 8);
        memcpy(*colors, vc.data(), vc.size() * 4);
        return 0;
    } catch (std::exception& e) { g_err = e.what(); return -1; }
}

// index::kmer_conservation for one read: malloc'd array of (start, num_kmers, color_set_id) triples
int fo_kmer_conservation(void* h, const char* seq, uint64_t len, uint64_t* n_out, uint32_t** triples) {
    auto* ix = static_cast<AnyIndex*>(h);
    std::vector<Index::Triple> t;
    ix->kmer_conservation(seq, len, t);
    *n_out = t.size();
    *triples = (uint32_t*)malloc(std::max<size_t>(1, t.size()) * 12);
    for (size_t i = 0; i < t.size(); ++i) {
        (*triples)[3 * i] = t[i].start_pos_in_query;
        (*triples)[3 * i + 1] = t[i].num_kmers;
        (*triples)[3 * i + 2] = t[i].color_set_id;
    }
    return 0;
}

// index::kmer_matches for one read: positive[num_kmers] (caller buffer, len - k + 1 bytes) and counts[num_colors]
int fo_kmer_matches(void* h, const char* seq, uint64_t len, uint8_t* positive, uint32_t* counts) {
    auto* ix = static_cast<AnyIndex*>(h);
    std::vector<uint8_t> p;
    std::vector<uint32_t> c(ix->colors.num_colors, 0);
    ix->kmer_matches(seq, len, p, c);
    if (!p.empty()) memcpy(positive, p.data(), p.size());
    memcpy(counts, c.data(), c.size() * 4);
    return 0;
}

}  // extern "C"
