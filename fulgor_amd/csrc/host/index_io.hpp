// Index ingestion for the MI355X engine.
//
// (1) The reference's text interchange format, written by `fulgor dump` and read by `fulgor load`
//     (src/index.cpp:59-120 and :122-305; README "Dump output format"):
//        <base>.metadata.txt  k= / num_kmers= / num_colors= / num_unitigs= / num_color_sets=
//        <base>.filenames.txt one filename per colour
//        <base>.unitigs.fa    "> color_set_id=<id>" + sequence, sorted by colour-set id
//        <base>.color_sets.txt "size=<n> c0 c1 ..." one set per line, id = line number
//     load_dump() mirrors index<ColorSets>::load: encode the sets with the hybrid codec, build the
//     k-mer dictionary over the unitigs, derive u2c from the headers.
// (2) An own binary container (".fgidx") holding the engine's HBM-ready arrays for fast reload.
//
// The reference's binary .fur/.mfur/.dfur/.mdfur files embed an SSHash dictionary whose layout is
// defined by a submodule that is not vendored; they are rejected loudly by open_index().
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include "codecs_build.hpp"
#include "dict_build.hpp"
#include "fur_format.hpp"
#include "hybrid_codec.hpp"

namespace fg {

// <base>.color_sets.txt (src/index.cpp:100-118 writes it: "size=<n> c0 c1 ..." one set per line, id = line number) -> the hybrid
// stream. A real salmonella_4546 dump is 4 GB of text holding 0.9 G integers: the file is mapped and cut at line starts into one
// piece per thread; every thread parses its lines and encodes them into a bit stream of its own (the codec has no state across
// sets), and the streams are joined bit by bit in line order — the same stream one encoder would have written front to back.
inline void load_color_sets_text(const std::string& path, uint64_t num_colors, uint64_t num_color_sets, HybridSets& out, unsigned nthreads = 0) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open color sets file");
    struct stat st;
    if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat color sets file"); }
    const uint64_t size = (uint64_t)st.st_size;
    if (size == 0) { ::close(fd); throw std::runtime_error("color sets file is short"); }
    const char* text = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (text == MAP_FAILED) throw std::runtime_error("cannot map color sets file");
    struct Unmap { const char* p; uint64_t n; ~Unmap() { munmap((void*)p, n); } } unmap{text, size};
    madvise((void*)text, size, MADV_SEQUENTIAL);
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = (unsigned)std::min<uint64_t>(nthreads, size / (1u << 20) + 1);
    std::vector<uint64_t> cut(T + 1, size);  // piece t = text bytes [cut[t], cut[t + 1]), both at line starts
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
        const uint64_t at = size / T * t;
        const char* nl = (const char*)memchr(text + at, '\n', size - at);
        cut[t] = nl ? (uint64_t)(nl - text) + 1 : size;
    }
    auto parallel = [&](auto fn) {
        if (T == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back(fn, t);
        for (auto& x : th) x.join();
    };
    // lines in front of every piece (a last line without a line end counts)
    std::vector<uint64_t> first_line(T + 1, 0);
    parallel([&](unsigned t) {
        uint64_t n = 0;
        const char* p = text + cut[t];
        const char* const e = text + cut[t + 1];
        while (p < e) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
            ++n;
            if (!nl) break;
            p = nl + 1;
        }
        first_line[t + 1] = n;
    });
    for (unsigned t = 0; t < T; ++t) first_line[t + 1] += first_line[t];
    if (first_line[T] < num_color_sets) throw std::runtime_error("color sets file is short");
    std::vector<HybridEncoder> enc(T);
    std::vector<std::string> err(T);
    parallel([&](unsigned t) {
        HybridEncoder& en = enc[t];
        en.init(num_colors);
        std::vector<uint32_t> v;
        const char* p = text + cut[t];
        const char* const e = text + cut[t + 1];
        for (uint64_t line = first_line[t]; p < e && line < num_color_sets; ++line) {  // (whatever follows the last set is not looked at)
            const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
            const char* le = nl ? nl : e;
            static const char key[] = "size=";
            const char* c = std::search(p, le, key, key + 5);
            if (c == le) { err[t] = "malformed color set line"; return; }
            c += 5;
            auto number = [&](uint64_t& x) {  // skips blanks, reads decimal digits; false if there are none
                while (c < le && (*c == ' ' || *c == '\t' || *c == '\r')) ++c;
                if (c >= le || *c < '0' || *c > '9') return false;
                x = 0;
                while (c < le && *c >= '0' && *c <= '9') { x = x * 10 + (uint64_t)(*c - '0'); if (x > (1ull << 40)) return false; ++c; }
                return true;
            };
            uint64_t sz = 0;
            if (!number(sz) || sz == 0 || sz > num_colors) { err[t] = "bad color set size"; return; }
            v.resize(sz);
            for (uint64_t j = 0; j < sz; ++j) {
                uint64_t x;
                if (!number(x)) { err[t] = "color set line is short"; return; }
                if (x >= num_colors || (j && x <= v[j - 1])) { err[t] = "color set not increasing"; return; }
                v[j] = (uint32_t)x;
            }
            en.encode(v.data(), v.size());
            p = nl ? nl + 1 : e;
        }
    });
    for (const std::string& e : err) if (!e.empty()) throw std::runtime_error(e);
    // join: piece t's stream goes to bit position base[t] of the whole
    std::vector<uint64_t> base(T + 1, 0), first_set(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) { base[t + 1] = base[t] + enc[t].bw.nbits; first_set[t + 1] = first_set[t] + (enc[t].offsets.size() - 1); }
    if (first_set[T] != num_color_sets) throw std::runtime_error("color sets file is short");
    out.num_colors = (uint32_t)num_colors;
    out.sparse_thr = enc[0].sparse_thr;
    out.dense_thr = enc[0].dense_thr;
    out.nbits = base[T];
    out.offsets.assign(num_color_sets + 1, 0);
    out.bits.assign((out.nbits + 63) / 64 + 4, 0);
    parallel([&](unsigned t) {
        const HybridEncoder& en = enc[t];
        for (size_t i = 0; i + 1 < en.offsets.size(); ++i) out.offsets[first_set[t] + i] = base[t] + en.offsets[i];
        const uint64_t nw = (en.bw.nbits + 63) / 64;
        if (!nw) return;
        const uint64_t d0 = base[t] >> 6, sh = base[t] & 63;
        uint64_t* dst = out.bits.data();
        // the first and the last word this piece touches may be shared with its neighbours: OR-ed in atomically; the others are its own
        const uint64_t last = (base[t] + en.bw.nbits - 1) >> 6;
        auto put = [&](uint64_t w, uint64_t bits) {
            if (!bits) return;
            if (w == d0 || w == last) __atomic_fetch_or(&dst[w], bits, __ATOMIC_RELAXED);
            else dst[w] |= bits;
        };
        for (uint64_t i = 0; i < nw; ++i) {
            const uint64_t x = en.bw.words[i];
            put(d0 + i, x << sh);
            if (sh) put(d0 + i + 1, x >> (64 - sh));
        }
    });
    out.offsets[num_color_sets] = out.nbits;
}

inline void load_dump(const std::string& base, HostIndex& idx, uint32_t m = 0, unsigned nthreads = 0, bool host_table = true) {
    uint64_t k = 0, num_kmers = 0, num_colors = 0, num_unitigs = 0, num_color_sets = 0;
    {
        std::ifstream in(base + ".metadata.txt");
        if (!in.is_open()) throw std::runtime_error("cannot open metadata file");
        std::string line;
        while (std::getline(in, line)) {
            size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            std::string key = line.substr(0, eq);
            uint64_t v = std::strtoull(line.c_str() + eq + 1, nullptr, 10);
            if (key == "k") k = v;
            else if (key == "num_kmers") num_kmers = v;
            else if (key == "num_colors") num_colors = v;
            else if (key == "num_unitigs") num_unitigs = v;
            else if (key == "num_color_sets") num_color_sets = v;
        }
    }
    if (k == 0 || num_colors == 0 || num_unitigs == 0 || num_color_sets == 0)
        throw std::runtime_error("incomplete metadata file");
    {
        std::ifstream in(base + ".filenames.txt");
        if (!in.is_open()) throw std::runtime_error("cannot open filenames file");
        idx.filenames.clear();
        std::string f;  // one filename per line (src/index.cpp:76); a path may contain blanks
        for (uint64_t i = 0; i < num_colors && std::getline(in, f); ++i) {
            while (!f.empty() && (f.back() == '\r' || f.back() == '\n')) f.pop_back();
            idx.filenames.push_back(f);
        }
        if (idx.filenames.size() != num_colors) throw std::runtime_error("filenames file is short");
    }
    // colour sets
    load_color_sets_text(base + ".color_sets.txt", num_colors, num_color_sets, idx.hybrid, nthreads);
    hybrid_build_blocks(idx.hybrid, nthreads);
    // unitigs
    std::string bases;
    std::vector<uint64_t> off(1, 0);
    std::vector<uint32_t> csid;
    {
        std::ifstream in(base + ".unitigs.fa");
        if (!in.is_open()) throw std::runtime_error("cannot open unitigs file");
        std::string header, seq;
        uint64_t prev = 0;
        for (uint64_t i = 0; i < num_unitigs; ++i) {
            if (!std::getline(in, header) || !std::getline(in, seq)) throw std::runtime_error("unitigs file is short");
            size_t p = header.find("color_set_id=");
            if (p == std::string::npos) throw std::runtime_error("malformed unitig header");
            uint64_t id = std::strtoull(header.c_str() + p + 13, nullptr, 10);
            if (id >= num_color_sets || id < prev) throw std::runtime_error("unitigs are not sorted by color_set_id");
            prev = id;
            while (!seq.empty() && (seq.back() == '\r' || seq.back() == ' ')) seq.pop_back();
            bases += seq;
            off.push_back(bases.size());
            csid.push_back((uint32_t)id);
        }
    }
    if (m == 0) m = k >= 15 ? (uint32_t)k - 14 : 1;  // k=31 -> m=17: 15 windows, contexts of 45 bases
    idx.type = IDX_HYBRID;
    build_dict(idx.dict, (uint32_t)k, m, bases.data(), bases.size(), off, csid, nthreads, host_table);
    if (num_kmers && idx.dict.num_kmers != num_kmers) throw std::runtime_error("num_kmers does not match metadata");
}

// ---- own binary container ------------------------------------------------------------------------
namespace detail {
static const char FGIDX_MAGIC[8] = {'F', 'G', 'I', 'D', 'X', '0', '0', '9'};  // 009: minimizer order of round 4 (the records are cut by it); 008: contexts of up to 45 bases (m = 17 at k = 31), smax in w3, 27-bit colour-set ids
template <typename T>
void wr(std::ofstream& o, const T& v) { o.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T>
void rd(std::ifstream& i, T& v) {
    i.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!i) throw std::runtime_error("truncated index file");
}
template <typename T>
void wrv(std::ofstream& o, const std::vector<T>& v) {
    uint64_t n = v.size();
    wr(o, n);
    if (n) o.write(reinterpret_cast<const char*>(v.data()), n * sizeof(T));
}
template <typename T>
void rdv(std::ifstream& i, std::vector<T>& v) {
    uint64_t n;
    rd(i, n);
    if (n > (1ULL << 40)) throw std::runtime_error("corrupt index file");
    v.resize(n);
    if (n) i.read(reinterpret_cast<char*>(v.data()), n * sizeof(T));
    if (!i) throw std::runtime_error("truncated index file");
}
}  // namespace detail

inline void save_binary(const HostIndex& idx, const std::string& path) {
    using namespace detail;
    std::ofstream o(path, std::ios::binary);
    if (!o.is_open()) throw std::runtime_error("cannot open output index file");
    o.write(FGIDX_MAGIC, 8);
    int32_t type = idx.type;
    wr(o, type);
    const Dict& d = idx.dict;
    wr(o, d.k); wr(o, d.m); wr(o, d.num_kmers); wr(o, d.total_bases); wr(o, d.seed);
    wrv(o, d.strings); wrv(o, d.records);  // the bucket table is rebuilt from the records at load
    wrv(o, d.unitig_off); wrv(o, d.unitig_csid);
    const HybridSets& h = idx.hybrid;
    wr(o, h.num_colors); wr(o, h.sparse_thr); wr(o, h.dense_thr); wr(o, h.nbits);
    wrv(o, h.offsets); wrv(o, h.bits);  // restart samples are an acceleration structure: rebuilt at load
    if (idx.type != IDX_HYBRID) {
        const GenericSets& g = idx.generic;
        wr(o, g.num_colors); wr(o, g.partition_size); wr(o, g.cluster_size); wr(o, g.num_partitions);
        wr(o, g.num_partial_sets); wr(o, g.num_clusters); wr(o, g.nbits);
        wrv(o, g.bits); wrv(o, g.ops); wrv(o, g.set_ops_off); wrv(o, g.set_ops); wrv(o, g.set_bytes);
    }
    uint64_t nf = idx.filenames.size();
    wr(o, nf);
    for (auto& f : idx.filenames) {
        std::vector<char> c(f.begin(), f.end());
        wrv(o, c);
    }
    if (!o) throw std::runtime_error("write error on index file");
}

// FULGOR_VERBOSE_LOAD=1: the stages of opening an index, with their times, on stderr
struct LoadClock {
    bool on = getenv("FULGOR_VERBOSE_LOAD") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (on) fprintf(stderr, "  open: %-34s %7.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};

// the container mapped: sections are found by walking their length words, their bytes copied where they are needed
struct MappedFile {
    const char* p = nullptr;
    uint64_t size = 0, at = 0;
    explicit MappedFile(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open index file");
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size == 0) { ::close(fd); throw std::runtime_error("cannot open index file"); }
        size = (uint64_t)st.st_size;
        p = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) { p = nullptr; throw std::runtime_error("cannot map index file"); }
    }
    ~MappedFile() { if (p) munmap((void*)p, size); }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    template <typename T>
    void rd(T& v) {
        if (at + sizeof(T) > size) throw std::runtime_error("truncated index file");
        memcpy(&v, p + at, sizeof(T));
        at += sizeof(T);
    }
    // a vector section: where its elements lie and how many (nothing is copied yet)
    template <typename T>
    struct View { const T* data = nullptr; uint64_t n = 0; void into(std::vector<T>& v) const { v.assign(data, data + n); } };
    template <typename T>
    View<T> view() {
        uint64_t n;
        rd(n);
        if (n > (1ULL << 40)) throw std::runtime_error("corrupt index file");
        if (at + n * sizeof(T) > size) throw std::runtime_error("truncated index file");
        View<T> v{reinterpret_cast<const T*>(p + at), n};
        at += n * sizeof(T);
        return v;
    }
};

inline void load_binary(const std::string& path, HostIndex& idx, bool host_table = true) {
    using namespace detail;
    LoadClock clk;
    MappedFile f(path);
    char magic[8] = {0};
    if (f.size >= 8) { memcpy(magic, f.p, 8); f.at = 8; }
    if (f.size >= 8 && std::memcmp(magic, FGIDX_MAGIC, 5) == 0 && std::memcmp(magic, FGIDX_MAGIC, 8) != 0)
        throw std::runtime_error("this .fgidx file has container version " + std::string(magic + 5, 3) + ", this build reads " + std::string(FGIDX_MAGIC + 5, 3) +
                                 ": rebuild the index from its dump files (the super-k-mer records depend on the build's minimizer order)");
    if (f.size < 8 || std::memcmp(magic, FGIDX_MAGIC, 8) != 0) throw std::runtime_error("not an .fgidx file (bad magic)");
    int32_t type;
    f.rd(type);
    idx.type = type;
    Dict& d = idx.dict;
    f.rd(d.k); f.rd(d.m); f.rd(d.num_kmers); f.rd(d.total_bases); f.rd(d.seed);
    const auto v_strings = f.view<uint64_t>();
    const auto v_records = f.view<uint32_t>();
    const auto v_unitig_off = f.view<uint64_t>();
    const auto v_unitig_csid = f.view<uint32_t>();
    HybridSets& h = idx.hybrid;
    f.rd(h.num_colors); f.rd(h.sparse_thr); f.rd(h.dense_thr); f.rd(h.nbits);
    const auto v_offsets = f.view<uint64_t>();
    const auto v_bits = f.view<uint64_t>();
    auto bad = [](const char* what) { throw std::runtime_error(std::string("corrupt index file: ") + what); };
    // The two halves of the container do not depend on each other: the dictionary's sections are copied out and checked on a thread of
    // their own while this one takes the colour sets and cuts their packed blocks (a truncated, stale or corrupt container must fail
    // here, not index out of bounds later).
    std::string dict_error;
    std::thread dict_thread([&] {
        try {
            v_strings.into(d.strings);
            v_records.into(d.records);
            v_unitig_off.into(d.unitig_off);
            v_unitig_csid.into(d.unitig_csid);
            try { check_dict_params(d.k, d.m); } catch (std::exception&) { bad("k / m"); }
            if (d.strings.size() < (d.total_bases + 31) / 32 + 2) bad("unitig strings shorter than total_bases");
            if (d.records.size() % REC_WORDS) bad("record array");
            if (d.unitig_off.size() != d.unitig_csid.size() + 1 || d.unitig_off.empty() || d.unitig_off[0] != 0 ||
                d.unitig_off.back() != d.total_bases) bad("unitig table");
            for (size_t u = 0; u + 1 < d.unitig_off.size(); ++u)
                if (d.unitig_off[u + 1] < d.unitig_off[u] + d.k) bad("unitig offsets");
            const uint64_t nsets = v_offsets.n ? v_offsets.n - 1 : 0;
            for (uint32_t c : d.unitig_csid)
                if (c >= nsets) bad("unitig colour-set id");
            for (uint64_t r = 0; r < d.num_records(); ++r)
                if ((d.records[r * REC_WORDS + 3] & REC_MAX_CSID) >= nsets) bad("record colour-set id");
            if (host_table) build_dict_table(d);
            else dict_table_geometry(d);  // (the table itself is built on the device: hip/dict_build.hip.h)
        } catch (std::exception& e) {
            dict_error = e.what();
        }
    });
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{dict_thread};
    v_offsets.into(h.offsets);
    v_bits.into(h.bits);
    if (h.offsets.empty() || h.offsets[0] > h.offsets.back() || h.offsets.back() != h.nbits) bad("colour-set offsets");
    for (size_t s = 0; s + 1 < h.offsets.size(); ++s)
        if (h.offsets[s + 1] <= h.offsets[s]) bad("colour-set offsets are not increasing");
    if (h.bits.size() * 64 < h.nbits) bad("colour-set stream shorter than nbits");
    if (h.num_colors == 0 || h.num_colors > BLK_MAX_COLORS) bad("num_colors");
    clk.lap("colour sets read and checked");
    h.bits.resize((h.nbits + 63) / 64 + 4, 0);  // the device reads up to 256 bits past a bitmap list
    hybrid_build_blocks(h);
    clk.lap("packed blocks of the gap lists");
    dict_thread.join();
    if (!dict_error.empty()) throw std::runtime_error(dict_error);
    clk.lap(host_table ? "dictionary read, checked, table built (beside the colour sets)" : "dictionary read and checked (beside the colour sets)");
    if (idx.type != IDX_HYBRID) {
        GenericSets& g = idx.generic;
        g.type = idx.type;
        f.rd(g.num_colors); f.rd(g.partition_size); f.rd(g.cluster_size); f.rd(g.num_partitions);
        f.rd(g.num_partial_sets); f.rd(g.num_clusters); f.rd(g.nbits);
        f.view<uint64_t>().into(g.bits);
        f.view<SetOp>().into(g.ops);
        f.view<uint64_t>().into(g.set_ops_off);
        f.view<uint32_t>().into(g.set_ops);
        f.view<uint32_t>().into(g.set_bytes);
        {
            if (g.num_colors != h.num_colors) bad("codec colour count");
            if (g.bits.size() * 64 < g.nbits) bad("codec arena shorter than nbits");
            if (g.set_ops_off.size() != h.offsets.size() || g.set_ops_off[0] != 0 || g.set_ops_off.back() != g.set_ops.size())
                bad("codec op lists");
            for (size_t s = 0; s + 1 < g.set_ops_off.size(); ++s)
                if (g.set_ops_off[s + 1] < g.set_ops_off[s]) bad("codec op offsets are not monotone");
            for (uint32_t o : g.set_ops)
                if (o >= g.ops.size()) bad("codec op index");
            for (const SetOp& o : g.ops)
                if (o.body > g.nbits || (uint64_t)o.base + o.np > g.num_colors || o.kind > OP_XOR_GAPS) bad("codec op");
            if (g.set_bytes.size() + 1 != g.set_ops_off.size()) bad("codec byte table");
        }
        build_generic_device(g);
    }
    uint64_t nf;
    f.rd(nf);
    idx.filenames.clear();
    for (uint64_t i = 0; i < nf; ++i) {
        const auto c = f.view<char>();
        idx.filenames.emplace_back(c.data, c.data + c.n);
    }
}

// A file in the reference's binary layout (fur_format.hpp: Fulgor-owned sections from the reference, primitive layouts from
// memory of upstream and NOT validated on a real file, k2u section = the engine's own block): what fgpu_save writes under
// a .fur / .mfur / .dfur / .mdfur name. A file written by the reference itself stops at its SSHash section.
inline void load_fur(const std::string& path, HostIndex& idx, unsigned nthreads = 0, bool host_table = true) {
    uint32_t psize = 0, csize = 0;
    fur::read_fur(path, idx, psize, csize);
    Dict& d = idx.dict;
    check_dict_params(d.k, d.m);
    if (d.unitig_off.empty() || d.unitig_off.back() != d.total_bases || d.strings.size() < (d.total_bases + 31) / 32)
        throw std::runtime_error("corrupt index file (unitigs)");
    std::string bases(d.total_bases, 'A');
    for (uint64_t i = 0; i < d.total_bases; ++i) bases[i] = "ACGT"[detail::string_base(d.strings, i)];
    const std::vector<uint64_t> off = d.unitig_off;
    const std::vector<uint32_t> csid = d.unitig_csid;
    const uint64_t num_kmers = d.num_kmers;
    build_dict(d, d.k, d.m, bases.data(), bases.size(), off, csid, nthreads, host_table);
    if (d.num_kmers != num_kmers) throw std::runtime_error("corrupt index file (num_kmers)");
    hybrid_build_blocks(idx.hybrid, nthreads);
    const int type = idx.type;
    if (type != IDX_HYBRID) {
        convert_sets(idx.hybrid, type, psize ? psize : idx.hybrid.num_colors, csize ? csize : 1, idx.generic);
        idx.generic.type = type;
    }
}

inline void save_fur(const HostIndex& idx, const std::string& path) {
    const bool have = idx.type != IDX_HYBRID;
    fur::write_fur(idx, path, have && idx.generic.partition_size ? idx.generic.partition_size : 160,
                   have && idx.generic.cluster_size ? idx.generic.cluster_size : 16);
}

// path dispatch: ".fgidx" container, a binary index in the reference's layout, or a dump basename
// host_table = false (a handle on a device): the dictionary's bucket table is not built on the host
inline void open_index(const std::string& path, HostIndex& idx, unsigned nthreads = 0, bool host_table = true) {
    if (ends_with(path, ".fgidx")) { load_binary(path, idx, host_table); return; }
    // suffix sniffing order of the reference CLI: mdfur, mfur, dfur, fur (tools/pseudoalign.cpp:294-306)
    if (ends_with(path, "fur")) { load_fur(path, idx, nthreads, host_table); return; }
    load_dump(path, idx, 0, nthreads, host_table);
}

}  // namespace fg
