// FASTA / FASTQ (plain or gzip) ingestion for the worker loop (src/ps_utils.cpp:245-305: the reference feeds
// its workers from FQFeeder's parser threads; read id = position in the file). Sequences come out exactly as
// kseq delivers them to the reference: header line skipped, sequence lines concatenated without line ends, the
// '+' line and as many quality characters as there are bases skipped.
//
// Two sources behind one interface, both delivering chunks of reads in file order while the caller's previous
// batches are on the GPU:
//   * plain files are mapped and cut into byte ranges that a pool of threads parses concurrently (a range
//     starts at the first record boundary at or behind its first byte: '>' at a line start, or '@' at a line
//     start whose line after next starts with '+'), and the chunks are handed out in range order;
//   * gzip streams cannot be entered in the middle: one thread inflates and parses;
//   * block-compressed gzip (BGZF, what bgzip / htslib write: a series of gzip members of at most 64 KB, each giving its
//     compressed size in an extra field) can: the members are inflated by the pool of threads into an anonymous buffer as
//     the ranges that need them come up, and the ranges are parsed as for a plain file. Any gzip reader, the reference's
//     included, reads such a file as one stream.
#pragma once
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace fg {

// ---- host buffers of the reader: pinned when the engine is there --------------------------------------------------------
// Parsed reads are written ONCE, by the thread that parses them, into the buffer the GPU's copy engine reads from (H2D copies out
// of pinned host memory run at PCIe speed and asynchronously; out of pageable memory the runtime stages them through a bounce
// buffer). The engine installs the allocator (hipHostMalloc) when it opens an index on a device; host-only tools, the CPU tests
// and the sanitizer builds keep malloc. Pinning costs tens of microseconds per megabyte, so released buffers wait in a
// process-wide pool for the next chunk, reader or run; what is still pooled at exit goes with the process.
struct HostAllocHooks {
    void* (*alloc)(size_t bytes, bool* pinned) = nullptr;  // nullptr: malloc
    void (*release)(void* p, bool pinned) = nullptr;
};
inline HostAllocHooks& host_alloc_hooks() { static HostAllocHooks h; return h; }

class SlabPool {
public:
    static SlabPool& get() { static SlabPool* p = new SlabPool(); return *p; }  // (never destroyed: no calls into HIP at exit)
    // a buffer of at least `want` bytes: the smallest pooled one that fits (and is not wastefully large), else a new one
    void* take(size_t want, size_t& got, bool& pinned) {
        // sizes in coarse steps (64 KB, 128 KB, ... 1 MB, then whole megabytes): the chunks of a file ask for almost, not exactly,
        // the same size range after range, and a slab that is a few bytes short would be pinned anew (0.16 ms per megabyte, and
        // the pinning stalls the copies in flight)
        if (want <= (1u << 20)) { size_t r = 1u << 16; while (r < want) r <<= 1; want = r; }
        else want = (want + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        {
            std::lock_guard<std::mutex> g(mu_);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].bytes >= want && free_[i].bytes <= 4 * want + (1u << 20) && (best == free_.size() || free_[i].bytes < free_[best].bytes)) best = i;
            if (best < free_.size()) {
                const Slab sl = free_[best];
                free_[best] = free_.back();
                free_.pop_back();
                held_ -= sl.bytes;
                got = sl.bytes;
                pinned = sl.pinned;
                return sl.p;
            }
        }
        const HostAllocHooks& h = host_alloc_hooks();
        void* p = nullptr;
        pinned = false;
        fresh_ += 1;
        fresh_bytes_ += want;
        if (h.alloc) p = h.alloc(want, &pinned);
        if (!p) { p = malloc(want); pinned = false; }
        if (!p) throw std::bad_alloc();
        got = want;
        return p;
    }
    void give(void* p, size_t bytes, bool pinned) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu_);
            if (held_ + bytes <= MAX_BYTES && free_.size() < MAX_SLABS) {
                free_.push_back(Slab{p, bytes, pinned});
                held_ += bytes;
                return;
            }
        }
        const HostAllocHooks& h = host_alloc_hooks();
        if (pinned && h.release) h.release(p, true); else free(p);
    }

    // buffers that had to be allocated (pinned) because the pool had none that fit: count and bytes since the process began
    uint64_t fresh_allocs() const { return fresh_.load(); }
    uint64_t fresh_bytes() const { return fresh_bytes_.load(); }

private:
    struct Slab { void* p; size_t bytes; bool pinned; };
    static constexpr size_t MAX_BYTES = (size_t)3 << 30, MAX_SLABS = 1024;
    std::atomic<uint64_t> fresh_{0}, fresh_bytes_{0};
    std::mutex mu_;
    std::vector<Slab> free_;
    size_t held_ = 0;
};

// grow-only array of PODs in a pooled (pinned) buffer; move-only
template <typename T>
class HostVec {
public:
    HostVec() {}
    HostVec(HostVec&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), bytes_(o.bytes_), pinned_(o.pinned_) { o.p_ = nullptr; o.n_ = o.cap_ = o.bytes_ = 0; }
    HostVec& operator=(HostVec&& o) noexcept {
        if (this != &o) {
            release();
            p_ = o.p_; n_ = o.n_; cap_ = o.cap_; bytes_ = o.bytes_; pinned_ = o.pinned_;
            o.p_ = nullptr; o.n_ = o.cap_ = o.bytes_ = 0;
        }
        return *this;
    }
    HostVec(const HostVec&) = delete;
    HostVec& operator=(const HostVec&) = delete;
    ~HostVec() { release(); }
    void release() {
        if (p_) SlabPool::get().give(p_, bytes_, pinned_);
        p_ = nullptr;
        n_ = cap_ = bytes_ = 0;
    }
    T* data() { return p_; }
    const T* data() const { return p_; }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    bool pinned() const { return pinned_; }
    void clear() { n_ = 0; }
    void set_size(size_t n) { n_ = n; }  // (n <= capacity)
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
    T& back() { return p_[n_ - 1]; }
    void reserve(size_t want) {
        if (want <= cap_) return;
        const size_t ask = std::max(want + want / 4, (size_t)(1u << 16) / sizeof(T));
        size_t got = 0;
        bool pin = false;
        T* q = (T*)SlabPool::get().take(ask * sizeof(T) + 1024, got, pin);  // + slack: the lookup kernel's over-read of padded reads is copied along
        if (n_) memcpy(q, p_, n_ * sizeof(T));
        const size_t keep = n_;
        release();
        p_ = q;
        n_ = keep;
        bytes_ = got;
        cap_ = (got - 1024) / sizeof(T);
        pinned_ = pin;
    }
    // room for `want` elements and no more than the pool's size step above it (reserve() adds a quarter for arrays that grow as they
    // are filled; a caller that knows an upper bound asks for exactly that: pinned memory costs 0.16 ms per megabyte the first time)
    void reserve_exact(size_t want) {
        if (want <= cap_) return;
        size_t got = 0;
        bool pin = false;
        T* q = (T*)SlabPool::get().take(want * sizeof(T) + 1024, got, pin);
        if (n_) memcpy(q, p_, n_ * sizeof(T));
        const size_t keep = n_;
        release();
        p_ = q;
        n_ = keep;
        bytes_ = got;
        cap_ = (got - 1024) / sizeof(T);
        pinned_ = pin;
    }
    void push_back(const T& v) {
        if (n_ == cap_) reserve(n_ + 1);
        p_[n_++] = v;
    }
    void append(const T* s, size_t n) {
        if (n_ + n > cap_) reserve(n_ + n);
        memcpy(p_ + n_, s, n * sizeof(T));
        n_ += n;
    }
    void resize(size_t n) {  // (new elements are not initialised)
        reserve(n);
        n_ = n;
    }
    void assign(size_t n, const T& v) {
        reserve(n);
        for (size_t i = 0; i < n; ++i) p_[i] = v;
        n_ = n;
    }

private:
    T* p_ = nullptr;
    size_t n_ = 0, cap_ = 0, bytes_ = 0;
    bool pinned_ = false;
};

struct FastxChunk {
    HostVec<char> bases;
    HostVec<uint64_t> offs;   // offs[0] = 0: positions inside `bases`
    std::vector<char> names;  // record names (header up to the first blank, as kseq's name), concatenated; kept only when asked for
    std::vector<uint64_t> name_offs{0};
    uint64_t max_len = 0;     // longest read of the chunk
    bool want_names = true;
    FastxChunk() {}  // (no buffer yet: an empty chunk is made and moved around all the time)
    FastxChunk(FastxChunk&&) = default;
    FastxChunk& operator=(FastxChunk&&) = default;
    uint64_t reads() const { return offs.size() ? offs.size() - 1 : 0; }
    // the parser's sink
    bool any() const { return reads() > 0; }
    void expect(uint64_t text_bytes, bool fastq) {  // (a FASTQ record spends as many bytes on qualities as on bases)
        bases.reserve_exact(fastq ? text_bytes / 2 + 64 : text_bytes);  // (upper bounds: they only grow for records of fewer than 250 bytes)
        offs.reserve_exact(text_bytes / 256 + 16);
    }
    // the two pool requests expect() makes for a range of `text_bytes` (what a caller pins ahead of a run: fgpu_prepare_host)
    static void slab_requests(uint64_t text_bytes, bool fastq, size_t& bases_bytes, size_t& offs_bytes) {
        bases_bytes = (size_t)(fastq ? text_bytes / 2 + 64 : text_bytes) + 1024;
        offs_bytes = (size_t)(text_bytes / 256 + 16) * sizeof(uint64_t) + 1024;
    }
    void on_name(const char* s, size_t n) {
        if (!want_names) return;
        size_t e = 0;
        while (e < n && s[e] != ' ' && s[e] != '\t') ++e;
        names.insert(names.end(), s, s + e);
        name_offs.push_back(names.size());
    }
    void on_seq(const char* s, size_t n) { bases.append(s, n); }
    void on_record(uint64_t len) {
        if (offs.size() == 0) offs.push_back(0);
        offs.push_back(bases.size());
        if (len > max_len) max_len = len;
    }
    void clear() {
        bases.clear();
        offs.clear();
        names.clear();
        name_offs.assign(1, 0);
        max_len = 0;
    }
};

// kseq's record grammar over a stream of lines: next(s, n) yields the next line without its terminator.
// odd (optional): set when the lines did not read as whole records — bit 0: text between a record and the next header, bit 1: a
// FASTQ record whose quality ends before its sequence does. A byte range of a file that was cut at true record boundaries never shows either;
// a range cut inside a wrapped FASTQ record does.
// Sink: any() (a record has been seen), on_name(s, n) (the header line without its first character), on_seq(s, n),
// on_record(len). FastxChunk collects the reads; CountSink only counts them (the same grammar, nothing copied).
template <typename NextLine, typename Sink, typename Emit>
void parse_fastx_records(NextLine&& next, Sink& c, Emit&& emit, unsigned* odd = nullptr) {
    const char* s;
    size_t n;
    bool have = next(s, n);
    bool any = c.any();
    while (have) {
        if (n == 0 || (s[0] != '>' && s[0] != '@')) {  // stray text before a header
            if (n && odd && any) *odd |= 1u;  // (in front of the first header: skipped, as kseq does)
            have = next(s, n);
            continue;
        }
        any = true;
        c.on_name(s + 1, n - 1);
        // sequence lines up to the next header or the '+' separator
        uint64_t len = 0;
        while ((have = next(s, n)) && !(n && (s[0] == '>' || s[0] == '@' || s[0] == '+'))) {
            c.on_seq(s, n);
            len += n;
        }
        c.on_record(len);
        if (have && s[0] == '+') {  // quality: as many characters as bases (may itself start with '@')
            uint64_t q = 0;
            while (q < len && (have = next(s, n))) q += n;
            if (q < len && odd) *odd |= 2u;
            have = next(s, n);
        }
        if (!emit(c)) return;
    }
}

// Four-line FASTQ records of the text M from `pos` on, up to `hi` (a record boundary or the end of the text): '@' header, one
// sequence line, a '+' line, a quality line of the sequence line's length.
// Stops in front of the first record that does not have this shape (wrapped lines, an empty sequence, a '>' header, a quality
// of another length, a text that ends early) and leaves it to parse_fastx_records, whose result for the records taken here is
// the same. last = false: `hi` is only where the bytes at hand end (a piece of a range), a record that touches it is not taken.
// Returns the position of the first record not taken.
template <typename Sink>
uint64_t parse_fastq4(const char* M, uint64_t pos, uint64_t hi, Sink& c, bool last = true) {
    const char* const fin = M + hi;
    const char* p = M + pos;
    while (p < fin && *p == '@') {
        const char* l1 = (const char*)memchr(p, '\n', (size_t)(fin - p));  // end of the header
        if (!l1 || l1 + 1 >= fin) break;
        const char* sq = l1 + 1;
        if (*sq == '>' || *sq == '@' || *sq == '+' || *sq == '\n' || *sq == '\r') break;
        const char* l2 = (const char*)memchr(sq, '\n', (size_t)(fin - sq));  // end of the sequence
        if (!l2 || l2 + 2 >= fin || l2[1] != '+') break;
        const char* l3 = l2[2] == '\n' ? l2 + 2 : (const char*)memchr(l2 + 2, '\n', (size_t)(fin - l2 - 2));  // end of the '+' line
        if (!l3) break;
        const size_t raw = (size_t)(l2 - sq);
        const char* e = l3 + 1 + raw;  // where the quality line ends if it is as long as the sequence line
        if (e > fin || (e < fin ? *e != '\n' : !last)) break;
        if (memchr(l3 + 1, '\n', raw)) break;  // (a shorter quality line: the grammar reads on into the next lines)
        size_t hn = (size_t)(l1 - p) - 1, len = raw;
        if (hn && p[hn] == '\r') --hn;
        if (sq[len - 1] == '\r' && --len == 0) break;
        c.on_name(p + 1, hn);
        c.on_seq(sq, len);
        c.on_record(len);
        p = e < fin ? e + 1 : fin;
    }
    return (uint64_t)(p - M);
}

struct CountSink {
    uint64_t n = 0;
    bool any() const { return n > 0; }
    void expect(uint64_t, bool) {}
    void on_name(const char*, size_t) {}
    void on_seq(const char*, size_t) {}
    void on_record(uint64_t) { ++n; }
};

// What the head of a FASTA / FASTQ text says about cutting it into byte ranges. kind: the first character of the first line
// that starts with '>' or '@' (kseq skips whatever stands in front of it), 0 if there is none. Returns false when the text
// cannot be cut at guessed boundaries: FASTQ records that are not exactly four lines (wrapped sequence / quality lines —
// there a quality line that begins with '@' is indistinguishable from a header), or no header at all. Such a file is
// parsed as one stream with the full grammar, as the reference does (kseq).
inline bool fastx_head_rangeable(const char* p, size_t n, char& kind) {
    kind = 0;
    size_t at = 0;
    auto line = [&](const char*& s, size_t& len) -> bool {  // complete lines only
        if (at >= n) return false;
        const char* nl = (const char*)memchr(p + at, '\n', n - at);
        if (!nl) return false;
        s = p + at;
        len = (size_t)(nl - s);
        at += len + 1;
        if (len && s[len - 1] == '\r') --len;
        return true;
    };
    const char* s;
    size_t len;
    bool have;
    while ((have = line(s, len)) && !(len && (s[0] == '>' || s[0] == '@'))) {}
    if (!have) {  // no complete header line in the head: a header without a line end still tells the kind
        if (at < n && (p[at] == '>' || p[at] == '@')) kind = p[at];
        return kind == '>';
    }
    kind = s[0];
    if (kind == '>') return true;
    for (int rec = 0; rec < 4096; ++rec) {  // '@' header in s: sequence, '+', quality of the same length, then the next header
        size_t ls, lq;
        const char* t;
        if (!line(t, ls)) return true;
        if (ls && t[0] == '+') return false;  // (an empty sequence: leave it to the stream parser)
        if (!line(t, lq)) return true;
        if (!(lq && t[0] == '+')) return false;  // a second sequence line
        if (!line(t, lq)) return true;
        if (lq != ls) return false;
        if (!line(s, len)) return true;
        if (!(len && s[0] == '@')) return false;
    }
    return true;
}
inline bool fastx_file_rangeable(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<char> head(1u << 20);
    const size_t got = fread(head.data(), 1, head.size(), f);
    fclose(f);
    char kind;
    return got == 0 || fastx_head_rangeable(head.data(), got, kind);
}

// Where the parser threads run: left to the scheduler. They can be confined (FULGOR_READER_AFFINITY = file | device | <node>) to the
// first hardware thread of every core of one NUMA node — the node that holds the file's cached pages (fastx_file_node samples a few),
// the node of the GPU, or a given one. Measured on two-socket, 256-thread hosts by taking turns inside one process
// (profiles/e2e_ab.py, profiles/r5/e2e_breakdown_r5.txt): scheduler 77 ms median / 54 ms best for ten million reads, file's node
// 82 / 64, device's node 86 / 64 — none of the confinements helps. (The whole PROCESS started under `taskset` on the cores of one
// node, file written from there too, runs 50 to 57 ms every time: what that buys is not the parser threads' placement.)
inline std::atomic<int>& fastx_preferred_node() { static std::atomic<int> n{-1}; return n; }
inline std::vector<int> parse_cpu_list(const std::string& text) {
    std::vector<int> out;
    size_t i = 0;
    while (i < text.size()) {
        while (i < text.size() && !isdigit((unsigned char)text[i])) ++i;
        if (i >= text.size()) break;
        int a = 0;
        while (i < text.size() && isdigit((unsigned char)text[i])) a = a * 10 + (text[i++] - '0');
        int b = a;
        if (i < text.size() && text[i] == '-') {
            ++i;
            b = 0;
            while (i < text.size() && isdigit((unsigned char)text[i])) b = b * 10 + (text[i++] - '0');
        }
        for (int c = a; c <= b && out.size() < 4096; ++c) out.push_back(c);
    }
    return out;
}
inline std::string read_small_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return std::string();
    char buf[4096];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    return std::string(buf, n);
}
// first hardware thread of every core of `node` that this process may run on (empty: unknown node, or fewer than 4 CPUs)
inline std::vector<int> fastx_node_cpus(int node) {
    std::vector<int> out;
    if (node < 0) return out;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
    for (int c : parse_cpu_list(read_small_file("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist"))) {
        if (c >= CPU_SETSIZE || !CPU_ISSET(c, &allowed)) continue;
        const std::vector<int> sib = parse_cpu_list(read_small_file("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/thread_siblings_list"));
        if (!sib.empty() && *std::min_element(sib.begin(), sib.end()) != c) continue;  // one hardware thread per core
        out.push_back(c);
    }
    if (out.size() < 4) out.clear();
    return out;
}
// NUMA node that holds (most of) the cached pages of bytes [begin, end) of an open file: a few pages across the range are mapped,
// touched and asked for their node (move_pages with no target nodes). -1: unknown (not Linux NUMA, nothing to sample).
inline int fastx_file_node(int fd, uint64_t begin, uint64_t end) {
    if (end <= begin) return -1;
    const long page = sysconf(_SC_PAGESIZE);
    constexpr int SAMPLES = 16;
    int votes[64] = {0};
    for (int i = 0; i < SAMPLES; ++i) {
        const uint64_t at = (begin + (end - begin) / SAMPLES * i) / (uint64_t)page * (uint64_t)page;
        void* m = mmap(nullptr, (size_t)page, PROT_READ, MAP_PRIVATE, fd, (off_t)at);
        if (m == MAP_FAILED) continue;
        volatile char touch = *(volatile const char*)m;
        (void)touch;
        void* pages[1] = {m};
        int status[1] = {-1};
        if (syscall(SYS_move_pages, 0, 1ul, pages, nullptr, status, 0) == 0 && status[0] >= 0 && status[0] < 64) votes[status[0]]++;
        munmap(m, (size_t)page);
    }
    int best = -1;
    for (int n = 0; n < 64; ++n) if (votes[n] > (best < 0 ? 0 : votes[best])) best = n;
    return best;
}
// confine the calling thread to `cpus` when they can seat `threads` threads
inline void fastx_confine_thread(const std::vector<int>& cpus, unsigned threads) {
    if (cpus.size() < threads) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

// a text whose records cannot be found from the middle (FASTQ with wrapped lines): the caller falls back to one stream
struct NotRangeable : std::runtime_error {
    using std::runtime_error::runtime_error;
};

class FastxSource {
public:
    virtual ~FastxSource() {}
    virtual bool pop(FastxChunk& c) = 0;  // next chunk in file order; false at end of file
    virtual void recycle(FastxChunk&& c) = 0;
    // number of records of the whole source by a walk over the record grammar that copies nothing, without consuming the
    // source; false: this source cannot (a stream has to be read to be counted)
    virtual bool count_records(unsigned, uint64_t&) { return false; }
    // record names are collected unless nobody will ask for them (the pseudoalignment loop never does)
    void set_want_names(bool on) { want_names_.store(on); }
    // what the parser threads spent (summed over the threads): nanoseconds parsing ranges, nanoseconds waiting because the
    // consumer was `window` ranges behind, text bytes and ranges parsed
    struct Stats { uint64_t parse_ns = 0, wait_ns = 0, bytes = 0, ranges = 0; unsigned threads = 0; };
    Stats stats() const { return Stats{parse_ns_.load(), wait_ns_.load(), bytes_.load(), ranges_.load(), nthreads_}; }

protected:
    std::atomic<bool> want_names_{true};
    std::atomic<uint64_t> parse_ns_{0}, wait_ns_{0}, bytes_{0}, ranges_{0};
    unsigned nthreads_ = 1;
};
inline uint64_t fastx_now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- gzip (or anything zlib reads): one background thread ---------------------------------------------------------
class StreamFastxSource : public FastxSource {
public:
    explicit StreamFastxSource(const std::string& path, uint64_t chunk_reads = 1 << 16) : chunk_reads_(chunk_reads) {
        f_ = gzopen(path.c_str(), "rb");  // transparent for files that are not gzip
        if (!f_) throw std::runtime_error("cannot open " + path);
        gzbuffer(f_, 1 << 20);
        buf_.resize(1 << 22);
        worker_ = std::thread([this] { produce(); });
    }
    ~StreamFastxSource() override {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_space_.notify_all();
        if (worker_.joinable()) worker_.join();
        if (f_) gzclose(f_);
    }
    bool pop(FastxChunk& c) override {
        {
            std::unique_lock<std::mutex> g(m_);
            cv_data_.wait(g, [this] { return !q_.empty() || done_; });
            if (!error_.empty()) throw std::runtime_error(error_);
            if (q_.empty()) return false;
            c = std::move(q_.front());
            q_.pop_front();
        }
        cv_space_.notify_one();
        return true;
    }
    void recycle(FastxChunk&& c) override {
        c.clear();
        std::lock_guard<std::mutex> g(m_);
        if (pool_.size() < 16) pool_.push_back(std::move(c));
    }

private:
    bool fill() {  // refill the line buffer; false at end of input
        if (pos_ < len_) memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
        len_ -= pos_;
        pos_ = 0;
        if (len_ == buf_.size()) buf_.resize(buf_.size() * 2);  // a single line longer than the buffer
        const int got = gzread(f_, buf_.data() + len_, (unsigned)std::min<size_t>(buf_.size() - len_, 1u << 30));
        if (got < 0) throw std::runtime_error("read error (corrupt gzip stream?)");
        if (got == 0) {  // end of input: zlib reports a stream that stops in the middle of a member only through gzerror
            int err = Z_OK;
            (void)gzerror(f_, &err);
            if (err == Z_BUF_ERROR || err == Z_DATA_ERROR) throw std::runtime_error("unexpected end of the gzip stream (truncated file?)");
        }
        len_ += (size_t)got;
        return got > 0;
    }
    bool line(const char*& s, size_t& n) {  // next line without its terminator; false at end of input
        for (;;) {
            const char* nl = (const char*)memchr(buf_.data() + pos_, '\n', len_ - pos_);
            if (nl) {
                s = buf_.data() + pos_;
                n = (size_t)(nl - s);
                pos_ += n + 1;
                if (n && s[n - 1] == '\r') --n;
                return true;
            }
            if (!fill()) {
                if (pos_ == len_) return false;
                s = buf_.data() + pos_;  // last line without a newline
                n = len_ - pos_;
                pos_ = len_;
                return true;
            }
        }
    }
    void produce() {
        try {
            FastxChunk c;
            c.want_names = want_names_.load();
            parse_fastx_records([this](const char*& s, size_t& n) { return line(s, n); }, c, [this](FastxChunk& ch) {
                if (ch.reads() == chunk_reads_) push(ch);
                std::lock_guard<std::mutex> g(m_);
                return !stop_;
            });
            if (c.reads()) push(c);
        } catch (std::exception& e) {
            std::lock_guard<std::mutex> g(m_);
            error_ = e.what();
        }
        {
            std::lock_guard<std::mutex> g(m_);
            done_ = true;
        }
        cv_data_.notify_all();
    }
    void push(FastxChunk& c) {
        {
            std::unique_lock<std::mutex> g(m_);
            cv_space_.wait(g, [this] { return q_.size() < 8 || stop_; });
            q_.push_back(std::move(c));
        }
        cv_data_.notify_one();
        std::lock_guard<std::mutex> g(m_);
        if (pool_.empty()) {
            c = FastxChunk();
        } else {
            c = std::move(pool_.back());
            pool_.pop_back();
        }
        c.want_names = want_names_.load();
    }

    gzFile f_ = nullptr;
    uint64_t chunk_reads_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
    std::thread worker_;
    std::mutex m_;
    std::condition_variable cv_data_, cv_space_;
    std::deque<FastxChunk> q_;
    std::vector<FastxChunk> pool_;  // emptied chunks, capacity kept
    bool done_ = false, stop_ = false;
    std::string error_;
};

// ---- plain files: mapped, parsed by a pool of threads, byte range by byte range -----------------------------------
class MappedFastxSource : public FastxSource {
public:
    // [begin, end): the part of the file this source delivers (byte positions; records that START inside it). The
    // multi-GPU driver gives every rank its own part.
    MappedFastxSource(const std::string& path, unsigned threads, uint64_t begin = 0, uint64_t end = ~0ULL,
                      uint64_t range_bytes = 8u << 20)
        : range_(range_bytes) {
        const uint64_t t_0 = fastx_now_ns();
        struct Trace { uint64_t t0; ~Trace() { if (getenv("FULGOR_TRACE_READER")) fprintf(stderr, "[reader] MappedFastxSource constructed in %.2f ms\n", (fastx_now_ns() - t0) / 1e6); } } trace{t_0};
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { close(fd_); throw std::runtime_error("cannot stat " + path); }
        size_ = (uint64_t)st.st_size;
        // The text is READ range by range into a buffer of the thread that parses the range (pread: one kernel copy out of the
        // page cache, 316 bytes per 150-base read). Mapping the file instead costs nothing while parsing but 60-90 ms for a 3 GB
        // file when the mapping goes (772 k page-table entries that up to a hundred threads have touched, under the process's
        // mmap lock: the next reader's mmap and its threads' stacks wait for it; measured, profiles/r5/e2e_breakdown_r5.txt).
        // FULGOR_READER_MMAP=1 maps (A/B measurements).
        const char* mm = getenv("FULGOR_READER_MMAP");
        use_pread_ = !mm || mm[0] == '2';
        map_windows_ = mm && mm[0] == '2';
        if (const char* e = getenv("FULGOR_READER_PIECE_KB")) {
            const long kb = atol(e);
            piece_ = kb > 0 ? (uint64_t)kb << 10 : ~0ULL;
        }
        if (size_ && !use_pread_) {
            map_ = (const char*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (map_ == MAP_FAILED) { close(fd_); throw std::runtime_error("cannot map " + path); }
            madvise((void*)map_, size_, MADV_SEQUENTIAL);
        }
        start(threads, begin, end);
    }
    ~MappedFastxSource() override {
        shutdown();
        // tearing down the page tables of a multi-gigabyte mapping that hundreds of threads have touched takes tens of
        // milliseconds (60-90 ms for 3 GB on a 256-thread host): nobody has to wait for it
        if (map_ && size_ && !use_pread_) {
            const void* m = map_;
            const size_t sz = size_;
            if (sz >= (size_t)64 << 20) std::thread([m, sz] { munmap((void*)m, sz); }).detach();
            else munmap((void*)m, sz);
        }
        if (fd_ >= 0) close(fd_);
    }
    bool pop(FastxChunk& c) override {
        std::unique_lock<std::mutex> g(m_);
        cv_data_.wait(g, [this] { return done_.count(next_out_) || next_out_ >= num_ranges_ || !error_.empty(); });
        if (!error_.empty()) throw std::runtime_error(error_);
        if (next_out_ >= num_ranges_) return false;
        c = std::move(done_[next_out_]);
        done_.erase(next_out_);
        const uint64_t r = next_out_++;
        g.unlock();
        cv_space_.notify_one();  // one range left the window: one parser thread may take the next (waking all of them for one slot costs 70 us per pop on a 64-thread pool)
        consumed(r);
        return true;
    }
    void recycle(FastxChunk&& c) override {
        c.clear();
        std::lock_guard<std::mutex> g(m_);
        if (pool_.size() < 64) pool_.push_back(std::move(c));
    }
    uint64_t size() const { return size_; }

protected:
    MappedFastxSource(uint64_t range_bytes) : range_(range_bytes) {}  // a derived source sets map_ / size_ and calls start()
    // hooks of a source whose bytes come into being on demand
    virtual void ensure(uint64_t, uint64_t) {}  // bytes [lo, hi) of map_ are about to be read
    virtual void consumed(uint64_t) {}          // range r has been handed out
    virtual bool lazy() const { return false; }
    void start(unsigned threads, uint64_t begin, uint64_t end) {
        if (size_) {  // FASTA or FASTQ, by the first header line: which line starts can open a record
            const uint64_t head = std::min<uint64_t>(size_, 1u << 20);
            Window w;
            if (!fastx_head_rangeable(base(w, 0, head), head, kind_))
                throw NotRangeable("this FASTQ text has records that are not four lines long (wrapped lines): it cannot be cut into byte ranges");
        }
        begin_ = std::min(begin, size_);
        end_ = std::min(end, size_);
        num_ranges_ = begin_ < end_ ? (end_ - begin_ + range_ - 1) / range_ : 0;
        window_ = std::max(1u, threads) + 8;  // ranges parsed ahead of the consumer (every one holds a pinned chunk)
        nthreads_ = std::max(1u, threads);
        {   // where the parser threads run: FULGOR_READER_AFFINITY = 0 (default: the scheduler's choice) | file | device | <node number>
            const char* e = getenv("FULGOR_READER_AFFINITY");
            const std::string mode = e ? e : "0";
            int node = -1;
            if (mode == "file") node = fd_ >= 0 && use_pread_ ? fastx_file_node(fd_, begin_, end_) : -1;
            else if (mode == "device") node = fastx_preferred_node().load();
            else if (mode != "0" && !mode.empty() && isdigit((unsigned char)mode[0])) node = atoi(mode.c_str());
            cpus_ = fastx_node_cpus(node);
            if (getenv("FULGOR_TRACE_READER")) fprintf(stderr, "[reader] parser threads: mode %s, node %d (device on %d), %zu cpus\n", mode.c_str(), node, fastx_preferred_node().load(), cpus_.size());
        }
        const uint64_t t_th = fastx_now_ns();
        for (unsigned t = 0; t < std::max(1u, threads) && t < num_ranges_; ++t) workers_.emplace_back([this] { work(); });
        if (getenv("FULGOR_TRACE_READER")) fprintf(stderr, "[reader] %zu parser threads started in %.2f ms\n", workers_.size(), (fastx_now_ns() - t_th) / 1e6);
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_space_.notify_all();
        cv_data_.notify_all();
        for (auto& w : workers_) w.join();
        workers_.clear();
    }
    // The bytes a thread looks at: M = base(w, a, b) makes the text bytes [a, b) readable as M[a] .. M[b - 1] (M is the mapping, or
    // the thread's own buffer moved back by the buffer's position in the text).
    struct Window {
        char* buf = nullptr;
        size_t cap = 0;
        uint64_t a = 0, b = 0;
        void* map = nullptr;
        size_t map_len = 0;
        ~Window() { free(buf); if (map) munmap(map, map_len); }
    };
    const char* base(Window& w, uint64_t a, uint64_t b) {
        if (!use_pread_) { ensure(a, b); return map_; }
        if (map_windows_) {
            if (a >= w.a && b <= w.b && w.map) return (const char*)w.map - w.a;
            if (w.map) munmap(w.map, w.map_len);
            const uint64_t a0 = a & ~(uint64_t)4095;
            w.map_len = (size_t)(b - a0);
            w.map = mmap(nullptr, w.map_len, PROT_READ, MAP_SHARED | MAP_POPULATE, fd_, (off_t)a0);
            if (w.map == MAP_FAILED) { w.map = nullptr; throw std::runtime_error("cannot map a range of the query file"); }
            w.a = a0;
            w.b = b;
            return (const char*)w.map - w.a;
        }
        if (a >= w.a && b <= w.b && w.buf) return w.buf - w.a;
        const size_t need = (size_t)(b - a);
        if (need > w.cap) {
            free(w.buf);
            w.cap = need + need / 8 + 4096;
            w.buf = (char*)malloc(w.cap);
            if (!w.buf) { w.cap = 0; throw std::bad_alloc(); }
        }
        size_t got = 0;
        while (got < need) {
            const ssize_t k = pread(fd_, w.buf + got, need - got, (off_t)(a + got));
            if (k < 0) { if (errno == EINTR) continue; throw std::runtime_error("cannot read the query file"); }
            if (k == 0) throw std::runtime_error("the query file shrank while it was read");
            got += (size_t)k;
        }
        w.a = a;
        w.b = b;
        return w.buf - w.a;
    }
    // first record boundary at or behind byte p within the window [.., wend) of the text M (wend if none): sure = false when the
    // window ended before the answer was certain. A record starts at a line start with '>' (FASTA files), or with '@' when the
    // four lines from there look like a FASTQ record: third line '+...', fourth as long as the second (a quality line may
    // begin with '@', but then the line after next is a sequence).
    uint64_t record_start_in(const char* M, uint64_t p, uint64_t wend, bool& sure) const {
        sure = true;
        if (p == 0) return 0;
        const char* nl = (const char*)memchr(M + p - 1, '\n', wend - (p - 1));  // p - 1: p itself may be a line start
        uint64_t q = nl ? (uint64_t)(nl - M) + 1 : wend;
        while (q < wend) {
            if (M[q] == '>' && kind_ != '@') return q;  // (in a FASTQ file a quality line may begin with '>')
            if (M[q] == '@' && kind_ != '>') {
                const char* fin = M + wend;
                const char* l1 = (const char*)memchr(M + q, '\n', wend - q);                    // end of the header
                const char* l2 = l1 ? (const char*)memchr(l1 + 1, '\n', (size_t)(fin - l1 - 1)) : nullptr;  // end of the sequence
                if (!l2) {  // the text ends within two lines: a last quality line that begins with '@', or a record cut off behind its
                    // header. Not taken for a record start: whoever parses the lines in front decides with the full grammar.
                    if (wend != size_) { sure = false; return q; }
                    return size_;
                }
                if (l2 + 1 < fin && l2[1] == '+') {
                    const char* l3 = (const char*)memchr(l2 + 1, '\n', (size_t)(fin - l2 - 1));   // end of the '+' line
                    if (!l3) { sure = wend == size_; return q; }
                    const char* l4 = (const char*)memchr(l3 + 1, '\n', (size_t)(fin - l3 - 1));
                    if (!l4 && wend != size_) { sure = false; return q; }
                    size_t ls = (size_t)(l2 - l1 - 1), lq = (size_t)((l4 ? l4 : fin) - l3 - 1);
                    if (ls && l1[ls] == '\r') --ls;
                    if (lq && l3[lq] == '\r') --lq;
                    if (ls == lq) return q;
                } else if (l2 + 1 >= fin && wend != size_) {
                    sure = false;
                    return q;
                }
            }
            const char* e = (const char*)memchr(M + q, '\n', wend - q);
            q = e ? (uint64_t)(e - M) + 1 : wend;
        }
        sure = wend == size_;
        return wend;
    }
    // records that start in range r: text bytes [lo, hi), both ends at record boundaries, readable through the returned M. One load
    // covers the range and what the two boundary searches look at beyond its ends (the margin grows in the rare case of a record
    // longer than the margin).
    const char* range_bounds(uint64_t r, uint64_t& lo, uint64_t& hi, Window& w) {
        const uint64_t a0 = begin_ + r * range_, b0 = r + 1 == num_ranges_ ? end_ : begin_ + (r + 1) * range_;
        for (uint64_t margin = 64u << 10;; margin *= 4) {
            const uint64_t la = a0 ? a0 - 1 : 0, lb = std::min(size_, b0 + margin);
            const char* M = base(w, la, lb);
            bool sure_lo, sure_hi = true;
            lo = record_start_in(M, a0, lb, sure_lo);
            hi = (r + 1 == num_ranges_ && end_ == size_) ? size_ : record_start_in(M, b0, lb, sure_hi);
            if ((sure_lo && sure_hi) || lb == size_) return M;
        }
    }

    // first record boundary at or behind byte p > 0 (the text's end if none): a small load around p
    uint64_t boundary_at(uint64_t p, Window& w) {
        for (uint64_t margin = 64u << 10;; margin *= 4) {
            const uint64_t lb = std::min(size_, p + margin);
            const char* M = base(w, p - 1, lb);
            bool sure;
            const uint64_t q = record_start_in(M, p, lb, sure);
            if (sure || lb == size_) return q;
        }
    }
    // The records that start in range r go to the sink; [lo, hi) = their text bytes. Plain four-line FASTQ records of a file that is
    // read (not mapped) are taken piece by piece: half a megabyte is read and parsed while it is still in the core's cache (an 8 MB
    // window is written to memory and read back: 566 against 449 M reads/s for 24 threads, 700 against 437 for 48;
    // profiles/r5/reader_pieces_r5.txt). Whatever the fast path leaves (other record shapes, FASTA) goes through the grammar.
    template <typename Sink>
    void parse_range(uint64_t r, Window& w, Sink& c, unsigned& odd, uint64_t& lo, uint64_t& hi) {
        const bool pieces = use_pread_ && !map_windows_ && kind_ == '@' && piece_ < range_;
        const char* M = nullptr;
        if (pieces) {
            const uint64_t a0 = begin_ + r * range_, b0 = r + 1 == num_ranges_ ? end_ : begin_ + (r + 1) * range_;
            lo = a0 ? boundary_at(a0, w) : 0;
            hi = (r + 1 == num_ranges_ && end_ == size_) ? size_ : boundary_at(b0, w);
        } else {
            M = range_bounds(r, lo, hi, w);
        }
        c.expect(hi - lo, kind_ == '@');
        uint64_t pos = lo;
        if (pieces) {
            for (uint64_t piece = piece_; pos < hi;) {
                const uint64_t pe = std::min(hi, pos + piece);
                const char* P = base(w, pos, pe);
                const uint64_t q = parse_fastq4(P, pos, pe, c, pe == hi);
                if (q == pos) break;  // not a plain four-line record, or one longer than the piece
                if (q - pos < (pe - pos) / 2 && piece < range_) piece *= 4;  // (long reads: fewer bytes read twice)
                pos = q;
            }
            if (pos < hi) M = base(w, pos, hi);
        } else if (kind_ == '@') {
            pos = parse_fastq4(M, pos, hi, c);
        }
        if (pos >= hi) return;
        parse_fastx_records(
            [&](const char*& s, size_t& n) {
                if (pos >= hi) return false;
                const char* nl = (const char*)memchr(M + pos, '\n', hi - pos);
                s = M + pos;
                n = nl ? (size_t)(nl - s) : (size_t)(hi - pos);
                pos += n + 1;
                if (n && s[n - 1] == '\r') --n;
                return true;
            },
            c, [](Sink&) { return true; }, &odd);
    }

public:
    bool count_records(unsigned threads, uint64_t& total) override {
        total = count_ranges(0, num_ranges_, threads);
        return true;
    }

protected:
    // records that start in ranges [r0, r1): the grammar walked by `threads` threads, nothing copied
    uint64_t count_ranges(uint64_t r0, uint64_t r1, unsigned threads) {
        std::atomic<uint64_t> next{r0}, sum{0};
        std::mutex em;
        std::string err;
        auto body = [&] {
            Window w;
            for (;;) {
                const uint64_t r = next++;
                if (r >= r1) return;
                try {
                    uint64_t lo, hi;
                    unsigned odd = 0;
                    CountSink cs;
                    parse_range(r, w, cs, odd, lo, hi);
                    sum += cs.n;
                } catch (std::exception& e) {
                    std::lock_guard<std::mutex> g(em);
                    err = e.what();
                    return;
                }
            }
        };
        std::vector<std::thread> th;
        const unsigned nt = (unsigned)std::min<uint64_t>(std::max(1u, threads), std::max<uint64_t>(1, r1 - r0));
        for (unsigned t = 1; t < nt; ++t) th.emplace_back([this, &body, nt] { fastx_confine_thread(cpus_, nt); body(); });
        body();
        for (auto& t : th) t.join();
        if (!err.empty()) throw std::runtime_error(err);
        return sum.load();
    }

    void work() {
        fastx_confine_thread(cpus_, nthreads_);
        Window w;
        for (;;) {
            uint64_t r;
            FastxChunk c;
            const uint64_t t_wait = fastx_now_ns();
            {
                std::unique_lock<std::mutex> g(m_);
                cv_space_.wait(g, [this] { return stop_ || (!paused_ && (next_in_ >= num_ranges_ || next_in_ < next_out_ + window_)); });
                if (stop_ || next_in_ >= num_ranges_) return;
                r = next_in_++;
                ++active_;
                if (!pool_.empty()) { c = std::move(pool_.back()); pool_.pop_back(); }
            }
            const uint64_t t_parse = fastx_now_ns();
            wait_ns_ += t_parse - t_wait;
            try {
                uint64_t lo, hi;
                unsigned odd = 0;
                c.want_names = want_names_.load();
                parse_range(r, w, c, odd, lo, hi);
                // a range that starts at a record boundary and ends at one holds whole records and nothing else
                if ((odd & 1u) || ((odd & 2u) && hi != size_))  // (a quality cut short by the end of the text is the file's business)
                    throw std::runtime_error("the query file does not parse as whole records in byte ranges (FASTQ with wrapped lines behind "
                                             "a four-line head?): bytes " + std::to_string(lo) + " to " + std::to_string(hi));
                bytes_ += hi - lo;
            } catch (std::exception& e) {
                std::lock_guard<std::mutex> g(m_);
                error_ = e.what();
            }
            parse_ns_ += fastx_now_ns() - t_parse;
            ranges_ += 1;
            {
                std::lock_guard<std::mutex> g(m_);
                done_[r] = std::move(c);
                --active_;
            }
            cv_data_.notify_all();
        }
    }
    // the parser threads take no new range until resume_parsers(); returns when none of them is inside a range
    void pause_parsers() {
        std::unique_lock<std::mutex> g(m_);
        paused_ = true;
        cv_data_.wait(g, [this] { return active_ == 0; });
    }
    void resume_parsers() {
        {
            std::lock_guard<std::mutex> g(m_);
            paused_ = false;
        }
        cv_space_.notify_all();
    }
    uint64_t handed_out() {
        std::lock_guard<std::mutex> g(m_);
        return next_out_;
    }

protected:
    int fd_ = -1;
    std::vector<int> cpus_;   // CPUs the parser threads are confined to (empty: left to the scheduler)
    bool use_pread_ = false;  // (derived sources hand out their own buffer as map_)
    uint64_t piece_ = 512u << 10;  // bytes read and parsed at a time inside a range (FULGOR_READER_PIECE_KB; 0: the whole range)
    bool map_windows_ = false;  // FULGOR_READER_MMAP=2: every thread maps the range it parses (and unmaps it behind itself)
    const char* map_ = nullptr;
    uint64_t size_ = 0, begin_ = 0, end_ = 0, range_, num_ranges_ = 0, window_ = 4;
    char kind_ = 0;  // '>' FASTA, '@' FASTQ, 0 unknown (both kinds of record start are looked for)

private:
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_data_, cv_space_;
    std::map<uint64_t, FastxChunk> done_;  // parsed ranges waiting for their turn
    std::vector<FastxChunk> pool_;
    uint64_t next_in_ = 0, next_out_ = 0;
    unsigned active_ = 0;  // parser threads inside a range
    bool stop_ = false, paused_ = false;
    std::string error_;
};

// ---- block-compressed gzip (BGZF): members inflated on demand by the same pool of threads -----------------------------
// A BGZF member: 10-byte gzip header with FLG.FEXTRA, XLEN, an extra subfield 'B' 'C' of two bytes holding the member's
// total size minus one, raw deflate data, CRC32, ISIZE (SAM specification, section 4.1; every member inflates to at most 64 KB).
inline bool bgzf_member(const unsigned char* p, uint64_t avail, uint64_t& total, uint64_t& data_at, uint64_t& data_len, uint32_t& isize) {
    if (avail < 28 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    const uint32_t xlen = p[10] | (p[11] << 8);
    if (avail < 12ull + xlen + 8) return false;
    uint32_t at = 0;
    total = 0;
    while (at + 4 <= xlen) {
        const unsigned char* f = p + 12 + at;
        const uint32_t slen = f[2] | (f[3] << 8);
        if (f[0] == 'B' && f[1] == 'C' && slen == 2 && at + 6 <= xlen) total = (uint64_t)(f[4] | (f[5] << 8)) + 1;
        at += 4 + slen;
    }
    if (total < 12ull + xlen + 8 || total > avail) return false;
    data_at = 12ull + xlen;
    data_len = total - data_at - 8;
    isize = p[total - 4] | (p[total - 3] << 8) | (p[total - 2] << 16) | ((uint32_t)p[total - 1] << 24);
    return true;
}

inline bool is_bgzf_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    unsigned char h[4096];
    const size_t got = fread(h, 1, sizeof h, f);
    fclose(f);
    if (got < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
    const uint32_t xlen = h[10] | (h[11] << 8);
    for (uint32_t at = 0; at + 4 <= xlen && 12 + at + 4 <= got;) {
        const unsigned char* s = h + 12 + at;
        const uint32_t slen = s[2] | (s[3] << 8);
        if (s[0] == 'B' && s[1] == 'C' && slen == 2) return true;
        at += 4 + slen;
    }
    return false;
}

// length of the text that the parts [begin, end) of a query file refer to: the file size, or for a block-compressed gzip file
// the inflated size (sum of the members' ISIZE fields); 0 with `parts = false` for an ordinary gzip stream
inline uint64_t fastx_text_size(const std::string& path, bool& parts) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) throw std::runtime_error("cannot stat " + path);
    parts = true;
    FILE* f0 = fopen(path.c_str(), "rb");
    if (!f0) throw std::runtime_error("cannot open " + path);
    unsigned char m[2] = {0, 0};
    const size_t got = fread(m, 1, 2, f0);
    fclose(f0);
    if (!(got == 2 && m[0] == 0x1f && m[1] == 0x8b)) {
        parts = fastx_file_rangeable(path);  // (FASTQ with wrapped lines: one stream)
        return (uint64_t)st.st_size;
    }
    if (!is_bgzf_file(path)) { parts = false; return 0; }
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    const uint64_t csize = (uint64_t)st.st_size;
    const unsigned char* cm = (const unsigned char*)mmap(nullptr, csize, PROT_READ, MAP_PRIVATE, fd, 0);
    if (cm == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + path); }
    uint64_t at = 0, u = 0;
    bool ok = true;
    while (at < csize) {
        uint64_t total, da, dl;
        uint32_t isize;
        if (!bgzf_member(cm + at, csize - at, total, da, dl, isize)) { ok = false; break; }
        u += isize;
        at += total;
    }
    munmap((void*)cm, csize);
    close(fd);
    if (!ok) throw std::runtime_error("corrupt block-compressed gzip file");
    return u;
}

// libdeflate, when the host has it (looked up at run time: only the shared object ships with most systems, no header), inflates a
// whole member 2-3 times faster than zlib's streaming inflate; without it zlib does the work.
struct LibDeflate {
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    int (*gzip_ex)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;  // one gzip member: bytes consumed / produced
    bool ok = false;
    LibDeflate() {
        if (getenv("FULGOR_NO_LIBDEFLATE")) return;
        void* h = nullptr;
        for (const char* name : {"libdeflate.so.0", "libdeflate.so", "libdeflate.so.1"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) return;
        alloc = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        release = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        gzip_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_gzip_decompress_ex");
        ok = alloc && decompress && release && crc;
    }
    static const LibDeflate& get() { static const LibDeflate l; return l; }
};

class BgzfFastxSource : public MappedFastxSource {
public:
    // [begin, end): positions in the INFLATED text (fastx_text_size gives its length), as for a plain file
    BgzfFastxSource(const std::string& path, unsigned threads, uint64_t begin = 0, uint64_t end = ~0ULL, uint64_t range_bytes = 8u << 20)
        : MappedFastxSource(range_bytes) {
        cfd_ = open(path.c_str(), O_RDONLY);
        if (cfd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(cfd_, &st) != 0 || st.st_size == 0) { close(cfd_); throw std::runtime_error("cannot stat " + path); }
        csize_ = (uint64_t)st.st_size;
        cmap_ = (const unsigned char*)mmap(nullptr, csize_, PROT_READ, MAP_PRIVATE, cfd_, 0);
        if (cmap_ == MAP_FAILED) { close(cfd_); throw std::runtime_error("cannot map " + path); }
        try {
            uint64_t at = 0, u = 0;
            while (at < csize_) {  // the members: a hop per member through the size fields
                Member m;
                uint64_t total;
                if (!bgzf_member(cmap_ + at, csize_ - at, total, m.data_at, m.data_len, m.isize))
                    throw std::runtime_error("corrupt block-compressed gzip file (member header at byte " + std::to_string(at) + ")");
                if (m.isize > 65536) throw std::runtime_error("corrupt block-compressed gzip file (member larger than 64 KB)");
                m.data_at += at;
                m.crc_at = at + total - 8;
                m.uoff = u;
                if (m.isize) members_.push_back(m);
                u += m.isize;
                at += total;
            }
            size_ = u;
            state_.reset(new std::atomic<unsigned char>[members_.size() + 1]);
            for (size_t i = 0; i <= members_.size(); ++i) state_[i].store(0);
            if (size_) {
                void* b = mmap(nullptr, size_ + 1, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
                if (b == MAP_FAILED) throw std::runtime_error("cannot reserve the buffer for the inflated file");
                madvise(b, size_ + 1, MADV_HUGEPAGE);  // (first touch by page: fewer faults for the inflating threads)
                map_ = (const char*)b;
            }
            start(threads, begin, end);  // (throws NotRangeable for wrapped FASTQ: the caller falls back to one stream)
        } catch (...) {
            shutdown();
            if (map_ && size_) munmap((void*)map_, size_ + 1);
            map_ = nullptr;  // (the base destructor finds nothing to unmap)
            size_ = 0;
            munmap((void*)cmap_, csize_);
            close(cfd_);
            throw;
        }
    }
    // Counting inflates what it walks: the parser threads are held while it runs, the ranges are counted a strip at a time, and what
    // lies behind a strip goes back to the system (members marked not inflated again), so that the count of a part keeps a few
    // dozen megabytes resident, not the part (N ranks of a multi-GPU run each count their part before they stream it). Only before
    // the first chunk has been handed out: behind the reader the text is gone.
    bool count_records(unsigned threads, uint64_t& total) override {
        if (handed_out() > 0) return false;
        pause_parsers();
        struct Resume { BgzfFastxSource* s; ~Resume() { s->resume_parsers(); } } resume{this};
        const uint64_t strip = std::max<uint64_t>(4, 2ull * std::max(1u, threads));
        uint64_t n = 0;
        size_t released = 0;  // members in front of this one are not inflated (any more)
        for (uint64_t r0 = 0; r0 < num_ranges_; r0 += strip) {
            const uint64_t r1 = std::min(num_ranges_, r0 + strip);
            n += count_ranges(r0, r1, threads);
            // nobody reads now; the next strip looks back one byte from its start: keep the member that holds it
            const uint64_t keep_from = r1 == num_ranges_ ? size_ : begin_ + r1 * range_ - 1;
            release_members(released, keep_from);
        }
        release_members(released, size_);  // (what the parser threads had inflated ahead as well: they inflate again what they need)
        total = n;
        return true;
    }
    ~BgzfFastxSource() override {
        shutdown();  // (the workers use this object's hooks: they end before it does)
        {   // (unmapping gigabytes that many threads have touched takes tens of milliseconds: not on the caller's time)
            const void* m = map_;
            const void* cm = cmap_;
            const size_t sz = map_ && size_ ? size_ + 1 : 0, csz = csize_;
            auto unmap = [m, sz, cm, csz] {
                if (sz) munmap((void*)m, sz);
                munmap((void*)cm, csz);
            };
            if (sz + csz >= ((size_t)64 << 20)) std::thread(unmap).detach();
            else unmap();
        }
        map_ = nullptr;
        size_ = 0;
        close(cfd_);
    }

protected:
    bool lazy() const override { return true; }
    void ensure(uint64_t lo, uint64_t hi) override {
        if (lo >= hi || members_.empty()) return;
        // first member that ends behind lo
        size_t a = 0, b = members_.size();
        while (a < b) {
            const size_t mid = (a + b) / 2;
            if (members_[mid].uoff + members_[mid].isize <= lo) a = mid + 1; else b = mid;
        }
        z_stream zs;
        bool init = false;
        const LibDeflate& ld = LibDeflate::get();
        void* fast = nullptr;
        struct Release { const LibDeflate& l; void*& d; ~Release() { if (d) l.release(d); } } release_fast{ld, fast};
        for (size_t i = a; i < members_.size() && members_[i].uoff < hi; ++i) {
            unsigned char st = 0;
            if (state_[i].compare_exchange_strong(st, 1)) {
                const Member& m = members_[i];
                if (ld.ok && (fast || (fast = ld.alloc()))) {
                    size_t got = 0;
                    const int rc = ld.decompress(fast, cmap_ + m.data_at, m.data_len, (void*)(map_ + m.uoff), m.isize, &got);
                    const unsigned char* t = cmap_ + m.crc_at;
                    const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                    const bool good = rc == 0 && got == m.isize && ld.crc(0, map_ + m.uoff, m.isize) == want_crc;
                    state_[i].store(good ? 2 : 3);
                    if (!good) {
                        if (init) inflateEnd(&zs);
                        throw std::runtime_error("corrupt block-compressed gzip file (member at inflated byte " + std::to_string(m.uoff) + ")");
                    }
                    continue;
                }
                if (!init) {
                    memset(&zs, 0, sizeof zs);
                    if (inflateInit2(&zs, -15) != Z_OK) { state_[i].store(3); throw std::runtime_error("zlib: inflateInit2 failed"); }
                    init = true;
                } else {
                    inflateReset(&zs);
                }
                zs.next_in = (Bytef*)(cmap_ + m.data_at);
                zs.avail_in = (uInt)m.data_len;
                zs.next_out = (Bytef*)(map_ + m.uoff);
                zs.avail_out = m.isize;
                const int rc = inflate(&zs, Z_FINISH);
                const unsigned char* t = cmap_ + m.crc_at;
                const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
                const bool ok = rc == Z_STREAM_END && zs.avail_out == 0 &&
                                (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)(map_ + m.uoff), m.isize) == want_crc;
                state_[i].store(ok ? 2 : 3);
                if (!ok) { inflateEnd(&zs); throw std::runtime_error("corrupt block-compressed gzip file (member at inflated byte " + std::to_string(m.uoff) + ")"); }
            } else {
                while ((st = state_[i].load()) == 1) std::this_thread::yield();  // another thread is inflating it
                if (st == 3) { if (init) inflateEnd(&zs); throw std::runtime_error("corrupt block-compressed gzip file"); }
            }
        }
        if (init) inflateEnd(&zs);
    }
    void consumed(uint64_t r) override {  // the inflated bytes of ranges well behind the reader go back to the system
        if (r < 2) return;
        const uint64_t page = 4096, lo = (begin_ + (r - 2) * range_ + page - 1) / page * page, hi = (begin_ + (r - 1) * range_) / page * page;
        if (hi > lo && hi <= size_) madvise((void*)(map_ + lo), hi - lo, MADV_DONTNEED);
    }
    // members [from, first member that reaches byte `keep_from` of the text): marked not inflated, their pages given back (those
    // that no kept member shares). Call only while no thread reads the text. `from` moves to the first member kept.
    void release_members(size_t& from, uint64_t keep_from) {
        size_t to = from;
        while (to < members_.size() && members_[to].uoff + members_[to].isize <= keep_from) ++to;
        if (to == from) return;
        for (size_t i = from; i < to; ++i) state_[i].store(0);
        const uint64_t page = 4096;
        const uint64_t lo = (members_[from].uoff + page - 1) / page * page;
        const uint64_t end = to < members_.size() ? members_[to].uoff : size_;
        const uint64_t hi = to < members_.size() ? end / page * page : (end + page - 1) / page * page;
        if (hi > lo) madvise((void*)(map_ + lo), hi - lo, MADV_DONTNEED);
        from = to;
    }

private:
    struct Member { uint64_t data_at, data_len, crc_at, uoff; uint32_t isize; };
    int cfd_ = -1;
    const unsigned char* cmap_ = nullptr;
    uint64_t csize_ = 0;
    std::vector<Member> members_;
    std::unique_ptr<std::atomic<unsigned char>[]> state_;
};

// ---- ordinary gzip of moderate size, when libdeflate is there: inflated in one go (2-3 times zlib's streaming rate), then
// parsed by the pool of threads like a plain file. Everything else about a gzip stream stays with StreamFastxSource.
class InflatedFastxSource : public MappedFastxSource {
public:
    static constexpr uint64_t MAX_COMPRESSED = 2ull << 30;  // the text (about four times that) is held in memory
    // nullptr: not applicable (no libdeflate, file too large, or anything unexpected: the caller streams it with zlib)
    static InflatedFastxSource* try_open(const std::string& path, unsigned threads) {
        const LibDeflate& ld = LibDeflate::get();
        if (!ld.ok || !ld.gzip_ex) return nullptr;
        struct stat st;
        if (stat(path.c_str(), &st) != 0 || st.st_size < 18 || (uint64_t)st.st_size > MAX_COMPRESSED) return nullptr;
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return nullptr;
        const uint64_t csize = (uint64_t)st.st_size;
        const unsigned char* cm = (const unsigned char*)mmap(nullptr, csize, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (cm == MAP_FAILED) return nullptr;
        // ISIZE of the last member bounds the text from below; members are inflated one after the other, the buffer grows
        uint64_t cap = std::max<uint64_t>(csize * 4, (uint64_t)(cm[csize - 4] | (cm[csize - 3] << 8) | (cm[csize - 2] << 16) | ((uint32_t)cm[csize - 1] << 24))) + 4096;
        char* buf = (char*)malloc(cap);
        void* d = buf ? ld.alloc() : nullptr;
        uint64_t in = 0, out = 0;
        bool good = buf && d;
        while (good && in < csize) {
            if (csize - in < 18 || cm[in] != 0x1f || cm[in + 1] != 0x8b) {  // trailing zero padding is tolerated, as gzip does; garbage is not
                for (uint64_t i = in; i < csize; ++i) good = good && cm[i] == 0;
                break;
            }
            size_t used = 0, made = 0;
            const int rc = ld.gzip_ex(d, cm + in, csize - in, buf + out, cap - out, &used, &made);
            if (rc == 3) {  // LIBDEFLATE_INSUFFICIENT_SPACE
                const uint64_t ncap = cap + cap / 2 + (64u << 20);
                char* nb = (char*)realloc(buf, ncap);
                if (!nb) { good = false; break; }
                buf = nb;
                cap = ncap;
                continue;
            }
            if (rc != 0) { good = false; break; }
            in += used;
            out += made;
        }
        if (d) ld.release(d);
        munmap((void*)cm, csize);
        if (!good) { free(buf); return nullptr; }
        {   // FASTQ with wrapped lines cannot be cut into ranges: the caller streams the file with the full grammar, as the reference does
            char kind = 0;
            if (out && !fastx_head_rangeable(buf, (size_t)std::min<uint64_t>(out, 1u << 20), kind)) { free(buf); return nullptr; }
        }
        try {
            return new InflatedFastxSource(buf, out, threads);
        } catch (...) {  // (the constructor has let go of the buffer)
            free(buf);
            return nullptr;
        }
    }
    ~InflatedFastxSource() override {
        shutdown();
        free(buf_);
        map_ = nullptr;
        size_ = 0;
    }

private:
    InflatedFastxSource(char* buf, uint64_t n, unsigned threads) : MappedFastxSource(8u << 20), buf_(buf) {
        map_ = buf;
        size_ = n;
        try {
            start(threads, 0, ~0ULL);
        } catch (...) {  // (the base destructor must not unmap a malloc'd buffer; try_open releases it)
            shutdown();
            map_ = nullptr;
            size_ = 0;
            buf_ = nullptr;
            throw;
        }
    }
    char* buf_;
};

inline bool is_gzip_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    unsigned char m[2] = {0, 0};
    const size_t got = fread(m, 1, 2, f);
    fclose(f);
    return got == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

// Batches of at most max_reads reads in file order out of a chunk source (a chunk that does not fit is split: the rest
// opens the next batch). The batch is assembled by several threads, slice by slice, straight into the caller's buffer
// (pinned host memory in the engine); record names are put together only when asked for.
class FastxReader {
public:
    explicit FastxReader(const std::string& path, unsigned threads = 0, uint64_t begin = 0, uint64_t end = ~0ULL) {
        const uint64_t t_0 = fastx_now_ns();
        struct Trace { uint64_t t0; ~Trace() { if (getenv("FULGOR_TRACE_READER")) fprintf(stderr, "[reader] FastxReader constructed in %.2f ms\n", (fastx_now_ns() - t0) / 1e6); } } trace{t_0};
        if (threads == 0) threads = default_threads();
        threads_ = threads;
        const uint64_t range_bytes = default_range_bytes();
        if (is_gzip_file(path)) {
            if (is_bgzf_file(path)) {
                // (a block-compressed FASTQ with wrapped lines cannot be cut into ranges: read whole, it falls back to the one-stream
                // reader, which takes block-compressed files as any gzip reader does; parts of it stay an error)
                try {
                    src_.reset(new BgzfFastxSource(path, threads, begin, end, range_bytes));
                } catch (const NotRangeable&) {
                    if (begin != 0 || end != ~0ULL) throw;
                    src_.reset(new StreamFastxSource(path));
                }
            } else {
                if (begin != 0 || end != ~0ULL) throw std::runtime_error("a gzip stream cannot be read in parts (a block-compressed one, as bgzip writes it, can)");
                FastxSource* whole = getenv("FULGOR_GZIP_STREAM") ? nullptr : InflatedFastxSource::try_open(path, threads);
                if (whole) src_.reset(whole);
                else src_.reset(new StreamFastxSource(path));
            }
        } else {
            // FASTQ with wrapped lines (records that are not four lines long) offers no record boundaries to guess: one stream,
            // kseq's full grammar, as the reference reads it
            if (fastx_file_rangeable(path)) src_.reset(new MappedFastxSource(path, threads, begin, end, range_bytes));
            else if (begin == 0 && end == ~0ULL) src_.reset(new StreamFastxSource(path));
            else throw std::runtime_error("a FASTQ file with wrapped lines cannot be read in parts");
        }
    }
    // parser threads when the caller names none: half of the host's hardware threads, at most 24 (FULGOR_READER_THREADS overrides);
    // the multi-GPU driver divides them among the ranks of a host. More do not help: the ranges are read with pread, and past 32
    // threads on one file the reads get in each other's way in the kernel (record count of a 3.2 GB FASTQ file on a 256-thread
    // host: 16 threads 74 GB/s, 32: 78, 64: 45, 128: 12; profiles/r5/e2e_grid_r5.txt), while 16 to 24 threads parse 250 M reads/s
    // and the worker loop behind them takes 200 M
    static unsigned default_threads() {
        if (const char* e = getenv("FULGOR_READER_THREADS")) { const long v = atol(e); if (v > 0) return (unsigned)std::min<long>(v, 1024); }
        return std::min(24u, std::max(1u, std::thread::hardware_concurrency() / 2));
    }
    // bytes of text per parsed range (= per chunk handed to the worker loop; FULGOR_READER_RANGE_KB overrides)
    static uint64_t default_range_bytes() {
        if (const char* e = getenv("FULGOR_READER_RANGE_KB")) { const long v = atol(e); if (v >= 4) return (uint64_t)v << 10; }
        // 8 MB less 40 KB: the bases of a FASTQ range (at most half its text) then fit a pooled slab of 4 MB, its offsets (one per 256
        // bytes of text at most, as reserved) one of 256 KB
        // (with 8 MB even: 5 MB and 512 KB; with the growth margin that reserve() adds, 6 MB: a third more to pin in a process's first run)
        // (4 MB: fewer pinned bytes but twice the slabs and copies: first run 370 instead of 240 ms, steady runs 58-90 instead of 51-69 ms)
        return (8u << 20) - (40u << 10);
    }
    // Chunk-level access for a worker loop that uploads the parsed ranges as they are (no second copy on the host): the next
    // non-empty chunk in file order, false at the end. Not to be mixed with next() on the same reader.
    bool pop_chunk(FastxChunk& c) {
        while (src_->pop(c)) {
            if (c.reads()) return true;
            src_->recycle(std::move(c));
            c = FastxChunk();
        }
        return false;
    }
    void recycle_chunk(FastxChunk&& c) { src_->recycle(std::move(c)); }
    FastxSource::Stats stats() const { return src_->stats(); }
    // records of this reader's part, counted by a walk over the record grammar (nothing copied, nothing consumed); false for a
    // source that has to be read to be counted (ordinary gzip, FASTQ with wrapped lines)
    bool count_records(uint64_t& n) { return src_->count_records(threads_, n); }
    void set_want_names(bool on) { src_->set_want_names(on); }
    unsigned threads() const { return threads_; }
    // Bases: clear(), reserve(bytes), data(), set_size(bytes); Offs: resize(n), data(), operator[]
    template <typename Bases, typename Offs>
    bool next(uint64_t max_reads, Bases& bases, Offs& offs) {
        // the chunks of the previous batch go back to the source; a partly used one stays in front
        for (size_t i = 0; i + (cur_partial_ ? 1 : 0) < held_.size(); ++i) src_->recycle(std::move(held_[i]));
        if (cur_partial_) {
            FastxChunk keep = std::move(held_.back());
            held_.clear();
            held_.push_back(std::move(keep));
        } else {
            held_.clear();
            cur_at_ = 0;
        }
        slices_.clear();
        uint64_t reads = 0, nbases = 0;
        size_t hi = 0;  // index in held_ of the chunk being sliced
        while (reads < max_reads) {
            if (hi == held_.size()) {
                FastxChunk c;
                if (!src_->pop(c)) break;
                if (c.reads() == 0) { src_->recycle(std::move(c)); continue; }
                held_.push_back(std::move(c));
                cur_at_ = 0;
            }
            const FastxChunk& c = held_[hi];
            const uint64_t take = std::min<uint64_t>(max_reads - reads, c.reads() - cur_at_);
            slices_.push_back(Slice{hi, cur_at_, take, nbases, reads});
            nbases += c.offs[cur_at_ + take] - c.offs[cur_at_];
            reads += take;
            cur_at_ += take;
            if (cur_at_ == c.reads()) { ++hi; cur_at_ = 0; cur_partial_ = false; }
            else cur_partial_ = true;
        }
        if (hi == held_.size()) cur_partial_ = false;
        bases.clear();
        bases.reserve(nbases);
        bases.set_size(nbases);
        offs.resize(reads + 1);
        offs[0] = 0;
        char* dst = bases.data();
        uint64_t* od = offs.data();
        auto copy = [&](size_t s0, size_t step) {
            for (size_t s = s0; s < slices_.size(); s += step) {
                const Slice& sl = slices_[s];
                const FastxChunk& c = held_[sl.chunk];
                const uint64_t b0 = c.offs[sl.first];
                memcpy(dst + sl.base_off, c.bases.data() + b0, c.offs[sl.first + sl.take] - b0);
                for (uint64_t i = 1; i <= sl.take; ++i) od[sl.read_off + i] = sl.base_off + c.offs[sl.first + i] - b0;
            }
        };
        const size_t nt = std::min<size_t>(std::min<size_t>(threads_, 16), slices_.size());
        if (nt <= 1) {
            copy(0, 1);
        } else {
            std::vector<std::thread> th;
            for (size_t t = 1; t < nt; ++t) th.emplace_back(copy, t, nt);
            copy(0, nt);
            for (auto& x : th) x.join();
        }
        return reads > 0;
    }
    // names of the records of the batch returned last (header up to the first blank), concatenated + offsets
    void names(std::vector<char>& names, std::vector<uint64_t>& name_offs) const {
        names.clear();
        name_offs.assign(1, 0);
        for (const Slice& sl : slices_) {
            const FastxChunk& c = held_[sl.chunk];
            const uint64_t n0 = c.name_offs[sl.first], n1 = c.name_offs[sl.first + sl.take], nb = names.size();
            names.insert(names.end(), c.names.begin() + n0, c.names.begin() + n1);
            for (uint64_t i = 1; i <= sl.take; ++i) name_offs.push_back(nb + c.name_offs[sl.first + i] - n0);
        }
    }
    // number of records left (consumes them)
    uint64_t count() {
        uint64_t n = 0;
        FastxChunk c;
        while (src_->pop(c)) {
            n += c.reads();
            src_->recycle(std::move(c));
            c = FastxChunk();
        }
        return n;
    }

private:
    struct Slice { size_t chunk; uint64_t first, take, base_off, read_off; };
    std::unique_ptr<FastxSource> src_;
    unsigned threads_ = 1;
    std::vector<FastxChunk> held_;  // chunks the current batch was cut from (alive until the next batch: names)
    std::vector<Slice> slices_;
    uint64_t cur_at_ = 0;           // first unused read of the last held chunk, if it is only partly used
    bool cur_partial_ = false;
};

}  // namespace fg
