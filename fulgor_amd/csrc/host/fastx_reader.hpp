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
//   * gzip streams cannot be entered in the middle: one thread inflates and parses.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <atomic>
#include <condition_variable>
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

struct FastxChunk {
    std::vector<char> bases;
    std::vector<uint64_t> offs{0};
    std::vector<char> names;  // record names (header up to the first blank, as kseq's name), concatenated
    std::vector<uint64_t> name_offs{0};
    uint64_t reads() const { return offs.size() - 1; }
    void clear() {
        bases.clear();
        offs.assign(1, 0);
        names.clear();
        name_offs.assign(1, 0);
    }
};

// kseq's record grammar over a stream of lines: next(s, n) yields the next line without its terminator
template <typename NextLine, typename Emit>
void parse_fastx_records(NextLine&& next, FastxChunk& c, Emit&& emit) {
    const char* s;
    size_t n;
    bool have = next(s, n);
    while (have) {
        if (n == 0 || (s[0] != '>' && s[0] != '@')) { have = next(s, n); continue; }  // stray text before a header
        {
            size_t e = 1;
            while (e < n && s[e] != ' ' && s[e] != '\t') ++e;
            c.names.insert(c.names.end(), s + 1, s + e);
            c.name_offs.push_back(c.names.size());
        }
        // sequence lines up to the next header or the '+' separator
        uint64_t len = 0;
        while ((have = next(s, n)) && !(n && (s[0] == '>' || s[0] == '@' || s[0] == '+'))) {
            c.bases.insert(c.bases.end(), s, s + n);
            len += n;
        }
        c.offs.push_back(c.bases.size());
        if (have && s[0] == '+') {  // quality: as many characters as bases (may itself start with '@')
            uint64_t q = 0;
            while (q < len && (have = next(s, n))) q += n;
            have = next(s, n);
        }
        if (!emit(c)) return;
    }
}

class FastxSource {
public:
    virtual ~FastxSource() {}
    virtual bool pop(FastxChunk& c) = 0;  // next chunk in file order; false at end of file
    virtual void recycle(FastxChunk&& c) = 0;
};

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
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) { close(fd_); throw std::runtime_error("cannot stat " + path); }
        size_ = (uint64_t)st.st_size;
        if (size_) {
            map_ = (const char*)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (map_ == MAP_FAILED) { close(fd_); throw std::runtime_error("cannot map " + path); }
            madvise((void*)map_, size_, MADV_SEQUENTIAL);
        }
        begin_ = std::min(begin, size_);
        end_ = std::min(end, size_);
        num_ranges_ = begin_ < end_ ? (end_ - begin_ + range_ - 1) / range_ : 0;
        window_ = 2 * std::max(1u, threads) + 2;
        for (unsigned t = 0; t < std::max(1u, threads) && t < num_ranges_; ++t) workers_.emplace_back([this] { work(); });
    }
    ~MappedFastxSource() override {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_space_.notify_all();
        cv_data_.notify_all();
        for (auto& w : workers_) w.join();
        // tearing down the page tables of a multi-gigabyte mapping that hundreds of threads have touched takes tens of
        // milliseconds (60-90 ms for 3 GB on a 256-thread host): nobody has to wait for it
        if (map_ && size_) {
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
        ++next_out_;
        g.unlock();
        cv_space_.notify_all();
        return true;
    }
    void recycle(FastxChunk&& c) override {
        c.clear();
        std::lock_guard<std::mutex> g(m_);
        if (pool_.size() < 64) pool_.push_back(std::move(c));
    }
    // first record boundary at or behind byte p (size_ if none). A record starts at a line start with '>', or with '@' when the
    // four lines from there look like a FASTQ record: third line '+...', fourth as long as the second (a quality line may
    // begin with '@', but then the line after next is a sequence). FASTQ files with wrapped sequences offer no such
    // boundaries: everything then falls to the first range, i.e. one thread parses the file with kseq's general grammar.
    uint64_t record_start(uint64_t p) const {
        if (p == 0) return 0;
        const char* nl = (const char*)memchr(map_ + p - 1, '\n', size_ - (p - 1));  // p - 1: p itself may be a line start
        uint64_t q = nl ? (uint64_t)(nl - map_) + 1 : size_;
        while (q < size_) {
            if (map_[q] == '>') return q;
            if (map_[q] == '@') {
                const char* fin = map_ + size_;
                const char* l1 = (const char*)memchr(map_ + q, '\n', size_ - q);                    // end of the header
                const char* l2 = l1 ? (const char*)memchr(l1 + 1, '\n', (size_t)(fin - l1 - 1)) : nullptr;  // end of the sequence
                if (!l2) return q;  // a truncated last record
                if (l2 + 1 < fin && l2[1] == '+') {
                    const char* l3 = (const char*)memchr(l2 + 1, '\n', (size_t)(fin - l2 - 1));   // end of the '+' line
                    if (!l3) return q;
                    const char* l4 = (const char*)memchr(l3 + 1, '\n', (size_t)(fin - l3 - 1));
                    size_t ls = (size_t)(l2 - l1 - 1), lq = (size_t)((l4 ? l4 : fin) - l3 - 1);
                    if (ls && l1[ls] == '\r') --ls;
                    if (lq && l3[lq] == '\r') --lq;
                    if (ls == lq) return q;
                }
            }
            const char* e = (const char*)memchr(map_ + q, '\n', size_ - q);
            q = e ? (uint64_t)(e - map_) + 1 : size_;
        }
        return size_;
    }
    uint64_t size() const { return size_; }

private:
    void work() {
        for (;;) {
            uint64_t r;
            FastxChunk c;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_space_.wait(g, [this] { return stop_ || next_in_ >= num_ranges_ || next_in_ < next_out_ + window_; });
                if (stop_ || next_in_ >= num_ranges_) return;
                r = next_in_++;
                if (!pool_.empty()) { c = std::move(pool_.back()); pool_.pop_back(); }
            }
            try {
                const uint64_t lo = record_start(begin_ + r * range_);
                const uint64_t hi = r + 1 == num_ranges_ ? (end_ == size_ ? size_ : record_start(end_)) : record_start(begin_ + (r + 1) * range_);
                uint64_t pos = lo;
                c.bases.reserve((hi - lo) / 2 + 64);
                parse_fastx_records(
                    [&](const char*& s, size_t& n) {
                        if (pos >= hi) return false;
                        const char* nl = (const char*)memchr(map_ + pos, '\n', hi - pos);
                        s = map_ + pos;
                        n = nl ? (size_t)(nl - s) : (size_t)(hi - pos);
                        pos += n + 1;
                        if (n && s[n - 1] == '\r') --n;
                        return true;
                    },
                    c, [](FastxChunk&) { return true; });
            } catch (std::exception& e) {
                std::lock_guard<std::mutex> g(m_);
                error_ = e.what();
            }
            {
                std::lock_guard<std::mutex> g(m_);
                done_[r] = std::move(c);
            }
            cv_data_.notify_all();
        }
    }

    int fd_ = -1;
    const char* map_ = nullptr;
    uint64_t size_ = 0, begin_ = 0, end_ = 0, range_, num_ranges_ = 0, window_ = 4;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_data_, cv_space_;
    std::map<uint64_t, FastxChunk> done_;  // parsed ranges waiting for their turn
    std::vector<FastxChunk> pool_;
    uint64_t next_in_ = 0, next_out_ = 0;
    bool stop_ = false;
    std::string error_;
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
        if (threads == 0) threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency() / 2));
        threads_ = threads;
        if (is_gzip_file(path)) {
            if (begin != 0 || end != ~0ULL) throw std::runtime_error("a gzip stream cannot be read in parts");
            src_.reset(new StreamFastxSource(path));
        } else {
            src_.reset(new MappedFastxSource(path, threads, begin, end));
        }
    }
    // Bases: clear(), reserve(bytes), data(), set_size(bytes)
    template <typename Bases>
    bool next(uint64_t max_reads, Bases& bases, std::vector<uint64_t>& offs) {
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
