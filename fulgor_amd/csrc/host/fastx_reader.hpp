// FASTA / FASTQ (plain or gzip) ingestion for the worker loop (src/ps_utils.cpp:245-305: the reference feeds
// its workers from FQFeeder's parser threads; read id = position in the file). One background thread inflates
// and parses into chunks of reads while the caller's previous batch is on the GPU; sequences come out exactly as
// kseq delivers them to the reference: header line skipped, sequence lines concatenated without line ends, the
// '+' line and as many quality characters as there are bases skipped.
#pragma once
#include <zlib.h>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace fg {

class FastxReader {
public:
    struct Chunk {
        std::vector<char> bases;
        std::vector<uint64_t> offs{0};
        std::vector<char> names;  // record names (header up to the first blank, as kseq's name), concatenated
        std::vector<uint64_t> name_offs{0};
        uint64_t reads() const { return offs.size() - 1; }
    };

    explicit FastxReader(const std::string& path, uint64_t chunk_reads = 1 << 16) : chunk_reads_(chunk_reads) {
        f_ = gzopen(path.c_str(), "rb");  // transparent for files that are not gzip
        if (!f_) throw std::runtime_error("cannot open " + path);
        gzbuffer(f_, 1 << 20);
        buf_.resize(1 << 22);
        worker_ = std::thread([this] { produce(); });
    }
    ~FastxReader() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_space_.notify_all();
        if (worker_.joinable()) worker_.join();
        if (f_) gzclose(f_);
    }

    // next batch of at most max_reads reads (at least one chunk unless the file is exhausted); false at end of file.
    // Buffers are recycled (the batch vectors by the caller, the chunk vectors through a pool): after the first
    // batches no fresh pages are touched.
    bool next(uint64_t max_reads, std::vector<char>& bases, std::vector<uint64_t>& offs, std::vector<char>* names = nullptr,
              std::vector<uint64_t>* name_offs = nullptr) {
        bases.clear();
        offs.assign(1, 0);
        if (names) { names->clear(); name_offs->assign(1, 0); }
        for (;;) {
            Chunk c;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_data_.wait(g, [this] { return !q_.empty() || done_; });
                if (!error_.empty()) throw std::runtime_error(error_);
                if (q_.empty()) break;
                if (offs.size() > 1 && offs.size() - 1 + q_.front().reads() > max_reads) break;
                c = std::move(q_.front());
                q_.pop_front();
            }
            cv_space_.notify_one();
            const uint64_t base = bases.size();
            bases.insert(bases.end(), c.bases.begin(), c.bases.end());
            for (size_t i = 1; i < c.offs.size(); ++i) offs.push_back(base + c.offs[i]);
            if (names) {
                const uint64_t nb = names->size();
                names->insert(names->end(), c.names.begin(), c.names.end());
                for (size_t i = 1; i < c.name_offs.size(); ++i) name_offs->push_back(nb + c.name_offs[i]);
            }
            c.bases.clear();
            c.offs.assign(1, 0);
            c.names.clear();
            c.name_offs.assign(1, 0);
            std::lock_guard<std::mutex> g(m_);
            if (pool_.size() < 16) pool_.push_back(std::move(c));
        }
        return offs.size() > 1;
    }

private:
    // refill the line buffer; returns false at end of input
    bool fill() {
        if (pos_ < len_) memmove(buf_.data(), buf_.data() + pos_, len_ - pos_);
        len_ -= pos_;
        pos_ = 0;
        if (len_ == buf_.size()) buf_.resize(buf_.size() * 2);  // a single line longer than the buffer
        const int got = gzread(f_, buf_.data() + len_, (unsigned)std::min<size_t>(buf_.size() - len_, 1u << 30));
        if (got < 0) throw std::runtime_error("read error (corrupt gzip stream?)");
        len_ += (size_t)got;
        return got > 0;
    }
    // next line without its terminator; false at end of input
    bool line(const char*& s, size_t& n) {
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
            Chunk c;
            const char* s;
            size_t n;
            bool have = line(s, n);
            while (have) {
                if (n == 0 || (s[0] != '>' && s[0] != '@')) { have = line(s, n); continue; }  // stray text before a header
                {
                    size_t e = 1;
                    while (e < n && s[e] != ' ' && s[e] != '\t') ++e;
                    c.names.insert(c.names.end(), s + 1, s + e);
                    c.name_offs.push_back(c.names.size());
                }
                // sequence lines up to the next header or the '+' separator
                uint64_t len = 0;
                while ((have = line(s, n)) && !(n && (s[0] == '>' || s[0] == '@' || s[0] == '+'))) {
                    c.bases.insert(c.bases.end(), s, s + n);
                    len += n;
                }
                c.offs.push_back(c.bases.size());
                if (have && s[0] == '+') {  // quality: as many characters as bases (may itself start with '@')
                    uint64_t q = 0;
                    while (q < len && (have = line(s, n))) q += n;
                    have = line(s, n);
                }
                if (c.reads() == chunk_reads_) push(c);
                {
                    std::lock_guard<std::mutex> g(m_);
                    if (stop_) return;
                }
            }
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
    void push(Chunk& c) {
        {
            std::unique_lock<std::mutex> g(m_);
            cv_space_.wait(g, [this] { return q_.size() < 8 || stop_; });
            q_.push_back(std::move(c));
        }
        cv_data_.notify_one();
        std::lock_guard<std::mutex> g(m_);
        if (pool_.empty()) {
            c = Chunk();
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
    std::deque<Chunk> q_;
    std::vector<Chunk> pool_;  // emptied chunks, capacity kept
    bool done_ = false, stop_ = false;
    std::string error_;
};

}  // namespace fg
