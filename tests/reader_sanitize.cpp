// Harness for tests/test_driver_cpu.py::test_reader_under_sanitizers: the query reader alone (plain and block-compressed), read three
// times over with a pool of threads, built with -fsanitize=thread and -fsanitize=address,undefined. Every repetition goes through the
// copying batch interface (FastxReader::next, as the k-mer tools use it) and through the chunk interface the streaming worker loop uses:
// the record count walked on the open handle (count_records), then several consumer threads that take turns popping chunks under a
// mutex, hold them for a while and hand them back (pop_chunk / recycle_chunk), with the pooled slabs shared by all of it.
#include "host/fastx_reader.hpp"
#include <cstdio>
struct Buf { std::vector<char> v; size_t n = 0; void clear() { n = 0; } void reserve(size_t w) { if (v.size() < w + 1024) v.resize(w + 1024); } char* data() { return v.data(); } void set_size(size_t l) { n = l; } };
int main(int argc, char** argv) {
    const unsigned threads = (unsigned)atoi(argv[2]);
    const uint64_t begin = argc > 3 ? strtoull(argv[3], 0, 10) : 0, end = argc > 4 ? strtoull(argv[4], 0, 10) : ~0ULL;
    for (int rep = 0; rep < 3; ++rep) {
        uint64_t reads = 0, bases = 0, h = 0;
        {
            fg::FastxReader r(argv[1], threads, begin, end);
            Buf b[4]; std::vector<uint64_t> offs[4];
            int cur = 0;
            while (r.next(5000, b[cur], offs[cur])) {
                reads += offs[cur].size() - 1; bases += b[cur].n;
                for (size_t i = 0; i < b[cur].n; i += 97) h = h * 1315423911u + (unsigned char)b[cur].data()[i];
                std::vector<char> nm; std::vector<uint64_t> no; r.names(nm, no);
                cur = (cur + 1) % 4;
            }
        }
        {
            fg::FastxReader r(argv[1], threads, begin, end);
            r.set_want_names(false);
            uint64_t counted = 0;
            const bool can_count = r.count_records(counted);
            std::mutex mu;
            uint64_t creads = 0, cbases = 0, seq = 0;
            std::vector<std::pair<uint64_t, uint64_t>> hashes;  // (sequence number, hash of the chunk's bases)
            auto consumer = [&] {
                std::vector<fg::FastxChunk> held;
                for (;;) {
                    fg::FastxChunk c;
                    uint64_t my;
                    {
                        std::lock_guard<std::mutex> g(mu);
                        if (!r.pop_chunk(c)) break;
                        my = seq++;
                        creads += c.reads(); cbases += c.bases.size();
                    }
                    uint64_t hh = 0;
                    for (size_t i = 0; i < c.bases.size(); i += 97) hh = hh * 31 + (unsigned char)c.bases[i];
                    if (c.offs[0] != 0 || c.offs[c.reads()] != c.bases.size()) { printf("bad offsets\n"); exit(2); }
                    { std::lock_guard<std::mutex> g(mu); hashes.emplace_back(my, hh); }
                    held.push_back(std::move(c));
                    if (held.size() == 3) { for (auto& x : held) r.recycle_chunk(std::move(x)); held.clear(); }
                }
                for (auto& x : held) r.recycle_chunk(std::move(x));
            };
            std::vector<std::thread> th;
            for (int t = 0; t < 3; ++t) th.emplace_back(consumer);
            for (auto& t : th) t.join();
            if (creads != reads || cbases != bases || (can_count && counted != reads)) { printf("chunk interface disagrees: %llu / %llu reads, counted %llu\n", (unsigned long long)creads, (unsigned long long)reads, (unsigned long long)counted); return 3; }
        }
        printf("reads %llu bases %llu hash %llx\n", (unsigned long long)reads, (unsigned long long)bases, (unsigned long long)h);
    }
}
