// Harness for tests/test_driver_cpu.py::test_reader_under_sanitizers: the query reader alone (plain and block-compressed), read three
// times over with a pool of threads, built with -fsanitize=thread and -fsanitize=address,undefined.
#include "host/fastx_reader.hpp"
#include <cstdio>
struct Buf { std::vector<char> v; size_t n = 0; void clear() { n = 0; } void reserve(size_t w) { if (v.size() < w + 1024) v.resize(w + 1024); } char* data() { return v.data(); } void set_size(size_t l) { n = l; } };
int main(int argc, char** argv) {
    for (int rep = 0; rep < 3; ++rep) {
        fg::FastxReader r(argv[1], (unsigned)atoi(argv[2]), argc > 3 ? strtoull(argv[3], 0, 10) : 0, argc > 4 ? strtoull(argv[4], 0, 10) : ~0ULL);
        Buf b[4]; std::vector<uint64_t> offs[4];
        uint64_t reads = 0, bases = 0, h = 0; int cur = 0;
        while (r.next(5000, b[cur], offs[cur])) {
            reads += offs[cur].size() - 1; bases += b[cur].n;
            for (size_t i = 0; i < b[cur].n; i += 97) h = h * 1315423911u + (unsigned char)b[cur].data()[i];
            std::vector<char> nm; std::vector<uint64_t> no; r.names(nm, no);
            cur = (cur + 1) % 4;
        }
        printf("reads %llu bases %llu hash %llx\n", (unsigned long long)reads, (unsigned long long)bases, (unsigned long long)h);
    }
}
