// How should a parser thread bring the text of its byte range into reach? (host side of the command-line path; no GPU involved)
//   pread of the whole 8 MB range into the thread's window, then parse (what the reader does)
//   pread in pieces of W bytes (the window stays in the core's L2), parse piece by piece
//   map the range (MAP_POPULATE), parse, unmap
// Every variant parses four-line FASTQ records of 150 bases into a chunk (bases + offsets), T threads taking ranges off a counter.
//   g++ -O3 -std=c++17 -pthread -o reader_pieces reader_pieces.cpp -lz && ./reader_pieces [million reads]
#include "../../fulgor_amd/csrc/host/fastx_reader.hpp"
#include <chrono>
using namespace fg;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const uint64_t n_reads = (argc > 1 ? atoll(argv[1]) : 10) * 1000000ull;
    const size_t REC = 316, RANGE_RECS = 26000;  // 8.2 MB per range
    char path[64];
    snprintf(path, sizeof path, "/dev/shm/reader_pieces_%d.fq", (int)getpid());
    {
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<char> block(REC * 10000);
        uint64_t x = 42;
        for (uint64_t r0 = 0; r0 < n_reads; r0 += 10000) {
            for (uint64_t i = 0; i < 10000; ++i) {
                char* p = block.data() + i * REC;
                snprintf(p, 13, "@r%09llu\n", (unsigned long long)(r0 + i));
                for (int j = 0; j < 150; ++j) { x = x * 6364136223846793005ull + 1442695040888963407ull; p[12 + j] = "ACGT"[x >> 62]; }
                memcpy(p + 162, "\n+\n", 3);
                memset(p + 165, 'I', 150);
                p[315] = '\n';
            }
            if (write(fd, block.data(), block.size()) != (ssize_t)block.size()) return 1;
        }
        close(fd);
    }
    int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const uint64_t n = (uint64_t)st.st_size;
    printf("%llu reads, %.2f GB of text on tmpfs, %u hardware threads\n", (unsigned long long)n_reads, n / 1e9, std::thread::hardware_concurrency());
    auto run = [&](const char* name, int nth, size_t W, int mode) {
        double best = 1e9;
        uint64_t got_total = 0;
        for (int rep = 0; rep < 4; ++rep) {
            std::atomic<uint64_t> next{0}, total{0};
            const double t0 = now_ms();
            std::vector<std::thread> th;
            for (int t = 0; t < nth; ++t) th.emplace_back([&] {
                char* win = mode == 0 ? (char*)malloc(W) : nullptr;
                FastxChunk c;
                c.want_names = false;
                for (;;) {
                    const uint64_t r = next++;
                    const uint64_t a = r * RANGE_RECS * REC, b = std::min<uint64_t>(n, a + RANGE_RECS * REC);
                    if (a >= n) break;
                    c.clear();
                    c.bases.reserve((b - a) / 2 + 64);
                    c.offs.reserve((b - a) / 256 + 16);
                    if (mode == 0) {
                        for (uint64_t p = a; p < b; p += W) {
                            const size_t len = std::min<uint64_t>(W, b - p);
                            size_t got = 0;
                            while (got < len) got += pread(fd, win + got, len - got, p + got);
                            uint64_t pos = 0;
                            if (parse_fastq4(win, 0, len, c) != len) abort();  // (pieces hold whole records here)
                        }
                    } else {
                        const uint64_t a0 = a & ~4095ull;
                        char* m = (char*)mmap(nullptr, b - a0, PROT_READ, MAP_SHARED | (mode == 1 ? MAP_POPULATE : 0), fd, a0);
                        const char* M = m - a0;
                        if (parse_fastq4(M, a, b, c) != b) abort();
                        munmap(m, b - a0);
                    }
                    total += c.reads();
                }
                free(win);
            });
            for (auto& t : th) t.join();
            best = std::min(best, now_ms() - t0);
            got_total = total.load();
        }
        printf("%-58s %3d threads: %7.1f ms  %6.1f M reads/s  %5.1f GB/s of text  (per thread %.2f GB/s)%s\n", name, nth, best, n_reads / best / 1e3, n / best / 1e6,
               n / best / 1e6 / nth, got_total == n_reads ? "" : "  WRONG COUNT");
    };
    for (int nth : {1, 8, 16, 24, 32, 48, 64}) {
        run("pread the 8 MB range, parse", nth, RANGE_RECS * REC, 0);
        run("pread pieces of 1 MB, parse each", nth, 3200 * REC, 0);
        run("pread pieces of 256 KB, parse each", nth, 800 * REC, 0);
        run("pread pieces of 64 KB, parse each", nth, 200 * REC, 0);
        run("map the range (populate), parse, unmap", nth, 0, 1);
        run("map the range (faults), parse, unmap", nth, 0, 2);
    }
    unlink(path);
    return 0;
}
