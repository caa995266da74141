// Store-pattern microbenchmark for k2b_expand: what does a write-only kernel reach when its stores look like the expansion
// kernel's — every wave writes runs of a few KB (one read's colours each) that begin at an arbitrary u32 of a CSR?
//  * block-contiguous fill (the ceiling) and the same shifted by 4 bytes;
//  * one run per wave at a time, the runs starting at multiples of 64 ... 1 u32 (the gap behind a run is never written), with
//    16-byte stores from the run's first u32 (MODE 0), a 16-byte-aligned body with scalar head and tail (MODE 1), 4-byte stores
//    (MODE 2), wave stores that cover whole 128-byte lines (MODE 3);
//  * the kernel's own shape: tickets of 32 consecutive runs per wave with the bench workload's result sizes, run by run or as
//    one stream of whole lines.
// Result (profiles/r4/k2b_stream_stores_r4b.txt): lines that are written in part cost (17 % at runs of 1570 u32, 35 % at 400),
// the width and alignment of the store instructions do not; a ticket as one stream reaches 5.26 against 4.65 TB/s.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

__global__ __launch_bounds__(256) void k_fill(uint32_t* __restrict__ out, size_t n16, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) *(u32x4_a4*)(out + 4 * i) = u32x4{v, v, v, v};
}

// runs: wave w of the grid takes runs w, w + nwaves, ...; run r covers u32 [off[r], off[r + 1]); MODE 0: 16-byte stores from the
// run's first u32 on (as k2b_expand does), MODE 1: scalar head up to the 16-byte boundary, aligned 16-byte body, scalar tail,
// MODE 2: 4-byte stores
template <int MODE>
__global__ __launch_bounds__(256) void k_runs(uint32_t* __restrict__ out, const uint64_t* __restrict__ off, size_t nruns, uint32_t v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = wave; r < nruns; r += nwaves) {
        uint64_t b = off[2 * r], e = off[2 * r + 1];
        if (MODE == 0) {
            uint64_t i = b + 4 * lane;
            for (; i + 4 <= e; i += 256) *(u32x4_a4*)(out + i) = u32x4{v, v, v, v};
            const uint64_t t = b + ((e - b) & ~3ull);
            if (t + lane < e) out[t + lane] = v;
        } else if (MODE == 1) {
            const uint64_t b4 = (b + 3) & ~3ull, e4 = e & ~3ull;
            if (b4 <= e4) {
                if (b + lane < b4) out[b + lane] = v;
                for (uint64_t i = b4 + 4 * lane; i < e4; i += 256) *(u32x4*)(out + i) = u32x4{v, v, v, v};
                if (e4 + lane < e) out[e4 + lane] = v;
            } else if (b + lane < e) out[b + lane] = v;
        } else if (MODE == 2) {
            for (uint64_t i = b + lane; i < e; i += 64) out[i] = v;
        } else {  // MODE 3: every wave store covers whole 128-byte lines: lanes in front of the run's first u32 / behind its last stay idle
            for (uint64_t j = (b & ~31ull) + 4 * lane; j < e; j += 256) {
                if (j >= b && j + 4 <= e) *(u32x4*)(out + j) = u32x4{v, v, v, v};
                else
                    for (int q = 0; q < 4; ++q)
                        if (j + q >= b && j + q < e) out[j + q] = v;
            }
        }
    }
}

// k2b_expand's shape: a wave takes TICKETS of 32 consecutive runs (reads) and writes them one after the other. STREAM = false: every
// run on its own, 16-byte stores from its first u32 (what the kernel does); STREAM = true: the ticket's runs as one stream in whole
// 128-byte lines (a run's last partial line waits for the head of the next run), partial lines only at the two ends of the ticket.
// (off[] = plain CSR here: run r = [off[r], off[r + 1]).)
// FAR: the tickets of neighbouring waves (one CU) lie 1/1024 of the buffer apart instead of side by side (address translation:
// the waves of a CU then write to as many pages as there are waves)
template <bool STREAM, bool FAR = false>
__global__ __launch_bounds__(256) void k_tickets(uint32_t* __restrict__ out, const uint64_t* __restrict__ off, size_t nruns, uint32_t v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t ntick = (nruns + 31) / 32, per = (ntick + 1023) / 1024;
    for (size_t t0 = wave; t0 < (FAR ? per * 1024 : ntick); t0 += nwaves) {
        const size_t tk = FAR ? (t0 % 1024) * per + t0 / 1024 : t0;
        if (tk >= ntick) continue;
        const size_t t = tk * 32;
        const size_t t1 = t + 32 < nruns ? t + 32 : nruns;
        if (STREAM) {
            const uint64_t b = off[t], e = off[t1];
            for (uint64_t j = (b & ~31ull) + 4 * lane; j < e; j += 256) {
                if (j >= b && j + 4 <= e) *(u32x4*)(out + j) = u32x4{v, v, v, v};
                else
                    for (int q = 0; q < 4; ++q)
                        if (j + q >= b && j + q < e) out[j + q] = v;
            }
        } else {
            for (size_t r = t; r < t1; ++r) {
                const uint64_t b = off[r], e = off[r + 1];
                uint64_t i = b + 4 * lane;
                for (; i + 4 <= e; i += 256) *(u32x4_a4*)(out + i) = u32x4{v, v, v, v};
                const uint64_t tl = b + ((e - b) & ~3ull);
                if (tl + lane < e) out[tl + lane] = v;
            }
        }
    }
}

static hipStream_t g_stream = nullptr;  // (argv[1] = n: a stream bound to the first n CUs of the CU mask)
template <class F>
void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, g_stream);
        launch();
        hipEventRecord(e1, g_stream);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-64s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best * 1e-6);
}

int main(int argc, char** argv) {
    if (argc > 1) {
        const int n = atoi(argv[1]);
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < n && c < 256; ++c) mask[c >> 5] |= 1u << (c & 31);
        if (hipExtStreamCreateWithCUMask(&g_stream, 8, mask) != hipSuccess) { printf("no CU mask\n"); return 1; }
        printf("stream bound to %d CUs\n", n);
    }
    const size_t bytes = 8ull << 30;
    uint32_t* a;
    hipMalloc(&a, bytes + 4096);
    hipMemset(a, 1, bytes + 4096);
    const size_t n16 = bytes / 16;
    for (int blocks : {2048, 8192}) {
        printf("grid %d x 256\n", blocks);
        timeit("block-contiguous fill, 16 B stores", bytes, [&] { k_fill<<<blocks, 256, 0, g_stream>>>(a, n16, 7); });
        timeit("the same, shifted by 4 bytes", bytes, [&] { k_fill<<<blocks, 256, 0, g_stream>>>(a + 1, n16, 7); });
    }
    // runs of `len` u32 (+- a third, seeded) whose starts are multiples of `align` u32 (1 = any u32: the CSR of the results;
    // the gap behind a run is never written)
    for (uint32_t len : {1570u, 400u, 4000u}) {
        for (uint32_t align : {64u, 32u, 16u, 8u, 4u, 1u}) {
            std::vector<uint64_t> off;  // run r = [off[2 r], off[2 r + 1])
            uint64_t at = 0, x = 88172645463325252ull, written = 0;
            while (true) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const uint64_t l = len * 2 / 3 + x % (len * 2 / 3 + 1);
                if ((at + l + align) * 4 > bytes) break;
                off.push_back(at);
                off.push_back(at + l);
                written += l;
                at = (at + l + align - 1) / align * align;
            }
            const size_t nruns = off.size() / 2;
            uint64_t* d_off;
            hipMalloc(&d_off, off.size() * 8);
            hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice);
            const double wb = (double)written * 4;
            char name[128];
            for (int blocks : {4096}) {
                snprintf(name, sizeof name, "runs of ~%u u32 at multiples of %u u32, 16 B stores from the start", len, align);
                timeit(name, wb, [&] { k_runs<0><<<blocks, 256, 0, g_stream>>>(a, d_off, nruns, 7); });
                if (len > 100) {
                    snprintf(name, sizeof name, "runs of ~%u u32 at multiples of %u u32, line-aligned wave stores", len, align);
                    timeit(name, wb, [&] { k_runs<3><<<blocks, 256, 0, g_stream>>>(a, d_off, nruns, 7); });
                }
                snprintf(name, sizeof name, "runs of ~%u u32 at multiples of %u u32, 4 B stores", len, align);
                timeit(name, wb, [&] { k_runs<2><<<blocks, 256, 0, g_stream>>>(a, d_off, nruns, 7); });
            }
            hipFree(d_off);
        }
    }
    // the bench workload's result sizes: 8 % empty, 52 % with 1..16 colours, 40 % large (129..4546, mean 1570), as a plain CSR
    {
        std::vector<uint64_t> off;
        uint64_t at = 0, x = 1234567ull;
        while (at * 4 + 20000 < bytes) {
            off.push_back(at);
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const uint32_t u = x % 100;
            uint64_t l = 0;
            if (u >= 8 && u < 60) l = 1 + (x >> 20) % 16;
            else if (u >= 60) l = 129 + (x >> 20) % 2883;
            at += l;
        }
        off.push_back(at);
        const size_t nruns = off.size() - 1;
        uint64_t* d_off;
        hipMalloc(&d_off, off.size() * 8);
        hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice);
        const double wb = (double)at * 4;
        for (int blocks : {2048, 4096}) {
            char name[128];
            snprintf(name, sizeof name, "tickets of 32 reads (bench sizes), run by run (grid %d)", blocks);
            timeit(name, wb, [&] { k_tickets<false><<<blocks, 256, 0, g_stream>>>(a, d_off, nruns, 7); });
            snprintf(name, sizeof name, "tickets of 32 reads (bench sizes), one stream of whole lines (grid %d)", blocks);
            timeit(name, wb, [&] { k_tickets<true><<<blocks, 256, 0, g_stream>>>(a, d_off, nruns, 7); });
            snprintf(name, sizeof name, "tickets of 32 reads, run by run, neighbouring waves far apart (grid %d)", blocks);
            timeit(name, wb, [&] { k_tickets<false, true><<<blocks, 256, 0, g_stream>>>(a, d_off, nruns, 7); });
        }
    }
    return 0;
}
