// Random-gather microbenchmark: what dictionary layouts can deliver per lookup on this part.
// One "lookup" is what one lane of the lookup kernel does for one minimizer run of a read.
//   A  8-byte slot from a 64 MB table, then a dependent 24-byte string window from a 27 MB array (round-1 layout, without the pilot)
//   B  16-bit pilot from a 2.6 MB table (L2), then a dependent 64-byte bucket from a table of S MB
//   C  64-byte bucket from a table of S MB, no pilot (open addressing)
//   D  16-byte record from a table of S MB, no pilot
//   hipcc --offload-arch=gfx950 -O3 -o gather_rates gather_rates.hip && ./gather_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }

constexpr int PER_LANE = 16;  // lookups per lane, independent (as the waves of the real kernel overlap)

__global__ __launch_bounds__(256, 8) void k_A(const uint64_t* __restrict__ slots, uint32_t ns, const uint64_t* __restrict__ str, uint32_t nw,
                                             uint32_t* sink, uint32_t salt) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc = 0;
    for (int i = 0; i < PER_LANE; ++i) {
        const uint32_t h = mix32(gid * PER_LANE + i + salt);
        const uint64_t e = slots[mulhi(h, ns)];
        const uint32_t p = mulhi(mix32(h ^ (uint32_t)e), nw - 3);
        acc ^= str[p] ^ str[p + 1] ^ str[p + 2];
    }
    if (acc == 0x1234567u) *sink = 1;
}
template <bool PILOT, int WORDS16>
__global__ __launch_bounds__(256, 8) void k_B(const uint16_t* __restrict__ pilots, uint32_t np, const u32x4* __restrict__ tab, uint32_t nb,
                                             uint32_t* sink, uint32_t salt) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < PER_LANE; ++i) {
        uint32_t h = mix32(gid * PER_LANE + i + salt);
        if (PILOT) h ^= pilots[mulhi(h, np)] * 0x9E3779B1u;
        const u32x4* b = tab + (size_t)mulhi(mix32(h), nb) * WORDS16;
#pragma unroll
        for (int w = 0; w < WORDS16; ++w) acc ^= b[w];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = 1;
}

template <class F>
float timeit(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(0);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        launch(r + 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t big = 2ull << 30;
    void* tab;
    hipMalloc(&tab, big);
    hipMemset(tab, 0x5a, big);
    uint16_t* pilots;
    hipMalloc(&pilots, 4 << 20);
    hipMemset(pilots, 0x11, 4 << 20);
    uint64_t* str;
    hipMalloc(&str, 32 << 20);
    hipMemset(str, 0x22, 32 << 20);
    uint32_t* sink;
    hipMalloc(&sink, 4);
    const uint32_t blocks = 16384;  // 4 M lanes x 16 = 67 M lookups
    const double lookups = (double)blocks * 256 * PER_LANE;
    {
        const uint32_t ns = (64u << 20) / 8, nw = (27u << 20) / 8;
        float ms = timeit([&](int s) { k_A<<<blocks, 256>>>((const uint64_t*)tab, ns, str, nw, sink, s * 77777u); });
        printf("A slot8(64MB)->string24(27MB)            %8.3f ms  %6.1f G lookups/s\n", ms, lookups / ms * 1e-6);
    }
    const uint32_t np = 1300000;
    for (size_t mb : {128, 192, 256, 384, 512, 1024, 2048}) {
        const uint32_t nb = (uint32_t)((mb << 20) / 64);
        float ms = timeit([&](int s) { k_B<true, 4><<<blocks, 256>>>(pilots, np, (const u32x4*)tab, nb, sink, s * 77777u); });
        printf("B pilot(2.6MB)->bucket64(%4zu MB)          %8.3f ms  %6.1f G lookups/s  %6.1f GB/s of 64-B lines\n", mb, ms, lookups / ms * 1e-6,
               lookups * 64 / ms * 1e-6);
        ms = timeit([&](int s) { k_B<false, 4><<<blocks, 256>>>(pilots, np, (const u32x4*)tab, nb, sink, s * 77777u); });
        printf("C bucket64(%4zu MB)                        %8.3f ms  %6.1f G lookups/s  %6.1f GB/s of 64-B lines\n", mb, ms, lookups / ms * 1e-6,
               lookups * 64 / ms * 1e-6);
        const uint32_t nr = (uint32_t)((mb << 20) / 16);
        ms = timeit([&](int s) { k_B<false, 1><<<blocks, 256>>>(pilots, np, (const u32x4*)tab, nr, sink, s * 77777u); });
        printf("D record16(%4zu MB)                        %8.3f ms  %6.1f G lookups/s\n", mb, ms, lookups / ms * 1e-6);
        ms = timeit([&](int s) { k_B<false, 2><<<blocks, 256>>>(pilots, np, (const u32x4*)tab, nr / 2, sink, s * 77777u); });
        printf("E bucket32(%4zu MB)                        %8.3f ms  %6.1f G lookups/s\n", mb, ms, lookups / ms * 1e-6);
    }
    return 0;
}
