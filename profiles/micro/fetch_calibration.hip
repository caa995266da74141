// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against kernels whose HBM-side byte count is known, in the
// access patterns of this engine (MI355X_MICROARCH.md: only wide coalesced streaming reads are established at 1/2).
//   read side : stream16 (16 B per lane, coalesced), stream4 (4 B per lane, coalesced), bucket64 (one random 64-byte bucket per
//               lane, as k1_lookup), desc32 (one random 32-byte descriptor per lane, as the set_desc gather), row576 (a random
//               576-byte row per wave, 36 lanes x 16 B, as the bitmap lists / result rows), word4 (one random 4-byte word per lane)
//   write side: fill16 (16 B per lane, coalesced), row576w (a 576-byte row per wave at consecutive rows, nontemporal, as k2a's
//               result rows), csr4 (consecutive runs of 1..64 words per wave at 4-byte alignment, as k2b's output), word4w (random 4 B)
// Every buffer is 4 GB (far beyond the 256 MB Infinity Cache) and every random address is touched once per launch.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calibration fetch_calibration.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calibration     (and again with --pmc WRITE_SIZE)
// The program prints the known byte counts; profiles/fetch_calibration.py divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// a permutation of [0, 2^bits): multiply by an odd constant and xorshift, so that every granule is touched exactly once
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t bits) {
    const uint32_t mask = bits == 32 ? 0xFFFFFFFFu : (1u << bits) - 1u;
    i = (i * 0x9E3779B1u) & mask;
    i ^= i >> (bits / 2 + 1);
    i = (i * 0x85EBCA6Bu) & mask;
    i ^= i >> (bits / 2);
    return i & mask;
}

__global__ void k_stream16(const u32x4* __restrict__ p, uint64_t n, uint32_t* sink) {
    u32x4 a = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a ^= p[i];
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) *sink = 1;
}
__global__ void k_stream4(const uint32_t* __restrict__ p, uint64_t n, uint32_t* sink) {
    uint32_t a = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a ^= p[i];
    if (a == 0x12345u) *sink = 1;
}
template <int V16>  // one random granule of V16 x 16 bytes per lane
__global__ void k_gather(const u32x4* __restrict__ p, uint32_t bits, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32x4* q = p + (size_t)perm(i, bits) * V16;
    u32x4 a = {0, 0, 0, 0};
#pragma unroll
    for (int v = 0; v < V16; ++v) a ^= q[v];
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) *sink = 1;
}
__global__ void k_word4(const uint32_t* __restrict__ p, uint32_t bits, uint32_t* sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (p[perm(i, bits)] == 0x12345u) *sink = 1;
}
__global__ void k_row576(const u32x4* __restrict__ p, uint32_t bits, uint32_t* sink) {  // wave = one random row of 36 x 16 B
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    u32x4 a = {0, 0, 0, 0};
    if (lane < 36) a = p[(size_t)perm(wave, bits) * 36 + lane];
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) *sink = 1;
}
__global__ void k_fill16(u32x4* __restrict__ p, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = u32x4{1, 2, 3, 4};
}
__global__ void k_row576w(u32x4* __restrict__ p, uint32_t rows) {  // consecutive rows, nontemporal
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave < rows && lane < 36) __builtin_nontemporal_store(u32x4{1, 2, 3, 4}, &p[(size_t)wave * 36 + lane]);
}
__global__ void k_csr4(uint32_t* __restrict__ p, uint32_t waves) {  // wave w writes a run of 1 + (w % 64) words; the runs abut (4-byte aligned, as a CSR)
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t m = wave & 63u;
    if (wave < waves && lane < 1 + m) p[(size_t)(wave >> 6) * 2080 + m * (m + 1) / 2 + lane] = lane;
}
__global__ void k_word4w(uint32_t* __restrict__ p, uint32_t bits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    p[perm(i, bits)] = i;
}

int main() {
    const size_t bytes = 4ull << 30;
    void* buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0x5a, bytes);
    uint32_t* sink;
    hipMalloc(&sink, 4);
    hipDeviceSynchronize();
    const uint32_t N = 1u << 24;  // lanes (or waves) of the random kernels
    printf("known bytes per launch (useful bytes | bytes of the 64-byte lines touched | of the 128-byte lines touched)\n");
    k_stream16<<<4096, 256>>>((const u32x4*)buf, bytes / 16, sink);
    printf("k_stream16 read %zu %zu %zu\n", bytes, bytes, bytes);
    k_stream4<<<4096, 256>>>((const uint32_t*)buf, bytes / 4, sink);
    printf("k_stream4 read %zu %zu %zu\n", bytes, bytes, bytes);
    k_gather<4><<<N / 256, 256>>>((const u32x4*)buf, 26, sink);  // 2^26 buckets of 64 B = 4 GB; 2^24 of them read
    printf("k_gather<4> read %zu %zu %zu\n", (size_t)N * 64, (size_t)N * 64, (size_t)N * 128);
    k_gather<2><<<N / 256, 256>>>((const u32x4*)buf, 27, sink);  // 32-byte descriptors
    printf("k_gather<2> read %zu %zu %zu\n", (size_t)N * 32, (size_t)N * 64, (size_t)N * 128);
    k_gather<1><<<N / 256, 256>>>((const u32x4*)buf, 28, sink);  // 16-byte records
    printf("k_gather<1> read %zu %zu %zu\n", (size_t)N * 16, (size_t)N * 64, (size_t)N * 128);
    k_word4<<<N / 256, 256>>>((const uint32_t*)buf, 30, sink);
    printf("k_word4 read %zu %zu %zu\n", (size_t)N * 4, (size_t)N * 64, (size_t)N * 128);
    {
        const uint32_t rows = 1u << 21;  // 2^21 of the 2^22 rows of 576 B that fit 2.4 GB
        k_row576<<<rows / 4, 256>>>((const u32x4*)buf, 22, sink);
        printf("k_row576 read %zu %zu %zu\n", (size_t)rows * 576, (size_t)rows * 576, (size_t)rows * 640);
        k_row576w<<<rows / 4, 256>>>((u32x4*)buf, rows);
        printf("k_row576w write %zu %zu %zu\n", (size_t)rows * 576, (size_t)rows * 576, (size_t)rows * 576);
    }
    k_fill16<<<4096, 256>>>((u32x4*)buf, bytes / 16);
    printf("k_fill16 write %zu %zu %zu\n", bytes, bytes, bytes);
    {
        const uint32_t waves = 1u << 22;
        size_t useful = 0;
        for (uint32_t w = 0; w < waves; ++w) useful += 4 * (1 + (w & 63u));
        k_csr4<<<waves / 4, 256>>>((uint32_t*)buf, waves);
        printf("k_csr4 write %zu %zu %zu\n", useful, useful, useful);
    }
    k_word4w<<<N / 256, 256>>>((uint32_t*)buf, 30);
    printf("k_word4w write %zu %zu %zu\n", (size_t)N * 4, (size_t)N * 64, (size_t)N * 128);
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
