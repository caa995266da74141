// Streaming-rate microbenchmark: what a kernel that only reads, only writes, or copies reaches on this part.
// The expand kernel (k2b) is write-dominated and the lookup / intersect kernels read-dominated; these are their ceilings.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o hbm_rates hbm_rates.hip && ./hbm_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fill(u32x4* __restrict__ out, size_t n, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = u32x4{v, v, v, v};
}
__global__ __launch_bounds__(256) void k_fill_nt(u32x4* __restrict__ out, size_t n, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        __builtin_nontemporal_store(u32x4{v, v, v, v}, &out[i]);
}
__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ out, size_t n, uint32_t v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ in, size_t n, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc ^= in[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = 1;
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

template <class F>
void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best * 1e-6);
}

int main() {
    const size_t bytes = 4ull << 30;
    void *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    uint32_t* sink;
    hipMalloc(&sink, 4);
    const size_t n16 = bytes / 16;
    for (int blocks : {2048, 8192, 32768}) {
        printf("grid %d x 256\n", blocks);
        timeit("fill 16 B stores", bytes, [&] { k_fill<<<blocks, 256>>>((u32x4*)a, n16, 7); });
        timeit("fill 16 B nontemporal stores", bytes, [&] { k_fill_nt<<<blocks, 256>>>((u32x4*)a, n16, 7); });
        timeit("fill 4 B stores", bytes, [&] { k_fill_u32<<<blocks, 256>>>((uint32_t*)a, bytes / 4, 7); });
        timeit("read 16 B loads", bytes, [&] { k_read<<<blocks, 256>>>((const u32x4*)a, n16, sink); });
        timeit("copy (read + write bytes)", 2.0 * bytes, [&] { k_copy<<<blocks, 256>>>((const u32x4*)a, (u32x4*)b, n16); });
    }
    timeit("hipMemsetAsync", bytes, [&] { hipMemsetAsync(a, 3, bytes, 0); });
    timeit("hipMemcpyAsync D2D (r + w)", 2.0 * bytes, [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
