// What the host link gives on this box, in the shapes the streaming worker loop uses (profiles/e2e_stream.py):
//   * pinning cost: hipHostMalloc / hipHostFree per size;
//   * H2D and D2H of 80 MB out of / into pinned memory: one copy, and 20 / 80 pieces on one stream;
//   * both directions at once on two streams; two H2D at once; with a compute kernel running beside them;
//   * the same transfers done by a kernel that reads / writes the pinned host memory itself (no copy engine).
//   hipcc --offload-arch=gfx950 -O3 -o pcie_rates pcie_rates.hip && ./pcie_rates
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_spin(uint32_t* out, uint32_t iters) {  // keeps the CUs busy (ALU only)
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0x12345u) *out = x;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t N = 80u << 20;
    for (size_t mb : {1, 4, 16, 64, 256}) {
        void* p = nullptr;
        double t0 = now_ms();
        CK(hipHostMalloc(&p, mb << 20, hipHostMallocDefault));
        double t1 = now_ms();
        memset(p, 1, mb << 20);
        double t2 = now_ms();
        CK(hipHostFree(p));
        double t3 = now_ms();
        printf("hipHostMalloc %4zu MB: %.2f ms (%.3f ms/MB), first touch %.2f ms, hipHostFree %.2f ms\n", mb, t1 - t0, (t1 - t0) / mb, t2 - t1, t3 - t2);
    }
    char *h_in, *h_out, *h_in2;
    CK(hipHostMalloc((void**)&h_in, N, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_in2, N, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_out, N, hipHostMallocDefault));
    memset(h_in, 3, N); memset(h_in2, 4, N); memset(h_out, 0, N);
    char *d_a, *d_b, *d_c;
    CK(hipMalloc((void**)&d_a, N)); CK(hipMalloc((void**)&d_b, N)); CK(hipMalloc((void**)&d_c, N));
    uint32_t* d_sink;
    CK(hipMalloc((void**)&d_sink, 4));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    auto wall = [&](const char* name, double bytes, auto body) {
        body();
        hipDeviceSynchronize();
        double best = 1e9;
        for (int r = 0; r < 5; ++r) {
            double t0 = now_ms();
            body();
            hipDeviceSynchronize();
            best = std::min(best, now_ms() - t0);
        }
        printf("%-78s %7.3f ms  %6.1f GB/s\n", name, best, bytes / best / 1e6);
    };
    auto pieces = [&](hipStream_t s, char* dst, const char* src, int k, hipMemcpyKind kind) {
        const size_t step = N / k;
        for (int i = 0; i < k; ++i) (void)hipMemcpyAsync(dst + i * step, src + i * step, step, kind, s);
    };
    wall("H2D 80 MB, one copy", N, [&] { pieces(s1, d_a, h_in, 1, hipMemcpyHostToDevice); });
    wall("H2D 80 MB, 20 pieces of 4 MB on one stream", N, [&] { pieces(s1, d_a, h_in, 20, hipMemcpyHostToDevice); });
    wall("H2D 80 MB, 80 pieces of 1 MB on one stream", N, [&] { pieces(s1, d_a, h_in, 80, hipMemcpyHostToDevice); });
    wall("D2H 80 MB, one copy", N, [&] { pieces(s1, h_out, d_b, 1, hipMemcpyDeviceToHost); });
    wall("D2H 80 MB, 20 pieces", N, [&] { pieces(s1, h_out, d_b, 20, hipMemcpyDeviceToHost); });
    wall("H2D 80 MB + D2H 80 MB at once (two streams)", 2.0 * N, [&] { pieces(s1, d_a, h_in, 1, hipMemcpyHostToDevice); pieces(s2, h_out, d_b, 1, hipMemcpyDeviceToHost); });
    wall("H2D 80 MB (20 pieces) + D2H 80 MB at once (two streams)", 2.0 * N, [&] { pieces(s1, d_a, h_in, 20, hipMemcpyHostToDevice); pieces(s2, h_out, d_b, 1, hipMemcpyDeviceToHost); });
    wall("two H2D of 80 MB at once (two streams)", 2.0 * N, [&] { pieces(s1, d_a, h_in, 1, hipMemcpyHostToDevice); pieces(s2, d_c, h_in2, 1, hipMemcpyHostToDevice); });
    wall("H2D + D2H at once, ALU kernel on all CUs beside them", 2.0 * N, [&] {
        hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s3, d_sink, 200000u);
        pieces(s1, d_a, h_in, 1, hipMemcpyHostToDevice); pieces(s2, h_out, d_b, 1, hipMemcpyDeviceToHost); });
    wall("(the ALU kernel alone)", 0, [&] { hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s3, d_sink, 200000u); });
    {   // do the copies make progress while a kernel holds every wave slot? HIP events on the copy streams
        hipEvent_t a1, b1, a2, b2;
        hipEventCreate(&a1); hipEventCreate(&b1); hipEventCreate(&a2); hipEventCreate(&b2);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s3, d_sink, 200000u);
            hipEventRecord(a1, s1); pieces(s1, d_a, h_in, 1, hipMemcpyHostToDevice); hipEventRecord(b1, s1);
            hipEventRecord(a2, s2); pieces(s2, h_out, d_b, 1, hipMemcpyDeviceToHost); hipEventRecord(b2, s2);
            double t0 = now_ms();
            hipStreamSynchronize(s1); double t1 = now_ms();
            hipStreamSynchronize(s2); double t2 = now_ms();
            hipDeviceSynchronize(); double t3 = now_ms();
            float e1 = 0, e2 = 0;
            hipEventElapsedTime(&e1, a1, b1); hipEventElapsedTime(&e2, a2, b2);
            printf("beside a 13 ms ALU kernel that fills every wave slot: H2D done after %.2f ms (event %.2f ms), D2H after %.2f ms (event %.2f ms), kernel after %.2f ms\n",
                   t1 - t0, e1, t2 - t0, e2, t3 - t0);
        }
        // a kernel copy beside the same ALU kernel (has to wait for wave slots)
        hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s3, d_sink, 200000u);
        hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s1, (const u32x4*)h_in, (u32x4*)d_a, N / 16);
        double t0 = now_ms();
        hipStreamSynchronize(s1); double t1 = now_ms();
        hipDeviceSynchronize(); double t3 = now_ms();
        printf("beside the same kernel: a copy KERNEL H2D done after %.2f ms, ALU kernel after %.2f ms\n", t1 - t0, t3 - t0);
    }
    for (int blocks : {64, 256, 1024}) {
        char name[128];
        snprintf(name, sizeof name, "kernel reads pinned host memory -> HBM, 80 MB, %d blocks", blocks);
        wall(name, N, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s1, (const u32x4*)h_in, (u32x4*)d_a, N / 16); });
        snprintf(name, sizeof name, "kernel writes HBM -> pinned host memory, 80 MB, %d blocks", blocks);
        wall(name, N, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s1, (const u32x4*)d_b, (u32x4*)h_out, N / 16); });
    }
    wall("kernel H2D (256 blocks) + kernel D2H (256 blocks) at once", 2.0 * N, [&] {
        hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s1, (const u32x4*)h_in, (u32x4*)d_a, N / 16);
        hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s2, (const u32x4*)d_b, (u32x4*)h_out, N / 16); });
    wall("copy-engine H2D + kernel D2H (256 blocks) at once", 2.0 * N, [&] {
        pieces(s1, d_a, h_in, 1, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s2, (const u32x4*)d_b, (u32x4*)h_out, N / 16); });
    // NUMA: where do the pinned pages live relative to this thread?
    printf("(pinned buffers: hipHostMallocDefault; run under `numactl --cpunodebind=0/1` to see the placement effect)\n");
    return 0;
}
