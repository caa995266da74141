// Which engine carries a device-to-host copy? A copy that makes progress while a kernel holds every wave slot of the chip is done
// by a copy engine (SDMA); one that has to wait for the kernel is a shader copy (__amd_rocclr_copyBuffer), which competes with the
// kernels of the pipeline for the CUs. The streamed worker loop has 5 workers x 4 streams: does the number of streams, the thread
// that issues the copy, or a kernel having written the source matter?
//   hipcc --offload-arch=gfx950 -O3 -o d2h_engine d2h_engine.hip && ./d2h_engine
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_spin(uint32_t* out, uint32_t iters) {
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 0x12345u) *out = x;
}
__global__ __launch_bounds__(256) void k_fill(uint32_t* p, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (uint32_t)i;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static const size_t N = 44u << 20;
static uint32_t* d_sink;
static hipStream_t s_spin;

// copy `bytes` on stream s while the slot-filling kernel runs; returns ms until the copy is done, and ms until the kernel is done
static void probe(const char* name, hipStream_t s, void* dst, const void* src, hipMemcpyKind kind) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_spin, d_sink, 200000u);
        const double t0 = now_ms();
        CK(hipMemcpyAsync(dst, src, N, kind, s));
        CK(hipStreamSynchronize(s));
        const double t1 = now_ms();
        CK(hipStreamSynchronize(s_spin));
        const double t2 = now_ms();
        if (rep) printf("%-100s copy done after %6.2f ms, kernel after %6.2f ms -> %s\n", name, t1 - t0, t2 - t0, t1 - t0 < 0.5 * (t2 - t0) ? "copy engine" : "SHADER (waited for the kernel)");
    }
}

int main() {
    char *h_a, *h_b;
    CK(hipHostMalloc((void**)&h_a, N, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_b, N, hipHostMallocDefault));
    memset(h_a, 1, N); memset(h_b, 2, N);
    char *d_a, *d_b;
    CK(hipMalloc((void**)&d_a, N)); CK(hipMalloc((void**)&d_b, N));
    CK(hipMalloc((void**)&d_sink, 4));
    CK(hipStreamCreateWithFlags(&s_spin, hipStreamNonBlocking));
    hipStream_t s_in, s_out, s_k;
    CK(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
    probe("H2D, kernel-free stream, 4 streams in the process", s_in, d_a, h_a, hipMemcpyHostToDevice);
    probe("D2H, kernel-free stream, 4 streams in the process", s_out, h_b, d_b, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s_k, (uint32_t*)d_b, N / 4, 7u);
    CK(hipStreamSynchronize(s_k));
    probe("D2H, kernel-free stream, source just written by a kernel on another stream (synchronised)", s_out, h_b, d_b, hipMemcpyDeviceToHost);
    {
        std::thread t([&] { CK(hipSetDevice(0)); probe("D2H, kernel-free stream, issued by another thread", s_out, h_b, d_b, hipMemcpyDeviceToHost); });
        t.join();
    }
    // the stream had a kernel once (long finished)
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, s_out, (uint32_t*)d_a, 1024, 1u);
    CK(hipStreamSynchronize(s_out));
    probe("D2H on a stream that ran a kernel earlier (finished)", s_out, h_b, d_b, hipMemcpyDeviceToHost);
    // many streams, as the worker loop has (5 results x 4 streams + the index's)
    std::vector<hipStream_t> many(24);
    for (auto& s : many) {
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    }
    hipStream_t s_out2;
    CK(hipStreamCreateWithFlags(&s_out2, hipStreamNonBlocking));
    probe("D2H, fresh kernel-free stream, 24 more idle streams exist", s_out2, h_b, d_b, hipMemcpyDeviceToHost);
    for (auto& s : many) hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, s, (uint32_t*)d_a, 1024, 1u);
    CK(hipDeviceSynchronize());
    probe("D2H, the same stream, after each of the 24 streams ran a kernel", s_out2, h_b, d_b, hipMemcpyDeviceToHost);
    probe("H2D, kernel-free stream, after each of the 24 streams ran a kernel", s_in, d_a, h_a, hipMemcpyHostToDevice);
    hipStream_t s_out3;
    CK(hipStreamCreateWithFlags(&s_out3, hipStreamNonBlocking));
    probe("D2H, stream created after all that", s_out3, h_b, d_b, hipMemcpyDeviceToHost);
    // two D2H at once (two workers finish together)
    {
        hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_spin, d_sink, 200000u);
        const double t0 = now_ms();
        CK(hipMemcpyAsync(h_b, d_b, N, hipMemcpyDeviceToHost, s_out3));
        CK(hipMemcpyAsync(h_a, d_a, N, hipMemcpyDeviceToHost, s_out2));
        CK(hipStreamSynchronize(s_out3));
        const double t1 = now_ms();
        CK(hipStreamSynchronize(s_out2));
        const double t2 = now_ms();
        CK(hipStreamSynchronize(s_spin));
        printf("two D2H at once on two kernel-free streams: done after %.2f and %.2f ms, kernel after %.2f ms\n", t1 - t0, t2 - t0, now_ms() - t0);
    }
    // D2H and H2D and a second D2H at once
    {
        hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_spin, d_sink, 200000u);
        const double t0 = now_ms();
        CK(hipMemcpyAsync(d_a, h_a, N, hipMemcpyHostToDevice, s_in));
        CK(hipMemcpyAsync(h_b, d_b, N, hipMemcpyDeviceToHost, s_out3));
        CK(hipStreamSynchronize(s_out3));
        const double t1 = now_ms();
        CK(hipStreamSynchronize(s_in));
        const double t2 = now_ms();
        CK(hipStreamSynchronize(s_spin));
        printf("D2H + H2D at once: D2H done after %.2f, H2D after %.2f ms, kernel after %.2f ms\n", t1 - t0, t2 - t0, now_ms() - t0);
    }
    return 0;
}
