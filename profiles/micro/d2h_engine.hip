// Which engine carries a device-to-host copy? A copy that makes progress while a kernel holds every wave slot of the chip is done
// by a copy engine (SDMA); one that has to wait for the kernel is a shader copy (__amd_rocclr_copyBuffer), which competes with the
// kernels of the pipeline for the CUs. The streamed worker loop has 5 workers x 4 streams: does the number of streams, the thread
// that issues the copy, or a kernel having written the source matter?
//   hipcc --offload-arch=gfx950 -O3 -o d2h_engine d2h_engine.hip -lhsa-runtime64 && ./d2h_engine
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
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
    // every hardware queue busy: a short kernel is pending on each of the 24 streams (the runtime multiplexes them onto its hardware
    // queues) when the copy out is issued on a kernel-free stream of normal / of high priority
    {
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        hipStream_t s_hi, s_hi_in;
        CK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, greatest));
        CK(hipStreamCreateWithPriority(&s_hi_in, hipStreamNonBlocking, greatest));
        uint32_t* d_sinks;
        CK(hipMalloc((void**)&d_sinks, 4 * 32));
        auto busy = [&] { for (size_t i = 0; i < many.size(); ++i) hipLaunchKernelGGL(k_spin, dim3(8), dim3(256), 0, many[i], d_sinks + i, 200000u); };
        busy(); probe("D2H, kernel-free stream of NORMAL priority, kernels pending on 24 other streams", s_out3, h_b, d_b, hipMemcpyDeviceToHost);
        CK(hipDeviceSynchronize());
        busy(); probe("D2H, kernel-free stream of HIGH priority (range %d..%d), kernels pending on 24 other streams", s_hi, h_b, d_b, hipMemcpyDeviceToHost);
        CK(hipDeviceSynchronize());
        busy(); probe("H2D, kernel-free stream of NORMAL priority, kernels pending on 24 other streams", s_in, d_a, h_a, hipMemcpyHostToDevice);
        CK(hipDeviceSynchronize());
        printf("(stream priorities: least %d, greatest %d)\n", least, greatest);
    }
    // the copy out through the HSA runtime on an engine of our choice: rate per engine, alone and beside 2 x 20 H2D pieces of the HIP runtime
    {
        hsa_amd_pointer_info_t pi;
        memset(&pi, 0, sizeof pi);
        pi.size = sizeof pi;
        hsa_agent_t gpu{}, cpu{};
        if (hsa_amd_pointer_info(d_b, &pi, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS) gpu = pi.agentOwner;
        memset(&pi, 0, sizeof pi);
        pi.size = sizeof pi;
        if (hsa_amd_pointer_info(h_b, &pi, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS) cpu = pi.agentOwner;
        uint32_t mask = 0, rec = 0;
        hsa_status_t st = hsa_amd_memory_copy_engine_status(cpu, gpu, &mask);
        hsa_amd_memory_get_preferred_copy_engine(cpu, gpu, &rec);
        printf("HSA: gpu agent %llx, cpu agent %llx, D2H engine status %d, free mask 0x%x, preferred mask 0x%x\n", (unsigned long long)gpu.handle, (unsigned long long)cpu.handle, (int)st, mask, rec);
        uint32_t mask_in = 0;
        hsa_amd_memory_copy_engine_status(gpu, cpu, &mask_in);
        printf("HSA: H2D free mask 0x%x\n", mask_in);
        hsa_signal_t sig;
        hsa_signal_create(1, 0, nullptr, &sig);
        hipStream_t s_in2;
        CK(hipStreamCreateWithFlags(&s_in2, hipStreamNonBlocking));
        char* d_c;
        CK(hipMalloc((void**)&d_c, N));
        for (uint32_t e = 1; e <= 0x8000u; e <<= 1) {
            if (!(mask & e)) continue;
            double alone = 0, beside = 0, h2d_done = 0;
            bool failed = false;
            for (int rep = 0; rep < 3 && !failed; ++rep) {
                hsa_signal_store_relaxed(sig, 1);
                const double t0 = now_ms();
                if (hsa_amd_memory_async_copy_on_engine(h_b, cpu, d_b, gpu, N, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)e, false) != HSA_STATUS_SUCCESS) { failed = true; break; }
                hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
                alone = now_ms() - t0;
            }
            for (int rep = 0; rep < 3 && !failed; ++rep) {
                hsa_signal_store_relaxed(sig, 1);
                const double t0 = now_ms();
                const size_t step = N / 20;
                for (int i = 0; i < 20; ++i) {
                    CK(hipMemcpyAsync(d_a + i * step, h_a + i * step, step, hipMemcpyHostToDevice, s_in));
                    CK(hipMemcpyAsync(d_c + i * step, h_a + i * step, step, hipMemcpyHostToDevice, s_in2));
                }
                if (hsa_amd_memory_async_copy_on_engine(h_b, cpu, d_b, gpu, N, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)e, false) != HSA_STATUS_SUCCESS) { failed = true; break; }
                hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
                beside = now_ms() - t0;
                CK(hipStreamSynchronize(s_in));
                CK(hipStreamSynchronize(s_in2));
                h2d_done = now_ms() - t0;
            }
            if (failed) printf("HSA D2H on engine 0x%04x: refused\n", e);
            else printf("HSA D2H of 44 MB on engine 0x%04x: alone %.2f ms (%.1f GB/s); issued behind 2 x 20 H2D pieces (88 MB, HIP): D2H done after %.2f ms, H2D after %.2f ms\n", e, alone, N / alone / 1e6, beside, h2d_done);
        }
    }
    // as in the worker loop: the copy in of two other batches is queued in pieces on two more streams when the copy out is issued
    {
        hipStream_t s_in2;
        CK(hipStreamCreateWithFlags(&s_in2, hipStreamNonBlocking));
        char* d_c;
        CK(hipMalloc((void**)&d_c, N));
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, s_spin, d_sink, 200000u);
            const double t0 = now_ms();
            const size_t step = N / 20;
            for (int i = 0; i < 20; ++i) {
                CK(hipMemcpyAsync(d_a + i * step, h_a + i * step, step, hipMemcpyHostToDevice, s_in));
                CK(hipMemcpyAsync(d_c + i * step, h_a + i * step, step, hipMemcpyHostToDevice, s_in2));
            }
            CK(hipMemcpyAsync(h_b, d_b, N, hipMemcpyDeviceToHost, s_out3));
            CK(hipStreamSynchronize(s_out3));
            const double t1 = now_ms();
            CK(hipStreamSynchronize(s_in));
            CK(hipStreamSynchronize(s_in2));
            const double t2 = now_ms();
            CK(hipStreamSynchronize(s_spin));
            if (rep) printf("D2H issued behind 2 x 20 H2D pieces on two other streams: D2H done after %.2f, the H2D after %.2f ms, kernel after %.2f ms\n", t1 - t0, t2 - t0, now_ms() - t0);
        }
    }
    // five threads, each: kernel on its own stream, synchronise, D2H on its own kernel-free stream (no slot-filling kernel: wall time)
    {
        std::vector<std::thread> th;
        const double t0 = now_ms();
        for (int t = 0; t < 5; ++t) th.emplace_back([&, t] {
            CK(hipSetDevice(0));
            hipStream_t sk, so;
            CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
            CK(hipStreamCreateWithFlags(&so, hipStreamNonBlocking));
            char *d, *h;
            CK(hipMalloc((void**)&d, N));
            CK(hipHostMalloc((void**)&h, N, hipHostMallocDefault));
            for (int it = 0; it < 8; ++it) {
                hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, sk, (uint32_t*)d, N / 4, (uint32_t)it);
                CK(hipStreamSynchronize(sk));
                CK(hipMemcpyAsync(h, d, N, hipMemcpyDeviceToHost, so));
                CK(hipStreamSynchronize(so));
            }
        });
        for (auto& t : th) t.join();
        printf("5 threads x 8 x (fill kernel, sync, D2H of 44 MB on the thread's kernel-free stream): %.2f ms (count __amd_rocclr_copyBuffer under rocprofv3 --kernel-trace)\n", now_ms() - t0);
    }
    return 0;
}
