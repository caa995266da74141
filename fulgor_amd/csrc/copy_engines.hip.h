// The bulk copies of the streamed path (reads up, records down) on copy engines of OUR choice, through the HSA runtime that the
// HIP runtime sits on. Why not hipMemcpyAsync: the HIP runtime picks a copy engine per call by what looks free at that moment.
// Measured on MI355X (profiles/micro/d2h_engine.hip, profiles/r5/d2h_engine_r5.txt; runtime log of the worker loop):
//   * of the 16 engines only four carry 55 GB/s over PCIe, the others 7-13 GB/s (they are built for the links between GPUs) — and
//     the runtime put one copy in fifty on a slow one;
//   * it puts copies of both directions on the same engine: an engine works in order, so a 44 MB copy out waits behind the twenty
//     pieces of another batch's copy in, and the link runs in one direction at a time (67 of the 94 GB/s it carries both ways).
// Here: the fast engines are found once per process by timing a 4 MB copy on each; the copies in of a batch go to one of two of them
// (by worker), every copy out to a third (first in, first out: the batch that has to be written first is copied first); the lowest
// fast engine is left to the HIP runtime's own small copies. If anything about this fails, the HIP calls are used.
#pragma once
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

class CopyEngines {
public:
    static CopyEngines& get() { static CopyEngines c; return c; }
    // (call with the device current) true: the bulk copies of this device go through h2d() / d2h(). One device per process — the first
    // that asks (one process per GPU is how the command line and the bench run); indexes on other devices use the HIP calls.
    bool usable(int device) {
        std::call_once(once_, [this, device] { device_ = device; init(); });
        return device == device_ && ok_.load(std::memory_order_relaxed);
    }
    hsa_signal_t new_signal() {
        hsa_signal_t s{0};
        if (hsa_signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) s.handle = 0;
        return s;
    }
    void free_signal(hsa_signal_t s) { if (s.handle) (void)hsa_signal_destroy(s); }
    // `copies` copies will count the signal down to zero
    void arm(hsa_signal_t s, int64_t copies) { hsa_signal_store_relaxed(s, copies); }
    // (a copy that was armed for but not issued)
    void disarm(hsa_signal_t s, int64_t copies) { if (copies) hsa_signal_subtract_relaxed(s, copies); }
    bool h2d(void* dst, const void* src, size_t n, hsa_signal_t s, unsigned lane) {
        return hsa_amd_memory_async_copy_on_engine(dst, gpu_, src, cpu_, n, 0, nullptr, s, (hsa_amd_sdma_engine_id_t)in_[lane % in_.size()], false) == HSA_STATUS_SUCCESS;
    }
    bool d2h(void* dst, const void* src, size_t n, hsa_signal_t s) {
        return hsa_amd_memory_async_copy_on_engine(dst, cpu_, src, gpu_, n, 0, nullptr, s, (hsa_amd_sdma_engine_id_t)out_, false) == HSA_STATUS_SUCCESS;
    }
    // returns when the armed copies have completed; throws when one of them failed (the runtime reports a failed asynchronous copy
    // by making the completion signal negative: what the copy was to bring must not be taken for there)
    void wait(hsa_signal_t s) {
        hsa_signal_value_t v;
        while ((v = hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED)) >= 1) {}
        if (v < 0) {
            disable();  // (the HIP calls from now on)
            throw std::runtime_error("a copy on a copy engine failed (completion signal " + std::to_string((long long)v) + ")");
        }
    }
    // the same without the verdict: on a path that is failing already, so that no engine still reads or writes a buffer that is let go
    void drain(hsa_signal_t s) noexcept {
        if (!s.handle) return;
        while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) >= 1) {
            if (hsa_signal_load_relaxed(s) >= 1 && ++spins_ > 30) break;  // (a minute: give up rather than hang on a dead engine)
        }
    }
    void disable() { ok_.store(false); }
    std::string report() const { return report_; }

private:
    std::once_flag once_;
    std::atomic<bool> ok_{false};
    std::atomic<unsigned> spins_{0};
    int device_ = -1;
    hsa_agent_t gpu_{0}, cpu_{0};
    std::vector<uint32_t> in_;
    uint32_t out_ = 0;
    std::string report_ = "copy engines: the HIP runtime's choice";

    // the engines that carry host traffic at full rate: those within half of the best (they differ by a factor of four from the
    // rest: 50 against 7-13 GB/s); three or four of them is what the hardware has, anything else is not trusted
public:
    static std::vector<uint32_t> classify(const std::vector<std::pair<uint32_t, double>>& rate) {
        double top = 0;
        for (auto& r : rate) top = std::max(top, r.second);
        std::vector<uint32_t> fast;
        for (auto& r : rate) if (r.second >= 0.5 * top) fast.push_back(r.first);
        if (fast.size() < 3 || fast.size() > 4) fast.clear();
        return fast;
    }

private:
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static bool owner_of(const void* p, hsa_agent_t& a) {
        hsa_amd_pointer_info_t pi;
        memset(&pi, 0, sizeof pi);
        pi.size = sizeof pi;
        if (hsa_amd_pointer_info(p, &pi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || pi.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
        a = pi.agentOwner;
        return a.handle != 0;
    }
    void init() {
        const char* e = getenv("FULGOR_COPY_ENGINES");  // 0: hipMemcpyAsync for everything (A/B measurements)
        if (e && e[0] == '0') return;
        constexpr size_t PROBE = 4u << 20;
        void *d = nullptr, *h = nullptr;
        hsa_signal_t sig{0};
        struct Release {
            void*& d; void*& h; hsa_signal_t& s;
            ~Release() { if (d) (void)hipFree(d); if (h) (void)hipHostFree(h); if (s.handle) (void)hsa_signal_destroy(s); }
        } release{d, h, sig};
        if (hipMalloc(&d, PROBE) != hipSuccess || hipHostMalloc(&h, PROBE, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipMemset(d, 0, PROBE) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return; }
        if (!owner_of(d, gpu_) || !owner_of(h, cpu_)) return;
        uint32_t mask = 0;
        if (hsa_amd_memory_copy_engine_status(cpu_, gpu_, &mask) != HSA_STATUS_SUCCESS || !mask) return;
        if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) { sig.handle = 0; return; }
        // What an earlier process of this user measured on this GPU (by PCI address and name) is taken over after ONE copy on the engine
        // for the records has confirmed its rate: timing all sixteen engines makes the runtime create a queue on each, 0.15 s of the
        // 0.3 s an index takes to open (profiles/r6/cli_cold_r6.txt). FULGOR_COPY_ENGINES_CACHE=0: always measure.
        std::string cache_path;
        {
            char bus[64] = {0};
            hipDeviceProp_t prop;
            const char* ce = getenv("FULGOR_COPY_ENGINES_CACHE");
            if (!(ce && ce[0] == '0') && hipDeviceGetPCIBusId(bus, sizeof bus, device_) == hipSuccess && hipGetDeviceProperties(&prop, device_) == hipSuccess) {
                std::string key = std::string(bus) + "_" + prop.name;
                for (char& c : key) if (!isalnum((unsigned char)c)) c = '_';
                cache_path = "/tmp/fulgor_amd_copy_engines_" + std::to_string((unsigned)getuid()) + "_" + key + ".txt";
            } else (void)hipGetLastError();
        }
        if (!cache_path.empty()) {
            const int cfd = ::open(cache_path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
            if (FILE* f = cfd >= 0 ? fdopen(cfd, "r") : nullptr) {
                unsigned in0 = 0, in1 = 0, out = 0, nin = 0;
                double rate = 0;
                char text[1024] = {0};
                const int got = fscanf(f, "v1 %u %x %x %x %lf\n", &nin, &in0, &in1, &out, &rate);
                if (got == 5 && fgets(text, sizeof text, f) && (nin == 1 || nin == 2) && out && (mask & out) && (mask & in0) && (nin == 1 || (mask & in1)) && rate > 0) {
                    double best = 0;
                    for (int rep = 0; rep < 2; ++rep) {
                        hsa_signal_store_relaxed(sig, 1);
                        const double t0 = now_ms();
                        if (hsa_amd_memory_async_copy_on_engine(h, cpu_, d, gpu_, PROBE, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)out, false) != HSA_STATUS_SUCCESS) { best = 0; break; }
                        try { wait(sig); } catch (std::exception&) { best = 0; break; }
                        best = std::max(best, PROBE / (now_ms() - t0) / 1e6);
                    }
                    if (best >= 0.5 * rate) {
                        fclose(f);
                        in_.assign(1, in0);
                        if (nin == 2) in_.push_back(in1);
                        out_ = out;
                        std::string t(text);
                        while (!t.empty() && (t.back() == '\n' || t.back() == '\r')) t.pop_back();
                        std::ostringstream os;
                        os << t << " (measured by an earlier process: " << cache_path << "; confirmed now: 0x" << std::hex << out << std::dec << " at " << (int)(best + 0.5) << " GB/s)";
                        report_ = os.str();
                        ok_.store(true);
                        return;
                    }
                }
                fclose(f);
                ok_.store(false);
            }
        }
        // One process per GPU probes at the same moment when a multi-GPU run starts: the probes take turns under a host-wide lock
        // (each is 64 copies of 4 MB, about 20 ms), so that a rank's engines are timed against an otherwise quiet host.
        int lock_fd = ::open("/tmp/fulgor_amd_copy_engines.lock", O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
        if (lock_fd >= 0 && flock(lock_fd, LOCK_EX) != 0) { ::close(lock_fd); lock_fd = -1; }
        struct Unlock { int fd; ~Unlock() { if (fd >= 0) { (void)flock(fd, LOCK_UN); ::close(fd); } } } unlock{lock_fd};
        std::vector<std::pair<uint32_t, double>> rate;  // engine, GB/s of a 4 MB copy out (best of four)
        for (uint32_t eng = 1; eng && eng <= mask; eng <<= 1) {
            if (!(mask & eng)) continue;
            double best = 0;
            for (int rep = 0; rep < 4; ++rep) {
                hsa_signal_store_relaxed(sig, 1);
                const double t0 = now_ms();
                if (hsa_amd_memory_async_copy_on_engine(h, cpu_, d, gpu_, PROBE, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)eng, false) != HSA_STATUS_SUCCESS) { best = 0; break; }
                try { wait(sig); } catch (std::exception&) { best = 0; break; }
                best = std::max(best, PROBE / (now_ms() - t0) / 1e6);
            }
            if (best > 0) rate.push_back({eng, best});
        }
        ok_.store(false);  // (wait() may have switched it off; it is decided below)
        std::ostringstream os;
        os << "copy engines (GB/s of a 4 MB copy out):";
        for (auto& r : rate) os << " 0x" << std::hex << r.first << std::dec << ":" << (int)(r.second + 0.5);
        const std::vector<uint32_t> fast = classify(rate);
        if (fast.size() >= 4) { in_ = {fast[1], fast[2]}; out_ = fast[3]; }
        else if (fast.size() == 3) { in_ = {fast[1]}; out_ = fast[2]; }
        else {
            // Ambiguous (a busy host, another process's traffic on the link): the engines that carry host traffic at full rate on every
            // MI355X measured so far are 0x1, 0x2, 0x4, 0x8; 0x1 stays with the HIP runtime's own small copies
            std::vector<uint32_t> dflt;
            for (uint32_t e : {0x2u, 0x4u, 0x8u})
                for (auto& r : rate) if (r.first == e) dflt.push_back(e);
            if (dflt.size() == 3) {
                in_ = {dflt[0], dflt[1]};
                out_ = dflt[2];
                os << "; classification ambiguous (" << fast.size() << " engines within half of the best): the default engines";
            } else {
                report_ = os.str() + "; classification ambiguous and the default engines are not there: the HIP runtime's choice";
                return;
            }
        }
        os << "; reads go up on";
        for (uint32_t x : in_) os << " 0x" << std::hex << x << std::dec;
        os << ", records come down on 0x" << std::hex << out_ << std::dec;
        report_ = os.str();
        ok_.store(true);
        if (!cache_path.empty()) {  // for the next process on this GPU (written whole, then renamed)
            double out_rate = 0;
            for (auto& r : rate) if (r.first == out_) out_rate = r.second;
            const std::string tmp = cache_path + "." + std::to_string((long)getpid());
            // (a new file of this user's, never through a link somebody else left under /tmp)
            const int tfd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
            if (FILE* f = tfd >= 0 ? fdopen(tfd, "w") : nullptr) {
                fprintf(f, "v1 %u %x %x %x %.1f\n%s\n", (unsigned)in_.size(), in_[0], in_.size() > 1 ? in_[1] : 0u, out_, out_rate, report_.c_str());
                fclose(f);
                if (rename(tmp.c_str(), cache_path.c_str()) != 0) (void)remove(tmp.c_str());
            }
        }
    }
};

}  // namespace
