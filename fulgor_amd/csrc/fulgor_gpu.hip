// libfulgor_gpu.so: C ABI (include/fulgor_gpu.h) over the HIP kernels. Host side = index ingestion,
// one-time upload to HBM, batch plumbing, HIP-event timing. No CPU execution path for queries: every
// query entry point launches the kernels in hip/kernels.hip.h or fails.
#include <hip/hip_runtime.h>
#include <unordered_map>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fulgor_gpu.h"
#include "host/fastx_reader.hpp"
#include "hip/kernels.hip.h"
#include "hip/dict_build.hip.h"
#include "host/formatters.hpp"
#include "host/index_io.hpp"
#include "copy_engines.hip.h"

using namespace fg;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // FULGOR_GUARD_ALLOC=1 (a debugging mode for the test suite): every buffer is exactly as long as asked for (rounded up to 256
    // bytes), ends where its mapping ends and is followed by 2 MB of reserved, unmapped addresses — a kernel that reads or writes past
    // a buffer faults there and then, instead of once in a dozen runs when the buffer happens to end a mapped block (the lookup
    // kernel's empty-ticket bug of round 5 hid that way for a round). HIP's virtual memory calls; plain structs, copied by value.
    // Address ranges are never handed back in this mode (see release()). Round 5: the whole GPU suite (130 tests) and the soak
    // against the oracle (profiles/soak_parity.py) run to their end in this mode: no kernel reads or writes past a buffer.
    void* guard_base = nullptr;
    size_t guard_mapped = 0, guard_reserved = 0;
    hipMemGenericAllocationHandle_t guard_handle{};
    static bool guard_mode() { static const bool g = getenv("FULGOR_GUARD_ALLOC") != nullptr; return g; }
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        release();
        static const bool trace = getenv("FULGOR_TRACE_ALLOC") != nullptr;  // (which buffer ends where a faulting address begins)
        if (guard_mode()) {
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = dev;
            size_t gran = 0;
            HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
            if (gran < 4096) gran = 2u << 20;
            const size_t want = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
            guard_mapped = (want + gran - 1) / gran * gran;
            guard_reserved = guard_mapped + gran;
            HIP_TRY(hipMemAddressReserve(&guard_base, guard_reserved, gran, nullptr, 0));
            HIP_TRY(hipMemCreate(&guard_handle, guard_mapped, &prop, 0));
            HIP_TRY(hipMemMap(guard_base, guard_mapped, 0, guard_handle, 0));
            hipMemAccessDesc acc{};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            HIP_TRY(hipMemSetAccess(guard_base, guard_mapped, &acc, 1));
            p = (char*)guard_base + (guard_mapped - want);
            cap = bytes;
            if (trace) fprintf(stderr, "[alloc] %p .. %p (%zu bytes for %zu asked, guarded) buffer object %p\n", p, (char*)p + want, want, bytes, (void*)this);
            return;
        }
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        if (trace) fprintf(stderr, "[alloc] %p .. %p (%zu bytes for %zu asked) buffer object %p\n", p, (char*)p + want, want, bytes, (void*)this);
    }
    void release() {
        if (guard_base) {
            (void)hipDeviceSynchronize();
            (void)hipMemUnmap(guard_base, guard_mapped);
            (void)hipMemRelease(guard_handle);
            // (the address range is NOT given back: a later buffer mapped at addresses that were unmapped a moment ago has faulted
            // inside its own range and returned zeros for its first pages on this driver — stale translations; the mode leaks
            // address space, of which a process has 128 TB)
            guard_base = nullptr;
        } else if (p) {
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

template <typename T>
void upload(DevBuf& b, const std::vector<T>& v, hipStream_t s) {
    b.ensure(std::max<size_t>(16, v.size() * sizeof(T)));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

constexpr size_t TICKET_BYTES = (2 * 8 + K2B_MAX_PARTS) * TICKET_STRIDE * sizeof(unsigned int);  // lookup, colour stage: 8 counters each; expansion: up to K2B_MAX_PARTS

const char* KERNEL_NAMES[FGPU_K_COUNT] = {"k1_lookup", "k2_intersect", "k3_union", "scan", "k2b_expand", "k_hits", "k_desc", "k_format", "k_order", "h2d", "d2h"};

// defaults of the fgpu_tune knobs from the environment (measurement: FULGOR_ORDER=0 takes the reads of a pass in file order,
// FULGOR_ORDER_MIN_READS sets the smallest pass that is ordered, FULGOR_SMALL=0 writes a bitmap row for every result)
uint64_t env_u64(const char* name, uint64_t dflt) {
    const char* e = getenv(name);
    return e && *e ? strtoull(e, nullptr, 10) : dflt;
}

}  // namespace

namespace {
// the reader's buffers (fg::HostVec, fg::SlabPool) in pinned host memory: H2D copies out of them run at PCIe speed and overlap
// with kernels; plain memory when there is no HIP device (host-only tools, CPU tests)
// opens whose device thread is still starting up (first stream, first pinned allocation, the copy engines' confirmation): pinning
// hundreds of megabytes at the same time doubles the time those first calls take (profiles/r6/cli_cold_r6.txt), so fgpu_prepare_host
// lets them finish first
std::atomic<int> g_device_startups{0};

void install_pinned_allocator() {
    static std::once_flag once;
    std::call_once(once, [] {
        HostAllocHooks& h = host_alloc_hooks();
        h.alloc = [](size_t bytes, bool* pinned) -> void* {
            void* p = nullptr;
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) { *pinned = true; return p; }
            (void)hipGetLastError();
            *pinned = false;
            return nullptr;  // (the pool falls back to malloc)
        };
        h.release = [](void* p, bool) { (void)hipHostFree(p); };
    });
}
}  // namespace

struct fgpu_index {
    HostIndex host;
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    // fgpu_tune
    uint64_t order_min_reads = env_u64("FULGOR_ORDER", 0) ? env_u64("FULGOR_ORDER_MIN_READS", 16384) : ~0ull;  // off: measured, no gain
    bool small_results = env_u64("FULGOR_SMALL", 1) != 0;
    bool dense_rows = env_u64("FULGOR_DENSE_ROWS", 1) != 0;  // use the dense rows (when they were built: d_rows)
    bool deduplicate = env_u64("FULGOR_DEDUPLICATE", 0) != 0;  // full intersection: every distinct id list of a pass once (--deduplicate)
    DevBuf d_table, d_bmp_rows, d_offsets, d_set_desc, d_blk_words, d_set_rank, d_rows;
    uint64_t table_buckets = 0;  // buckets of d_table (hashed region, tail, overflow region)
    DevBuf d_gops, d_gset_ops_off, d_gset_ops, d_garena, d_gblk_hdr, d_gblk_words, d_gset_bytes;
    DevDict dd{};
    DevColors dc{};
    DevGeneric dg{};
    // timing
    bool timing = false;
    double ms[FGPU_K_COUNT] = {0};
    uint64_t launches[FGPU_K_COUNT] = {0};
    struct Pending { int kernel; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    // device buffers of released read batches, kept for the next upload: the command-line path uploads and releases a batch
    // every few milliseconds from several host threads, and hipMalloc / hipFree synchronise the whole device (every stream)
    std::mutex reads_mu;
    std::vector<std::pair<DevBuf, DevBuf>> reads_pool;  // (bases, offsets)
    static constexpr size_t READS_POOL_MAX = 8;
    std::mutex tmu;  // results on different streams may be driven from different host threads
    // results of finished host-buffer calls (fgpu_full_intersection / fgpu_threshold_union: the calls of a reference worker, one per
    // chunk of reads from several threads), kept for the next call: creating a result is three streams and a pinned allocation, its
    // device buffers are a dozen hipMalloc — 2 ms for a chunk of a thousand reads whose kernels take 0.1
    std::mutex host_mu;
    std::vector<fgpu_result*> host_results;
    static constexpr size_t HOST_RESULTS_MAX = 4;
    static constexpr size_t HOST_RESULT_KEEP_BYTES = (size_t)1 << 30;  // a result that has grown beyond this is not kept (the index holds on to 4 GB at most)

    hipEvent_t get_event() {
        std::lock_guard<std::mutex> g(tmu);
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        return e;
    }
    // call only after the recording streams have been synchronised
    void add_timing(int kernel, double t_ms) {  // (a copy timed by the host's clock: the copy engines driven through the HSA runtime have no HIP events)
        std::lock_guard<std::mutex> g(tmu);
        ms[kernel] += t_ms;
        launches[kernel] += 1;
    }
    void collect_timing(std::vector<Pending>& mine) {
        std::lock_guard<std::mutex> g(tmu);
        for (auto& p : mine) {
            float t = 0;
            HIP_TRY(hipEventElapsedTime(&t, p.a, p.b));
            ms[p.kernel] += t;
            launches[p.kernel] += 1;
            event_pool.push_back(p.a);
            event_pool.push_back(p.b);
        }
        mine.clear();
    }
};

// RAII bracket: records HIP events around a kernel on the engine's stream
struct fgpu_result;
struct Timed {
    fgpu_index* ix;
    int kernel;
    hipStream_t stream;
    std::vector<fgpu_index::Pending>* sink;
    hipEvent_t a = nullptr, b = nullptr;
    bool on = false;  // (kernel < 0: a launch inside another bracket)
    Timed(fgpu_index* i, fgpu_result* r, int k, bool lookup = false);
    Timed(fgpu_index* i, fgpu_result* r, int k, hipStream_t on_stream);
    ~Timed() {
        if (on) { (void)hipEventRecord(b, stream); sink->push_back({kernel, a, b}); }
    }
};

struct fgpu_reads {
    fgpu_index* ix = nullptr;
    DevBuf d_bases, d_offs;
    uint64_t n = 0;
    // lengths: every read of the batch `uni_len` bases long (the common case: offsets and k-mer prefix sums are products, nothing is
    // kept per read), or the two vectors
    bool uniform = false;
    uint64_t uni_len = 0, uni_nk = 0;
    std::vector<uint64_t> cum_kmers;  // prefix sums of max(0, len-k+1)
    std::vector<uint64_t> h_offs;
    // (a batch of the streaming worker loop keeps neither: it is only ever processed whole, and knows its two totals)
    bool whole_only = false;
    uint64_t whole_kmers = 0, whole_bases = 0;
    uint64_t kmers_before(uint64_t i) const { return whole_only ? (i == n ? whole_kmers : 0) : uniform ? i * uni_nk : cum_kmers[i]; }
    uint64_t bases_before(uint64_t i) const { return whole_only ? (i == n ? whole_bases : 0) : uniform ? i * uni_len : h_offs[i]; }
    uint32_t max_kmers = 0;
    uint64_t max_total_kmers = 0;  // longest read, in k-mers (not capped by segmentation)
    // reads with more than SEG_KMERS k-mers are cut into overlapping segments (k-1 shared bases), copied one after
    // the other into the device base buffer (d_bases / d_seg_offs replace the reads' own bases and offsets), that the
    // lookup kernel treats as units; seg_first[r] = first segment of read r (n + 1 entries)
    bool has_long = false;
    std::vector<uint64_t> seg_first, seg_start, seg_end;
    DevBuf d_seg_offs, d_seg_first;
    // the last lookup every result queued on these reads (fgpu_run_lookup returns with the kernel in flight): fgpu_reads_free waits
    // for them before the buffers go back to the pool
    mutable std::mutex use_mu;
    mutable std::vector<std::pair<const fgpu_result*, hipEvent_t>> uses;
};
constexpr uint32_t SEG_KMERS = 512;  // what the 4-window lookup kernel takes as one unit

struct fgpu_result {
    fgpu_index* ix = nullptr;
    hipStream_t stream = nullptr;                 // every result owns a stream: passes on different results overlap
    // CU partition (FULGOR_CU_SPLIT, fgpu_run_lookup / fgpu_run_colours): the lookup kernel of a pass runs on its own stream, bound to
    // one part of the CUs, the colour kernels on `stream`, bound to the others; ev_lookup orders the two. Without a partition both
    // are the same stream.
    hipStream_t stream_lookup = nullptr;
    hipEvent_t ev_lookup = nullptr;
    // Copies between pinned host memory and the device go through streams that never run a kernel: behind a kernel on the same
    // stream the runtime performs a copy with a shader (__amd_rocclr_copyBuffer, seen in the kernel trace), which queues for the CUs
    // that the lookup and formatter kernels of the other batches occupy; on a kernel-free stream it goes to a copy engine and
    // runs beside them (profiles/micro/pcie_rates.hip: 80 MB up or down in 1.5 ms beside a kernel that holds every wave slot).
    hipStream_t stream_in = nullptr, stream_out = nullptr;
    // ... unless the copy engines are driven directly (copy_engines.hip.h): then the streams idle and these count the copies down
    hsa_signal_t sig_in{0}, sig_out{0};
    unsigned lane = 0;  // which of the engines for copies in this result's batches use
    std::vector<fgpu_index::Pending> pending;     // timing events recorded on that stream
    DevBuf d_nids, d_npos, d_idoff, d_ids_pool, d_cnt_pool, d_cursor, d_bitmap, d_counts, d_offsets, d_block_sums,
        d_block_mapped, d_totals, d_colors, d_acct, d_partial, d_tickets, d_idcsr, d_desc, d_kmer_ids, d_scores;
    DevBuf d_nids2, d_npos2, d_idoff2, d_ids_pool2, d_cnt_pool2;  // long reads: merged per-read lists (stage_lookup)
    DevBuf d_order_keys, d_order_hist, d_order_off, d_order;      // locality order of a pass (k_order_*)
    uint64_t order_hist_sets = 0;  // entries of d_order_hist that are known to be zero
    DevBuf d_small;                // results of at most SMALL_RESULT colours as colours (small_mode)
    // --deduplicate (stage_colors): order of the reads by id list, groups of equal lists, the results of one list per group
    DevBuf d_dd_hash, d_dd_hash2, d_dd_idx, d_dd_idx2, d_dd_head, d_dd_goff, d_dd_group, d_dd_nids, d_dd_idoff, d_dd_bitmap, d_dd_counts, d_dd_small, d_dd_tmp;
    uint64_t dd_groups = 0;        // distinct id lists of the last deduplicated pass
    bool small_mode = false;       // the last pass left no bitmap row for results of 0..SMALL_RESULT colours
    bool want_kmer_ids = false, want_scores = false;
    uint64_t total_ids = 0;
    uint64_t* h_totals = nullptr;  // pinned {total colours, mapped reads, ids used}
    uint64_t n = 0, total = 0, mapped = 0, total_kmers = 0, total_bases = 0;
    uint32_t id_stride = 0;
    uint64_t pool_units = 0;  // slabs of id_stride entries in d_ids_pool / d_cnt_pool (reads, or segments of long reads)
    uint32_t hit_rows = 0;  // rows of d_partial filled by the last expand launch (0: no colours, nothing to add)
    DevBuf d_fmt_sizes, d_fmt_off, d_fmt_out;  // device-side formatter
    char* h_fmt = nullptr;                     // pinned host copy of the formatted records (recycled; a slab of the process-wide pool)
    size_t h_fmt_cap = 0;
    bool h_fmt_pinned = false;
    bool hits_folded = true;  // false: too many colours for the expand kernel's LDS histogram; k_hits counts from the bitmaps
    // The u32 colour lists (CSR: d_offsets + d_colors) are materialised on demand (stage_expand): a pass leaves the result rows, the
    // small-result slots, the sizes and the CSR offsets; fgpu_result_expand, fgpu_result_download, the ascii / binary formatters and
    // the host-buffer calls run k2b_expand, the compressed formatter and the counters work from the rows
    bool csr_valid = false;
    // a worker loop that knows its largest batch says so: the buffers are sized for it at their first use instead of growing batch by
    // batch (growing a device buffer synchronises the device; growing the pinned output buffer stalls the copies in flight)
    uint64_t reserve_reads = 0;
    uint32_t max_kmers_in_batch = 0xFFFFFFFFu;  // bound on #positive k-mers of any read (unknown for id-only calls)
    bool have_ids = false;
};

Timed::Timed(fgpu_index* i, fgpu_result* r, int k, bool lookup) : Timed(i, r, k, lookup ? r->stream_lookup : r->stream) {}
Timed::Timed(fgpu_index* i, fgpu_result* r, int k, hipStream_t on_stream) : ix(i), kernel(k), stream(on_stream), sink(&r->pending) {
    on = ix->timing && k >= 0;
    if (on) { a = ix->get_event(); b = ix->get_event(); HIP_TRY(hipEventRecord(a, stream)); }
}

namespace {

// rank of every colour set by the number of k-mers that carry it (rarest first): the sort key of k_order_keys. Only the optional
// locality order of a pass reads it (off by default): computed when that is first asked for, not at every open.
void ensure_set_rank(fgpu_index* ix) {
    if (ix->d_set_rank.p) return;
    const Dict& d = ix->host.dict;
    const uint64_t ns = ix->host.hybrid.num_sets();
    std::vector<uint64_t> weight(ns, 0);
    for (uint64_t u = 0; u < d.num_unitigs(); ++u)
        if (d.unitig_csid[u] < ns) weight[d.unitig_csid[u]] += d.unitig_off[u + 1] - d.unitig_off[u] - d.k + 1;
    std::vector<uint32_t> by_weight(ns), rank(ns);
    for (uint64_t i = 0; i < ns; ++i) by_weight[i] = (uint32_t)i;
    std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t a, uint32_t b) { return weight[a] < weight[b]; });
    for (uint64_t i = 0; i < ns; ++i) rank[by_weight[i]] = (uint32_t)i;
    upload(ix->d_set_rank, rank, ix->stream);
    HIP_TRY(hipStreamSynchronize(ix->stream));
}

// The dictionary table in HBM, built there from the records (hip/dict_build.hip.h) by the rule the host builder follows
// (common/dict_place.h): sort by minimizer, sort by home bucket (both stable: the order is (home, minimizer, record number)),
// count the overflow buckets every hashed bucket needs, scan, and let every hashed bucket write itself and its overflow run.
void build_table_on_device(fgpu_index* ix) {
    const Dict& d = ix->host.dict;
    hipStream_t s = ix->stream;
    const uint64_t nrec = d.num_records();
    const uint64_t nb_hashed = (uint64_t)d.num_buckets + DICT_TAIL_BUCKETS;
    static_assert(DICT_TAIL_BUCKETS == DICT_TAIL_BUCKETS_, "one tail length");
    DevBuf recs, keyA, keyB, idxA, idxB, homeA, homeB, posA, posB, bstart, nb_over, over_off, tmp;
    struct Release { std::vector<DevBuf*> v; ~Release() { for (DevBuf* b : v) b->release(); } }
        release{{&recs, &keyA, &keyB, &idxA, &idxB, &homeA, &homeB, &posA, &posB, &bstart, &nb_over, &over_off, &tmp}};
    upload(recs, d.records, s);
    for (DevBuf* b : {&keyA, &keyB}) b->ensure(std::max<uint64_t>(1, nrec) * 8);
    for (DevBuf* b : {&idxA, &idxB, &homeA, &homeB, &posA, &posB}) b->ensure(std::max<uint64_t>(1, nrec) * 4);
    for (DevBuf* b : {&bstart, &nb_over, &over_off}) b->ensure((nb_hashed + 1) * 4);
    const uint32_t grid = (uint32_t)ix->num_cus * 8;
    uint32_t total_over = 0;
    if (nrec) {
        hipLaunchKernelGGL(k_dict_keys, dim3(grid), dim3(256), 0, s, recs.as<uint32_t>(), nrec, d.k, d.m, keyA.as<unsigned long long>(), idxA.as<uint32_t>());
        size_t need = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keyA.as<unsigned long long>(), keyB.as<unsigned long long>(), idxA.as<uint32_t>(), idxB.as<uint32_t>(),
                                          (size_t)nrec, 0u, 32u + d.m, s));
        tmp.ensure(need + 256);
        need = tmp.cap;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, need, keyA.as<unsigned long long>(), keyB.as<unsigned long long>(), idxA.as<uint32_t>(), idxB.as<uint32_t>(),
                                          (size_t)nrec, 0u, 32u + d.m, s));
        hipLaunchKernelGGL(k_dict_homes, dim3(grid), dim3(256), 0, s, keyB.as<unsigned long long>(), nrec, d.seed, d.num_buckets, homeA.as<uint32_t>(), posA.as<uint32_t>());
        uint32_t home_bits = 1;
        while (home_bits < 32 && (1ull << home_bits) < d.num_buckets) ++home_bits;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, homeA.as<uint32_t>(), homeB.as<uint32_t>(), posA.as<uint32_t>(), posB.as<uint32_t>(), (size_t)nrec, 0u, home_bits, s));
        tmp.ensure(need + 256);
        need = tmp.cap;
        HIP_TRY(rocprim::radix_sort_pairs(tmp.p, need, homeA.as<uint32_t>(), homeB.as<uint32_t>(), posA.as<uint32_t>(), posB.as<uint32_t>(), (size_t)nrec, 0u, home_bits, s));
    }
    HIP_TRY(hipMemsetAsync(bstart.p, 0xFF, (nb_hashed + 1) * 4, s));
    if (nrec)
        hipLaunchKernelGGL(k_dict_gather, dim3(grid), dim3(256), 0, s, homeB.as<uint32_t>(), posB.as<uint32_t>(), keyB.as<unsigned long long>(), idxB.as<uint32_t>(), nrec,
                           keyA.as<unsigned long long>(), idxA.as<uint32_t>(), bstart.as<uint32_t>());
    hipLaunchKernelGGL(k_dict_count, dim3(grid), dim3(256), 0, s, homeB.as<uint32_t>(), keyA.as<unsigned long long>(), nrec, bstart.as<uint32_t>(), nb_hashed, nb_over.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    {
        size_t need = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, need, nb_over.as<uint32_t>(), over_off.as<uint32_t>(), 0u, (size_t)nb_hashed, rocprim::plus<uint32_t>(), s));
        tmp.ensure(need + 256);
        need = tmp.cap;
        HIP_TRY(rocprim::exclusive_scan(tmp.p, need, nb_over.as<uint32_t>(), over_off.as<uint32_t>(), 0u, (size_t)nb_hashed, rocprim::plus<uint32_t>(), s));
        uint32_t last[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(&last[0], over_off.as<uint32_t>() + (nb_hashed - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(&last[1], nb_over.as<uint32_t>() + (nb_hashed - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        total_over = last[0] + last[1];
    }
    if (nb_hashed + total_over >= DICT_MAX_BUCKETS) throw std::runtime_error("dictionary table exceeds 2^31 buckets");
    ix->d_table.ensure((nb_hashed + total_over) * (uint64_t)BUCKET_WORDS * 4);
    ix->table_buckets = nb_hashed + total_over;
    hipLaunchKernelGGL(k_dict_fill, dim3(grid), dim3(256), 0, s, recs.as<uint32_t>(), homeB.as<uint32_t>(), keyA.as<unsigned long long>(), idxA.as<uint32_t>(), nrec,
                       bstart.as<uint32_t>(), over_off.as<uint32_t>(), nb_hashed, ix->d_table.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
}

// what the device gets about the colour sets besides the packed blocks, prepared on the host (no device call: fgpu_open does this
// while the device starts up)
struct HostForms {
    std::vector<ListDesc> sd;     // one resolved descriptor per colour set: everything a kernel needs about a list behind one gather
    std::vector<uint32_t> rows;   // the bitmap lists as aligned rows of w32 words, in id order
    uint32_t w32 = 0;
};
void build_host_forms(const HybridSets& h, HostForms& f) {
    const uint32_t w32 = ((h.num_colors + 31) / 32 + 3) & ~3u;  // result bitmaps move as 128-bit groups
    f.w32 = w32;
    // (on all threads: 0.85 M descriptors and 155 MB of bitmap rows cut out of the bit stream took a tenth of a second on one)
    const uint64_t ns = h.num_sets();
    std::vector<ListDesc>& sd = f.sd;
    sd.resize(ns);
    const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), ns / 4096 + 1);
    auto parallel = [&](auto fn) {  // fn(thread, first id, last id + 1)
        if (T == 1) { fn(0u, (uint64_t)0, ns); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] { fn(t, ns * t / T, ns * (t + 1) / T); });
        for (auto& x : th) x.join();
    };
    const auto is_bitmap = [&](uint64_t id) { return h.set_size[id] >= h.sparse_thr && h.set_size[id] < h.dense_thr; };
    // the bitmap lists leave the bit stream (arbitrary bit offsets) for aligned rows of w32 words, in id order
    std::vector<uint64_t> first_row(T + 1, 0);
    parallel([&](unsigned t, uint64_t a, uint64_t b) {
        uint64_t nb = 0;
        for (uint64_t id = a; id < b; ++id) nb += is_bitmap(id);
        first_row[t + 1] = nb;
    });
    for (unsigned t = 0; t < T; ++t) first_row[t + 1] += first_row[t];
    std::vector<uint32_t>& rows = f.rows;
    rows.assign(first_row[T] * w32 + 4, 0u);
    const auto stream_bits = [&](uint64_t pos, uint32_t len) -> uint32_t {  // len <= 32 bits at bit `pos` of the stream
        const uint64_t w = pos >> 6, sh = pos & 63;
        uint64_t v = h.bits[w] >> sh;
        if (sh + len > 64) v |= h.bits[w + 1] << (64 - sh);
        return len == 32 ? (uint32_t)v : (uint32_t)v & ((1u << len) - 1u);
    };
    parallel([&](unsigned t, uint64_t a, uint64_t b) {
        uint64_t next_row = first_row[t];
        for (uint64_t id = a; id < b; ++id) {
            const uint32_t size = h.set_size[id];
            ListDesc& d = sd[id];
            d.score = 0;
            d.id = (uint32_t)id;
            if (is_bitmap(id)) {  // bitmap list: its row
                const uint64_t body = h.offsets[id] + delta_code_bits(size);
                uint32_t* row = rows.data() + next_row * w32;
                for (uint32_t c0 = 0; c0 < h.num_colors; c0 += 32) row[c0 >> 5] = stream_bits(body + c0, std::min(32u, h.num_colors - c0));
                d.begin = next_row++ * w32;
                d.soff = 0;
                d.ncodes = 0;
                d.meta = (uint32_t)D_ENC_BITMAP;
            } else {  // gap-coded on the host, packed blocks here
                d.begin = h.blk_wbase[id];
                d.ncodes = (uint32_t)(h.blk_first[id + 1] - h.blk_first[id]);
                d.soff = d.ncodes == 1 ? h.blk_hdr[h.blk_first[id]] : h.blk_first[id];  // single block: the header itself
                d.meta = (uint32_t)(size < h.sparse_thr ? D_ENC_DELTA_GAPS : D_ENC_COMPLEMENT);
            }
        }
    });
}

void upload_index(fgpu_index* ix, const HostForms& forms) {
    const Dict& d = ix->host.dict;
    const HybridSets& h = ix->host.hybrid;
    hipStream_t s = ix->stream;
    LoadClock clk;
    // the whole k-mer dictionary: one table of 64-byte buckets, built on the device from the records (or uploaded, when the host built it)
    if (d.table.empty()) { build_table_on_device(ix); clk.lap("dictionary table (built on the device)"); }
    else { upload(ix->d_table, d.table, s); ix->table_buckets = d.table.size() / BUCKET_WORDS; HIP_TRY(hipStreamSynchronize(s)); clk.lap("dictionary table (uploaded)"); }
    upload(ix->d_offsets, h.offsets, s);
    const uint32_t w32 = forms.w32;
    upload(ix->d_set_desc, forms.sd, s);
    upload(ix->d_bmp_rows, forms.rows, s);
    if (h.blk_words.size() >= (1ull << 32)) throw std::runtime_error("colour sets too large: the packed blocks exceed 2^32 words");  // BlockLane::word
    upload(ix->d_blk_words, h.blk_words, s);
    HIP_TRY(hipStreamSynchronize(s));
    clk.lap("upload of descriptors, bitmap rows, packed blocks");
    ix->dd = DevDict{ix->d_table.as<uint32_t>(), d.num_buckets, d.k, d.m, d.seed};
    ix->dc = DevColors{ix->d_bmp_rows.as<uint32_t>(), ix->d_offsets.as<uint64_t>(), ix->d_set_desc.as<ListDesc>(),
                       ix->d_blk_words.as<uint32_t>(), h.num_colors, h.sparse_thr, h.dense_thr, w32};
    // Dense rows (k2r_intersect): every colour set as a plain bitmap row, built on the device from the forms above, while
    // they fit the budget (default: a quarter of the device's memory; FULGOR_ROWS_MAX_BYTES, 0 = never). Any number of colours:
    // a row of more than 32768 (4 groups of 128 bits per lane) goes through the kernels in tiles
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const uint64_t budget = env_u64("FULGOR_ROWS_MAX_BYTES", total_b / 4);
        const uint64_t need = (uint64_t)h.num_sets() * w32 * 4;
        if (h.num_sets() && need <= budget && need + (1ull << 30) <= free_b && (size_t)4 * w32 * 4 <= 160 * 1024) {  // (k_rows_build: four rows in LDS per block, up to 327680 colours)
            ix->d_rows.ensure(need + 64);
            const size_t lds = (size_t)4 * w32 * 4;
            if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_rows_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const uint32_t grid = (uint32_t)std::min<uint64_t>((h.num_sets() + 3) / 4, (uint64_t)ix->num_cus * 8);
            hipLaunchKernelGGL(k_rows_build, dim3(grid), dim3(256), lds, s, ix->dc, (uint64_t)h.num_sets(), ix->d_rows.as<uint32_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s));
            clk.lap("dense rows (built on the device)");
        }
    }
}

static_assert(sizeof(GenOpDev) == sizeof(ListDesc), "host and device op layouts must match");

void upload_generic(fgpu_index* ix) {
    const GenericSets& g = ix->host.generic;
    hipStream_t s = ix->stream;
    // the device holds the ops in their device form (spans / packed blocks); the encoded arena stays on the host
    upload(ix->d_gops, g.dev_ops, s);
    upload(ix->d_gset_ops_off, g.set_ops_off, s);
    upload(ix->d_gset_ops, g.dev_set_ops, s);
    upload(ix->d_garena, g.dev_span, s);
    upload(ix->d_gblk_hdr, g.dev_blk_hdr, s);
    if (g.dev_blk_words.size() >= (1ull << 32)) throw std::runtime_error("colour sets too large: the packed blocks exceed 2^32 words");  // BlockLane::word
    upload(ix->d_gblk_words, g.dev_blk_words, s);
    upload(ix->d_gset_bytes, g.set_bytes, s);
    HIP_TRY(hipStreamSynchronize(s));
    ix->dg = DevGeneric{ix->d_gset_ops_off.as<uint64_t>(), ix->d_gset_ops.as<uint32_t>(), ix->d_garena.as<uint4>(),
                        ix->d_gops.as<ListDesc>(), ix->d_gblk_hdr.as<uint64_t>(), ix->d_gblk_words.as<uint32_t>(),
                        g.num_colors, ix->dc.w32};
}

// waves per block such that the dynamic LDS request fits; throws if one wave does not fit a CU
uint32_t pick_waves(size_t per_wave, const void* kernel) {
    const size_t LDS_CU = 160 * 1024;
    if (per_wave > LDS_CU) throw std::runtime_error("colour count too large for the per-wave LDS layout of this kernel");
    uint32_t w = 4;
    while (w > 1 && w * per_wave > 64 * 1024) w >>= 1;
    if (w * per_wave > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)per_wave));
    return w;
}

// persistent grid: exactly as many blocks as can be resident (a larger grid-stride grid would run a
// second, partially filled round), fewer when there is not enough work
template <typename K>
uint32_t resident_grid(K kernel, uint64_t units, uint32_t per_block, int num_cus, int block_threads, size_t dyn_lds) {
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dyn_lds));
    if (per_cu < 1) per_cu = 1;
    uint64_t need = (units + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
    return (uint32_t)std::max<uint64_t>(1, std::min(need, cap));
}

void stage_lookup_on(fgpu_index* ix, const fgpu_reads* rd, uint64_t first, uint64_t count, fgpu_result* res) {
    hipStream_t s = res->stream_lookup;
    if (rd->whole_only && (first != 0 || count != rd->n)) throw std::runtime_error("internal error: a streamed batch is processed whole");
    res->total_kmers = rd->kmers_before(first + count) - rd->kmers_before(first);
    res->total_bases = rd->bases_before(first + count) - rd->bases_before(first);
    // units = reads, or segments when the batch holds reads longer than SEG_KMERS k-mers
    const bool seg = rd->has_long;
    const uint64_t u_first = seg ? rd->seg_first[first] : first;
    const uint64_t units = seg ? rd->seg_first[first + count] - u_first : count;
    res->n = count;
    res->max_kmers_in_batch = (uint32_t)std::min<uint64_t>(rd->max_total_kmers, 0xFFFFFFFFull);
    const uint64_t cap_units = std::max(units, seg ? (uint64_t)0 : res->reserve_reads);
    res->d_nids.ensure(cap_units * 4 + 16);
    res->d_npos.ensure(cap_units * 4 + 16);
    res->d_idoff.ensure(cap_units * 8 + 16);
    res->d_tickets.ensure(TICKET_BYTES);  // 8 padded work counters for each persistent launch of a pass
    HIP_TRY(hipMemsetAsync(res->d_tickets.p, 0, TICKET_BYTES, s));
    const uint32_t stride = std::max<uint32_t>(1, rd->max_kmers);  // at most one id per k-mer
    res->id_stride = stride;
    res->pool_units = units;
    res->d_ids_pool.ensure(cap_units * (uint64_t)stride * 4 + 64);  // (k2r_intersect reads the ids eight at a time)
    res->d_cnt_pool.ensure(cap_units * (uint64_t)stride * 4 + 64);
    res->have_ids = true;
    uint32_t* kmer_out = nullptr;
    if (res->want_kmer_ids) {
        res->d_kmer_ids.ensure(units * (uint64_t)stride * 4 + 16);
        kmer_out = res->d_kmer_ids.as<uint32_t>();
    }
    if (units == 0) return;
    const bool w13 = ix->dd.k - ix->dd.m + 1 == K1_WFIX;  // (the window count the kernel unrolls for)
    // units of at most 128 k-mers: one window each; up to 512 k-mers (250- to 500-base reads, segments of longer reads): 2 to 4 windows
    {
        auto launch_short = [&](auto kernel) {
            const uint32_t grid = resident_grid(kernel, units, 4, ix->num_cus, 256, 0);
            Timed t(ix, res, FGPU_K_LOOKUP, true);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, ix->dd, rd->d_bases.as<uint8_t>(),
                               seg ? rd->d_seg_offs.as<uint64_t>() : rd->d_offs.as<uint64_t>(), seg ? u_first : first, units,
                               res->d_nids.as<uint32_t>(), res->d_npos.as<uint32_t>(), res->d_idoff.as<uint64_t>(),
                               res->d_ids_pool.as<uint32_t>(), res->d_cnt_pool.as<uint32_t>(), stride, res->d_tickets.as<unsigned int>(),
                               kmer_out);
        };
        const bool ko = kmer_out != nullptr;
        // tables of more than 2^26 buckets (about 120 M distinct 31-mers): the WIDE instantiations (FULGOR_DICT_WIDE=1 forces them: tests)
        static const bool force_wide = env_u64("FULGOR_DICT_WIDE", 0) != 0;
        const bool wide = force_wide || ix->table_buckets > DICT_NARROW_BUCKETS;
        // units of up to 128 / 192 / 256 / 384 / 512 k-mers (158 / 222 / 286 / 414 / 542 bases at k = 31)
        const uint32_t quarters = (std::max<uint32_t>(rd->max_kmers, 1) + 63) / 64;
        const int hsel = quarters <= 2 ? 2 : (quarters == 3 ? 3 : (quarters == 4 ? 4 : (quarters <= 6 ? 6 : (quarters <= 8 ? 8 : 0))));
#define FG_K1_PICK(H, WIDE_)                                                                   \
        do {                                                                                   \
            if (w13 && !ko) launch_short(k1_lookup<true, H, false, WIDE_>);                     \
            else if (w13) launch_short(k1_lookup<true, H, true, WIDE_>);                        \
            else if (!ko) launch_short(k1_lookup<false, H, false, WIDE_>);                      \
            else launch_short(k1_lookup<false, H, true, WIDE_>);                                \
        } while (0)
        // (reads of up to 144 bases: 114 k-mers look at 128 m-mer positions, two rounds of 64; FULGOR_K1_SHORT=0: the general instantiation)
        static const bool allow_short = env_u64("FULGOR_K1_SHORT", 1) != 0;
        const bool short_units = allow_short && hsel == 2 && w13 && !ko && !wide && rd->max_kmers + K1_WFIX - 1 <= 128;
        if (short_units) launch_short(k1_lookup<true, 2, false, false, true>);
        else if (hsel == 2 && !wide) FG_K1_PICK(2, false);
        else if (hsel == 3 && !wide) FG_K1_PICK(3, false);
        else if (hsel == 4 && !wide) FG_K1_PICK(4, false);
        else if (hsel == 6 && !wide) FG_K1_PICK(6, false);
        else if (hsel == 8 && !wide) FG_K1_PICK(8, false);
        else if (hsel == 2) FG_K1_PICK(2, true);
        else if (hsel == 3) FG_K1_PICK(3, true);
        else if (hsel == 4) FG_K1_PICK(4, true);
        else if (hsel == 6) FG_K1_PICK(6, true);
        else if (hsel == 8) FG_K1_PICK(8, true);
#undef FG_K1_PICK
        else throw std::runtime_error("internal error: lookup unit longer than 512 k-mers");
        HIP_TRY(hipGetLastError());
    }
    if (!seg) return;
    // long reads: merge the per-segment id lists of every read on the device (k_merge_segments) into a second set
    // of slabs, which then take the place of the per-segment ones
    res->d_nids2.ensure(count * 4 + 16);
    res->d_npos2.ensure(count * 4 + 16);
    res->d_idoff2.ensure(count * 8 + 16);
    res->d_ids_pool2.ensure(units * (uint64_t)stride * 4 + 64);
    res->d_cnt_pool2.ensure(units * (uint64_t)stride * 4 + 64);
    {
        Timed t(ix, res, FGPU_K_LOOKUP, true);
        const uint32_t mgrid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((count + 3) / 4, (uint64_t)ix->num_cus * 8));
        hipLaunchKernelGGL(k_merge_segments, dim3(mgrid), dim3(256), 0, s, res->d_nids.as<uint32_t>(), res->d_npos.as<uint32_t>(),
                           res->d_ids_pool.as<uint32_t>(), res->d_cnt_pool.as<uint32_t>(), stride, rd->d_seg_first.as<uint64_t>(),
                           first, count, res->d_nids2.as<uint32_t>(), res->d_npos2.as<uint32_t>(), res->d_idoff2.as<uint64_t>(),
                           res->d_ids_pool2.as<uint32_t>(), res->d_cnt_pool2.as<uint32_t>());
        HIP_TRY(hipGetLastError());
    }
    std::swap(res->d_nids, res->d_nids2);
    std::swap(res->d_npos, res->d_npos2);
    std::swap(res->d_idoff, res->d_idoff2);
    std::swap(res->d_ids_pool, res->d_ids_pool2);
    std::swap(res->d_cnt_pool, res->d_cnt_pool2);
}

// the lookup of a pass, queued on the result's lookup stream; what follows on the result's main stream waits for it
void stage_lookup(fgpu_index* ix, const fgpu_reads* rd, uint64_t first, uint64_t count, fgpu_result* res) {
    stage_lookup_on(ix, rd, first, count, res);
    {   // the reads are in use until this point of the stream (fgpu_reads_free waits for it)
        std::lock_guard<std::mutex> g(rd->use_mu);
        hipEvent_t ev = nullptr;
        for (auto& u : rd->uses) if (u.first == res) ev = u.second;
        if (!ev) {
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            rd->uses.emplace_back(res, ev);
        }
        HIP_TRY(hipEventRecord(ev, res->stream_lookup));
    }
    if (res->stream_lookup != res->stream) {
        HIP_TRY(hipEventRecord(res->ev_lookup, res->stream_lookup));
        HIP_TRY(hipStreamWaitEvent(res->stream, res->ev_lookup, 0));
    }
}

// exclusive scan of n u32 sizes into n+1 u64 offsets; totals -> d_totals {sum, #nonzero}
void run_scan(fgpu_index* ix, fgpu_result* res, const uint32_t* sizes, uint64_t n, uint64_t* offsets, uint64_t* totals = nullptr,
              int timed_as = FGPU_K_SCAN) {
    hipStream_t s = res->stream;
    const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE, cnb = (std::max(n, res->reserve_reads) + SCAN_TILE - 1) / SCAN_TILE;
    res->d_block_sums.ensure(std::max<uint64_t>(1, cnb) * 8);
    res->d_block_mapped.ensure(std::max<uint64_t>(1, cnb) * 8);
    res->d_totals.ensure(32);
    Timed t(ix, res, timed_as);
    hipLaunchKernelGGL(scan_block_sums, dim3((uint32_t)nb), dim3(256), 0, s, sizes, n, res->d_block_sums.as<uint64_t>(),
                       res->d_block_mapped.as<uint64_t>());
    hipLaunchKernelGGL(scan_top, dim3(1), dim3(256), 0, s, res->d_block_sums.as<uint64_t>(), res->d_block_mapped.as<uint64_t>(), nb,
                       totals ? totals : res->d_totals.as<uint64_t>());
    hipLaunchKernelGGL(scan_apply, dim3((uint32_t)nb), dim3(256), 0, s, sizes, n, res->d_block_sums.as<uint64_t>(), offsets);
    HIP_TRY(hipGetLastError());
}

// per-read id lists (nids + source offsets into ids/cnt arrays) -> compact CSR of resolved descriptors
// Per-read descriptor lists (colour-set id + score) for the generic codecs. The hybrid kernels gather
// DevColors::set_desc themselves from the lookup kernel's id slab (no scan, no descriptor array, no host
// round trip), so they skip this.
void stage_descriptors(fgpu_index* ix, fgpu_result* res, uint64_t max_total_ids, int algo) {
    (void)algo;
    if (ix->host.type == IDX_HYBRID || (ix->d_rows.p && ix->dense_rows)) return;  // (the dense rows serve every codec)
    hipStream_t s = res->stream;
    const uint64_t n = res->n;
    res->d_idcsr.ensure((n + 1) * 8 + 16);
    res->total_ids = 0;
    if (n == 0) return;
    run_scan(ix, res, res->d_nids.as<uint32_t>(), n, res->d_idcsr.as<uint64_t>());
    // size the descriptor array exactly (one small D2H copy) instead of by the k-mer upper bound
    HIP_TRY(hipMemcpyAsync(res->h_totals, res->d_totals.p, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    res->total_ids = res->h_totals[0];
    if (res->total_ids > max_total_ids) throw std::runtime_error("internal error: more colour-set ids than k-mers");
    res->d_desc.ensure(std::max<uint64_t>(1, res->total_ids) * sizeof(ListDesc));
    Timed t(ix, res, FGPU_K_DESC);
    const uint64_t threads = n * 16;
    hipLaunchKernelGGL(k_desc, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, res->d_nids.as<uint32_t>(),
                       res->d_idoff.as<uint64_t>(), res->d_ids_pool.as<uint32_t>(),
                       res->have_ids ? res->d_cnt_pool.as<uint32_t>() : (const uint32_t*)nullptr, res->d_idcsr.as<uint64_t>(), n,
                       res->d_desc.as<ListDesc>());
    HIP_TRY(hipGetLastError());
}

// Locality order of the pass (k_order_*): the reads sorted by the rarest colour set among their ids. nullptr when the pass
// is too small to gain from it (or FULGOR_ORDER=0).
const uint32_t* stage_order(fgpu_index* ix, fgpu_result* res) {
    const uint64_t n = res->n;
    if (n < ix->order_min_reads || n >= (1ull << 32)) return nullptr;
    hipStream_t s = res->stream;
    const uint64_t ns = ix->host.hybrid.num_sets();
    {
        std::lock_guard<std::mutex> g(ix->reads_mu);
        ensure_set_rank(ix);
    }
    res->d_order_keys.ensure(n * 4 + 16);
    res->d_order.ensure(n * 4 + 16);
    res->d_order_off.ensure((ns + 2) * 8 + 64);
    if (res->order_hist_sets < ns + 1) {  // (the scatter leaves the histogram all zero)
        res->d_order_hist.ensure((ns + 1) * 4 + 16);
        HIP_TRY(hipMemsetAsync(res->d_order_hist.p, 0, (ns + 1) * 4, s));
    }
    res->order_hist_sets = 0;  // dirty from k_order_keys on; known zero again once the scatter has been queued (a failure in between leaves it dirty: cleared on the next pass)
    uint64_t* totals = res->d_order_off.as<uint64_t>() + (ns + 2);  // scratch behind the offsets (keeps d_totals intact)
    Timed t(ix, res, FGPU_K_ORDER);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)ix->num_cus * 8);  // (both kernels: the same slices)
    hipLaunchKernelGGL(k_order_keys, dim3(grid), dim3(256), 0, s, res->d_nids.as<uint32_t>(), res->d_idoff.as<uint64_t>(),
                       res->d_ids_pool.as<uint32_t>(), ix->d_set_rank.as<uint32_t>(), (uint32_t)ns, n, res->d_order_keys.as<uint32_t>(),
                       res->d_order_hist.as<uint32_t>());
    run_scan(ix, res, res->d_order_hist.as<uint32_t>(), ns + 1, res->d_order_off.as<uint64_t>(), totals, -1);  // (inside this bracket)
    hipLaunchKernelGGL(k_order_scatter, dim3(grid), dim3(256), 0, s, res->d_order_keys.as<uint32_t>(), (uint32_t)ns, n,
                       res->d_order_off.as<uint64_t>(), res->d_order_hist.as<uint32_t>(), res->d_order.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    res->order_hist_sets = ns + 1;
    return res->d_order.as<uint32_t>();
}

void stage_colors(fgpu_index* ix, int algo, double tau, fgpu_result* res) {
    hipStream_t s = res->stream;
    const uint64_t n = res->n;
    const uint32_t W = ix->dc.w32;
    const uint64_t cap_n = std::max(n, res->reserve_reads);
    res->d_bitmap.ensure(cap_n * W * 4 + 16);
    res->d_counts.ensure(cap_n * 4 + 16);
    res->d_offsets.ensure((cap_n + 1) * 8 + 16);
    res->total = res->mapped = 0;
    res->hit_rows = 0;
    res->small_mode = false;
    res->csr_valid = false;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(res->d_offsets.p, 0, 8, s));
        HIP_TRY(hipStreamSynchronize(s));
        return;
    }
    // the per-colour hit histogram rides along in the expand kernel's LDS while it fits (16-bit counters, about W * 64 bytes);
    // for larger collections the expand kernel runs without it and k_hits counts from the bitmaps on demand
    const size_t stage_lds = (K2B_THREADS / 64) * K2B_STAGE_BYTES + 16;  // + the block's ticket counter
    const bool hits_fold = stage_lds + k2b_hist_region(W) <= 80 * 1024;  // two blocks per CU
    if (ix->host.type != IDX_HYBRID && !(ix->d_rows.p && ix->dense_rows)) {
        const bool uni = algo == FGPU_THRESHOLD_UNION;
        if (!uni && algo != FGPU_FULL_INTERSECTION) throw std::runtime_error("unknown algorithm");
        if (uni && res->max_kmers_in_batch > 32767)  // biased 16-bit score counters at most
            throw std::runtime_error("threshold-union on the meta / differential codecs supports reads of at most 32767 k-mers");
        const bool plain8 = res->max_kmers_in_batch > 127 && res->max_kmers_in_batch <= 255;  // (as for the hybrid union)
        const int bits = res->max_kmers_in_batch <= 255 ? 8 : 16;
        const size_t per_wave = wave_scratch_bytes_compact() + (size_t)(uni ? G_SETS_UNION : G_SETS) * W * 4 +
                                (uni ? (size_t)W * 4 * bits : (size_t)W * 4);
        uint32_t* scores_out = nullptr;
        if (uni && res->want_scores) {
            res->d_scores.ensure(n * (uint64_t)ix->dc.n * 4 + 16);
            scores_out = res->d_scores.as<uint32_t>();
        }
        auto launch = [&](auto kernel) {
            const uint32_t wpb = pick_waves(per_wave, (const void*)kernel);
            const uint32_t grid = resident_grid(kernel, n, wpb, ix->num_cus, 64 * wpb, wpb * per_wave);
            Timed t(ix, res, uni ? FGPU_K_UNION : FGPU_K_INTERSECT);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wpb), wpb * per_wave, s, ix->dg, res->d_npos.as<uint32_t>(),
                               res->d_idcsr.as<uint64_t>(), res->d_desc.as<ListDesc>(), tau, n, res->d_bitmap.as<uint32_t>(),
                               res->d_counts.as<uint32_t>(), res->d_tickets.as<unsigned int>() + 8 * TICKET_STRIDE, scores_out);
            HIP_TRY(hipGetLastError());
        };
        if (!uni) launch(k_generic<false, 16>);
        else if (plain8) launch(k_generic<true, 8, false>);
        else if (bits == 8) launch(k_generic<true, 8>);
        else launch(k_generic<true, 16>);
    } else if (algo == FGPU_FULL_INTERSECTION) {
        // two reads per group (a second EXCL plane) while 8 waves per SIMD still fit the LDS with it
        const bool pair = (size_t)32 * k2a_wave_bytes(W, true) <= 160 * 1024;
        const size_t per_wave = k2a_wave_bytes(W, pair);
        const uint32_t* order = stage_order(ix, res);
        // small results travel as colours when a row is much larger than SMALL_RESULT colours, every such result is of the
        // sparse kind of the compressed formatter, and nobody needs the rows (k_hits counts from them when the expand
        // kernel's histogram does not fit)
        res->small_mode = ix->small_results && W >= 32 && SMALL_RESULT < ix->host.hybrid.sparse_thr && hits_fold && W / 4 <= 256;  // (a row of one tile in k2r_intersect)
        uint32_t* small_out = nullptr;
        if (res->small_mode) {
            res->d_small.ensure(cap_n * SMALL_RESULT * 4 + 16);
            small_out = res->d_small.as<uint32_t>();
        }
        // what the intersection kernel reads and writes: the reads' own lists and results, or (--deduplicate) one list per group of
        // reads with equal lists and that group's result
        uint64_t nl = n;
        const uint32_t* in_nids = res->d_nids.as<uint32_t>();
        const uint64_t* in_idoff = res->d_idoff.as<uint64_t>();
        uint32_t* out_bitmap = res->d_bitmap.as<uint32_t>();
        uint32_t* out_counts = res->d_counts.as<uint32_t>();
        uint32_t* out_small = small_out;
        const bool dedup = ix->deduplicate && n > 1 && n < (1ull << 32);
        res->dd_groups = 0;
        if (dedup) {
            const uint32_t g256 = (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)ix->num_cus * 16);
            for (DevBuf* b : {&res->d_dd_hash, &res->d_dd_hash2}) b->ensure(cap_n * 8 + 16);
            for (DevBuf* b : {&res->d_dd_idx, &res->d_dd_idx2, &res->d_dd_head, &res->d_dd_group}) b->ensure(cap_n * 4 + 16);
            res->d_dd_goff.ensure((cap_n + 3) * 8 + 16);
            Timed t(ix, res, FGPU_K_ORDER);
            hipLaunchKernelGGL(k_dd_hash, dim3(g256), dim3(256), 0, s, in_nids, in_idoff, res->d_ids_pool.as<uint32_t>(), n,
                               res->d_dd_hash.as<unsigned long long>(), res->d_dd_idx.as<uint32_t>());
            size_t need = 0;
            HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, res->d_dd_hash.as<unsigned long long>(), res->d_dd_hash2.as<unsigned long long>(), res->d_dd_idx.as<uint32_t>(),
                                              res->d_dd_idx2.as<uint32_t>(), (size_t)n, 0u, 64u, s));
            res->d_dd_tmp.ensure(need + 256);
            need = res->d_dd_tmp.cap;
            HIP_TRY(rocprim::radix_sort_pairs(res->d_dd_tmp.p, need, res->d_dd_hash.as<unsigned long long>(), res->d_dd_hash2.as<unsigned long long>(), res->d_dd_idx.as<uint32_t>(),
                                              res->d_dd_idx2.as<uint32_t>(), (size_t)n, 0u, 64u, s));
            hipLaunchKernelGGL(k_dd_heads, dim3(g256), dim3(256), 0, s, res->d_dd_hash2.as<unsigned long long>(), res->d_dd_idx2.as<uint32_t>(), in_nids, in_idoff,
                               res->d_ids_pool.as<uint32_t>(), n, res->d_dd_head.as<uint32_t>());
            uint64_t* totals = res->d_dd_goff.as<uint64_t>() + (n + 1);  // scratch behind the offsets (keeps d_totals intact)
            run_scan(ix, res, res->d_dd_head.as<uint32_t>(), n, res->d_dd_goff.as<uint64_t>(), totals, -1);  // (inside this bracket)
            uint64_t h[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(h, totals, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            nl = h[0];
            res->dd_groups = nl;
            res->d_dd_nids.ensure(nl * 4 + 16);
            res->d_dd_idoff.ensure(nl * 8 + 16);
            res->d_dd_bitmap.ensure(nl * W * 4 + 16);
            res->d_dd_counts.ensure(nl * 4 + 16);
            if (small_out) res->d_dd_small.ensure(nl * SMALL_RESULT * 4 + 16);
            hipLaunchKernelGGL(k_dd_groups, dim3(g256), dim3(256), 0, s, res->d_dd_idx2.as<uint32_t>(), res->d_dd_head.as<uint32_t>(), res->d_dd_goff.as<uint64_t>(),
                               in_nids, in_idoff, n, res->d_dd_group.as<uint32_t>(), res->d_dd_nids.as<uint32_t>(), res->d_dd_idoff.as<uint64_t>());
            HIP_TRY(hipGetLastError());
            in_nids = res->d_dd_nids.as<uint32_t>();
            in_idoff = res->d_dd_idoff.as<uint64_t>();
            out_bitmap = res->d_dd_bitmap.as<uint32_t>();
            out_counts = res->d_dd_counts.as<uint32_t>();
            out_small = small_out ? res->d_dd_small.as<uint32_t>() : nullptr;
        }
        const uint32_t* use_order = dedup ? nullptr : order;  // (the locality order is an order of the reads, not of the groups)
        auto launch = [&](auto kernel) {
            const uint32_t wpb = pick_waves(per_wave, (const void*)kernel);
            const uint32_t grid = resident_grid(kernel, nl, wpb, ix->num_cus, 64 * wpb, wpb * per_wave);
            Timed t(ix, res, FGPU_K_INTERSECT);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wpb), wpb * per_wave, s, ix->dc, in_nids,
                               in_idoff, res->d_ids_pool.as<uint32_t>(), nl, out_bitmap,
                               out_counts, res->d_tickets.as<unsigned int>() + 8 * TICKET_STRIDE, use_order, out_small);
            HIP_TRY(hipGetLastError());
        };
        auto launch_rows = [&](auto kernel) {
            const uint32_t grid = resident_grid(kernel, nl, 4, ix->num_cus, 256, 0);
            Timed t(ix, res, FGPU_K_INTERSECT);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, ix->d_rows.as<u32x4>(), W, in_nids,
                               in_idoff, res->d_ids_pool.as<uint32_t>(), nl, out_bitmap,
                               out_counts, res->d_tickets.as<unsigned int>() + 8 * TICKET_STRIDE, use_order, out_small);
            HIP_TRY(hipGetLastError());
        };
        if (ix->d_rows.p && ix->dense_rows) {
            if (W / 4 <= 64) launch_rows(k2r_intersect<1>);
            else if (W / 4 <= 128) launch_rows(k2r_intersect<2>);
            else launch_rows(k2r_intersect<4>);
        } else if (pair) launch(k2a_intersect<true>);
        else launch(k2a_intersect<false>);
        if (dedup) {  // every read takes the result of its group
            Timed t(ix, res, FGPU_K_ORDER);
            const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 3) / 4, (uint64_t)ix->num_cus * 16);
            hipLaunchKernelGGL(k_dd_fanout, dim3(grid), dim3(256), 0, s, res->d_dd_group.as<uint32_t>(), res->d_dd_bitmap.as<uint32_t>(), res->d_dd_counts.as<uint32_t>(),
                               out_small ? res->d_dd_small.as<uint32_t>() : (const uint32_t*)nullptr, n, W, res->d_bitmap.as<uint32_t>(), res->d_counts.as<uint32_t>(),
                               small_out);
            HIP_TRY(hipGetLastError());
        }
    } else if (algo == FGPU_THRESHOLD_UNION) {
        // score counters from the longest read of the batch: biased 8-bit up to 127 k-mers, plain 8-bit up to 255, biased
        // 16-bit up to 32767, else 32-bit
        const bool plain8 = res->max_kmers_in_batch > 127 && res->max_kmers_in_batch <= 255;
        const int bits = res->max_kmers_in_batch <= 255 ? 8 : (res->max_kmers_in_batch <= 32767 ? 16 : 32);
        const size_t per_wave = (size_t)W * 4 * bits + k3a_scratch_bytes();
        auto launch = [&](auto kernel) {
            const uint32_t wpb = pick_waves(per_wave, (const void*)kernel);
            const uint32_t grid = resident_grid(kernel, n, wpb, ix->num_cus, 64 * wpb, wpb * per_wave);
            Timed t(ix, res, FGPU_K_UNION);
            uint32_t* scores_out = nullptr;
            if (res->want_scores) {
                res->d_scores.ensure(n * (uint64_t)ix->dc.n * 4 + 16);
                scores_out = res->d_scores.as<uint32_t>();
            }
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * wpb), wpb * per_wave, s, ix->dc, res->d_npos.as<uint32_t>(),
                               res->d_nids.as<uint32_t>(), res->d_idoff.as<uint64_t>(), res->d_ids_pool.as<uint32_t>(),
                               res->d_cnt_pool.as<uint32_t>(), tau, n, res->d_bitmap.as<uint32_t>(),
                               res->d_counts.as<uint32_t>(), res->d_tickets.as<unsigned int>() + 8 * TICKET_STRIDE, scores_out);
            HIP_TRY(hipGetLastError());
        };
        // on dense rows: k3r_union
        auto launch_rows = [&](auto kernel) {
            const uint32_t grid = resident_grid(kernel, n, 4, ix->num_cus, 256, 0);
            Timed t(ix, res, FGPU_K_UNION);
            uint32_t* scores_out = nullptr;
            if (res->want_scores) {
                res->d_scores.ensure(n * (uint64_t)ix->dc.n * 4 + 16);
                scores_out = res->d_scores.as<uint32_t>();
            }
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, ix->d_rows.as<uint32_t>(), W, ix->dc.n, res->d_npos.as<uint32_t>(),
                               res->d_nids.as<uint32_t>(), res->d_idoff.as<uint64_t>(), res->d_ids_pool.as<uint32_t>(),
                               res->d_cnt_pool.as<uint32_t>(), tau, n, res->d_bitmap.as<uint32_t>(), res->d_counts.as<uint32_t>(),
                               res->d_tickets.as<unsigned int>() + 8 * TICKET_STRIDE, scores_out);
            HIP_TRY(hipGetLastError());
        };
        if (ix->d_rows.p && ix->dense_rows) {
            const bool sc = res->want_scores;
            if (plain8) { if (sc) launch_rows(k3r_union<8, false, true>); else launch_rows(k3r_union<8, false>); }
            else if (bits == 8) { if (sc) launch_rows(k3r_union<8, true, true>); else launch_rows(k3r_union<8>); }
            else if (bits == 16) { if (sc) launch_rows(k3r_union<16, true, true>); else launch_rows(k3r_union<16>); }
            else if (sc) launch_rows(k3r_union<32, true, true>);
            else launch_rows(k3r_union<32>);
        } else if (plain8) launch(k3a_union<8, false>);
        else if (bits == 8) launch(k3a_union<8>);
        else if (bits == 16) launch(k3a_union<16>);
        else launch(k3a_union<32>);
    } else {
        throw std::runtime_error("unknown algorithm");
    }
    run_scan(ix, res, res->d_counts.as<uint32_t>(), n, res->d_offsets.as<uint64_t>());
    HIP_TRY(hipMemcpyAsync(res->h_totals, res->d_totals.p, 16, hipMemcpyDeviceToHost, s));
    res->hits_folded = hits_fold;
    HIP_TRY(hipStreamSynchronize(s));
    res->total = res->h_totals[0];
    res->mapped = res->h_totals[1];
    if (ix->timing) ix->collect_timing(res->pending);
}

// The u32 colour lists of the last pass (k2b_expand: result rows / small-result slots -> CSR colours), for the consumers that
// read them: fgpu_result_expand, fgpu_result_download, the ascii / binary formatters, the host-buffer calls. The compressed
// formatter (src/ps_utils.cpp:168-237 works from the colours of a read one by one; here from its row) and the counters do not.
// The per-colour hit histogram of the pass rides along while it fits the kernel's LDS.
void stage_expand(fgpu_index* ix, fgpu_result* res) {
    if (res->csr_valid) return;
    hipStream_t s = res->stream;
    const uint64_t n = res->n;
    const uint32_t W = ix->dc.w32;
    res->hit_rows = 0;
    if (n == 0 || res->total == 0) { res->d_colors.ensure(16); res->csr_valid = true; return; }
    // the per-colour hit histogram rides along in the expand kernel's LDS while it fits (16-bit counters, about W * 64 bytes);
    // for larger collections the expand kernel runs without it and k_hits counts from the bitmaps on demand
    const size_t stage_lds = (K2B_THREADS / 64) * K2B_STAGE_BYTES + 16;  // + the block's ticket counter
    const size_t lds = res->hits_folded ? stage_lds + k2b_hist_region(W) : stage_lds;
    // 16-bit hit counters: a block takes tickets for at most block_cap reads (k2b_expand), and the grid is large enough for
    // the caps of the blocks of every ticket partition to exceed its reads by a quarter
    static const uint32_t block_cap = [] {
        const char* e = getenv("FULGOR_EXPAND_BLOCK_CAP");  // test knob: a small cap exercises the limit on small batches
        const long v = e ? atol(e) : 0;
        return (uint32_t)(v >= 32 && v <= 65504 ? v : 65504);
    }();
    const uint32_t cap_grid = (uint32_t)(n / (block_cap - block_cap / 4) + 1) + 8;
    const uint32_t grid = std::max<uint32_t>(resident_grid(k2b_expand, n, K2B_THREADS / 64, ix->num_cus, K2B_THREADS, lds),
                                             res->hits_folded ? cap_grid : 1u);
    if (res->hits_folded) res->d_partial.ensure((size_t)grid * W * 32 * 4);
    auto launch = [&](DevBuf& colors, bool probe = false) {
        HIP_TRY(hipMemsetAsync(res->d_tickets.as<unsigned int>() + 16 * TICKET_STRIDE, 0, K2B_MAX_PARTS * TICKET_STRIDE * sizeof(unsigned int), s));
        hipLaunchKernelGGL(probe ? k_probe_allocation : k2b_expand, dim3(grid), dim3(K2B_THREADS), lds, s, res->d_bitmap.as<uint32_t>(), res->d_counts.as<uint32_t>(),
                           res->d_offsets.as<uint64_t>(), n, W, colors.as<uint32_t>(),
                           res->d_tickets.as<unsigned int>() + 16 * TICKET_STRIDE,
                           res->hits_folded ? res->d_partial.as<uint32_t>() : (uint32_t*)nullptr, res->d_totals.as<uint64_t>(),
                           (uint64_t)(colors.cap / 4), block_cap,
                           res->small_mode ? res->d_small.as<uint32_t>() : (const uint32_t*)nullptr);
        HIP_TRY(hipGetLastError());
    };
    if (res->total > res->d_colors.cap / 4) {
        const size_t bytes = res->total * 4 + res->total + 16;  // 25 % headroom: later passes of the same size fit
        res->d_colors.ensure(bytes);
        // Which memory the driver hands out for the colour lists decides how fast this kernel stores into them: one allocation in three is
        // of a kind on which it takes 5 to 14 % longer (6.0 / 6.3 ms full intersection, 11.7 / 12.7 threshold union, 12.0 / 14.0 on dense
        // results; one in eight is 8 % FASTER on the full intersection; profiles/r6/k2b_allocation_r6.txt), and the time belongs to the
        // allocation for as long as it lives. FULGOR_EXPAND_LOTTERY=<n> (OFF by default) tries n more allocations of a LARGE buffer, each
        // timed on this very pass (its second run: the first pays for the first use of the memory), and keeps the fastest. It is off
        // because allocating and freeing tens of gigabytes costs 0.7 s (34 GB) to 2.5 s (105 GB) per candidate — the driver clears the
        // memory —, i.e. the 0.3 to 2 ms it can take off a pass of 10 M reads are earned back after thousands of passes: a knob for a
        // service that keeps one result for days, not a default, and not what bench.py measures.
        static const uint64_t extra = env_u64("FULGOR_EXPAND_LOTTERY", 0);
        if (extra && res->d_colors.cap >= (2ull << 30) && !DevBuf::guard_mode()) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            auto timed = [&](DevBuf& b) {
                launch(b, true);
                HIP_TRY(hipEventRecord(e0, s));
                launch(b, true);
                HIP_TRY(hipEventRecord(e1, s));
                HIP_TRY(hipEventSynchronize(e1));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                return ms;
            };
            static const bool trace = getenv("FULGOR_TRACE_ALLOC") != nullptr;
            const auto lot0 = std::chrono::steady_clock::now();
            float best = timed(res->d_colors);
            if (trace) fprintf(stderr, "[lottery] colour lists of %zu bytes: first allocation %.3f ms\n", res->d_colors.cap, best);
            for (uint64_t c = 0; c < extra; ++c) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < res->d_colors.cap + (8ull << 30)) break;
                DevBuf cand;
                try { cand.ensure(bytes); } catch (...) { (void)hipGetLastError(); break; }
                float t = 0;
                try { t = timed(cand); } catch (...) { cand.release(); throw; }
                if (trace) fprintf(stderr, "[lottery] candidate %llu: %.3f ms\n", (unsigned long long)c + 1, t);
                if (t < best) { std::swap(cand, res->d_colors); best = t; }
                cand.release();
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (trace) fprintf(stderr, "[lottery] kept %.3f ms; the lottery took %.1f ms\n", best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lot0).count());
        }
    }
    {
        Timed t(ix, res, FGPU_K_EXPAND);
        launch(res->d_colors);
    }
    if (res->hits_folded) res->hit_rows = grid;
    HIP_TRY(hipStreamSynchronize(s));
    res->csr_valid = true;
    if (ix->timing) ix->collect_timing(res->pending);
}

// the result's pinned output buffer: a slab of the process-wide pool (pinned once, at 0.16 ms per megabyte, and kept for the next
// result, reader or run of the process; fgpu_prepare_host pins them ahead of the first run)
void take_out_buffer(fgpu_result* res, size_t want) {
    if (res->h_fmt) SlabPool::get().give(res->h_fmt, res->h_fmt_cap, res->h_fmt_pinned);
    res->h_fmt = nullptr;
    res->h_fmt_cap = 0;
    install_pinned_allocator();
    size_t got = 0;
    bool pinned = false;
    res->h_fmt = (char*)SlabPool::get().take(want, got, pinned);
    res->h_fmt_cap = got;
    res->h_fmt_pinned = pinned;
}

// what a pass of `reads` reads of at most max_kmers k-mers each asks of a result's device buffers, asked for in one go (the stages ask
// again with the same expressions: growing is idempotent). For a worker loop that knows its batch size before its first batch.
void reserve_result(fgpu_index* ix, fgpu_result* res, uint64_t reads, uint32_t max_kmers, int format, uint64_t out_bytes) {
    const uint32_t W = ix->dc.w32;
    const uint32_t stride = std::max<uint32_t>(1, max_kmers);
    res->d_nids.ensure(reads * 4 + 16);
    res->d_npos.ensure(reads * 4 + 16);
    res->d_idoff.ensure(reads * 8 + 16);
    res->d_tickets.ensure(TICKET_BYTES);
    res->d_ids_pool.ensure(reads * (uint64_t)stride * 4 + 64);
    res->d_cnt_pool.ensure(reads * (uint64_t)stride * 4 + 64);
    res->d_bitmap.ensure(reads * W * 4 + 16);
    res->d_counts.ensure(reads * 4 + 16);
    res->d_offsets.ensure((reads + 1) * 8 + 16);
    if (ix->small_results && W >= 32) res->d_small.ensure(reads * SMALL_RESULT * 4 + 16);
    const uint64_t snb = (reads + SCAN_TILE - 1) / SCAN_TILE;
    res->d_block_sums.ensure(std::max<uint64_t>(1, snb) * 8);
    res->d_block_mapped.ensure(std::max<uint64_t>(1, snb) * 8);
    res->d_totals.ensure(32);
    if (format == FGPU_FMT_COMPRESSED) {
        const uint64_t cnb = (reads + CFMT_BLOCK_READS - 1) / CFMT_BLOCK_READS;
        res->d_fmt_sizes.ensure((reads + 3 * cnb) * 4 + 64);
        res->d_fmt_off.ensure((cnb + 1) * 8 + 48 + reads * 4);
    }
    if (out_bytes) {
        if (format == FGPU_FMT_COMPRESSED) res->d_fmt_out.ensure(out_bytes + 64);
        if (out_bytes > res->h_fmt_cap) take_out_buffer(res, out_bytes);
    }
}

template <typename F>
int guarded(F f) {
    try {
        f();
        return 0;
    } catch (std::bad_alloc&) {
        return fail(-ENOMEM, "out of host memory");
    } catch (std::exception& e) {
        return fail(-EIO, e.what());
    }
}

}  // namespace

void fgpu_stream_cache_release(fgpu_index* ix);  // stream_pipeline.hip.h

extern "C" {

const char* fgpu_last_error(void) { return g_err.c_str(); }
const char* fgpu_kernel_name(int kernel) { return kernel >= 0 && kernel < FGPU_K_COUNT ? KERNEL_NAMES[kernel] : ""; }

int fgpu_open(const char* path, int device, fgpu_index** out) {
    if (!path || !out) return fail(-EINVAL, "null argument");
    *out = nullptr;
    fgpu_index* ix = nullptr;
    int rc = guarded([&] {
        if (device != FGPU_HOST_ONLY && device < 0) throw std::runtime_error("invalid device ordinal");
        ix = new fgpu_index();
        ix->device = device;
        // The device side of opening (runtime start-up: a third of a second in a fresh process; stream, pinned-memory path, the timing
        // of the copy engines) runs on its own thread WHILE this one reads the container and prepares the host forms.
        std::string dev_error;
        std::thread dev_init;
        LoadClock clk_dev;
        if (device != FGPU_HOST_ONLY) g_device_startups.fetch_add(1);
        if (device != FGPU_HOST_ONLY)
            dev_init = std::thread([&] {
                struct Done { ~Done() { g_device_startups.fetch_sub(1); } } done;
                try {
                    int ndev = 0;  // (the first call into the runtime: this is where a fresh process spends its start-up time)
                    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
                        throw std::runtime_error("no HIP device available: the pseudoalignment engine has no CPU execution path");
                    if (device >= ndev) throw std::runtime_error("invalid device ordinal");
                    clk_dev.lap("device: runtime start-up (hipGetDeviceCount)");
                    HIP_TRY(hipSetDevice(device));
                    hipDeviceProp_t prop;
                    HIP_TRY(hipGetDeviceProperties(&prop, device));
                    ix->num_cus = prop.multiProcessorCount;
                    {   // the NUMA node of the device: the query reader keeps its threads on that node's cores (pinned host memory lives there)
                        char bus[64] = {0};
                        if (hipDeviceGetPCIBusId(bus, sizeof bus, device) == hipSuccess) {
                            std::string b(bus);
                            for (auto& ch : b) ch = (char)tolower((unsigned char)ch);
                            const std::string t = read_small_file("/sys/bus/pci/devices/" + b + "/numa_node");
                            if (!t.empty() && atoi(t.c_str()) >= 0) fastx_preferred_node().store(atoi(t.c_str()));
                        } else (void)hipGetLastError();
                    }
                    clk_dev.lap("device: properties");
                    HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
                    clk_dev.lap("device: first stream");
                    {   // the runtime sets up its pinned-memory path at the first hipHostMalloc of a process (50 ms): here, not in the first batch of reads
                        void* warm = nullptr;
                        if (hipHostMalloc(&warm, 4096, hipHostMallocDefault) == hipSuccess) (void)hipHostFree(warm);
                        else (void)hipGetLastError();
                    }
                    install_pinned_allocator();  // (the reader's chunks and the results' output buffers: pinned slabs of one pool)
                    clk_dev.lap("device: first pinned allocation");
                    (void)CopyEngines::get().usable(device);  // (times a small copy on every copy engine, once per process)
                    clk_dev.lap("device: copy engines timed");
                } catch (std::exception& e) {
                    dev_error = e.what();
                }
            });
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{dev_init};
        // (a handle on a device builds the dictionary's bucket table in HBM; FULGOR_DICT_ON_HOST=1: on the host and uploaded, for A/B measurements)
        // (FULGOR_INGEST_THREADS: threads that ingest a dump, 0 = all; what is built does not depend on it)
        open_index(path, ix->host, (unsigned)env_u64("FULGOR_INGEST_THREADS", 0), device == FGPU_HOST_ONLY || env_u64("FULGOR_DICT_ON_HOST", 0) != 0);
        if (device == FGPU_HOST_ONLY) return;  // ingestion / export / save only; queries are refused
        LoadClock clk;
        HostForms forms;
        build_host_forms(ix->host.hybrid, forms);
        clk.lap("descriptors and bitmap rows (host)");
        dev_init.join();
        if (!dev_error.empty()) throw std::runtime_error(dev_error);
        clk.lap("waiting for the device start-up");
        HIP_TRY(hipSetDevice(device));
        upload_index(ix, forms);
        if (ix->host.type != IDX_HYBRID) upload_generic(ix);
    });
    if (rc) { fgpu_close(ix); return rc; }  // (what the device thread created so far goes with it)
    *out = ix;
    return 0;
}

void fgpu_close(fgpu_index* ix) {
    if (!ix) return;
    if (ix->device == FGPU_HOST_ONLY) { delete ix; return; }
    (void)hipSetDevice(ix->device);
    fgpu_stream_cache_release(ix);  // the results the streaming worker loop keeps with the index
    for (fgpu_result* r : ix->host_results) fgpu_result_free(r);
    ix->host_results.clear();
    for (DevBuf* b : {&ix->d_table, &ix->d_bmp_rows, &ix->d_offsets, &ix->d_set_rank, &ix->d_rows,
                      &ix->d_set_desc, &ix->d_blk_words, &ix->d_gops, &ix->d_gset_ops_off, &ix->d_gset_ops,
                      &ix->d_garena, &ix->d_gblk_hdr, &ix->d_gblk_words, &ix->d_gset_bytes})
        b->release();
    for (auto& pr : ix->reads_pool) { pr.first.release(); pr.second.release(); }
    for (auto e : ix->event_pool) (void)hipEventDestroy(e);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

int fgpu_save(const fgpu_index* ix, const char* path) {
    if (!ix || !path) return fail(-EINVAL, "null argument");
    return guarded([&] {
        if (ends_with(path, "fur")) {  // the reference's layout (fur_format.hpp: not validated on a real file)
            fprintf(stderr, "fulgor_amd: WARNING: %s is written in the reference's section order with this engine's own k-mer dictionary "
                            "block and unverified bit-vector layouts: the reference's `fulgor` cannot read it. Use `dump` to hand an "
                            "index to the reference.\n", path);
            save_fur(ix->host, path);
        }
        else save_binary(ix->host, path);
    });
}

int fgpu_info(const fgpu_index* ix, uint64_t* k, uint64_t* num_colors, uint64_t* num_color_sets, uint64_t* num_unitigs,
              uint64_t* num_kmers, int* index_type) {
    if (!ix) return fail(-EINVAL, "null argument");
    if (k) *k = ix->host.dict.k;
    if (num_colors) *num_colors = ix->host.hybrid.num_colors;
    if (num_color_sets) *num_color_sets = ix->host.hybrid.num_sets();
    if (num_unitigs) *num_unitigs = ix->host.dict.num_unitigs();
    if (num_kmers) *num_kmers = ix->host.dict.num_kmers;
    if (index_type) *index_type = ix->host.type;
    return 0;
}

// Output arrays of the host-buffer calls come from the process-wide pool of pinned slabs while they are small enough to be worth
// pinning (the copy out of the device runs at the link's speed into pinned memory, at a fifth of it into pageable memory), and go
// back to it in fgpu_free; larger ones, and every other output of the library, are malloc'd.
namespace {
struct OutRegistry {
    std::mutex mu;
    std::unordered_map<void*, std::pair<size_t, bool>> held;  // pointer -> (slab bytes, pinned)
};
OutRegistry& out_registry() { static OutRegistry* r = new OutRegistry(); return *r; }
constexpr size_t OUT_POOLED_MAX = (size_t)512 << 20;
void* out_alloc(size_t bytes) {
    bytes = std::max<size_t>(bytes, 1);
    if (bytes > OUT_POOLED_MAX) return malloc(bytes);
    install_pinned_allocator();
    size_t got = 0;
    bool pinned = false;
    void* p = SlabPool::get().take(bytes, got, pinned);
    OutRegistry& r = out_registry();
    std::lock_guard<std::mutex> g(r.mu);
    r.held[p] = {got, pinned};
    return p;
}
}  // namespace

void fgpu_free(void* p) {
    if (!p) return;
    {
        OutRegistry& r = out_registry();
        std::unique_lock<std::mutex> g(r.mu);
        auto it = r.held.find(p);
        if (it != r.held.end()) {
            const std::pair<size_t, bool> sl = it->second;
            r.held.erase(it);
            g.unlock();
            SlabPool::get().give(p, sl.first, sl.second);
            return;
        }
    }
    free(p);
}

int fgpu_convert(fgpu_index* ix, int index_type, uint32_t partition_size, uint32_t cluster_size) {
    if (!ix) return fail(-EINVAL, "null argument");
    return guarded([&] {
        if (index_type == IDX_HYBRID) {
            ix->host.generic = GenericSets();
        } else {
            convert_sets(ix->host.hybrid, index_type, partition_size, cluster_size, ix->host.generic);
            // every set must decode to the same colours through its ops, in the encoded and in the device form
            // (cheap sample: every 97th set)
            std::vector<uint32_t> a, b, c;
            for (uint64_t id = 0; id < ix->host.hybrid.num_sets(); id += 97) {
                hybrid_decode(ix->host.hybrid, id, a);
                generic_decode(ix->host.generic, id, b);
                generic_decode_device(ix->host.generic, id, c);
                if (a != b || a != c) throw std::runtime_error("codec conversion self-check failed");
            }
        }
        ix->host.type = index_type;
        if (ix->device != FGPU_HOST_ONLY) {
            HIP_TRY(hipSetDevice(ix->device));
            if (index_type != IDX_HYBRID) upload_generic(ix);
        }
    });
}

#ifdef FG_K1_STATS
int fgpu_debug_k1_stats(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fg::k1_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return -EIO;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fg::k1_stats), z, sizeof(z)) != hipSuccess) return -EIO;
    }
    return 0;
}
#endif

int fgpu_selfcheck(const fgpu_index* cix, uint64_t unitig_stride) {
    if (!cix) return fail(-EINVAL, "null argument");
    fgpu_index* ix = const_cast<fgpu_index*>(cix);
    return guarded([&] {
        Dict& d = ix->host.dict;
        if (ix->device != FGPU_HOST_ONLY && d.table.empty()) {
            // the table this handle queries was built on the device: bring it here, build the host's from the same records, and the
            // two must be the same bytes; then the walk below goes through the DEVICE's table
            HIP_TRY(hipSetDevice(ix->device));
            std::vector<uint32_t> dev(ix->table_buckets * BUCKET_WORDS);
            HIP_TRY(hipMemcpy(dev.data(), ix->d_table.p, dev.size() * 4, hipMemcpyDeviceToHost));
            build_dict_table(d);
            if (d.table.size() != dev.size() || memcmp(d.table.data(), dev.data(), dev.size() * 4) != 0) {
                d.table.clear();
                d.table.shrink_to_fit();
                throw std::runtime_error("dictionary self-check failed (the table built on the device differs from the host's)");
            }
            d.table.swap(dev);
            struct Drop { Dict& d; ~Drop() { d.table.clear(); d.table.shrink_to_fit(); } } drop{d};
            verify_dict(d, unitig_stride ? unitig_stride : 1);
            return;
        }
        verify_dict(d, unitig_stride ? unitig_stride : 1);
    });
}

#define NEED_DEVICE(ix)                                                                                         \
    if ((ix)->device == FGPU_HOST_ONLY)                                                                         \
        return fail(-ENODEV, "index was opened host-only (device = -1): queries need a GPU, there is no CPU path")

int fgpu_reads_upload(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, fgpu_reads** out) {
    if (!ix || !offs || !out || (n && !bases)) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    *out = nullptr;
    fgpu_reads* rd = nullptr;
    int rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        rd = new fgpu_reads();
        rd->ix = ix;
        rd->n = n;
        const uint32_t k = ix->host.dict.k;
        // all reads of one length (one pass of differences, no per-read state kept)? Otherwise the prefix sums per read.
        {
            const uint64_t len0 = n ? offs[1] - offs[0] : 0;
            uint64_t differ = 0;
            for (uint64_t i = 0; i < n; ++i) differ |= (offs[i + 1] - offs[i]) ^ len0;
            // (equal steps DOWN are equal unsigned differences too: the offsets of a batch also add up)
            if (n && !differ && offs[0] == 0 && len0 <= ~0ull / n && offs[n] == len0 * n) {
                rd->uniform = true;
                rd->uni_len = len0;
                rd->uni_nk = len0 >= k ? len0 - k + 1 : 0;
                rd->max_total_kmers = rd->uni_nk;
                rd->max_kmers = (uint32_t)std::min<uint64_t>(rd->uni_nk, SEG_KMERS);
                rd->has_long = rd->uni_nk > SEG_KMERS;
            }
        }
        if (!rd->uniform || rd->has_long) {
            rd->uniform = false;
            rd->h_offs.assign(offs, offs + n + 1);
            rd->cum_kmers.assign(n + 1, 0);
            for (uint64_t i = 0; i < n; ++i) {
                if (offs[i + 1] < offs[i]) throw std::runtime_error("read offsets are not monotone");
                uint64_t len = offs[i + 1] - offs[i];
                uint64_t nk = len >= k ? len - k + 1 : 0;
                if (nk > SEG_KMERS) rd->has_long = true;
                rd->max_total_kmers = std::max(rd->max_total_kmers, nk);
                rd->max_kmers = std::max<uint32_t>(rd->max_kmers, (uint32_t)std::min<uint64_t>(nk, SEG_KMERS));
                rd->cum_kmers[i + 1] = rd->cum_kmers[i] + nk;
            }
        }
        const uint64_t nb = offs[n];
        if (rd->has_long) {
            rd->seg_first.assign(1, 0);
            std::vector<char> seg_bases;
            std::vector<uint64_t> seg_offs(1, 0);
            seg_bases.reserve(nb + nb / 16 + 64);
            auto add_segment = [&](uint64_t b0, uint64_t b1) {
                rd->seg_start.push_back(b0);
                rd->seg_end.push_back(b1);
                seg_bases.insert(seg_bases.end(), bases + b0, bases + b1);
                seg_offs.push_back(seg_bases.size());
            };
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t len = offs[i + 1] - offs[i];
                const uint64_t nk = len >= k ? len - k + 1 : 0;
                if (nk <= SEG_KMERS) add_segment(offs[i], offs[i + 1]);
                else
                    for (uint64_t s0 = 0; s0 < nk; s0 += SEG_KMERS) add_segment(offs[i] + s0, offs[i] + std::min(nk, s0 + SEG_KMERS) + k - 1);
                rd->seg_first.push_back(rd->seg_start.size());
            }
            upload(rd->d_seg_offs, seg_offs, ix->stream);
            upload(rd->d_seg_first, rd->seg_first, ix->stream);
            rd->d_bases.ensure(seg_bases.size() + 1024);  // the lookup kernel reads up to 576 bases past a unit's start unconditionally
            if (!seg_bases.empty())
                HIP_TRY(hipMemcpyAsync(rd->d_bases.p, seg_bases.data(), seg_bases.size(), hipMemcpyHostToDevice, ix->stream));
            HIP_TRY(hipStreamSynchronize(ix->stream));  // seg_bases / seg_offs are released at the end of this block
        } else {
            {
                std::lock_guard<std::mutex> g(ix->reads_mu);
                // the smallest pooled pair that is large enough, else the largest one (it grows below)
                size_t best = ix->reads_pool.size();
                for (size_t i = 0; i < ix->reads_pool.size(); ++i) {
                    const auto& c = ix->reads_pool[i];
                    const bool fits = c.first.cap >= nb + 1024 && c.second.cap >= (n + 1) * 8;
                    if (best == ix->reads_pool.size()) { best = i; continue; }
                    const auto& b = ix->reads_pool[best];
                    const bool bfits = b.first.cap >= nb + 1024 && b.second.cap >= (n + 1) * 8;
                    if (fits != bfits ? fits : (fits ? c.first.cap < b.first.cap : c.first.cap > b.first.cap)) best = i;
                }
                if (best < ix->reads_pool.size()) {
                    rd->d_bases = ix->reads_pool[best].first;
                    rd->d_offs = ix->reads_pool[best].second;
                    ix->reads_pool.erase(ix->reads_pool.begin() + best);
                }
            }
            rd->d_bases.ensure(nb + 1024);
            rd->d_offs.ensure((n + 1) * 8);
            if (nb) HIP_TRY(hipMemcpyAsync(rd->d_bases.p, bases, nb, hipMemcpyHostToDevice, ix->stream));
            HIP_TRY(hipMemcpyAsync(rd->d_offs.p, offs, (n + 1) * 8, hipMemcpyHostToDevice, ix->stream));
            HIP_TRY(hipStreamSynchronize(ix->stream));
        }
    });
    if (rc) { fgpu_reads_free(rd); return rc; }
    *out = rd;
    return 0;
}

void fgpu_reads_free(fgpu_reads* rd) {
    if (!rd) return;
    (void)hipSetDevice(rd->ix->device);
    for (auto& u : rd->uses) {  // lookups still in flight on these buffers (fgpu_run_lookup returns at once)
        (void)hipEventSynchronize(u.second);
        (void)hipEventDestroy(u.second);
    }
    rd->uses.clear();
    if (!rd->has_long && rd->d_bases.p && rd->d_offs.p) {  // back to the pool (the kernels that read them have completed)
        std::lock_guard<std::mutex> g(rd->ix->reads_mu);
        if (rd->ix->reads_pool.size() < fgpu_index::READS_POOL_MAX) {
            rd->ix->reads_pool.emplace_back(rd->d_bases, rd->d_offs);
            rd->d_bases = DevBuf();
            rd->d_offs = DevBuf();
        }
    }
    rd->d_bases.release();
    rd->d_offs.release();
    rd->d_seg_offs.release();
    rd->d_seg_first.release();
    delete rd;
}

int fgpu_result_create(fgpu_index* ix, fgpu_result** out) {
    if (!ix || !out) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    fgpu_result* r = new fgpu_result();
    r->ix = ix;
    int rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipHostMalloc((void**)&r->h_totals, 32));
        // FULGOR_CU_SPLIT=<n>: CUs [0, n) of the device's CU mask serve the lookup streams, the others the colour streams (a pass's
        // lookup then overlaps another pass's colour stage without the two kernels sharing a CU: fgpu_run_lookup / fgpu_run_colours).
        // FULGOR_CU_RANGE=<lo>:<hi> binds both to CUs [lo, hi) (scaling measurements). (Streams with a CU mask are blocking streams:
        // they synchronise implicitly with the legacy default stream, the others do not; split and unsplit timings of a caller
        // that also uses the default stream are therefore not strictly comparable.)
        const uint64_t split = env_u64("FULGOR_CU_SPLIT", 0);
        const char* range = getenv("FULGOR_CU_RANGE");
        auto masked = [&](hipStream_t* st, uint32_t lo, uint32_t hi) {
            std::vector<uint32_t> mask((ix->num_cus + 31) / 32, 0u);
            for (uint32_t c = lo; c < hi && c < (uint32_t)ix->num_cus; ++c) mask[c >> 5] |= 1u << (c & 31);
            HIP_TRY(hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()));
        };
        if (split > 0 && split < (uint64_t)ix->num_cus) {
            masked(&r->stream_lookup, 0, (uint32_t)split);
            masked(&r->stream, (uint32_t)split, (uint32_t)ix->num_cus);
            HIP_TRY(hipEventCreateWithFlags(&r->ev_lookup, hipEventDisableTiming));
        } else if (range && strchr(range, ':')) {
            const long lo = atol(range), hi = atol(strchr(range, ':') + 1);
            if (!(lo >= 0 && lo < hi && hi <= (long)ix->num_cus))
                throw std::runtime_error("FULGOR_CU_RANGE=" + std::string(range) + ": need 0 <= lo < hi <= " + std::to_string(ix->num_cus));
            masked(&r->stream, (uint32_t)lo, (uint32_t)hi);
            r->stream_lookup = r->stream;
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
            r->stream_lookup = r->stream;
        }
        r->d_totals.ensure(32);
        HIP_TRY(hipStreamCreateWithFlags(&r->stream_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&r->stream_out, hipStreamNonBlocking));
        if (CopyEngines::get().usable(ix->device)) {
            static std::atomic<unsigned> lanes{0};
            r->lane = lanes++;
            r->sig_in = CopyEngines::get().new_signal();
            r->sig_out = CopyEngines::get().new_signal();
        }
    });
    if (rc) { fgpu_result_free(r); return rc; }  // (what was created so far goes with it)
    *out = r;
    return 0;
}

static size_t result_device_bytes(const fgpu_result* cr) {
    fgpu_result* r = const_cast<fgpu_result*>(cr);
    size_t total = 0;
    for (DevBuf* b : {&r->d_nids, &r->d_npos, &r->d_idoff, &r->d_ids_pool, &r->d_cnt_pool, &r->d_cursor, &r->d_bitmap,
                      &r->d_counts, &r->d_offsets, &r->d_block_sums, &r->d_block_mapped, &r->d_totals, &r->d_colors, &r->d_acct, &r->d_partial, &r->d_tickets, &r->d_idcsr, &r->d_desc, &r->d_kmer_ids, &r->d_scores, &r->d_fmt_sizes, &r->d_fmt_off, &r->d_fmt_out,
                      &r->d_nids2, &r->d_npos2, &r->d_idoff2, &r->d_ids_pool2, &r->d_cnt_pool2, &r->d_order_keys, &r->d_order_hist,
                      &r->d_order_off, &r->d_order, &r->d_small, &r->d_dd_hash, &r->d_dd_hash2, &r->d_dd_idx, &r->d_dd_idx2, &r->d_dd_head, &r->d_dd_goff,
                      &r->d_dd_group, &r->d_dd_nids, &r->d_dd_idoff, &r->d_dd_bitmap, &r->d_dd_counts, &r->d_dd_small, &r->d_dd_tmp})
        total += b->cap;
    return total;
}

void fgpu_result_free(fgpu_result* r) {
    if (!r) return;
    (void)hipSetDevice(r->ix->device);
    for (DevBuf* b : {&r->d_nids, &r->d_npos, &r->d_idoff, &r->d_ids_pool, &r->d_cnt_pool, &r->d_cursor, &r->d_bitmap,
                      &r->d_counts, &r->d_offsets, &r->d_block_sums, &r->d_block_mapped, &r->d_totals, &r->d_colors, &r->d_acct, &r->d_partial, &r->d_tickets, &r->d_idcsr, &r->d_desc, &r->d_kmer_ids, &r->d_scores, &r->d_fmt_sizes, &r->d_fmt_off, &r->d_fmt_out,
                      &r->d_nids2, &r->d_npos2, &r->d_idoff2, &r->d_ids_pool2, &r->d_cnt_pool2, &r->d_order_keys, &r->d_order_hist,
                      &r->d_order_off, &r->d_order, &r->d_small, &r->d_dd_hash, &r->d_dd_hash2, &r->d_dd_idx, &r->d_dd_idx2, &r->d_dd_head, &r->d_dd_goff,
                      &r->d_dd_group, &r->d_dd_nids, &r->d_dd_idoff, &r->d_dd_bitmap, &r->d_dd_counts, &r->d_dd_small, &r->d_dd_tmp})
        b->release();
    if (r->h_totals) (void)hipHostFree(r->h_totals);
    if (r->h_fmt) SlabPool::get().give(r->h_fmt, r->h_fmt_cap, r->h_fmt_pinned);
    if (r->stream_lookup && r->stream_lookup != r->stream) (void)hipStreamDestroy(r->stream_lookup);
    CopyEngines::get().free_signal(r->sig_in);
    CopyEngines::get().free_signal(r->sig_out);
    if (r->stream_in) (void)hipStreamDestroy(r->stream_in);
    if (r->stream_out) (void)hipStreamDestroy(r->stream_out);
    if (r->ev_lookup) (void)hipEventDestroy(r->ev_lookup);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    for (auto& p : r->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    delete r;
}

int fgpu_run(fgpu_index* ix, const fgpu_reads* rd, uint64_t first, uint64_t count, int algo, double tau, fgpu_result* res) {
    if (!ix || !rd || !res) return fail(-EINVAL, "null argument");
    if (first > rd->n || count > rd->n - first) return fail(-EINVAL, "read range out of bounds");
    if (algo == FGPU_THRESHOLD_UNION && !(tau > 0.0 && tau <= 1.0))
        return fail(-EINVAL, "threshold must be a float in (0.0,1.0]");  // tools/pseudoalign.cpp:275-278
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        stage_lookup(ix, rd, first, count, res);
        stage_descriptors(ix, res, res->total_kmers, algo);
        stage_colors(ix, algo, tau, res);
    });
}

int fgpu_run_lookup(fgpu_index* ix, const fgpu_reads* rd, uint64_t first, uint64_t count, fgpu_result* res) {
    if (!ix || !rd || !res) return fail(-EINVAL, "null argument");
    if (first > rd->n || count > rd->n - first) return fail(-EINVAL, "read range out of bounds");
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        stage_lookup(ix, rd, first, count, res);  // queued, not waited for
    });
}

int fgpu_run_colours(fgpu_index* ix, int algo, double tau, fgpu_result* res) {
    if (!ix || !res) return fail(-EINVAL, "null argument");
    if (!res->have_ids) return fail(-EINVAL, "fgpu_run_colours: no lookup has been run on this result");
    if (algo == FGPU_THRESHOLD_UNION && !(tau > 0.0 && tau <= 1.0))
        return fail(-EINVAL, "threshold must be a float in (0.0,1.0]");  // tools/pseudoalign.cpp:275-278
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        stage_descriptors(ix, res, res->total_kmers, algo);
        stage_colors(ix, algo, tau, res);
    });
}

int fgpu_result_expand(fgpu_result* res) {
    if (!res) return fail(-EINVAL, "null argument");
    return guarded([&] {
        HIP_TRY(hipSetDevice(res->ix->device));
        stage_expand(res->ix, res);
    });
}

int fgpu_result_sizes(const fgpu_result* r, uint64_t* num_reads, uint64_t* total_colors, uint64_t* num_mapped) {
    if (!r) return fail(-EINVAL, "null argument");
    if (num_reads) *num_reads = r->n;
    if (total_colors) *total_colors = r->total;
    if (num_mapped) *num_mapped = r->mapped;
    return 0;
}

int fgpu_result_download(const fgpu_result* r, uint64_t* offsets, uint32_t* colors) {
    if (!r || !offsets) return fail(-EINVAL, "null argument");
    return guarded([&] {
        HIP_TRY(hipSetDevice(r->ix->device));
        if (colors) stage_expand(r->ix, const_cast<fgpu_result*>(r));
        HIP_TRY(hipMemcpy(offsets, r->d_offsets.p, (r->n + 1) * 8, hipMemcpyDeviceToHost));
        if (r->total && colors) HIP_TRY(hipMemcpy(colors, r->d_colors.p, r->total * 4, hipMemcpyDeviceToHost));
    });
}

int fgpu_result_format_view(const fgpu_result* r, int format, uint32_t first_read_id, const char** out, uint64_t* out_len) {
    if (!r || !out || !out_len) return fail(-EINVAL, "null argument");
    if (format != FGPU_FMT_ASCII && format != FGPU_FMT_BINARY && format != FGPU_FMT_COMPRESSED) return fail(-EINVAL, "unknown format");
    fgpu_result* res = const_cast<fgpu_result*>(r);
    fgpu_index* ix = r->ix;
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        hipStream_t s = res->stream;
        const uint64_t n = res->n;
        uint64_t bytes = 0;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 3) / 4, (uint64_t)ix->num_cus * 16);
        if (format != FGPU_FMT_COMPRESSED) stage_expand(ix, res);  // ascii and binary records are made of the u32 colour lists
        if (n && format == FGPU_FMT_COMPRESSED) {
            // from the result bitmaps: bits per record -> offsets inside blocks of CFMT_BLOCK_READS records -> block offsets
            const uint32_t W = ix->dc.w32, nc = ix->dc.n;
            const uint32_t sthr = ix->host.hybrid.sparse_thr, dthr = ix->host.hybrid.dense_thr;
            const uint64_t nb = (n + CFMT_BLOCK_READS - 1) / CFMT_BLOCK_READS;
            const uint64_t cn = std::max(n, res->reserve_reads), cnb = (cn + CFMT_BLOCK_READS - 1) / CFMT_BLOCK_READS;
            res->d_fmt_sizes.ensure((cn + 3 * cnb) * 4 + 64);
            res->d_fmt_off.ensure((cnb + 1) * 8 + 48 + cn * 4);
            uint32_t* bits = res->d_fmt_sizes.as<uint32_t>();
            uint32_t* block_bits = bits + n;
            uint32_t* block_bytes = block_bits + nb;
            uint64_t* block_off = res->d_fmt_off.as<uint64_t>();
            uint64_t* totals = block_off + (nb + 2);
            uint32_t* rec_off = (uint32_t*)(block_off + (nb + 6));
            {
                Timed t(ix, res, FGPU_K_FORMAT);
                hipLaunchKernelGGL(k_cfmt_sizes, dim3(grid), dim3(256), 0, s, res->d_bitmap.as<uint32_t>(), res->d_counts.as<uint32_t>(),
                                   n, W, nc, sthr, dthr, first_read_id, bits, res->small_mode ? res->d_small.as<uint32_t>() : (const uint32_t*)nullptr);
                hipLaunchKernelGGL(k_cfmt_blocks, dim3((uint32_t)nb), dim3(CFMT_BLOCK_READS), 0, s, bits, n, rec_off, block_bits,
                                   block_bytes);
            }
            run_scan(ix, res, block_bytes, nb, block_off, totals);
            uint64_t h[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(h, totals, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            bytes = h[0];
            res->d_fmt_out.ensure(std::max<uint64_t>(bytes, n < res->reserve_reads ? bytes / n * res->reserve_reads * 9 / 8 : 0) + 64);
            HIP_TRY(hipMemsetAsync(res->d_fmt_out.p, 0, bytes + 64, s));
            Timed t(ix, res, FGPU_K_FORMAT);
            // per-wave LDS stage of one record: the n bits of a bitmap record or the codes of at most n/4 gaps, plus its
            // header codes and the alignment slack (larger records, if any, are written code by code)
            const uint32_t cap_words = (uint32_t)std::min<uint64_t>(((uint64_t)nc * 3 / 2 + 256 + 63) / 64, 1024);
            hipLaunchKernelGGL(k_cfmt_write, dim3(grid), dim3(256), (size_t)4 * cap_words * 8, s, res->d_bitmap.as<uint32_t>(),
                               res->d_counts.as<uint32_t>(), n, W, nc, sthr, dthr, first_read_id, bits, rec_off, block_bits, block_off,
                               res->d_fmt_out.as<unsigned long long>(), cap_words,
                               res->small_mode ? res->d_small.as<uint32_t>() : (const uint32_t*)nullptr);
            HIP_TRY(hipGetLastError());
        } else if (n && format == FGPU_FMT_BINARY) {
            bytes = 8 * n + 4 * res->total;
            res->d_fmt_out.ensure(bytes);
            Timed t(ix, res, FGPU_K_FORMAT);
            hipLaunchKernelGGL(k_fmt_binary_write, dim3(grid), dim3(256), 0, s, res->d_offsets.as<uint64_t>(), res->d_colors.as<uint32_t>(),
                               n, first_read_id, res->d_fmt_out.as<uint32_t>());
            HIP_TRY(hipGetLastError());
        } else if (n) {
            res->d_fmt_sizes.ensure(n * 4 + 16);
            res->d_fmt_off.ensure((n + 1) * 8 + 48);
            {
                Timed t(ix, res, FGPU_K_FORMAT);
                hipLaunchKernelGGL(k_fmt_ascii_sizes, dim3(grid), dim3(256), 0, s, res->d_offsets.as<uint64_t>(),
                                   res->d_colors.as<uint32_t>(), n, first_read_id, res->d_fmt_sizes.as<uint32_t>());
            }
            uint64_t* totals = res->d_fmt_off.as<uint64_t>() + (n + 2);  // scratch behind the offsets (keeps d_totals intact)
            run_scan(ix, res, res->d_fmt_sizes.as<uint32_t>(), n, res->d_fmt_off.as<uint64_t>(), totals);
            uint64_t h[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(h, totals, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            bytes = h[0];
            res->d_fmt_out.ensure(bytes + 16);
            Timed t(ix, res, FGPU_K_FORMAT);
            hipLaunchKernelGGL(k_fmt_ascii_write, dim3(grid), dim3(256), 0, s, res->d_offsets.as<uint64_t>(), res->d_colors.as<uint32_t>(),
                               n, first_read_id, res->d_fmt_off.as<uint64_t>(), res->d_fmt_out.as<unsigned char>());
            HIP_TRY(hipGetLastError());
        }
        if (bytes > res->h_fmt_cap) {  // pinned: the copy runs at PCIe speed and the buffer is reused by later passes
            size_t want = bytes + bytes / 4 + 4096;
            if (n && n < res->reserve_reads) want = std::max<size_t>(want, bytes / n * res->reserve_reads * 5 / 4 + 4096);
            take_out_buffer(res, want);
        }
        HIP_TRY(hipStreamSynchronize(s));  // the records are complete; their copy out runs on the kernel-free stream (a copy engine)
        if (bytes) {
            CopyEngines& ce = CopyEngines::get();
            bool done = false;
            if (res->sig_out.handle && ce.usable(ix->device)) {  // the engine that carries every copy out, first in first out
                const uint64_t t0 = fastx_now_ns();
                ce.arm(res->sig_out, 1);
                if (ce.d2h(res->h_fmt, res->d_fmt_out.p, bytes, res->sig_out)) {
                    ce.wait(res->sig_out);
                    done = true;
                    if (ix->timing) ix->add_timing(FGPU_K_D2H, (fastx_now_ns() - t0) / 1e6);
                } else {
                    ce.disable();
                }
            }
            if (!done) {
                {
                    Timed t(ix, res, FGPU_K_D2H, res->stream_out);
                    HIP_TRY(hipMemcpyAsync(res->h_fmt, res->d_fmt_out.p, bytes, hipMemcpyDeviceToHost, res->stream_out));
                }
                HIP_TRY(hipStreamSynchronize(res->stream_out));
            }
        }
        if (ix->timing) ix->collect_timing(res->pending);
        *out = res->h_fmt ? res->h_fmt : "";
        *out_len = bytes;
    });
}

int fgpu_result_format(const fgpu_result* r, int format, uint32_t first_read_id, char** out, uint64_t* out_len) {
    if (!out) return fail(-EINVAL, "null argument");
    const char* view = nullptr;
    const int rc = fgpu_result_format_view(r, format, first_read_id, &view, out_len);
    if (rc) return rc;
    *out = (char*)malloc(std::max<uint64_t>(1, *out_len));
    if (!*out) return fail(-ENOMEM, "out of host memory");
    memcpy(*out, view, *out_len);
    return 0;
}

int fgpu_result_accumulate_hits(fgpu_index* ix, const fgpu_result* r, void* device_u64_hits) {
    if (!ix || !r || !device_u64_hits) return fail(-EINVAL, "null argument");
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        if (r->n) {
            const uint32_t W = ix->dc.w32;
            Timed t(ix, const_cast<fgpu_result*>(r), FGPU_K_HITS);
            fgpu_result* rw = const_cast<fgpu_result*>(r);
            if (r->total && (!r->hits_folded || !r->csr_valid)) {
                // no histogram from the expand kernel (large collections, or the colour lists of this pass were never asked
                // for): count from the result rows now, and from the slots of the results that travel as colours
                const uint32_t rows = (uint32_t)std::min<uint64_t>(256, (r->n + 63) / 64);
                rw->d_partial.ensure((size_t)rows * W * 32 * 4);
                const uint32_t* small = r->small_mode ? r->d_small.as<uint32_t>() : (const uint32_t*)nullptr;
                hipLaunchKernelGGL(k_hits, dim3(rows), dim3(256), 0, r->stream, r->d_bitmap.as<uint32_t>(), r->d_counts.as<uint32_t>(), r->n, W,
                                   rw->d_partial.as<uint32_t>(), small);
                if (small)
                    hipLaunchKernelGGL(k_hits_small, dim3((uint32_t)std::min<uint64_t>((r->n * SMALL_RESULT + 255) / 256, (uint64_t)ix->num_cus * 16)),
                                       dim3(256), 0, r->stream, r->d_counts.as<uint32_t>(), small, r->n, (unsigned long long*)device_u64_hits);
                rw->hit_rows = rows;
            }
            // one row of per-colour counts per block (of the expand kernel or of k_hits): sum the rows into the totals
            if (r->hit_rows)
                hipLaunchKernelGGL(k_hits_reduce, dim3((ix->dc.n + 255) / 256, HITS_ROW_GROUPS), dim3(256), 0, r->stream,
                                   r->d_partial.as<uint32_t>(), r->hit_rows, W, ix->dc.n, (unsigned long long*)device_u64_hits);
            hipLaunchKernelGGL(k_add_totals, dim3(1), dim3(64), 0, r->stream, (unsigned long long*)device_u64_hits, ix->dc.n,
                               r->n, r->d_totals.as<uint64_t>());
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(r->stream));
        if (ix->timing) ix->collect_timing(const_cast<fgpu_result*>(r)->pending);
    });
}

int fgpu_result_checksum(fgpu_result* r, uint64_t* from_lists, uint64_t* from_rows) {
    if (!r || !from_lists || !from_rows) return fail(-EINVAL, "null argument");
    fgpu_index* ix = r->ix;
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        stage_expand(ix, r);
        hipStream_t s = r->stream;
        r->d_acct.ensure(64);
        HIP_TRY(hipMemsetAsync(r->d_acct.p, 0, 48, s));
        unsigned long long* out = r->d_acct.as<unsigned long long>();
        if (r->n && r->total) {
            const uint32_t grid = (uint32_t)ix->num_cus * 8;
            hipLaunchKernelGGL(k_checksum_lists, dim3(grid), dim3(256), 0, s, r->d_colors.as<uint32_t>(), r->d_offsets.as<uint64_t>() + r->n, out);
            hipLaunchKernelGGL(k_checksum_rows, dim3(grid), dim3(256), 0, s, r->d_bitmap.as<uint32_t>(), r->d_counts.as<uint32_t>(),
                               r->d_offsets.as<uint64_t>(), r->n, ix->dc.w32,
                               r->small_mode ? r->d_small.as<uint32_t>() : (const uint32_t*)nullptr, out + 3);
            HIP_TRY(hipGetLastError());
        }
        uint64_t h[6];
        HIP_TRY(hipMemcpyAsync(h, out, 48, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < 3; ++i) { from_lists[i] = h[i]; from_rows[i] = h[3 + i]; }
    });
}

int fgpu_result_algorithmic_bytes(const fgpu_result* r, uint64_t* list_bytes, uint64_t* output_bytes, uint64_t* lookup_bytes) {
    if (!r) return fail(-EINVAL, "null argument");
    fgpu_index* ix = r->ix;
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        uint64_t acct[2] = {0, 0};
        if (r->n) {
            const_cast<fgpu_result*>(r)->d_acct.ensure(16);
            HIP_TRY(hipMemsetAsync(r->d_acct.p, 0, 16, r->stream));
            hipLaunchKernelGGL(k_account, dim3(1024), dim3(256), 0, r->stream, ix->dc, r->d_nids.as<uint32_t>(),
                               r->d_idoff.as<uint64_t>(), r->d_ids_pool.as<uint32_t>(), r->d_counts.as<uint32_t>(), r->n,
                               r->d_acct.as<unsigned long long>(),
                               // (a re-encoded index that answers from the dense rows is charged the hybrid lists the rows were built from,
                               // not the longer chains of partial lists its codec would have walked)
                               ix->host.type == IDX_HYBRID || (ix->d_rows.p && ix->dense_rows) ? (const uint32_t*)nullptr : ix->d_gset_bytes.as<uint32_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(acct, r->d_acct.p, 16, hipMemcpyDeviceToHost, r->stream));
            HIP_TRY(hipStreamSynchronize(r->stream));
        }
        if (list_bytes) *list_bytes = acct[0];
        if (output_bytes) *output_bytes = acct[1];
        // ceil(L/4) bytes of 2-bit read + one 8-byte record per k-mer
        if (lookup_bytes) *lookup_bytes = (r->total_bases + 3) / 4 + 8 * r->total_kmers;
    });
}

int fgpu_device_report(fgpu_index* ix, char** out) {
    if (!ix || !out) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, sizeof bus, ix->device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
        char line[512];
        snprintf(line, sizeof line, "device %d (%s, %s, %d CUs), %.1f of %.1f GB free, host NUMA node %d; ", ix->device, prop.name, bus,
                 prop.multiProcessorCount, free_b / 1e9, total_b / 1e9, fastx_preferred_node().load());
        const std::string text = std::string(line) + (CopyEngines::get().usable(ix->device) ? "" : "(copies through the HIP runtime) ") + CopyEngines::get().report();
        *out = (char*)malloc(text.size() + 1);
        if (!*out) throw std::bad_alloc();
        memcpy(*out, text.c_str(), text.size() + 1);
    });
}

int fgpu_copy_engines_classify(const uint32_t* engines, const double* gb_per_s, uint32_t n, uint32_t* fast, uint32_t* num_fast) {
    if (!engines || !gb_per_s || !fast || !num_fast) return fail(-EINVAL, "null argument");
    std::vector<std::pair<uint32_t, double>> rate;
    for (uint32_t i = 0; i < n; ++i) rate.push_back({engines[i], gb_per_s[i]});
    const std::vector<uint32_t> f = CopyEngines::classify(rate);
    for (size_t i = 0; i < f.size(); ++i) fast[i] = f[i];
    *num_fast = (uint32_t)f.size();
    return 0;
}

int fgpu_tune(fgpu_index* ix, int knob, uint64_t value) {
    if (!ix) return fail(-EINVAL, "null argument");
    if (knob == FGPU_TUNE_ORDER_MIN_READS) ix->order_min_reads = value;
    else if (knob == FGPU_TUNE_SMALL_RESULTS) ix->small_results = value != 0;
    else if (knob == FGPU_TUNE_DENSE_ROWS) ix->dense_rows = value != 0;
    else if (knob == FGPU_TUNE_DEDUPLICATE) ix->deduplicate = value != 0;
    else return fail(-EINVAL, "unknown knob");
    return 0;
}

int fgpu_result_distinct_lists(const fgpu_result* r, uint64_t* num_lists) {
    if (!r || !num_lists) return fail(-EINVAL, "null argument");
    *num_lists = r->dd_groups;
    return 0;
}

int fgpu_timing_enable(fgpu_index* ix, int on) {
    if (!ix) return fail(-EINVAL, "null argument");
    ix->timing = on != 0;
    return 0;
}
int fgpu_timing_reset(fgpu_index* ix) {
    if (!ix) return fail(-EINVAL, "null argument");
    for (int i = 0; i < FGPU_K_COUNT; ++i) { ix->ms[i] = 0; ix->launches[i] = 0; }
    return 0;
}
int fgpu_timing_get(fgpu_index* ix, int kernel, double* total_ms, uint64_t* launches) {
    if (!ix || kernel < 0 || kernel >= FGPU_K_COUNT) return fail(-EINVAL, "bad argument");
    if (total_ms) *total_ms = ix->ms[kernel];
    if (launches) *launches = ix->launches[kernel];
    return 0;
}

// ---- host-buffer convenience calls -------------------------------------------------------------------
static size_t result_device_bytes(const fgpu_result* r);
static int run_host(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, int algo, double tau,
                    uint64_t** out_offsets, uint32_t** out_colors) {
    if (!ix || !out_offsets || !out_colors) return fail(-EINVAL, "null argument");
    *out_offsets = nullptr;
    *out_colors = nullptr;
    fgpu_reads* rd = nullptr;
    fgpu_result* res = nullptr;
    int rc = fgpu_reads_upload(ix, bases, offs, n, &rd);
    if (!rc) {  // a result a finished call left behind, or a new one
        std::lock_guard<std::mutex> g(ix->host_mu);
        if (!ix->host_results.empty()) { res = ix->host_results.back(); ix->host_results.pop_back(); }
    }
    if (!rc && !res) rc = fgpu_result_create(ix, &res);
    if (!rc) rc = fgpu_run(ix, rd, 0, n, algo, tau, res);
    if (!rc) {  // (fgpu_result_download materialises the colour lists)
        uint64_t* o = nullptr;
        uint32_t* c = nullptr;
        rc = guarded([&] {
            o = (uint64_t*)out_alloc((n + 1) * 8);
            c = (uint32_t*)out_alloc(res->total * 4);
            if (!o || !c) throw std::bad_alloc();
        });
        if (!rc) rc = fgpu_result_download(res, o, c);
        if (rc) { fgpu_free(o); fgpu_free(c); } else { *out_offsets = o; *out_colors = c; }
    }
    if (res && !rc && result_device_bytes(res) <= fgpu_index::HOST_RESULT_KEEP_BYTES) {
        std::lock_guard<std::mutex> g(ix->host_mu);
        if (ix->host_results.size() < fgpu_index::HOST_RESULTS_MAX) { ix->host_results.push_back(res); res = nullptr; }
    }
    fgpu_result_free(res);
    fgpu_reads_free(rd);
    return rc;
}

int fgpu_full_intersection(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, uint64_t** out_offsets,
                           uint32_t** out_colors) {
    return run_host(ix, bases, offs, n, FGPU_FULL_INTERSECTION, 0.0, out_offsets, out_colors);
}

int fgpu_threshold_union(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, double tau,
                         uint64_t** out_offsets, uint32_t** out_colors) {
    return run_host(ix, bases, offs, n, FGPU_THRESHOLD_UNION, tau, out_offsets, out_colors);
}

int fgpu_fetch_color_set_ids(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, uint64_t** out_offsets,
                             uint32_t** out_ids) {
    if (!ix || !out_offsets || !out_ids) return fail(-EINVAL, "null argument");
    *out_offsets = nullptr;
    *out_ids = nullptr;
    fgpu_reads* rd = nullptr;
    fgpu_result* res = nullptr;
    int rc = fgpu_reads_upload(ix, bases, offs, n, &rd);
    if (!rc) rc = fgpu_result_create(ix, &res);
    if (!rc) rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        stage_lookup(ix, rd, 0, n, res);
        HIP_TRY(hipStreamSynchronize(res->stream));
        if (ix->timing) ix->collect_timing(res->pending);
        // dense CSR on the device first (scan of the list sizes + gather): only the ids that exist cross PCIe
        hipStream_t s = res->stream;
        res->d_idcsr.ensure((n + 1) * 8 + 16);
        uint64_t used = 0;
        if (n) {
            run_scan(ix, res, res->d_nids.as<uint32_t>(), n, res->d_idcsr.as<uint64_t>());
            HIP_TRY(hipMemcpyAsync(res->h_totals, res->d_totals.p, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            used = res->h_totals[0];
        }
        uint64_t* o = (uint64_t*)malloc((n + 1) * 8);
        uint32_t* v = (uint32_t*)malloc(std::max<uint64_t>(1, used) * 4);
        if (!o || !v) { free(o); free(v); throw std::bad_alloc(); }
        o[0] = 0;
        if (n) {
            res->d_colors.ensure(std::max<uint64_t>(1, used) * 4);
            hipLaunchKernelGGL(k_gather_ids, dim3((uint32_t)((n * 16 + 255) / 256)), dim3(256), 0, s, res->d_nids.as<uint32_t>(),
                               res->d_idoff.as<uint64_t>(), res->d_ids_pool.as<uint32_t>(), res->d_idcsr.as<uint64_t>(), n,
                               res->d_colors.as<uint32_t>());
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(o, res->d_idcsr.p, (n + 1) * 8, hipMemcpyDeviceToHost, s));
            if (used) HIP_TRY(hipMemcpyAsync(v, res->d_colors.p, used * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        *out_offsets = o;
        *out_ids = v;
    });
    fgpu_result_free(res);
    fgpu_reads_free(rd);
    return rc;
}

int fgpu_intersect_ids(fgpu_index* ix, const uint32_t* ids, const uint64_t* id_offs, uint64_t n, uint64_t** out_offsets,
                       uint32_t** out_colors) {
    if (!ix || !id_offs || !out_offsets || !out_colors) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    *out_offsets = nullptr;
    *out_colors = nullptr;
    fgpu_result* res = nullptr;
    int rc = fgpu_result_create(ix, &res);
    if (!rc) rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        const uint64_t ns = ix->host.hybrid.num_sets();
        std::vector<uint32_t> nids(n);
        for (uint64_t r = 0; r < n; ++r) {
            if (id_offs[r + 1] < id_offs[r]) throw std::runtime_error("id offsets are not monotone");
            nids[r] = (uint32_t)(id_offs[r + 1] - id_offs[r]);
        }
        for (uint64_t i = 0; i < id_offs[n]; ++i)
            if (ids[i] >= ns) throw std::runtime_error("colour-set id out of range");
        res->n = n;
        res->d_nids.ensure(n * 4 + 16);
        res->d_npos.ensure(n * 4 + 16);
        res->d_idoff.ensure(n * 8 + 16);
        res->d_ids_pool.ensure(id_offs[n] * 4 + 64);
        if (n) {
            HIP_TRY(hipMemcpy(res->d_nids.p, nids.data(), n * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(res->d_idoff.p, id_offs, n * 8, hipMemcpyHostToDevice));
        }
        if (id_offs[n]) HIP_TRY(hipMemcpy(res->d_ids_pool.p, ids, id_offs[n] * 4, hipMemcpyHostToDevice));
        res->d_tickets.ensure(TICKET_BYTES);
        HIP_TRY(hipMemsetAsync(res->d_tickets.p, 0, TICKET_BYTES, res->stream));
        res->have_ids = false;
        stage_descriptors(ix, res, id_offs[n], FGPU_FULL_INTERSECTION);
        stage_colors(ix, FGPU_FULL_INTERSECTION, 0.0, res);
        stage_expand(ix, res);
        uint64_t* o = (uint64_t*)malloc((n + 1) * 8);
        uint32_t* c = (uint32_t*)malloc(std::max<uint64_t>(1, res->total) * 4);
        if (!o || !c) { free(o); free(c); throw std::bad_alloc(); }
        HIP_TRY(hipMemcpy(o, res->d_offsets.p, (n + 1) * 8, hipMemcpyDeviceToHost));
        if (res->total) HIP_TRY(hipMemcpy(c, res->d_colors.p, res->total * 4, hipMemcpyDeviceToHost));
        *out_offsets = o;
        *out_colors = c;
    });
    fgpu_result_free(res);
    return rc;
}

// ---- k-mer level queries --------------------------------------------------------------------
int fgpu_kmer_color_set_ids(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, uint64_t** out_offsets,
                            uint32_t** out_ids) {
    if (!ix || !out_offsets || !out_ids) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    *out_offsets = nullptr;
    *out_ids = nullptr;
    fgpu_reads* rd = nullptr;
    fgpu_result* res = nullptr;
    int rc = fgpu_reads_upload(ix, bases, offs, n, &rd);
    if (!rc) rc = fgpu_result_create(ix, &res);
    if (!rc) rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        res->want_kmer_ids = true;
        stage_lookup(ix, rd, 0, n, res);
        HIP_TRY(hipStreamSynchronize(res->stream));
        if (ix->timing) ix->collect_timing(res->pending);
        const uint32_t k = ix->host.dict.k;
        const uint64_t stride = std::max<uint32_t>(1, rd->max_kmers);
        const uint64_t units = rd->has_long ? rd->seg_first[n] : n;
        std::vector<uint32_t> raw(units * stride);
        if (units) HIP_TRY(hipMemcpy(raw.data(), res->d_kmer_ids.p, units * stride * 4, hipMemcpyDeviceToHost));
        uint64_t* o = (uint64_t*)malloc((n + 1) * 8);
        uint32_t* v = (uint32_t*)malloc(std::max<uint64_t>(1, rd->kmers_before(n)) * 4);
        if (!o || !v) { free(o); free(v); throw std::bad_alloc(); }
        for (uint64_t r = 0; r <= n; ++r) o[r] = rd->kmers_before(r);
        for (uint64_t r = 0; r < n; ++r) {
            // segments of a long read hold consecutive, non-overlapping k-mer ranges
            const uint64_t u0 = rd->has_long ? rd->seg_first[r] : r, u1 = rd->has_long ? rd->seg_first[r + 1] : r + 1;
            uint64_t at = o[r];
            for (uint64_t u = u0; u < u1; ++u) {
                const uint64_t len = rd->has_long ? rd->seg_end[u] - rd->seg_start[u] : offs[r + 1] - offs[r];
                const uint64_t nk = len >= k ? len - k + 1 : 0;
                memcpy(v + at, raw.data() + u * stride, nk * 4);
                at += nk;
            }
        }
        *out_offsets = o;
        *out_ids = v;
    });
    fgpu_result_free(res);
    fgpu_reads_free(rd);
    return rc;
}

int fgpu_kmer_matches(fgpu_index* ix, const char* bases, const uint64_t* offs, uint64_t n, uint32_t** out_counts) {
    if (!ix || !out_counts) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    *out_counts = nullptr;
    fgpu_reads* rd = nullptr;
    fgpu_result* res = nullptr;
    int rc = fgpu_reads_upload(ix, bases, offs, n, &rd);
    if (!rc) rc = fgpu_result_create(ix, &res);
    if (!rc) rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        res->want_scores = true;
        stage_lookup(ix, rd, 0, n, res);
        stage_descriptors(ix, res, res->total_kmers, FGPU_THRESHOLD_UNION);
        stage_colors(ix, FGPU_THRESHOLD_UNION, 1.0, res);  // the threshold only shapes the (discarded) bitmap
        const uint64_t nc = ix->dc.n;
        uint32_t* c = (uint32_t*)malloc(std::max<uint64_t>(1, n * nc) * 4);
        if (!c) throw std::bad_alloc();
        if (n) HIP_TRY(hipMemcpy(c, res->d_scores.p, n * nc * 4, hipMemcpyDeviceToHost));
        *out_counts = c;
    });
    fgpu_result_free(res);
    fgpu_reads_free(rd);
    return rc;
}

// ---- the two per-k-mer tools as line emitters (tools/kmer_conservation.cpp:10-56, tools/kmer_matches.cpp:10-57) -----------------
// One batch of records in, the tool's output lines out: the lookup (and, for kmer-matches, the un-thresholded union scores) on the
// device, the text on all host threads. The reference's workers keep their buffers across records: a record shorter than k leaves
// index::kmer_matches' outputs untouched (src/kmer_matches.cpp:11), so its line repeats the previous record's flags and counts — the
// emitter carries that state from batch to batch, as ONE worker reading the file in order sees it.
struct fgpu_kmer_emitter {
    fgpu_index* ix = nullptr;
    int tool = 0;
    fgpu_result* res = nullptr;
    std::vector<uint8_t> prev_flags;    // kmer-matches: what the last record of at least k bases left behind
    std::vector<uint32_t> prev_counts;
    // buffers of a batch, kept from one add to the next (tens of megabytes each: allocated anew per batch they cost more in page
    // faults than the formatting — the first index of a process ran at 2 M records/s, the second, with a warmed-up allocator, at 8 M)
    HostVec<uint32_t> raw, counts;      // what comes down from the device: pinned slabs of the pool
    std::vector<uint64_t> ko;
    std::vector<uint32_t> ki;
    std::vector<std::string> parts;
};

}  // extern "C"

namespace {

inline void put_u32(std::string& o, uint32_t x) {
    char d[10];
    int n = 0;
    do { d[n++] = (char)('0' + x % 10u); x /= 10u; } while (x);
    while (n) o.push_back(d[--n]);
}

// per-k-mer colour-set ids of an uploaded batch on the host: o[r] .. o[r + 1] index v (0xFFFFFFFF = negative k-mer)
void kmer_ids_to_host(fgpu_index* ix, const fgpu_reads* rd, fgpu_result* res, const uint64_t* offs, uint64_t n, std::vector<uint64_t>& o, std::vector<uint32_t>& v,
                      HostVec<uint32_t>& raw) {
    const uint32_t k = ix->host.dict.k;
    const uint64_t stride = std::max<uint32_t>(1, rd->max_kmers);
    const uint64_t units = rd->has_long ? rd->seg_first[n] : n;
    raw.clear();
    raw.resize(std::max<uint64_t>(1, units * stride));
    if (units) HIP_TRY(hipMemcpy(raw.data(), res->d_kmer_ids.p, units * stride * 4, hipMemcpyDeviceToHost));
    o.resize(n + 1);
    for (uint64_t r = 0; r <= n; ++r) o[r] = rd->kmers_before(r);
    v.resize(std::max<uint64_t>(1, o[n]));
    for (uint64_t r = 0; r < n; ++r) {
        // segments of a long read hold consecutive, non-overlapping k-mer ranges
        const uint64_t u0 = rd->has_long ? rd->seg_first[r] : r, u1 = rd->has_long ? rd->seg_first[r + 1] : r + 1;
        uint64_t at = o[r];
        for (uint64_t u = u0; u < u1; ++u) {
            const uint64_t len = rd->has_long ? rd->seg_end[u] - rd->seg_start[u] : offs[r + 1] - offs[r];
            const uint64_t nk = len >= k ? len - k + 1 : 0;
            memcpy(v.data() + at, raw.data() + u * stride, nk * 4);
            at += nk;
        }
    }
}

template <typename F>
void parallel_ranges(uint64_t n, F fn) {  // fn(thread, begin, end) over [0, n)
    const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), n / 64 + 1);
    if (T == 1) { fn(0u, (uint64_t)0, n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] { fn(t, n * t / T, n * (t + 1) / T); });
    for (auto& x : th) x.join();
}
unsigned parallel_ranges_threads(uint64_t n) { return (unsigned)std::min<uint64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), n / 64 + 1); }

}  // namespace

extern "C" {

int fgpu_kmer_emitter_create(fgpu_index* ix, int tool, fgpu_kmer_emitter** out) {
    if (!ix || !out) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    if (tool != FGPU_TOOL_KMER_CONSERVATION && tool != FGPU_TOOL_KMER_MATCHES) return fail(-EINVAL, "unknown tool");
    *out = nullptr;
    auto* e = new fgpu_kmer_emitter();
    e->ix = ix;
    e->tool = tool;
    if (fgpu_result_create(ix, &e->res)) { delete e; return -EIO; }
    e->prev_counts.assign(ix->host.hybrid.num_colors, 0u);
    *out = e;
    return 0;
}

void fgpu_kmer_emitter_free(fgpu_kmer_emitter* e) {
    if (!e) return;
    fgpu_result_free(e->res);
    delete e;
}

static int kmer_emitter_batch(fgpu_kmer_emitter* e, const char* bases, const uint64_t* offs, uint64_t n, const char* names, const uint64_t* name_offs,
                              char** out, int out_fd, uint64_t* out_len);

int fgpu_kmer_emitter_add(fgpu_kmer_emitter* e, const char* bases, const uint64_t* offs, uint64_t n, const char* names, const uint64_t* name_offs,
                          char** out, uint64_t* out_len) {
    if (!out) return fail(-EINVAL, "null argument");
    return kmer_emitter_batch(e, bases, offs, n, names, name_offs, out, -1, out_len);
}

int fgpu_kmer_emitter_write(fgpu_kmer_emitter* e, const char* bases, const uint64_t* offs, uint64_t n, const char* names, const uint64_t* name_offs,
                            int out_fd, uint64_t* out_len) {
    if (out_fd < 0) return fail(-EINVAL, "bad file descriptor");
    return kmer_emitter_batch(e, bases, offs, n, names, name_offs, nullptr, out_fd, out_len);
}

static int kmer_emitter_batch(fgpu_kmer_emitter* e, const char* bases, const uint64_t* offs, uint64_t n, const char* names, const uint64_t* name_offs,
                              char** out, int out_fd, uint64_t* out_len) {
    if (!e || !offs || !name_offs || !out_len || (n && (!bases || !names))) return fail(-EINVAL, "null argument");
    if (out) *out = nullptr;
    *out_len = 0;
    fgpu_index* ix = e->ix;
    fgpu_reads* rd = nullptr;
    int rc = fgpu_reads_upload(ix, bases, offs, n, &rd);
    if (!rc) rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        fgpu_result* res = e->res;
        const uint32_t k = ix->host.dict.k;
        const uint64_t nc = ix->dc.n;
        res->want_kmer_ids = true;
        res->want_scores = e->tool == FGPU_TOOL_KMER_MATCHES;
        stage_lookup(ix, rd, 0, n, res);
        install_pinned_allocator();
        HostVec<uint32_t>& counts = e->counts;
        if (e->tool == FGPU_TOOL_KMER_MATCHES) {
            stage_descriptors(ix, res, res->total_kmers, FGPU_THRESHOLD_UNION);
            stage_colors(ix, FGPU_THRESHOLD_UNION, 1.0, res);  // (the threshold only shapes the discarded bitmap; the scores are the counts)
            counts.clear();
            counts.resize(std::max<uint64_t>(1, n * nc));
            if (n) HIP_TRY(hipMemcpy(counts.data(), res->d_scores.p, n * nc * 4, hipMemcpyDeviceToHost));
        } else {
            HIP_TRY(hipStreamSynchronize(res->stream));
            if (ix->timing) ix->collect_timing(res->pending);
        }
        std::vector<uint64_t>& ko = e->ko;
        std::vector<uint32_t>& ki = e->ki;
        // A batch without long reads is formatted straight out of the slab that came down (unit r's ids at r * stride): packing them
        // first was one thread copying 31 MB per batch of 65536 records, half the batch's time. Long reads (segments) are packed.
        const bool direct = !rd->has_long;
        const uint64_t stride = std::max<uint32_t>(1, rd->max_kmers);
        if (direct) {
            e->raw.clear();
            e->raw.resize(std::max<uint64_t>(1, n * stride));
            if (n) HIP_TRY(hipMemcpy(e->raw.data(), res->d_kmer_ids.p, n * stride * 4, hipMemcpyDeviceToHost));
        } else {
            kmer_ids_to_host(ix, rd, res, offs, n, ko, ki, e->raw);
        }
        auto ids_of = [&](uint64_t r, uint64_t& nk) -> const uint32_t* {
            if (direct) {
                const uint64_t len = offs[r + 1] - offs[r];
                nk = len >= k ? len - k + 1 : 0;
                return e->raw.data() + r * stride;
            }
            nk = ko[r + 1] - ko[r];
            return ki.data() + ko[r];
        };
        const unsigned T = parallel_ranges_threads(n);
        std::vector<std::string>& parts = e->parts;
        if (parts.size() < T) parts.resize(T);
        for (std::string& p_ : parts) p_.clear();  // (capacity kept)
        if (e->tool == FGPU_TOOL_KMER_CONSERVATION) {
            // `name <tab> #triples [<tab>(start num_kmers color_set_id)]...`: maximal runs of consecutive positive k-mers with one colour-set id
            parallel_ranges(n, [&](unsigned t, uint64_t a, uint64_t b) {
                std::string& o = parts[t];
                std::string body;
                for (uint64_t r = a; r < b; ++r) {
                    o.append(names + name_offs[r], names + name_offs[r + 1]);
                    o.push_back('\t');
                    body.clear();
                    uint32_t triples = 0;
                    uint64_t nk;
                    const uint32_t* id = ids_of(r, nk);
                    for (uint64_t i = 0; i < nk;) {
                        uint64_t j = i + 1;
                        while (j < nk && id[j] == id[i]) ++j;
                        if (id[i] != 0xFFFFFFFFu) {
                            ++triples;
                            body += "\t(";
                            put_u32(body, (uint32_t)i);
                            body.push_back(' ');
                            put_u32(body, (uint32_t)(j - i));
                            body.push_back(' ');
                            put_u32(body, id[i]);
                            body.push_back(')');
                        }
                        i = j;
                    }
                    put_u32(o, triples);
                    o += body;
                    o.push_back('\n');
                }
            });
        } else {
            // `name <tab> #k-mers [<tab>0|1 per k-mer] [<tab>count per colour]`; src[r] = the record whose flags and counts line r shows
            std::vector<int64_t> src(n);
            int64_t last = -1;
            for (uint64_t r = 0; r < n; ++r) {
                if (offs[r + 1] - offs[r] >= k) last = (int64_t)r;
                src[r] = last;
            }
            parallel_ranges(n, [&](unsigned t, uint64_t a, uint64_t b) {
                std::string& o = parts[t];
                for (uint64_t r = a; r < b; ++r) {
                    o.append(names + name_offs[r], names + name_offs[r + 1]);
                    o.push_back('\t');
                    if (src[r] < 0) {  // (no record of at least k bases yet in this batch: what the batches before left)
                        put_u32(o, (uint32_t)e->prev_flags.size());
                        for (uint8_t f : e->prev_flags) { o.push_back('\t'); o.push_back(f ? '1' : '0'); }
                        for (uint32_t c : e->prev_counts) { o.push_back('\t'); put_u32(o, c); }
                    } else {
                        const uint64_t q = (uint64_t)src[r];
                        uint64_t nk;
                        const uint32_t* id = ids_of(q, nk);
                        put_u32(o, (uint32_t)nk);
                        for (uint64_t i = 0; i < nk; ++i) { o.push_back('\t'); o.push_back(id[i] != 0xFFFFFFFFu ? '1' : '0'); }
                        const uint32_t* c = counts.data() + q * nc;
                        for (uint64_t j = 0; j < nc; ++j) { o.push_back('\t'); put_u32(o, c[j]); }
                    }
                    o.push_back('\n');
                }
            });
            if (last >= 0) {  // the state the next batch starts from
                const uint64_t q = (uint64_t)last;
                uint64_t nk;
                const uint32_t* id = ids_of(q, nk);
                e->prev_flags.resize(nk);
                for (uint64_t i = 0; i < nk; ++i) e->prev_flags[i] = id[i] != 0xFFFFFFFFu;
                e->prev_counts.assign(counts.data() + q * nc, counts.data() + (q + 1) * nc);
            }
        }
        uint64_t total = 0;
        for (const std::string& p : parts) total += p.size();
        if (out) {
            char* buf = (char*)malloc(std::max<uint64_t>(1, total));
            if (!buf) throw std::bad_alloc();
            uint64_t at = 0;
            for (const std::string& p : parts) { memcpy(buf + at, p.data(), p.size()); at += p.size(); }
            *out = buf;
        } else {  // straight to the file, the threads' pieces in order (kmer-matches writes 14 KB per record at 4546 colours: no second copy)
            for (const std::string& p : parts) {
                const char* q = p.data();
                size_t left = p.size();
                while (left) {
                    const ssize_t w = ::write(out_fd, q, left);
                    if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("cannot write the output: ") + strerror(errno)); }
                    q += w;
                    left -= (size_t)w;
                }
            }
        }
        *out_len = total;
    });
    fgpu_reads_free(rd);
    return rc;
}

// ---- output formatters (host) ---------------------------------------------------------------------------
struct fgpu_formatter {
    int format;
    CompressedFormatter comp;
};

int fgpu_formatter_create(int format, uint64_t num_colors, fgpu_formatter** out, char** header, uint64_t* header_len) {
    if (!out || !header || !header_len) return fail(-EINVAL, "null argument");
    if (format < FGPU_FMT_ASCII || format > FGPU_FMT_COMPRESSED)
        return fail(-EINVAL, "Unknown output format. Supported formats: ascii, binary, compressed.");  // tools/pseudoalign.cpp:317-320
    return guarded([&] {
        auto* f = new fgpu_formatter();
        f->format = format;
        std::string h;
        if (format == FGPU_FMT_COMPRESSED) f->comp.init((uint32_t)num_colors, h);
        *header = (char*)malloc(std::max<size_t>(1, h.size()));
        memcpy(*header, h.data(), h.size());
        *header_len = h.size();
        *out = f;
    });
}

int fgpu_formatter_add(fgpu_formatter* f, uint32_t first_id, const uint64_t* offsets, const uint32_t* colors, uint64_t n,
                       char** out, uint64_t* out_len) {
    if (!f || !offsets || !out || !out_len) return fail(-EINVAL, "null argument");
    return guarded([&] {
        std::string s;
        if (f->format == FGPU_FMT_ASCII) format_ascii(first_id, offsets, colors, n, s);
        else if (f->format == FGPU_FMT_BINARY) format_binary(first_id, offsets, colors, n, s);
        else f->comp.add_batch(first_id, offsets, colors, n, s);
        *out = (char*)malloc(std::max<size_t>(1, s.size()));
        if (!*out) throw std::bad_alloc();
        memcpy(*out, s.data(), s.size());
        *out_len = s.size();
    });
}

int fgpu_formatter_finish(fgpu_formatter* f, char** out, uint64_t* out_len) {
    if (!f || !out || !out_len) return fail(-EINVAL, "null argument");
    int rc = guarded([&] {
        std::string s;
        if (f->format == FGPU_FMT_COMPRESSED) f->comp.finish(s);
        *out = (char*)malloc(std::max<size_t>(1, s.size()));
        memcpy(*out, s.data(), s.size());
        *out_len = s.size();
    });
    delete f;
    return rc;
}

// ---- query reader -------------------------------------------------------------------------------------------
struct fgpu_fastx {
    static constexpr int RING = 4;  // batches alive at a time: a worker loop keeps several passes in flight
    FastxReader reader;
    HostVec<char> bases[RING];
    HostVec<uint64_t> offs[RING];
    std::vector<char> names;
    std::vector<uint64_t> name_offs{0};
    int cur = RING - 1;
    fgpu_fastx(const char* path, unsigned threads, uint64_t begin, uint64_t end) : reader(path, threads, begin, end) {}
};

int fgpu_fastx_open(const char* path, fgpu_fastx** out) { return fgpu_fastx_open_part(path, 0, 0, ~0ULL, out); }

int fgpu_fastx_open_part(const char* path, unsigned threads, uint64_t begin, uint64_t end, fgpu_fastx** out) {
    if (!path || !out) return fail(-EINVAL, "null argument");
    *out = nullptr;
    return guarded([&] {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) install_pinned_allocator();
        else (void)hipGetLastError();
        *out = new fgpu_fastx(path, threads, begin, end);
    });
}

int fgpu_fastx_text_size(const char* path, uint64_t* size, int* can_be_read_in_parts) {
    if (!path || !size || !can_be_read_in_parts) return fail(-EINVAL, "null argument");
    return guarded([&] {
        bool parts = false;
        *size = fastx_text_size(path, parts);
        *can_be_read_in_parts = parts ? 1 : 0;
    });
}

int fgpu_fastx_count(const char* path, unsigned threads, uint64_t begin, uint64_t end, uint64_t* num_reads) {
    if (!path || !num_reads) return fail(-EINVAL, "null argument");
    return guarded([&] {
        FastxReader r(path, threads, begin, end);
        const uint64_t total = r.count();
        *num_reads = total;
    });
}

int fgpu_fastx_count_part(fgpu_fastx* f, uint64_t* num_reads) {
    if (!f || !num_reads) return fail(-EINVAL, "null argument");
    return guarded([&] {
        if (!f->reader.count_records(*num_reads))
            throw std::runtime_error("this query file has to be read to be counted (a gzip stream, or FASTQ with wrapped lines)");
    });
}

int fgpu_fastx_next(fgpu_fastx* f, uint64_t max_reads, const char** bases, const uint64_t** offs, uint64_t* n) {
    if (!f || !bases || !offs || !n) return fail(-EINVAL, "null argument");
    if (max_reads == 0) return fail(-EINVAL, "max_reads must be positive");
    return guarded([&] {
        f->cur = (f->cur + 1) % fgpu_fastx::RING;
        HostVec<char>& b = f->bases[f->cur];
        f->reader.next(max_reads, b, f->offs[f->cur]);
        *n = f->offs[f->cur].size() - 1;
        b.reserve(b.size() + 1);
        *bases = b.data();
        *offs = f->offs[f->cur].data();
    });
}

int fgpu_fastx_names(fgpu_fastx* f, const char** names, const uint64_t** name_offs) {
    if (!f || !names || !name_offs) return fail(-EINVAL, "null argument");
    return guarded([&] {
        f->reader.names(f->names, f->name_offs);
        f->names.reserve(f->names.size() + 1);
        *names = f->names.data();
        *name_offs = f->name_offs.data();
    });
}

void fgpu_fastx_close(fgpu_fastx* f) { delete f; }
int fgpu_fastx_ring(void) { return fgpu_fastx::RING; }

// ---- export ---------------------------------------------------------------------------------------------
int fgpu_export_sizes(const fgpu_index* ix, uint64_t* unitig_bases, uint64_t* num_unitigs, uint64_t* color_words,
                      uint64_t* color_bits, uint64_t* num_sets) {
    if (!ix) return fail(-EINVAL, "null argument");
    if (unitig_bases) *unitig_bases = ix->host.dict.total_bases;
    if (num_unitigs) *num_unitigs = ix->host.dict.num_unitigs();
    if (color_words) *color_words = (ix->host.hybrid.nbits + 63) / 64;
    if (color_bits) *color_bits = ix->host.hybrid.nbits;
    if (num_sets) *num_sets = ix->host.hybrid.num_sets();
    return 0;
}

int fgpu_export(const fgpu_index* ix, char* unitig_bases, uint64_t* unitig_off, uint32_t* unitig_csid, uint64_t* color_words,
                uint64_t* color_offsets, uint32_t* thresholds) {
    if (!ix) return fail(-EINVAL, "null argument");
    const Dict& d = ix->host.dict;
    const HybridSets& h = ix->host.hybrid;
    if (unitig_bases)
        for (uint64_t i = 0; i < d.total_bases; ++i) {
            const uint64_t w = d.strings[i >> 5];
            const uint32_t c = (uint32_t)((w >> (i & 31)) & 1) | (uint32_t)(((w >> (32 + (i & 31))) & 1) << 1);
            unitig_bases[i] = "ACGT"[c];
        }
    if (unitig_off) memcpy(unitig_off, d.unitig_off.data(), d.unitig_off.size() * 8);
    if (unitig_csid) memcpy(unitig_csid, d.unitig_csid.data(), d.unitig_csid.size() * 4);
    if (color_words) memcpy(color_words, h.bits.data(), ((h.nbits + 63) / 64) * 8);
    if (color_offsets) memcpy(color_offsets, h.offsets.data(), h.offsets.size() * 8);
    if (thresholds) { thresholds[0] = h.num_colors; thresholds[1] = h.sparse_thr; thresholds[2] = h.dense_thr; }
    return 0;
}

int fgpu_dump(const fgpu_index* ix, const char* basename) {
    if (!ix || !basename) return fail(-EINVAL, "null argument");
    return guarded([&] {
        const Dict& d = ix->host.dict;
        const HybridSets& h = ix->host.hybrid;
        const std::string base(basename);
        auto open = [&](const char* suffix) {
            FILE* f = fopen((base + suffix).c_str(), "w");
            if (!f) throw std::runtime_error("cannot open output file");
            return f;
        };
        FILE* f = open(".metadata.txt");
        fprintf(f, "k=%u\nnum_kmers=%llu\nnum_colors=%u\nnum_unitigs=%llu\nnum_color_sets=%llu\n", d.k,
                (unsigned long long)d.num_kmers, h.num_colors, (unsigned long long)d.num_unitigs(), (unsigned long long)h.num_sets());
        fclose(f);
        f = open(".filenames.txt");
        for (uint32_t i = 0; i < h.num_colors; ++i)
            fprintf(f, "%s\n", i < ix->host.filenames.size() ? ix->host.filenames[i].c_str() : "");
        fclose(f);
        f = open(".unitigs.fa");
        std::string seq;
        for (uint64_t u = 0; u < d.num_unitigs(); ++u) {
            fprintf(f, "> color_set_id=%u\n", d.unitig_csid[u]);
            seq.clear();
            for (uint64_t i = d.unitig_off[u]; i < d.unitig_off[u + 1]; ++i) {
                const uint64_t w = d.strings[i >> 5];
                seq.push_back("ACGT"[(uint32_t)((w >> (i & 31)) & 1) | (uint32_t)(((w >> (32 + (i & 31))) & 1) << 1)]);
            }
            seq.push_back('\n');
            fwrite(seq.data(), 1, seq.size(), f);
        }
        fclose(f);
        f = open(".color_sets.txt");
        {   // 0.9 G integers for a salmonella_4546-sized index: strips of sets are decoded and formatted by all threads, written in order
            const uint64_t ns = h.num_sets();
            const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::min(32u, std::thread::hardware_concurrency())), ns / 1024 + 1);
            const uint64_t CH = 8192;
            std::vector<std::string> bufs(T);
            auto put_u32 = [](std::string& o, uint32_t x) {
                char d[10];
                int n = 0;
                do { d[n++] = (char)('0' + x % 10u); x /= 10u; } while (x);
                while (n) o.push_back(d[--n]);
            };
            for (uint64_t g0 = 0; g0 < ns; g0 += (uint64_t)T * CH) {
                auto work = [&](unsigned t) {
                    std::string& o = bufs[t];
                    o.clear();
                    std::vector<uint32_t> set;
                    const uint64_t a = std::min(ns, g0 + t * CH), b = std::min(ns, a + CH);
                    for (uint64_t id = a; id < b; ++id) {
                        hybrid_decode(h, id, set);
                        o += "size=";
                        put_u32(o, (uint32_t)set.size());
                        o.push_back(' ');
                        for (size_t j = 0; j < set.size(); ++j) {
                            put_u32(o, set[j]);
                            if (j + 1 != set.size()) o.push_back(' ');
                        }
                        o.push_back('\n');
                    }
                };
                std::vector<std::thread> th;
                for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
                work(0);
                for (auto& x : th) x.join();
                for (unsigned t = 0; t < T; ++t)
                    if (!bufs[t].empty() && fwrite(bufs[t].data(), 1, bufs[t].size(), f) != bufs[t].size()) { fclose(f); throw std::runtime_error("write error on the color sets file"); }
            }
        }
        fclose(f);
    });
}

}  // extern "C"

#include "stream_pipeline.hip.h"
