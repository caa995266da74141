// The worker loop of `fulgor pseudoalign` on the host side of the C ABI: pseudoalign_orchestrator / pseudoalign_worker
// (tools/pseudoalign.cpp:12-89) with the per-read loop replaced by batched passes. Included at the end of fulgor_gpu.hip.
//
//   parser threads (host/fastx_reader.hpp)  --ranges parsed into pinned chunks, in file order-->
//   W workers, each with its own result (= its own HIP stream):
//       take the next batch (whole chunks, at most batch_reads reads; read ids = file order)
//       H2D of every chunk as it lies (one copy for its bases, one for its offsets; the offsets are rebased on the device)
//       k1_lookup -> colour stage (no u32 colour lists) -> device-side formatter -> D2H into the result's pinned buffer
//       write the records when it is this batch's turn (file order), hand the chunks back to the reader
//
// While one worker waits for its copy out, another one's kernels run and a third one's reads go up: the stages of different
// batches overlap on the copy engines and the CUs. Nothing is copied twice on the host: a base is written once by the thread
// that parses it (into pinned memory) and read once by the GPU's copy engine.
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <sstream>
#include <thread>

namespace {

uint64_t now_ns() { return fastx_now_ns(); }

struct StreamBatch {
    std::vector<FastxChunk> chunks;
    uint64_t seq = 0, first_id = 0, reads = 0, bases = 0, max_len = 0;
};

struct StreamBatchLog {
    uint64_t seq, reads, bases, out_bytes;
    uint64_t t_begin, t_acquired, t_copied, t_issued, t_colours, t_formatted, t_turn, t_written;  // ns since the start of the run
};

// what a worker keeps between runs (device buffers sized for one batch, a result = a stream): allocating and freeing device
// memory synchronises the whole device, so they stay with the index
struct StreamWorkerState {
    fgpu_result* res = nullptr;
    DevBuf d_bases, d_offs;
};

struct StreamRun {
    fgpu_index* ix;
    FastxReader* rd;
    int fd, algo, format;
    double tau;
    uint64_t batch_reads;
    // input side: batches are cut from the chunk sequence one at a time
    std::mutex in_mu;
    uint64_t next_seq = 0, next_id = 0;
    bool eof = false, have_carry = false;
    FastxChunk carry;
    // output side: records leave in file order
    std::mutex out_mu;
    std::condition_variable out_cv;
    uint64_t out_turn = 0;
    // failure of any worker ends the run
    std::atomic<bool> failed{false};
    std::mutex err_mu;
    std::string error;
    std::atomic<uint64_t> reads{0}, mapped{0}, out_bytes{0};
    uint64_t t0 = 0, slabs0 = 0, slab_bytes0 = 0;
    std::mutex log_mu;
    std::vector<StreamBatchLog> log;

    void fail_with(const std::string& msg) {
        {
            std::lock_guard<std::mutex> g(err_mu);
            if (error.empty()) error = msg;
        }
        failed.store(true);
        out_cv.notify_all();
    }

    // the next batch: whole chunks in file order, at most `batch_reads` reads (a single larger chunk is a batch of its own).
    // The first batches are smaller: the pipeline fills sooner (its first records leave after a quarter of the time).
    bool acquire(StreamBatch& b) {
        std::lock_guard<std::mutex> g(in_mu);
        if (eof && !have_carry) return false;
        const uint64_t limit = std::min<uint64_t>(batch_reads, (uint64_t)65536 << std::min<uint64_t>(next_seq, 16));
        b.chunks.clear();
        b.reads = b.bases = b.max_len = 0;
        for (;;) {
            FastxChunk c;
            if (have_carry) { c = std::move(carry); have_carry = false; }
            else if (eof || !rd->pop_chunk(c)) { eof = true; break; }
            if (b.reads && b.reads + c.reads() > limit) { carry = std::move(c); have_carry = true; break; }
            b.reads += c.reads();
            b.bases += c.bases.size();
            b.max_len = std::max(b.max_len, c.max_len);
            b.chunks.push_back(std::move(c));
            if (b.reads >= limit) break;
        }
        if (b.reads == 0) return false;
        b.seq = next_seq++;
        b.first_id = next_id;
        next_id += b.reads;
        return true;
    }
};

void write_all(int fd, const char* p, size_t n) {
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error(std::string("cannot write the output: ") + strerror(errno));
        }
        p += w;
        n -= (size_t)w;
    }
}

// one batch through the device; returns the formatted records (a view into the result's pinned buffer)
void stream_one_batch(StreamRun& run, StreamWorkerState& w, StreamBatch& b, StreamBatchLog& lg, const char*& out, uint64_t& out_len) {
    fgpu_index* ix = run.ix;
    fgpu_result* res = w.res;
    const uint32_t k = ix->host.dict.k;
    const uint64_t max_nk = b.max_len >= k ? b.max_len - k + 1 : 0;
    fgpu_reads* uploaded = nullptr;  // (long reads: the general upload path)
    fgpu_reads local;
    const fgpu_reads* reads = &local;
    if (max_nk > SEG_KMERS) {
        // reads longer than one lookup unit are cut into overlapping segments by fgpu_reads_upload: gather the batch for it
        std::vector<char> bases(b.bases);
        std::vector<uint64_t> offs(b.reads + 1, 0);
        uint64_t at = 0, r = 0;
        for (const FastxChunk& c : b.chunks) {
            memcpy(bases.data() + at, c.bases.data(), c.bases.size());
            for (uint64_t i = 1; i <= c.reads(); ++i) offs[r + i] = at + c.offs[i];
            at += c.bases.size();
            r += c.reads();
        }
        if (fgpu_reads_upload(ix, bases.data(), offs.data(), b.reads, &uploaded)) throw std::runtime_error(fgpu_last_error());
        reads = uploaded;
    } else {
        hipStream_t s = res->stream_lookup, sin = res->stream_in;
        w.d_bases.ensure(b.bases + 1024);  // the lookup kernel reads up to 576 bases past a unit's start unconditionally
        w.d_offs.ensure((b.reads + 1) * 8);
        {   // every chunk as it lies: two copies per chunk, beside the kernels of the other batches, on a copy engine for copies in
            // (copy_engines.hip.h) or, failing that, on the result's kernel-free copy stream
            CopyEngines& ce = CopyEngines::get();
            const bool direct = res->sig_in.handle && ce.usable(ix->device);
            const uint64_t t_h2d = now_ns();
            // (an engine reads the host memory as it is mapped for the device: only buffers of the pinned pool go that way — a buffer
            // that the pool handed out before an index was open, or beyond its cap, is plain memory, which the HIP call stages)
            int64_t armed = 0;
            if (direct) {
                for (const FastxChunk& c : b.chunks) if (c.reads()) armed += (c.bases.size() && c.bases.pinned() ? 1 : 0) + (c.offs.pinned() ? 1 : 0);
                ce.arm(res->sig_in, armed);
            }
            bool engines = direct, staged = false;
            auto piece = [&](void* dst, const void* src, size_t n, bool pinned) {
                if (!n) return;
                if (engines && pinned) {
                    if (ce.h2d(dst, src, n, res->sig_in, res->lane)) { --armed; return; }
                    engines = false;  // (refused: this piece and the rest through the HIP runtime; the engines are left alone from now on)
                    ce.disable();
                }
                staged = true;
                HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, sin));
            };
            // (if anything below throws while copies are in flight, the chunks' pinned slabs must not go back to the pool under an
            // engine that still reads them: the copies issued so far are waited for first)
            struct DrainOnFailure {
                CopyEngines& ce; fgpu_result* res; int64_t& armed; bool direct, done = false;
                ~DrainOnFailure() {
                    if (done) return;
                    if (direct) { ce.disarm(res->sig_in, armed); ce.drain(res->sig_in); }
                    (void)hipStreamSynchronize(res->stream_in);
                }
            } drain_on_failure{ce, res, armed, direct};
            {
                Timed t(ix, res, direct ? -1 : FGPU_K_H2D, sin);
                uint64_t at = 0, r = 0;
                bool first = true;
                for (const FastxChunk& c : b.chunks) {
                    // a range inside one long record holds no read: its chunk has no offsets at all (not even the leading 0 — copying
                    // "offs[0]" of such a chunk at the head of a batch put whatever the buffer held in front of the batch's offsets)
                    if (!c.reads()) continue;
                    if (c.bases.size()) piece(w.d_bases.as<char>() + at, c.bases.data(), c.bases.size(), c.bases.pinned());
                    // (the first chunk with reads brings offs[0] = 0 along)
                    if (first) piece(w.d_offs.as<uint64_t>(), c.offs.data(), (c.reads() + 1) * 8, c.offs.pinned());
                    else piece(w.d_offs.as<uint64_t>() + r + 1, c.offs.data() + 1, c.reads() * 8, c.offs.pinned());
                    first = false;
                    at += c.bases.size();
                    r += c.reads();
                }
            }
            if (direct) {
                ce.disarm(res->sig_in, armed);  // (pieces that went the other way)
                armed = 0;
                ce.wait(res->sig_in);
                if (ix->timing) ix->add_timing(FGPU_K_H2D, (now_ns() - t_h2d) / 1e6);
            }
            if (!direct || staged) HIP_TRY(hipStreamSynchronize(sin));
            drain_on_failure.done = true;
        }
        lg.t_copied = now_ns() - run.t0;
        {   // offsets of a chunk count from the chunk's first base: add its position in the batch
            uint64_t at = 0, r = 0;
            RebaseTable tab;
            tab.count = 0;
            auto flush = [&] {
                if (!tab.count) return;
                tab.first_read[tab.count] = r;
                const uint64_t n = r - tab.first_read[0];
                hipLaunchKernelGGL(k_offs_rebase, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 1024)), dim3(256), 0, s, w.d_offs.as<uint64_t>(), tab);
                tab.count = 0;
            };
            for (const FastxChunk& c : b.chunks) {
                if (!c.reads()) continue;
                tab.first_read[tab.count] = r;
                tab.base[tab.count] = at;
                ++tab.count;
                at += c.bases.size();
                r += c.reads();
                if (tab.count == REBASE_RANGES) flush();
            }
            flush();
            HIP_TRY(hipGetLastError());
        }
        local.ix = ix;
        local.n = b.reads;
        local.d_bases = w.d_bases;  // (borrowed: a fgpu_reads on the stack releases nothing)
        local.d_offs = w.d_offs;
        local.whole_only = true;
        local.whole_bases = b.bases;
        local.whole_kmers = b.bases;  // an upper bound: what bounds the id lists; the byte accounting is not asked of a streamed batch
        local.max_total_kmers = max_nk;
        local.max_kmers = (uint32_t)max_nk;
    }
    struct FreeUploaded { fgpu_reads* r; ~FreeUploaded() { if (r) fgpu_reads_free(r); } } free_uploaded{uploaded};
    stage_lookup_on(ix, reads, 0, b.reads, res);
    if (res->stream_lookup != res->stream) {
        HIP_TRY(hipEventRecord(res->ev_lookup, res->stream_lookup));
        HIP_TRY(hipStreamWaitEvent(res->stream, res->ev_lookup, 0));
    }
    lg.t_issued = now_ns() - run.t0;
    stage_descriptors(ix, res, res->total_kmers, run.algo);
    stage_colors(ix, run.algo, run.tau, res);  // (waits for the pass: the chunks' copies have completed)
    lg.t_colours = now_ns() - run.t0;
    run.mapped += res->mapped;
    out = nullptr;
    out_len = 0;
    if (run.fd >= 0) {
        if (b.first_id + b.reads > 0xFFFFFFFFull) throw std::runtime_error("more than 2^32 reads: read ids are 32-bit in the reference's output formats");
        if (fgpu_result_format_view(res, run.format, (uint32_t)b.first_id, &out, &out_len)) throw std::runtime_error(fgpu_last_error());
    }
    lg.t_formatted = now_ns() - run.t0;
}

void stream_worker(StreamRun& run, StreamWorkerState& w) {
    try {
        HIP_TRY(hipSetDevice(run.ix->device));
        StreamBatch b;
        for (;;) {
            if (run.failed.load()) return;
            StreamBatchLog lg{};
            lg.t_begin = now_ns() - run.t0;
            if (!run.acquire(b)) return;
            lg.t_acquired = now_ns() - run.t0;
            lg.seq = b.seq;
            lg.reads = b.reads;
            lg.bases = b.bases;
            const char* out = nullptr;
            uint64_t out_len = 0;
            stream_one_batch(run, w, b, lg, out, out_len);
            for (FastxChunk& c : b.chunks) run.rd->recycle_chunk(std::move(c));
            b.chunks.clear();
            {   // records leave in file order
                std::unique_lock<std::mutex> g(run.out_mu);
                run.out_cv.wait(g, [&] { return run.out_turn == b.seq || run.failed.load(); });
                if (run.failed.load()) return;
                lg.t_turn = now_ns() - run.t0;
                if (out_len) write_all(run.fd, out, out_len);
                run.out_turn = b.seq + 1;
            }
            run.out_cv.notify_all();
            lg.t_written = now_ns() - run.t0;
            lg.out_bytes = out_len;
            run.reads += b.reads;
            run.out_bytes += out_len;
            std::lock_guard<std::mutex> g(run.log_mu);
            run.log.push_back(lg);
        }
    } catch (std::exception& e) {
        run.fail_with(e.what());
    }
}

std::mutex g_report_mu;
std::string g_stream_report;

}  // namespace

// the worker states an index keeps between runs
struct fgpu_stream_cache {
    std::mutex mu;
    std::vector<StreamWorkerState> idle;
};

namespace {
std::mutex g_caches_mu;
std::map<fgpu_index*, fgpu_stream_cache*> g_caches;

fgpu_stream_cache* cache_of(fgpu_index* ix) {
    std::lock_guard<std::mutex> g(g_caches_mu);
    auto it = g_caches.find(ix);
    if (it != g_caches.end()) return it->second;
    return g_caches[ix] = new fgpu_stream_cache();
}
}  // namespace

// called by fgpu_close
void fgpu_stream_cache_release(fgpu_index* ix) {
    fgpu_stream_cache* c = nullptr;
    {
        std::lock_guard<std::mutex> g(g_caches_mu);
        auto it = g_caches.find(ix);
        if (it == g_caches.end()) return;
        c = it->second;
        g_caches.erase(it);
    }
    for (StreamWorkerState& w : c->idle) {
        fgpu_result_free(w.res);
        w.d_bases.release();
        w.d_offs.release();
    }
    delete c;
}

extern "C" {

int fgpu_pseudoalign_stream(fgpu_index* ix, fgpu_fastx* query, int out_fd, int algo, double tau, int format, uint64_t first_read_id,
                            int write_header, uint64_t batch_reads, unsigned workers, uint64_t* num_reads, uint64_t* num_mapped) {
    if (!ix || !query) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    if (algo != FGPU_FULL_INTERSECTION && algo != FGPU_THRESHOLD_UNION) return fail(-EINVAL, "unknown algorithm");
    if (algo == FGPU_THRESHOLD_UNION && !(tau > 0.0 && tau <= 1.0))
        return fail(-EINVAL, "threshold must be a float in (0.0,1.0]");  // tools/pseudoalign.cpp:275-278
    if (format != FGPU_FMT_ASCII && format != FGPU_FMT_BINARY && format != FGPU_FMT_COMPRESSED)
        return fail(-EINVAL, "Unknown output format. Supported formats: ascii, binary, compressed.");  // tools/pseudoalign.cpp:317-320
    // defaults from the grid of profiles/e2e_stream.py (profiles/r5/e2e_grid_r5.txt): 16 to 32 parser threads, 4 to 6 workers and
    // batches of 2^18 or 2^19 reads all land within the run-to-run spread of each other; smaller batches fill the pipeline sooner
    // and pin less host memory
    // default batch: 2^18 reads for the compressed records; 2^15 for ascii / binary, whose records are an order of magnitude larger
    // (3 KB per read on a 4546-colour collection: the copy out is the whole run at any batch size — 18.7 M reads/s = 54.6 GB/s
    // from 2^14 to 2^18, profiles/r5/e2e_formats_r5.txt — and five result buffers of 2^18 reads would pin 4 GB of host memory)
    if (batch_reads == 0) batch_reads = env_u64("FULGOR_STREAM_BATCH", format == FGPU_FMT_COMPRESSED ? 1u << 18 : 1u << 15);
    if (workers == 0) workers = (unsigned)env_u64("FULGOR_STREAM_WORKERS", 5);
    workers = std::min(workers, 16u);
    std::vector<StreamWorkerState> states;
    fgpu_stream_cache* cache = nullptr;
    StreamRun run;
    int rc = guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        run.ix = ix;
        run.rd = &query->reader;
        run.fd = out_fd;
        run.algo = algo;
        run.format = format;
        run.tau = tau;
        run.batch_reads = batch_reads;
        run.next_id = first_read_id;
        run.t0 = now_ns();
        run.slabs0 = SlabPool::get().fresh_allocs();
        run.slab_bytes0 = SlabPool::get().fresh_bytes();
        query->reader.set_want_names(false);
        if (out_fd >= 0 && write_header && format == FGPU_FMT_COMPRESSED) {  // the file header (src/ps_utils.cpp:158-164)
            CompressedFormatter f;
            std::string h;
            f.init(ix->host.hybrid.num_colors, h);
            write_all(out_fd, h.data(), h.size());
        }
        cache = cache_of(ix);
        {
            std::lock_guard<std::mutex> g(cache->mu);
            while (states.size() < workers && !cache->idle.empty()) { states.push_back(cache->idle.back()); cache->idle.pop_back(); }
        }
        while (states.size() < workers) {
            StreamWorkerState w;
            if (fgpu_result_create(ix, &w.res)) throw std::runtime_error(fgpu_last_error());
            states.push_back(w);
        }
        for (auto& w : states) w.res->reserve_reads = batch_reads + batch_reads / 16;
        std::vector<std::thread> th;
        for (unsigned i = 1; i < workers; ++i) th.emplace_back([&run, &states, i] { stream_worker(run, states[i]); });
        stream_worker(run, states[0]);
        for (auto& t : th) t.join();
        if (run.failed.load()) throw std::runtime_error(run.error);
    });
    if (cache) {
        std::lock_guard<std::mutex> g(cache->mu);
        for (auto& w : states) cache->idle.push_back(w);
    } else {
        for (auto& w : states) { fgpu_result_free(w.res); w.d_bases.release(); w.d_offs.release(); }
    }
    if (rc) return rc;
    if (num_reads) *num_reads = run.reads.load();
    if (num_mapped) *num_mapped = run.mapped.load();
    {   // the run's timeline, for fgpu_last_stream_report
        const uint64_t t_end = now_ns() - run.t0;
        std::sort(run.log.begin(), run.log.end(), [](const StreamBatchLog& a, const StreamBatchLog& b) { return a.seq < b.seq; });
        const FastxSource::Stats st = query->reader.stats();
        std::ostringstream o;
        o << "stream: " << run.reads.load() << " reads, " << run.log.size() << " batches, " << workers << " workers, " << st.threads
          << " parser threads, " << t_end / 1e6 << " ms, " << run.out_bytes.load() << " output bytes\n";
        o << "parser: " << st.bytes << " text bytes in " << st.ranges << " ranges; per thread " << st.parse_ns / 1e6 / std::max(1u, st.threads)
          << " ms parsing, " << st.wait_ns / 1e6 / std::max(1u, st.threads) << " ms waiting for the workers; host buffers pinned anew during the run: "
          << SlabPool::get().fresh_allocs() - run.slabs0 << " (" << (SlabPool::get().fresh_bytes() - run.slab_bytes0) / 1e6 << " MB)\n";
        o << CopyEngines::get().report() << "\n";
        o << "# seq reads bases out_bytes | ms since start: begin acquired copied-in issued colours formatted+copied-out turn written\n";
        for (const StreamBatchLog& l : run.log)
            o << l.seq << " " << l.reads << " " << l.bases << " " << l.out_bytes << " | " << l.t_begin / 1e6 << " " << l.t_acquired / 1e6 << " " << l.t_copied / 1e6 << " "
              << l.t_issued / 1e6 << " " << l.t_colours / 1e6 << " " << l.t_formatted / 1e6 << " " << l.t_turn / 1e6 << " " << l.t_written / 1e6 << "\n";
        std::lock_guard<std::mutex> g(g_report_mu);
        g_stream_report = o.str();
    }
    return 0;
}

int fgpu_prepare_host(int device, unsigned reader_threads, unsigned workers, uint64_t batch_reads, uint64_t text_bytes_per_read, int fastq,
                      uint64_t out_bytes_per_read, uint64_t total_text_bytes) {
    if (device < 0) return fail(-EINVAL, "invalid device ordinal");
    return guarded([&] {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            throw std::runtime_error("no HIP device available: the pseudoalignment engine has no CPU execution path");
        if (device >= ndev) throw std::runtime_error("invalid device ordinal");
        HIP_TRY(hipSetDevice(device));
        install_pinned_allocator();
        // (an index that is being opened finishes starting its device first: the two would slow each other down)
        for (int waited = 0; g_device_startups.load() > 0 && waited < 2000; ++waited) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (reader_threads == 0) reader_threads = FastxReader::default_threads();
        if (workers == 0) workers = (unsigned)env_u64("FULGOR_STREAM_WORKERS", 5);
        if (batch_reads == 0) batch_reads = env_u64("FULGOR_STREAM_BATCH", 1u << 18);
        const uint64_t range = FastxReader::default_range_bytes();
        // chunks alive at a time: the ranges parsed ahead of the workers (the reader's window) and the batches in flight
        const uint64_t per_batch = (batch_reads * std::max<uint64_t>(1, text_bytes_per_read) + range - 1) / range + 1;
        uint64_t chunks = std::min<uint64_t>((uint64_t)reader_threads + 8 + workers * per_batch, 256);
        if (total_text_bytes) {  // a small query: no more chunks than it has ranges, no larger batches than it has reads
            chunks = std::min<uint64_t>(chunks, total_text_bytes / range + 2);
            batch_reads = std::min<uint64_t>(batch_reads, total_text_bytes / std::max<uint64_t>(1, text_bytes_per_read) + 1);
        }
        size_t bb = 0, ob = 0;
        FastxChunk::slab_requests(range, fastq != 0, bb, ob);
        struct Held { void* p; size_t bytes; bool pinned; };
        std::vector<Held> held;
        std::mutex mu;
        // (pinning is a system call per buffer that the runtime does not serialise: a few threads share the list)
        std::vector<size_t> want;
        for (uint64_t c = 0; c < chunks; ++c) { want.push_back(bb); want.push_back(ob); }
        const uint64_t out_bytes = out_bytes_per_read ? (batch_reads + batch_reads / 16) * out_bytes_per_read * 5 / 4 + 4096 : 0;
        for (unsigned w = 0; out_bytes && w < workers; ++w) want.push_back((size_t)out_bytes);
        std::atomic<size_t> next{0};
        std::string error;
        auto work = [&] {
            try {
                (void)hipSetDevice(device);
                for (size_t i; (i = next++) < want.size();) {
                    Held h{nullptr, 0, false};
                    h.p = SlabPool::get().take(want[i], h.bytes, h.pinned);
                    std::lock_guard<std::mutex> g(mu);
                    held.push_back(h);
                }
            } catch (std::exception& e) {
                std::lock_guard<std::mutex> g(mu);
                error = e.what();
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < 4; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        for (const Held& h : held) SlabPool::get().give(h.p, h.bytes, h.pinned);  // (all taken first: a slab given back early would be handed out again)
        if (!error.empty()) throw std::runtime_error(error);
    });
}

int fgpu_stream_prepare(fgpu_index* ix, int format, uint64_t batch_reads, unsigned workers, uint32_t max_read_bases, uint64_t out_bytes_per_read) {
    if (!ix) return fail(-EINVAL, "null argument");
    NEED_DEVICE(ix);
    if (format != FGPU_FMT_ASCII && format != FGPU_FMT_BINARY && format != FGPU_FMT_COMPRESSED) return fail(-EINVAL, "unknown format");
    return guarded([&] {
        HIP_TRY(hipSetDevice(ix->device));
        if (batch_reads == 0) batch_reads = env_u64("FULGOR_STREAM_BATCH", format == FGPU_FMT_COMPRESSED ? 1u << 18 : 1u << 15);
        if (workers == 0) workers = (unsigned)env_u64("FULGOR_STREAM_WORKERS", 5);
        workers = std::min(workers, 16u);
        const uint32_t k = ix->host.dict.k;
        const uint32_t max_kmers = std::min<uint32_t>(max_read_bases >= k ? max_read_bases - k + 1 : 1, SEG_KMERS);
        const uint64_t reads = batch_reads + batch_reads / 16;
        fgpu_stream_cache* cache = cache_of(ix);
        std::vector<StreamWorkerState> fresh;
        {
            std::lock_guard<std::mutex> g(cache->mu);
            while (fresh.size() + cache->idle.size() < workers) fresh.emplace_back();
        }
        try {
            for (StreamWorkerState& w : fresh) {
                if (fgpu_result_create(ix, &w.res)) throw std::runtime_error(fgpu_last_error());
                w.res->reserve_reads = reads;
                reserve_result(ix, w.res, reads, max_kmers, format, out_bytes_per_read ? reads * out_bytes_per_read * 5 / 4 + 4096 : 0);
                w.d_bases.ensure(reads * (uint64_t)std::max<uint32_t>(max_read_bases, 1) + 1024);
                w.d_offs.ensure((reads + 1) * 8);
            }
        } catch (...) {
            for (StreamWorkerState& w : fresh) { fgpu_result_free(w.res); w.d_bases.release(); w.d_offs.release(); }
            throw;
        }
        std::lock_guard<std::mutex> g(cache->mu);
        for (StreamWorkerState& w : fresh) cache->idle.push_back(w);
    });
}

int fgpu_last_stream_report(char** out) {
    if (!out) return fail(-EINVAL, "null argument");
    std::lock_guard<std::mutex> g(g_report_mu);
    *out = (char*)malloc(g_stream_report.size() + 1);
    if (!*out) return fail(-ENOMEM, "out of host memory");
    memcpy(*out, g_stream_report.c_str(), g_stream_report.size() + 1);
    return 0;
}

}  // extern "C"
