// HIP kernels of the pseudoalignment hot path for gfx950 (CDNA4, wave64). Integer/bit work only.
//
//   k1_lookup      (k1_lookup.hip.h) read -> minimizer runs -> one bucket fetch per run -> sorted distinct
//                  colour-set ids (+ multiplicities). Replaces index::fetch_color_set_ids and the k-mer streaming half
//                  of pseudoalign_threshold_union (ps_full_intersection.cpp:334-374,
//                  ps_threshold_union.cpp:327-387) including u2c (index.hpp:37).
//   k_merge_segments  reads longer than 512 k-mers: id lists of their segments -> one list per read
//   k2a_intersect  hybrid `intersect` (ps_full_intersection.cpp:32-127) -> result bitmap + size
//   k3a_union      hybrid `merge`     (ps_threshold_union.cpp:16-40)   -> result bitmap + size
//   k_generic      meta / differential / meta-differential: intersect and merge (ps_full_intersection.cpp:129-332,
//                  ps_threshold_union.cpp:42-318) over the device form of those codecs (host/codecs_build.hpp)
//   scan_*         sizes -> CSR offsets
//   k2b_expand     bitmap -> sorted u32 colour list (the vector<uint32_t> the reference returns)
//   k_hits*        per-colour hit counts over a batch
//   k_fmt_* / k_cfmt_*  the three output formats of psa_formatter (ascii, binary, compressed) on the device
//   k_account      algorithmic bytes of a pass, outside the timed region
//
// One wavefront owns one read. Cross-lane steps use ballot / mbcnt / shuffles; per-wave scratch lives
// in LDS; there is no inter-workgroup communication inside a launch.
#pragma once
#include <hip/hip_runtime.h>
#include "../common/kmer_common.h"

namespace fg {

struct DevColors {
    const uint32_t* bmp_words;  // the bitmap lists as rows of w32 words, 16-byte aligned, zero behind colour n - 1 (the
                                // stream packs them at arbitrary bit offsets; a row moves as 128-bit groups without shifts)
    const uint64_t* offsets;
    const struct ListDesc* set_desc;  // one resolved descriptor per colour set (built at upload)
    const uint32_t* blk_words;        // packed blocks of the gap-coded lists, headers in front of the data (host/hybrid_codec.hpp)
    uint32_t n, sparse_thr, dense_thr;
    uint32_t w32;  // 32-bit words per result bitmap, rounded up to a multiple of 4
};

constexpr uint32_t NEG = 0xFFFFFFFFu;
constexpr uint32_t SMALL_RESULT = 16;  // results of at most this many colours may travel as colours, not as a bitmap row (k2a -> k2b)
enum { D_ENC_NONE = -1, D_ENC_DELTA_GAPS = 0, D_ENC_BITMAP = 1, D_ENC_COMPLEMENT = 2 };

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// First thing in a kernel that reads buffers which a copy engine driven through the HSA runtime may have filled (copy_engines.hip.h):
// the HIP runtime orders and fences the copies IT issues — it makes the first dispatch behind one of its copies invalidate the caches
// — and knows nothing of these. Without this the lookup kernel read offsets of the batch before (lines left in an XCD's L2) and
// faulted, once in a few hundred batches. System-scope acquire: the vector caches and the non-local lines of the wave's L2
// (buffer_inv sc0 sc1), and the scalar cache.
__device__ __forceinline__ void foreign_writes_acquire() {
#ifndef FG_NO_FOREIGN_ACQUIRE  // (variant build: what the fence costs the lookup kernel)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __builtin_amdgcn_s_dcache_inv();
#endif
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, uint32_t src) {  // src wave-uniform
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
}

// order LDS traffic of one wave: LDS ops of a wave complete in issue order; the wait + compiler
// barrier makes earlier writes visible to later reads by other lanes of the same wave
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));  // 16-byte global access at any 4-byte boundary

// LDS accesses by byte address. Plain pointers into a kernel's dynamic LDS block cost an addition of the block's (link-time)
// address per access, and `base[v >> 5]` becomes shift, mask, add; from an explicit byte address it is shift, shift-add.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ lds_u16* lds16(uint32_t a) { return (lds_u16*)(uintptr_t)a; }
__device__ __forceinline__ lds_u32* lds32(uint32_t a) { return (lds_u32*)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void lds_add(uint32_t a, uint32_t v) { __hip_atomic_fetch_add(lds32(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_or(uint32_t a, uint32_t v) { __hip_atomic_fetch_or(lds32(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_xor(uint32_t a, uint32_t v) { __hip_atomic_fetch_xor(lds32(a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// byte address of the word that holds bit v of a bit plane at byte address `plane` (wave-uniform)
__device__ __forceinline__ uint32_t lds_bit_word(uint32_t plane, uint32_t v) {
    uint32_t a;
    asm("v_lshrrev_b32 %0, 5, %1\n\tv_lshl_add_u32 %0, %0, 2, %2" : "=&v"(a) : "v"(v), "s"(plane));
    return a;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
// inclusive prefix sum across the 64 lanes with DPP adds (pure VALU, no LDS crossbar round trips):
// row_shr 1,2,3 of the input, then row_shr 4 / 8 of the partial sums inside each 16-lane row, then the last
// lane of row 0/2 broadcast into row 1/3 (row_bcast:15) and lane 31 into rows 2,3 (row_bcast:31).
// Lanes masked off by row/bank masks and lanes without a source receive 0.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    uint32_t r = v;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xF, 0xF, true);  // row_shr:3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xF, 0xE, true);  // row_shr:4, banks 1-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xF, 0xC, true);  // row_shr:8, banks 2-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1,3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2,3
    return r;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}
// number of set bits of a wave-uniform 64-bit mask below this lane
__device__ __forceinline__ uint32_t mask_rank(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Dynamic work distribution for the persistent one-wave-per-read kernels. The batch is cut into 8
// static partitions (label = blockIdx & 7, which the dispatcher happens to place on one XCD each; only
// speed depends on that) and every wave pulls `batch` reads at a time from its partition's
// counter; a wave whose partition is used up goes on with the partitions behind it (the partitions of a pass
// taken in locality order differ a lot in cost: all unmapped reads sit in the last one). No tail from uneven
// reads or from a grid larger than what is actually resident.
constexpr uint32_t TICKET_STRIDE = 64;  // counters live 256 bytes apart (separate L2 channels)
struct WorkQueue {
    unsigned int* counters;  // 8 * TICKET_STRIDE words, zeroed before every launch
    uint64_t n;
    uint32_t batch;          // reads per pull
    uint32_t max_parts = 8;  // (k2b_expand: FG_K2B_PARTS, its counters have room for K2B_MAX_PARTS)
    __device__ __forceinline__ bool pull(uint64_t& first, uint32_t& count) const {
        const uint32_t parts = min(max_parts, gridDim.x);
        const uint64_t per = (n + parts - 1) / parts;
        uint32_t part = blockIdx.x % parts;
        for (uint32_t tries = 0; tries < parts; ++tries) {
            const uint64_t lo = min(n, part * per), hi = min(n, lo + per);
            unsigned int t = 0;
            if (lane_id() == 0) t = atomicAdd(&counters[part * TICKET_STRIDE], batch);
            t = __builtin_amdgcn_readfirstlane(t);
            first = lo + t;
            if (first < hi) {
                count = (uint32_t)min((uint64_t)batch, hi - first);
                return true;
            }
            part = part + 1 == parts ? 0u : part + 1;
        }
        return false;
    }
};

}  // namespace fg
#include "k1_lookup.hip.h"
namespace fg {

// ---------------------------------------------------------------------------------------------
// Long reads: the lookup kernel ran on segments of at most 512 k-mers; this kernel merges the sorted id
// lists of the segments of every read into what one pass over the whole read would have produced: sorted
// distinct ids, summed multiplicities, summed positive counts (fetch_color_set_ids sorts and deduplicates
// over the whole read: ps_full_intersection.cpp:361-373). One wave per read. Up to 64 segments (32768
// k-mers): lane = one segment with a cursor into its sorted list, every step emits the smallest id under
// the cursors. Up to 512 segments: the same with the cursors in LDS, several segments per lane. More segments:
// repeated minimum extraction over all the lists (quadratic, reads beyond 262144 k-mers only).
// The merged list of read r is written to out_ids/out_cnt at the slab offset of its first segment.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t MERGE_MAX_SEGMENTS = 512;  // 262144 k-mers; longer reads: repeated minimum extraction
__global__ __launch_bounds__(256) void k_merge_segments(const uint32_t* __restrict__ seg_nids, const uint32_t* __restrict__ seg_npos,
                                                        const uint32_t* __restrict__ seg_ids, const uint32_t* __restrict__ seg_cnt,
                                                        uint32_t stride, const uint64_t* __restrict__ seg_first, uint64_t first,
                                                        uint64_t n_reads, uint32_t* __restrict__ out_nids,
                                                        uint32_t* __restrict__ out_npos, uint64_t* __restrict__ out_idoff,
                                                        uint32_t* __restrict__ out_ids, uint32_t* __restrict__ out_cnt) {
    __shared__ uint32_t s_cur[4][MERGE_MAX_SEGMENTS];
    __shared__ uint32_t s_len[4][MERGE_MAX_SEGMENTS];
    const int lane = lane_id();
    uint32_t* cur = s_cur[threadIdx.x >> 6];
    uint32_t* len = s_len[threadIdx.x >> 6];
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t u_first = seg_first[first];
    for (uint64_t r = wave; r < n_reads; r += n_waves) {
        const uint64_t u0 = seg_first[first + r] - u_first, u1 = seg_first[first + r + 1] - u_first;
        const uint64_t base = u0 * stride;
        uint32_t pos = 0;
        for (uint64_t u = u0 + lane; u < u1; u += 64) pos += seg_npos[u];
        pos = wave_sum_u32(pos);
        uint32_t out = 0;
        if (u1 - u0 <= 64) {
            const bool mine = u0 + lane < u1;
            const uint32_t my_n = mine ? seg_nids[u0 + lane] : 0u;
            const uint64_t my_base = (u0 + (mine ? lane : 0)) * stride;
            uint32_t cur = 0;
            for (;;) {
                const uint32_t v = cur < my_n ? seg_ids[my_base + cur] : NEG;
                const uint32_t wm = wave_min_u32(v);
                if (wm == NEG) break;
                const bool hit = v == wm;
                const uint32_t total = wave_sum_u32(hit ? seg_cnt[my_base + cur] : 0u);
                if (hit) ++cur;  // ids are distinct within a segment
                if (lane == 0) {
                    out_ids[base + out] = wm;
                    out_cnt[base + out] = total;
                }
                ++out;
            }
        } else if (u1 - u0 <= MERGE_MAX_SEGMENTS) {
            // the same merge with the cursors in LDS: lane l owns segments l, l + 64, ...
            const uint32_t nseg = (uint32_t)(u1 - u0);
            for (uint32_t sg = lane; sg < nseg; sg += 64) {
                cur[sg] = 0;
                len[sg] = seg_nids[u0 + sg];
            }
            for (;;) {
                uint32_t lm = NEG;
                for (uint32_t sg = lane; sg < nseg; sg += 64) {
                    const uint32_t c = cur[sg];
                    if (c < len[sg]) lm = min(lm, seg_ids[(u0 + sg) * stride + c]);
                }
                const uint32_t wm = wave_min_u32(lm);
                if (wm == NEG) break;
                uint32_t c_sum = 0;
                for (uint32_t sg = lane; sg < nseg; sg += 64) {
                    const uint32_t c = cur[sg];
                    if (c < len[sg] && seg_ids[(u0 + sg) * stride + c] == wm) {
                        c_sum += seg_cnt[(u0 + sg) * stride + c];
                        cur[sg] = c + 1;
                    }
                }
                c_sum = wave_sum_u32(c_sum);
                if (lane == 0) {
                    out_ids[base + out] = wm;
                    out_cnt[base + out] = c_sum;
                }
                ++out;
            }
        } else {
            uint32_t last = 0;
            bool have_last = false;
            for (;;) {
                uint32_t lm = NEG;
                for (uint64_t u = u0; u < u1; ++u) {
                    const uint32_t nu = seg_nids[u];
                    for (uint32_t j = lane; j < nu; j += 64) {
                        const uint32_t v = seg_ids[u * stride + j];
                        if (!have_last || v > last) lm = min(lm, v);
                    }
                }
                const uint32_t wm = wave_min_u32(lm);
                if (wm == NEG) break;
                uint32_t c = 0;
                for (uint64_t u = u0; u < u1; ++u) {
                    const uint32_t nu = seg_nids[u];
                    for (uint32_t j = lane; j < nu; j += 64) c += seg_ids[u * stride + j] == wm ? seg_cnt[u * stride + j] : 0u;
                }
                c = wave_sum_u32(c);
                if (lane == 0) {
                    out_ids[base + out] = wm;
                    out_cnt[base + out] = c;
                }
                last = wm;
                have_last = true;
                ++out;
            }
        }
        if (lane == 0) {
            out_nids[r] = out;
            out_npos[r] = pos;
            out_idoff[r] = base;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-list descriptors of the colour kernels
// ---------------------------------------------------------------------------------------------
struct ListHeader {
    uint64_t begin, body, soff;  // bitmap list: bit offsets of the list / of its bitmap. Gap-coded list: begin = first
                                 // data word in blk_words, soff = first block header
    uint32_t ncodes, size;
    int type;
};

// One resolved colour list (32 bytes). DevColors::set_desc holds one per colour set (score = 0, id = its
// index), so that a kernel reaches everything it needs about a list with ONE gather. k_desc, a flat kernel with
// one thread per (read, list) pair, writes (id, score) descriptors in per-read order for the generic codecs;
// the hybrid kernels gather set_desc themselves (the full intersection one read ahead).
struct __attribute__((aligned(16))) ListDesc {
    uint64_t begin;   // bitmap list: first word of its row in bmp_words; gap-coded list: first data word in blk_words
    uint64_t soff;    // gap-coded list: index of its first block header — or, for a single-block list, the header itself
    uint32_t ncodes;  // blocks of a gap-coded list (0 for bitmap lists)
    uint32_t meta;    // encoding
    int32_t score;    // positive k-mers that produced this id (threshold-union)
    uint32_t id;      // colour-set id
};
__device__ __forceinline__ int desc_type(const ListDesc& d) { return (int)(d.meta & 0xFFu); }
__device__ __forceinline__ uint64_t desc_body(const ListDesc& d) { return d.begin + (d.meta >> 8); }

// per-read id lists out of the lookup kernel's fixed-stride slabs into one dense CSR (what fetch_color_set_ids hands to the
// host: only the ids that exist cross PCIe, not the slab pool). 16 threads per read.
__global__ __launch_bounds__(256) void k_gather_ids(const uint32_t* __restrict__ nids, const uint64_t* __restrict__ src_off,
                                                    const uint32_t* __restrict__ ids_src, const uint64_t* __restrict__ dst_off,
                                                    uint64_t n_reads, uint32_t* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = t >> 4;
    if (r >= n_reads) return;
    const uint32_t cnt = nids[r];
    const uint64_t so = src_off[r], dso = dst_off[r];
    for (uint32_t j = (uint32_t)t & 15u; j < cnt; j += 16) out[dso + j] = ids_src[so + j];
}

// ---------------------------------------------------------------------------------------------
// Locality order of a pass. Reads drawn from the same locus share their colour sets, and a pass of millions of
// reads fetches every list dozens of times — in file order those fetches are spread over the whole launch and
// miss the L2. The colour kernels therefore take the reads of a pass sorted by the RAREST colour set among a
// read's ids (set_rank: position of a colour set when the sets are sorted by the number of k-mers that carry
// them; reads without ids come last): reads that share a rare set overlap on the same few unitigs. Counting
// sort, three light kernels: keys + histogram, scan (scan_*), scatter. The histogram is left all zero by the
// scatter, ready for the next pass. The order inside a bucket is whatever the atomics give: results go to the
// read's own row, so nothing observable depends on it.
// ---------------------------------------------------------------------------------------------
// The most common colour sets (the ORDER_HOT highest ranks) and the bucket of the reads without ids receive thousands of
// reads each — atomics on one address serialise — so every block counts those keys in LDS and touches their global
// counters once. Persistent blocks, each over a contiguous slice of the reads.
constexpr uint32_t ORDER_HOT = 1023;  // + the bucket of reads without ids = 1024 LDS bins
__global__ __launch_bounds__(256) void k_order_keys(const uint32_t* __restrict__ nids, const uint64_t* __restrict__ idoff,
                                                    const uint32_t* __restrict__ ids_pool, const uint32_t* __restrict__ set_rank,
                                                    uint32_t num_sets, uint64_t n_reads, uint32_t* __restrict__ keys,
                                                    uint32_t* __restrict__ hist) {
    __shared__ uint32_t hot[ORDER_HOT + 1];
    for (uint32_t i = threadIdx.x; i <= ORDER_HOT; i += blockDim.x) hot[i] = 0;
    __syncthreads();
    const uint32_t hot_lo = num_sets > ORDER_HOT ? num_sets - ORDER_HOT : 0u;  // keys hot_lo .. num_sets
    const uint64_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint64_t r0 = min(n_reads, (uint64_t)blockIdx.x * per), r1 = min(n_reads, r0 + per);
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const uint32_t cnt = nids[r];
        const uint64_t off = idoff[r];
        uint32_t key = num_sets;
        for (uint32_t i = 0; i < cnt; ++i) key = min(key, set_rank[ids_pool[off + i]]);
        keys[r] = key;
        if (key >= hot_lo) atomicAdd(&hot[key - hot_lo], 1u);
        else atomicAdd(&hist[key], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= ORDER_HOT; i += blockDim.x)
        if (hot[i] && hot_lo + i <= num_sets) atomicAdd(&hist[hot_lo + i], hot[i]);
}
// same grid as k_order_keys (the slices must be the same)
__global__ __launch_bounds__(256) void k_order_scatter(const uint32_t* __restrict__ keys, uint32_t num_sets, uint64_t n_reads,
                                                       const uint64_t* __restrict__ bucket_off, uint32_t* __restrict__ hist,
                                                       uint32_t* __restrict__ order) {
    __shared__ uint32_t hot_cnt[ORDER_HOT + 1], hot_base[ORDER_HOT + 1];
    for (uint32_t i = threadIdx.x; i <= ORDER_HOT; i += blockDim.x) hot_cnt[i] = 0;
    __syncthreads();
    const uint32_t hot_lo = num_sets > ORDER_HOT ? num_sets - ORDER_HOT : 0u;
    const uint64_t per = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint64_t r0 = min(n_reads, (uint64_t)blockIdx.x * per), r1 = min(n_reads, r0 + per);
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const uint32_t key = keys[r];
        if (key >= hot_lo) atomicAdd(&hot_cnt[key - hot_lo], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= ORDER_HOT; i += blockDim.x) {  // this block's places in every hot bucket
        const uint32_t c = hot_cnt[i];
        hot_base[i] = c ? atomicSub(&hist[hot_lo + i], c) - c : 0u;
        hot_cnt[i] = 0;
    }
    __syncthreads();
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
        const uint32_t key = keys[r];
        uint32_t place;
        if (key >= hot_lo) place = hot_base[key - hot_lo] + atomicAdd(&hot_cnt[key - hot_lo], 1u);
        else place = atomicSub(&hist[key], 1u) - 1u;  // counts down to zero
        order[bucket_off[key] + place] = (uint32_t)r;
    }
}

__global__ __launch_bounds__(256) void k_desc(const uint32_t* __restrict__ nids,
                                              const uint64_t* __restrict__ src_off, const uint32_t* __restrict__ ids_src,
                                              const uint32_t* __restrict__ cnt_src, const uint64_t* __restrict__ dst_off,
                                              uint64_t n_reads, ListDesc* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t r = t >> 4;
    if (r >= n_reads) return;
    const uint32_t cnt = nids[r];
    const uint64_t so = src_off[r], dso = dst_off[r];
    for (uint32_t j = (uint32_t)t & 15u; j < cnt; j += 16) {
        const uint32_t id = ids_src[so + j];
        ListDesc d{0, 0, 0, (uint32_t)D_ENC_NONE & 0xFFu, 0, id};
        d.score = cnt_src ? (int32_t)cnt_src[so + j] : 0;
        out[dso + j] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// Packed blocks: wave-parallel decode
// ---------------------------------------------------------------------------------------------
// The gap-coded lists of one read are flattened into one sequence of 64-value blocks. Lane s of the wave
// resolves block s (its list by a search over the inclusive block counts `pref`, then its header), and the
// wave then walks the blocks one per step: the step's parameters come out of lane q with v_readlane (so
// they are scalars), lane i extracts value i with one funnel shift, and the data words of the next two
// blocks are already in flight while a block is consumed.
struct BlockLane {      // per lane: the flattened block this lane resolved
    uint32_t word;        // its first data word, as an index into the block words (checked at upload: fewer than 2^32 words)
    uint32_t start;       // value of field 0 (plus a caller-defined bias, e.g. an LDS plane offset in bits)
    uint32_t meta;        // width | (count-1) << 5 | caller flags << 11
    uint32_t extra;       // caller-defined (the list's score for the threshold union)
};

// first i in [0,64) with pref[i] > t (pref non-decreasing, pref[63] > t)
__device__ __forceinline__ uint32_t upper_slot(const uint32_t* pref, uint32_t t) {
    uint32_t lo = 0, hi = 63;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (pref[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// list owning flattened block s, given the inclusive block counts of the (at most 64) lists in the lanes:
// the last list whose exclusive count is <= s. Lists are few (5 on average), so a scalar walk beats a search.
// (`excl` must have been computed by all lanes: v_readlane reads lanes that are masked off here.)
__device__ __forceinline__ uint32_t owner_list(uint32_t excl, uint32_t nlists, uint32_t s) {
    uint32_t owner = 0;
    for (uint32_t i = 1; i < nlists; ++i) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)excl, i);
        owner = s >= e ? i : owner;
    }
    return owner;
}

// Every lane requests the two words that hold its field of block q (word `lane` of a bitmap chunk). The
// request is unconditional (lanes past the block's count read inside the 64 padding words of blk_words) so
// that the number of loads in flight is known at compile time and the wait before a block is consumed leaves
// the two younger requests outstanding. Returns the block's meta word (a scalar) for the step that consumes it.
__device__ __forceinline__ uint32_t block_fetch(const uint32_t* words, const BlockLane& b, uint32_t q, int lane, uint2& w) {
    const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)b.word, q);
    const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)b.meta, q);
    uint32_t width = mt & 31u;
    width += width == BLK_CHUNK_WIDTH;  // a chunk is read as 32-bit fields
    typedef const uint32_t __attribute__((address_space(1))) * global_words;  // keeps the request a global_load (vmcnt only)
    // scalar base (words + first) plus a 32-bit lane offset: the address needs no vector arithmetic beyond the offset
    const global_words p = (global_words)(words + first) + (__umul24((uint32_t)lane, width) >> 5);
    w = make_uint2(p[0], p[1]);
    return mt;
}

template <typename F, typename H, typename G>
__device__ __forceinline__ void block_consume(const BlockLane& b, uint32_t q, uint32_t mt, int lane, uint2 w, F& per_value, H& per_word,
                                              G& after_block) {
    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)b.start, q);
    const uint32_t ex = (uint32_t)__builtin_amdgcn_readlane((int)b.extra, q);
    const uint32_t width = mt & 31u;
    // the field is cut out by all lanes (not only count of them) so that the wait for `w` is unconditional
    uint32_t f = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w.y, w.x, __umul24((uint32_t)lane, width)), 0, width);
    uint32_t x = w.x;
    asm volatile("" : "+v"(f), "+v"(x));  // (keeps the compiler from sinking the extraction into the branches below)
    if ((uint32_t)lane <= ((mt >> 5) & 63u)) {
        if (width == BLK_CHUNK_WIDTH) per_word((st >> 5) + (uint32_t)lane, x, ex);
        else per_value(st + f, ex);
    }
    after_block(mt >> 11);
}

// per_value(v, extra of the block) for every value of the offset blocks, per_word(word index, bits, extra) for
// every word of the bitmap chunks among blocks [0, steps) held by the lanes of `b` (steps >= 1);
// after_block(flags) once per block (wave-uniform). Three register pairs (and the blocks' meta words, scalars) rotate by
// unrolling, not by moves, so a block is consumed while the requests of the next two are in flight.
template <typename F, typename H, typename G>
__device__ __forceinline__ void run_blocks(const uint32_t* words, const BlockLane& b, uint32_t steps, int lane, F per_value, H per_word,
                                           G after_block) {
    const uint32_t last = steps - 1;
    uint2 c0, c1, c2;
    uint32_t m0 = block_fetch(words, b, 0, lane, c0), m1 = block_fetch(words, b, min(1u, last), lane, c1), m2;
    uint32_t q = 0;
    while (true) {
        m2 = block_fetch(words, b, min(q + 2, last), lane, c2);
        block_consume(b, q, m0, lane, c0, per_value, per_word, after_block);
        if (++q > last) break;
        m0 = block_fetch(words, b, min(q + 2, last), lane, c0);
        block_consume(b, q, m1, lane, c1, per_value, per_word, after_block);
        if (++q > last) break;
        m1 = block_fetch(words, b, min(q + 2, last), lane, c1);
        block_consume(b, q, m2, lane, c2, per_value, per_word, after_block);
        if (++q > last) break;
    }
}

// per-wave LDS carve shared by k2a / k3a
struct WaveScratch {
    uint64_t* h_begin;
    uint64_t* h_body;
    uint64_t* h_soff;
    uint32_t* h_ncodes;
    uint32_t* pref;
    int32_t* h_score;
};
__host__ __device__ inline uint32_t wave_scratch_bytes() { return 64 * (8 * 3 + 4 * 3); }
// the hybrid kernels do without h_body (a bitmap list keeps its body offset in h_soff, which only gap-coded lists
// use otherwise): 512 bytes less per wave, one more wave per SIMD for the threshold union's counter planes
__host__ __device__ inline uint32_t wave_scratch_bytes_compact() { return wave_scratch_bytes() - 64 * 8; }
__device__ __forceinline__ WaveScratch carve_scratch(unsigned char* p) {
    WaveScratch s;
    s.h_begin = (uint64_t*)p;
    s.h_soff = s.h_begin + 64;
    s.h_ncodes = (uint32_t*)(s.h_soff + 64);
    s.pref = s.h_ncodes + 64;
    s.h_score = (int32_t*)(s.pref + 64);
    s.h_body = (uint64_t*)(s.h_score + 64);  // last: absent from the compact layout
    return s;
}

// ---------------------------------------------------------------------------------------------
// K2a: full intersection of hybrid colour sets -> bitmap + cardinality
// ---------------------------------------------------------------------------------------------
// Semantics of `intersect` (ps_full_intersection.cpp:32-127): the result is the set intersection of the
// given lists. Per read (one wave) the kernel builds an EXCLUSION bitmap in LDS and returns its complement:
//   bitmap lists      : EXCL |= ~list, word-wise
//   complemented lists: every missing colour sets its bit in EXCL
//   sparse lists      : the list sets its bits in the plane T; after its last block EXCL |= ~T and T = 0
// so every value of every gap-coded list ORs one bit into LDS, and all blocks of all lists of the read run
// through one loop (run_blocks), one block of up to 64 values per step.
__device__ __forceinline__ uint4 or_not(uint4 e, uint4 x) { return make_uint4(e.x | ~x.x, e.y | ~x.y, e.z | ~x.z, e.w | ~x.w); }

// PAIR: two consecutive reads of a ticket with at most 64 lists between them go through the kernel as ONE group — their
// descriptors arrive with one gather, their blocks run through one block loop into two EXCL planes — so that the fetch
// chain ids -> descriptors -> block headers -> blocks, which bounds this kernel, is paid once for both.
// (Without PAIR there is one EXCL plane: collections whose planes are too large for a fourth one.)
__host__ __device__ inline uint32_t k2a_wave_bytes(uint32_t W, bool pair) { return ((pair ? 2u : 1u) + 2u) * W * 4u + wave_scratch_bytes_compact(); }

template <bool PAIR>
__global__ __launch_bounds__(256, 8) void k2a_intersect(DevColors c, const uint32_t* __restrict__ nids,
                                                     const uint64_t* __restrict__ idoff, const uint32_t* __restrict__ ids_pool,
                                                     uint64_t n_reads, uint32_t* __restrict__ out_bitmap,
                                                     uint32_t* __restrict__ out_count, unsigned int* tickets,
                                                     const uint32_t* __restrict__ order, uint32_t* __restrict__ small_out) {
    // order != nullptr: the reads are taken in that order (k_order_*: reads that share their rarest colour set are
    // neighbours, so the lists of a ticket are fetched once and found in the L2 by the reads behind); results land at
    // the read's own row whatever the order.
    // small_out != nullptr: a result of at most SMALL_RESULT colours is written as colours into the read's slot of
    // small_out (SMALL_RESULT u32 per read) INSTEAD of its bitmap row, which is then left untouched.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t W = c.w32, W4 = W >> 2;  // W is a multiple of 4: the bitmaps move as 128-bit groups
    constexpr uint32_t NEX = PAIR ? 2u : 1u;  // EXCL planes (reads of a group)
    unsigned char* mine = smem + (size_t)wv * k2a_wave_bytes(W, PAIR);
    WaveScratch sc = carve_scratch(mine);
    uint32_t* EXCL = (uint32_t*)(mine + wave_scratch_bytes_compact());
    uint4* EX4 = (uint4*)EXCL;     // plane p of the group at EX4 + p * W4
    uint4* T4 = EX4 + NEX * W4;    // the plane T, all zero between sparse lists
    uint4* INIT4 = T4 + W4;        // what EXCL starts from: colours >= n excluded
    const uint32_t excl_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(EXCL));  // LDS byte address of EXCL
    constexpr uint32_t BATCH = 8;
    constexpr uint32_t SPARSE_FLAG = 0x80000000u, SECOND_FLAG = 0x40000000u, NBLK_MASK = 0x3FFFFFFFu;
    const WorkQueue wq{tickets, n_reads, BATCH};
    uint64_t t_first;
    uint32_t t_count;
    const uint32_t n = c.n, tail_word = n >> 5, tail_mask = ~((1u << (n & 31u)) - 1u);
    const ListDesc none{0, 0, 0, (uint32_t)D_ENC_NONE & 0xFFu, 0, 0};

    for (uint32_t g4 = lane; g4 < W4; g4 += 64) {
        T4[g4] = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t w = 4 * g4;
        INIT4[g4] = make_uint4(w < tail_word ? 0u : (w == tail_word ? tail_mask : 0xFFFFFFFFu),
                               w + 1 < tail_word ? 0u : (w + 1 == tail_word ? tail_mask : 0xFFFFFFFFu),
                               w + 2 < tail_word ? 0u : (w + 2 == tail_word ? tail_mask : 0xFFFFFFFFu),
                               w + 3 < tail_word ? 0u : (w + 3 == tail_word ? tail_mask : 0xFFFFFFFFu));
    }
    wave_lds_sync();

    while (wq.pull(t_first, t_count)) {
        // The colour-set ids of the next group are requested while a group is processed; its descriptors (one 32-byte gather
        // per list) are fetched at the top of the group. (Requesting the descriptors a group ahead as well was slower, in
        // registers and through an LDS stage: DESIGN.md §8.)
        const uint64_t tl = min(t_first + (uint64_t)lane, n_reads - 1);
        // the read behind place `lane` of the ticket (a pass has fewer than 2^32 reads); kept in LDS, not in a register:
        // it is needed once per read, for the stores
        uint32_t* const t_read = (uint32_t*)sc.h_score;  // (the score column of the scratch is the union kernel's)
        uint32_t cnt_l;
        uint64_t off_l;
        {
            const uint32_t rd_l = order ? order[tl] : (uint32_t)tl;
            cnt_l = (uint32_t)lane < t_count ? nids[rd_l] : 0u;  // lanes past the ticket: empty reads
            off_l = idoff[rd_l];
            t_read[lane] = rd_l;
        }
        // the group that starts with the ticket's read i: its list counts, and whether read i + 1 belongs to it
        // (t_count <= BATCH < 62: the lanes exist and read as empty past the ticket)
        auto group_at = [&](uint32_t i, uint32_t& a0, uint32_t& a1) -> bool {
            a0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, i);
            a1 = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, i + 1);
            return PAIR && i + 1 < t_count && a0 != 0 && a1 != 0 && a0 + a1 <= 64u;
        };
        auto fetch_ids = [&](uint32_t i, uint32_t a0, uint32_t a1, bool two) -> uint32_t {  // lane = list of the group (first 64)
            uint32_t id = 0;
            if ((uint32_t)lane < a0) id = ids_pool[readlane_u64(off_l, i) + lane];
            else if (two && (uint32_t)lane < a0 + a1) id = ids_pool[readlane_u64(off_l, i + 1) + lane - a0];
            return id;
        };
        uint32_t c0, c1;
        bool pair = group_at(0, c0, c1);
        uint32_t id1 = fetch_ids(0, c0, c1, pair);
        for (uint32_t ri = 0; ri < t_count;) {
            const uint32_t nr = pair ? 2u : 1u;
            const uint32_t nlist = pair ? c0 + c1 : c0;  // lists of the group (a single read may have more than 64)
            ListDesc dcur = none;
            if ((uint32_t)lane < nlist) dcur = c.set_desc[id1];
            uint32_t n0, n1;
            const bool npair = group_at(ri + nr, n0, n1);
            const uint32_t id2 = fetch_ids(ri + nr, n0, n1, npair);
            const uint64_t off = readlane_u64(off_l, ri);
            const uint32_t second = (pair && (uint32_t)lane >= c0) ? SECOND_FLAG : 0u;  // this lane's list belongs to the group's second read
            if (nlist == 0) {
                const uint64_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_read[ri]);
                if (!small_out) {
                    uint4* bm4 = (uint4*)(out_bitmap + r * W);
                    for (uint32_t g4 = lane; g4 < W4; g4 += 64) bm4[g4] = make_uint4(0u, 0u, 0u, 0u);
                }
                if (lane == 0) out_count[r] = 0;
            } else {
                for (uint32_t p = 0; p < nr; ++p)
                    for (uint32_t g4 = lane; g4 < W4; g4 += 64) EX4[p * W4 + g4] = INIT4[g4];
                for (uint32_t g = 0; g < nlist; g += 64) {
                    ListDesc d = dcur;
                    if (g) {  // more than 64 lists (a read on its own): rare, fetched in place
                        d = none;
                        if (g + lane < nlist) d = c.set_desc[ids_pool[off + g + lane]];
                    }
                    const int type = (int)(int8_t)(d.meta & 0xFFu);
                    const uint32_t nblk = d.ncodes;  // 0 for bitmap lists
                    const uint32_t incl = wave_incl_scan_u32(nblk);
                    const uint32_t excl = incl - nblk;
                    const uint32_t total_blk = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    sc.h_begin[lane] = d.begin; sc.h_soff[lane] = type == D_ENC_BITMAP ? desc_body(d) : d.soff;
                    sc.h_ncodes[lane] = nblk | (type == D_ENC_DELTA_GAPS ? SPARSE_FLAG : 0u) | second;
                    sc.pref[lane] = incl;
                    wave_lds_sync();

                    // bitmap lists: exclude what they do not contain (lane = 128 bits of the list)
                    uint64_t mb = __ballot(type == D_ENC_BITMAP);
                    while (mb) {
                        const int src = __builtin_ctzll(mb);
                        mb &= mb - 1;
                        const uint4* row = (const uint4*)(c.bmp_words + sc.h_soff[src]);
                        uint4* ex = EX4 + ((PAIR && (sc.h_ncodes[src] & SECOND_FLAG)) ? W4 : 0u);
                        for (uint32_t g4 = lane; g4 < W4; g4 += 64) ex[g4] = or_not(ex[g4], row[g4]);
                    }
                    wave_lds_sync();

                    for (uint32_t s0 = 0; s0 < total_blk; s0 += 64) {
                        BlockLane bl{0u, 0u, 0u, 0u};
                        const uint32_t s = s0 + lane;
                        if (s < total_blk) {
                            const uint32_t i = owner_list(excl, min(64u, nlist - g), s);
                            const uint32_t nb = sc.h_ncodes[i];
                            const uint32_t j = s - (sc.pref[i] - (nb & NBLK_MASK));
                            // a single-block list carries its block header in the descriptor itself (one fetch less)
                            const uint64_t hd = (nb & NBLK_MASK) == 1 ? sc.h_soff[i] : ((const uint64_t*)(c.blk_words + sc.h_begin[i]))[j];
                            const bool sparse = (nb & SPARSE_FLAG) != 0;
                            const uint32_t plane = (PAIR && (nb & SECOND_FLAG)) ? 1u : 0u;
                            bl.word = (uint32_t)sc.h_begin[i] + blk_rel_word(hd);
                            // bit index relative to EXCL: the list's own plane, or T for a sparse list
                            bl.start = blk_start(hd) + (sparse ? NEX : plane) * W * 32u;
                            bl.meta = blk_width(hd) | ((blk_count(hd) - 1u) << 5) |
                                      ((sparse && j + 1 == (nb & NBLK_MASK)) ? 1u << 11 : 0u) | (plane << 12);
                        }
                        run_blocks(c.blk_words, bl, min(64u, total_blk - s0), lane,
                                   [&](uint32_t v, uint32_t) { lds_or(lds_bit_word(excl_at, v), 1u << (v & 31)); },
                                   [&](uint32_t wi, uint32_t x, uint32_t) { atomicOr(&EXCL[wi], x); },
                                   [&](uint32_t flags) {  // bit 0: last block of a sparse list, bit 1: the list's plane
                                       if (flags & 1u) {  // a colour absent from this sparse list is excluded; T goes back to zero
                                           uint4* ex = EX4 + ((PAIR && (flags & 2u)) ? W4 : 0u);
                                           wave_lds_sync();
                                           for (uint32_t g4 = lane; g4 < W4; g4 += 64) {
                                               ex[g4] = or_not(ex[g4], T4[g4]);
                                               T4[g4] = make_uint4(0u, 0u, 0u, 0u);
                                           }
                                           wave_lds_sync();
                                       }
                                   });
                    }
                    wave_lds_sync();
                }
                for (uint32_t p = 0; p < nr; ++p) {
                    const uint64_t rp = (uint32_t)__builtin_amdgcn_readfirstlane((int)t_read[ri + p]);
                    uint4* bm4 = (uint4*)(out_bitmap + rp * W);
                    uint32_t pc = 0;
                    if (!small_out) {  // count and store in one pass
                        for (uint32_t g4 = lane; g4 < W4; g4 += 64) {
                            const uint4 e = EX4[p * W4 + g4];
                            const uint4 x = make_uint4(~e.x, ~e.y, ~e.z, ~e.w);
                            // streamed past the L2 (nontemporal): the rows are read back by another kernel, the L2 is for the lists
                            __builtin_nontemporal_store((u32x4){x.x, x.y, x.z, x.w}, (u32x4*)&bm4[g4]);
                            pc += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
                        }
                        pc = wave_sum_u32(pc);
                    } else {
                        for (uint32_t g4 = lane; g4 < W4; g4 += 64) {
                            const uint4 e = EX4[p * W4 + g4];
                            pc += __popc(~e.x) + __popc(~e.y) + __popc(~e.z) + __popc(~e.w);
                        }
                        pc = wave_sum_u32(pc);
                        if (pc > SMALL_RESULT) {
                            for (uint32_t g4 = lane; g4 < W4; g4 += 64) {
                                const uint4 e = EX4[p * W4 + g4];
                                __builtin_nontemporal_store((u32x4){~e.x, ~e.y, ~e.z, ~e.w}, (u32x4*)&bm4[g4]);
                            }
                        } else if (pc) {  // the colours themselves, no row
                            uint32_t* so = small_out + rp * SMALL_RESULT;
                            uint32_t at = 0;  // colours in the groups of earlier rounds
                            for (uint32_t g0 = 0; g0 < W4; g0 += 64) {
                                uint4 x = make_uint4(0u, 0u, 0u, 0u);
                                if (g0 + lane < W4) {
                                    const uint4 e = EX4[p * W4 + g0 + lane];
                                    x = make_uint4(~e.x, ~e.y, ~e.z, ~e.w);
                                }
                                const uint32_t m = __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
                                const uint32_t incl = wave_incl_scan_u32(m);
                                uint32_t pos = at + incl - m;
                                const uint32_t c0w = (g0 + (uint32_t)lane) * 128u;
                                for (uint32_t y = x.x; y; y &= y - 1) so[pos++] = c0w + (uint32_t)__builtin_ctz(y);
                                for (uint32_t y = x.y; y; y &= y - 1) so[pos++] = c0w + 32u + (uint32_t)__builtin_ctz(y);
                                for (uint32_t y = x.z; y; y &= y - 1) so[pos++] = c0w + 64u + (uint32_t)__builtin_ctz(y);
                                for (uint32_t y = x.w; y; y &= y - 1) so[pos++] = c0w + 96u + (uint32_t)__builtin_ctz(y);
                                at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            }
                        }
                    }
                    if (lane == 0) out_count[rp] = pc;
                }
                wave_lds_sync();
            }
            ri += nr;
            id1 = id2;
            c0 = n0;
            c1 = n1;
            pair = npair;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dense rows: every colour set as a plain bitmap row of w32 words (16-byte aligned, zero behind colour n - 1).
// 288 GB of HBM hold the rows of collections far beyond the benchmark's (0.85 M sets x 576 bytes = 0.49 GB), and a
// row needs no descriptor, no block header and no decoder: its address is the colour-set id times the row size, the
// intersection is an AND of registers. The rows are built on the device at upload from the packed blocks / bitmap
// rows of the hybrid device form (k_rows_build), which stays the form of collections whose rows do not fit.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_build(DevColors c, uint64_t num_sets, uint32_t* __restrict__ rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rb[];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t W = c.w32, W4 = W >> 2;
    uint32_t* T = (uint32_t*)smem_rb + (size_t)wv * W;
    uint4* T4 = (uint4*)T;
    const uint32_t t_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(T));
    const uint32_t n = c.n, tail_word = n >> 5, tail_mask = (1u << (n & 31u)) - 1u;  // valid colours of the last word
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t id = wave; id < num_sets; id += nwaves) {
        const ListDesc d = c.set_desc[id];
        const int type = desc_type(d);
        uint4* out4 = (uint4*)(rows + id * W);
        if (type == D_ENC_BITMAP) {
            const uint4* row = (const uint4*)(c.bmp_words + desc_body(d));
            for (uint32_t g4 = lane; g4 < W4; g4 += 64) out4[g4] = row[g4];
            continue;
        }
        for (uint32_t g4 = lane; g4 < W4; g4 += 64) T4[g4] = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
        const uint32_t nblk = d.ncodes;
        for (uint32_t s0 = 0; s0 < nblk; s0 += 64) {
            BlockLane bl{0u, 0u, 0u, 0u};
            const uint32_t sb = s0 + lane;
            if (sb < nblk) {
                const uint64_t hd = nblk == 1 ? d.soff : ((const uint64_t*)(c.blk_words + d.begin))[sb];
                bl.word = (uint32_t)d.begin + blk_rel_word(hd);
                bl.start = blk_start(hd);
                bl.meta = blk_width(hd) | ((blk_count(hd) - 1u) << 5);
            }
            run_blocks(c.blk_words, bl, min(64u, nblk - s0), lane,
                       [&](uint32_t v, uint32_t) { lds_or(lds_bit_word(t_at, v), 1u << (v & 31)); },
                       [&](uint32_t wi, uint32_t x, uint32_t) { atomicOr(&T[wi], x); }, [](uint32_t) {});
        }
        wave_lds_sync();
        for (uint32_t w = lane; w < W; w += 64) {
            uint32_t x = T[w];
            if (type == D_ENC_COMPLEMENT) x = ~x & (w < tail_word ? 0xFFFFFFFFu : (w == tail_word ? tail_mask : 0u));
            rows[id * W + w] = x;
        }
        wave_lds_sync();
    }
}

// K2r: full intersection over dense rows -> bitmap + cardinality (or, for small results, the colours themselves).
// Semantics of `intersect` (ps_full_intersection.cpp:32-127): the set intersection of the given lists. One wave per read,
// lane = 128 bits of the colour space (G groups of them for more than 8192 colours); the rows of up to UNROLL lists are in
// flight at once, the accumulator never leaves the registers, no LDS. Reads come in file order (the locality order of a pass,
// k_order_*, is off by default: it cut the fetched bytes and bought no time).
// AND of the rows of the first K of eight colour sets into acc: K x G unconditional 16-byte loads back to back, then the ANDs —
// no control flow between the requests, so that all of them are in flight together. The ids are scalars: a row's address is a
// scalar base plus the lane's offset.
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef u32x8 u32x8_a4 __attribute__((aligned(4)));
template <int K, int G>
__device__ __forceinline__ void rows_and(const u32x4* __restrict__ rows4, uint32_t W4, const u32x8 id, const uint32_t (&lo)[G], u32x4 (&acc)[G]) {
    u32x4 x[K][G];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const u32x4* row = rows4 + (uint64_t)id[j] * W4;
#pragma unroll
        for (int q = 0; q < G; ++q) x[j][q] = row[lo[q]];
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int q = 0; q < G; ++q) acc[q] &= x[j][q];
}

template <int G>
__global__ __launch_bounds__(256, G == 1 ? 8 : (G == 2 ? 6 : 4)) void k2r_intersect(  // (G groups per lane: 8 waves per SIMD would spill from two groups on)
const u32x4* __restrict__ rows4, uint32_t W, const uint32_t* __restrict__ nids,
                                                     const uint64_t* __restrict__ idoff, const uint32_t* __restrict__ ids_pool,
                                                     uint64_t n_reads, uint32_t* __restrict__ out_bitmap,
                                                     uint32_t* __restrict__ out_count, unsigned int* tickets,
                                                     const uint32_t* __restrict__ order, uint32_t* __restrict__ small_out) {
#ifndef FG_K2R_TICKET  // (variant builds: reads per ticket. 8 / 16 / 32: 3.76 / 3.42 / 3.41 ms; the next read's ids requested a read ahead: 3.43; the next read's first six ROWS requested before the current result is counted and stored, hand-written loads, 7 waves per SIMD: 3.45 — the kernel is bound by the bytes it fetches)
#define FG_K2R_TICKET 16
#endif
    constexpr uint32_t BATCH = FG_K2R_TICKET;
    const int lane = lane_id();
    const uint32_t W4 = W >> 2;
    const WorkQueue wq{tickets, n_reads, BATCH};
    uint64_t t_first;
    uint32_t t_count;
    // the 128-bit group(s) of this lane; lanes past the row load its last group (and are cleared at the end)
    uint32_t lo[G];
#pragma unroll
    for (int q = 0; q < G; ++q) lo[q] = min(q * 64 + (uint32_t)lane, W4 - 1);
    // The ids of a read are fetched with SCALAR loads, eight at a time (wave-uniform address, read-only data): they arrive
    // through the scalar cache and their own counter, so no vector-memory wait — which would also wait for the stores of the
    // read before, one in-order counter — stands between two reads. (The eight words may reach past the read's list, into
    // the slack behind the id pool at worst; only the first nl are used.)
    typedef const __attribute__((address_space(4))) u32x8_a4* ids8_ptr;
    while (wq.pull(t_first, t_count)) {
        const uint64_t tl = min(t_first + (uint64_t)lane, n_reads - 1);
        const uint32_t rd_l = order ? order[tl] : (uint32_t)tl;  // the read behind place `lane` of the ticket
        const uint32_t cnt_l = (uint32_t)lane < t_count ? nids[rd_l] : 0u;
        const uint64_t off_l = idoff[rd_l];
        // (the scalars of read ri + 1 are taken out of the lanes at the end of read ri: the ticket's loads are waited for once,
        // in front of the loop, not at the top of every read — where the wait would also cover the stores of the read before)
        uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, 0);
        uint64_t r = (uint32_t)__builtin_amdgcn_readlane((int)rd_l, 0);
        const uint32_t* ids = ids_pool + readlane_u64(off_l, 0);
        for (uint32_t ri = 0; ri < t_count; ++ri) {
            // A row wider than G groups per lane (more than 32768 colours: G = 4) goes through in TILES of 64 G groups: the ids are
            // fetched again for every tile (scalar loads out of the scalar cache), the cardinality adds up over the tiles. One trip otherwise.
            uint32_t pc = 0;
            uint32_t t0 = 0;
            do {
                if (G == 4) {
#pragma unroll
                    for (int q = 0; q < G; ++q) lo[q] = min(t0 + q * 64 + (uint32_t)lane, W4 - 1);
                }
                u32x4 acc[G];
#pragma unroll
                for (int q = 0; q < G; ++q) acc[q] = nl && t0 + q * 64 + (uint32_t)lane < W4 ? u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu} : u32x4{0u, 0u, 0u, 0u};
                auto round8 = [&](const uint32_t* p, uint32_t left) {  // (left wave-uniform) the rows of up to eight lists in one round
                    const u32x8 id = *(ids8_ptr)p;
                    switch (min(left, 8u)) {
                        case 1: rows_and<1, G>(rows4, W4, id, lo, acc); break;
                        case 2: rows_and<2, G>(rows4, W4, id, lo, acc); break;
                        case 3: rows_and<3, G>(rows4, W4, id, lo, acc); break;
                        case 4: rows_and<4, G>(rows4, W4, id, lo, acc); break;
                        case 5: rows_and<5, G>(rows4, W4, id, lo, acc); break;
                        case 6: rows_and<6, G>(rows4, W4, id, lo, acc); break;
                        case 7: rows_and<7, G>(rows4, W4, id, lo, acc); break;
                        default: rows_and<8, G>(rows4, W4, id, lo, acc); break;
                    }
                };
                if (nl) round8(ids, nl);
                if (nl > 8)  // (a read in twenty)
                    for (uint32_t i = 8; i < nl; i += 8) round8(ids + i, nl - i);
                uint32_t pct = 0;
#pragma unroll
                for (int q = 0; q < G; ++q) pct += __popc(acc[q].x) + __popc(acc[q].y) + __popc(acc[q].z) + __popc(acc[q].w);
                pct = wave_sum_u32(pct);
                pc += pct;
                if (small_out && pct <= SMALL_RESULT) {  // (wave-uniform; small_out only when a row is ONE tile) the colours themselves, no row
                    if (pct) {
                        uint32_t* so = small_out + r * SMALL_RESULT;
                        uint32_t at = 0;
#pragma unroll
                        for (int q = 0; q < G; ++q) {
                            const u32x4 x = acc[q];
                            const uint32_t mq = __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
                            const uint32_t incl = wave_incl_scan_u32(mq);
                            uint32_t pos = at + incl - mq;
                            const uint32_t c0w = (q * 64 + (uint32_t)lane) * 128u;
                            for (uint32_t y = x.x; y; y &= y - 1) so[pos++] = c0w + (uint32_t)__builtin_ctz(y);
                            for (uint32_t y = x.y; y; y &= y - 1) so[pos++] = c0w + 32u + (uint32_t)__builtin_ctz(y);
                            for (uint32_t y = x.z; y; y &= y - 1) so[pos++] = c0w + 64u + (uint32_t)__builtin_ctz(y);
                            for (uint32_t y = x.w; y; y &= y - 1) so[pos++] = c0w + 96u + (uint32_t)__builtin_ctz(y);
                            at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        }
                    }
                } else {
                    // streamed past the L2 (nontemporal): the result rows are read back by another kernel, the L2 is for the lists
                    u32x4* bm4 = (u32x4*)(out_bitmap + r * W);
#pragma unroll
                    for (int q = 0; q < G; ++q)
                        if (t0 + q * 64 + (uint32_t)lane < W4)
                            __builtin_nontemporal_store((u32x4){acc[q].x, acc[q].y, acc[q].z, acc[q].w}, &bm4[t0 + q * 64 + lane]);
                }
            } while (G == 4 && (t0 += 64u * G) < W4);
            if (lane == 0) out_count[r] = pc;
            const uint32_t nx = min(ri + 1, 63u);  // (lanes past the ticket hold empty reads)
            nl = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, nx);
            r = (uint32_t)__builtin_amdgcn_readlane((int)rd_l, nx);
            ids = ids_pool + readlane_u64(off_l, nx);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3a: threshold union of hybrid colour sets -> bitmap + cardinality
// ---------------------------------------------------------------------------------------------
// `merge` (ps_threshold_union.cpp:16-40): scores[c] += score for members of sparse/bitmap lists,
// -= score for the missing colours of complemented lists while min_score is lowered by that score;
// keep c iff scores[c] >= min_score. min_score = uint64(double(#positive k-mers) * tau) (:389).
// Scores live in LDS as small biased counters: BITS = 8 for reads of at most 127 k-mers (4 per word, 8
// planes of W words), 16 up to 32767 k-mers (2 per word, 16 planes), 32 beyond; colour c -> plane c % PLANES, word
// c / 32, field (c % 32) / PLANES: the colours of bitmap word x that live in plane q are (x >> q) & ONES, one
// shift and one mask without any multiply, and the result bits of plane q go back as (sign bits) << q. The counters
// start at HALF - min_score + (total score of complemented lists), so that
//     score[c] >= min_score   <=>   counter[c] >= HALF   <=>   top bit of the field set,
// and they never leave [0, 2*HALF) because 0 <= min_score <= #positive k-mers < HALF. A signed score is
// added as a 32-bit two's complement shifted to the field: fields cannot borrow from each other.
// signed score sv for the colours of bitmap word x that live in counter plane q: flags times the score with one
// 32-bit multiply (full rate on gfx950, profiles/r1/valu_rates_r1q.txt). flag * |sv| < 2^BITS stays inside its field,
// and a negative sv gives the two's complement of the whole word, which is what the counters are added with
template <int BITS>
__device__ __forceinline__ uint32_t counter_spread(uint32_t x, uint32_t q, uint32_t sv) {
    constexpr uint32_t ONES = BITS == 8 ? 0x01010101u : (BITS == 16 ? 0x00010001u : 1u);
    return ((x >> q) & ONES) * sv;
}

constexpr uint32_t K3A_GROUP = 16;
__host__ __device__ inline uint32_t k3a_scratch_bytes() { return K3A_GROUP * (8 + 8 + 4 + 4 + 4); }

// BIASED = false (8-bit only): reads of 128 to 255 k-mers. The counters hold the plain scores (0 <= score <= #positive
// k-mers <= 255 at every moment, because they start at the total score of the complemented lists), and the final
// pass compares every byte with min_score by the carry out of byte + (256 - min_score).
template <int BITS, bool BIASED = true>
__global__ __launch_bounds__(256, BITS == 8 ? 8 : (BITS == 16 ? 4 : 2)) void k3a_union(DevColors c, const uint32_t* __restrict__ npos,
                                                                  const uint32_t* __restrict__ nids, const uint64_t* __restrict__ idoff,
                                                                  const uint32_t* __restrict__ ids_pool,
                                                                  const uint32_t* __restrict__ cnt_pool, double tau, uint64_t n_reads,
                                                                  uint32_t* __restrict__ out_bitmap, uint32_t* __restrict__ out_count,
                                                                  unsigned int* tickets, uint32_t* __restrict__ scores_out) {
    // scores_out != nullptr: also store score[c] (= #positive k-mers of the read whose colour set contains c,
    // the `counts` of index::kmer_matches, src/kmer_matches.cpp:7-30) for every colour: n u32 per read
    constexpr uint32_t PER = 32 / BITS;            // counters per word
    constexpr uint32_t PLANES = 32 / PER;          // planes (one result word = PLANES counter words)
    constexpr uint32_t HALF = 1u << (BITS - 1);
    constexpr uint32_t ONES = BITS == 8 ? 0x01010101u : (BITS == 16 ? 0x00010001u : 1u);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t W = c.w32;
    const uint32_t n = c.n;
    // lists are taken in groups of K3A_GROUP (a read has 5 on average, 12 at the 99th percentile): a 448-byte scratch,
    // so that with the 8 counter planes of the 8-bit variant 8 waves fit a SIMD's share of the LDS
    const uint32_t per_wave = W * PLANES * 4 + k3a_scratch_bytes();
    unsigned char* mine = smem + (size_t)wv * per_wave;
    WaveScratch sc;
    sc.h_begin = (uint64_t*)mine;
    sc.h_soff = sc.h_begin + K3A_GROUP;
    sc.h_ncodes = (uint32_t*)(sc.h_soff + K3A_GROUP);
    sc.pref = sc.h_ncodes + K3A_GROUP;
    sc.h_score = (int32_t*)(sc.pref + K3A_GROUP);
    sc.h_body = nullptr;
    uint32_t* SC = (uint32_t*)(mine + k3a_scratch_bytes());
    const WorkQueue wq{tickets, n_reads, 8};
    uint64_t t_first;
    uint32_t t_count;

    const ListDesc none{0, 0, 0, 0xFFu, 0, 0};
    while (wq.pull(t_first, t_count)) {
    // ids and multiplicities straight from the lookup kernel's slab, one 32-byte descriptor gather per list; as in
    // k2a the ids of read i + 2 and the descriptors of read i + 1 are requested while read i is processed
    const uint64_t rl = min(t_first + (uint64_t)lane, n_reads - 1);
    const uint32_t cnt_l = (uint32_t)lane < t_count ? nids[rl] : 0u;
    const uint64_t off_l = idoff[rl];
    auto fetch_ids = [&](uint32_t i, uint32_t& id, uint32_t& mult) {
        const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, i);
        id = mult = 0;
        if ((uint32_t)lane < min(cn, K3A_GROUP)) {
            const uint64_t p = readlane_u64(off_l, i) + lane;
            id = ids_pool[p];
            mult = cnt_pool[p];
        }
    };
    auto fetch_desc = [&](uint32_t i, uint32_t id, uint32_t mult) -> ListDesc {
        const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, i);
        ListDesc dd = none;
        if ((uint32_t)lane < min(cn, K3A_GROUP)) {
            dd = c.set_desc[id];
            dd.score = (int32_t)mult;
        }
        return dd;
    };
    uint32_t id1, mu1, id2, mu2;
    fetch_ids(0, id1, mu1);
    ListDesc dcur = fetch_desc(0, id1, mu1);
    fetch_ids(1, id1, mu1);
    for (uint32_t ri = 0; ri < t_count; ++ri) {
        const uint64_t r = t_first + ri;
        const uint64_t off = readlane_u64(off_l, ri);
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, ri);
        fetch_ids(ri + 2, id2, mu2);
        const ListDesc dnext = fetch_desc(ri + 1, id1, mu1);
        const ListDesc d_first = dcur;
        dcur = dnext;
        id1 = id2;
        mu1 = mu2;
        uint32_t* bm = out_bitmap + r * W;
        if (cnt == 0) {
            for (uint32_t w = lane; w < W; w += 64) bm[w] = 0;
            if (lane == 0) out_count[r] = 0;
            if (scores_out)
                for (uint32_t cc = lane; cc < n; cc += 64) scores_out[r * (uint64_t)n + cc] = 0;
            continue;
        }
        auto load_desc = [&](uint32_t g) -> ListDesc {  // more than K3A_GROUP lists: rare, fetched in place
            ListDesc d = none;
            if (g + lane < cnt && (uint32_t)lane < K3A_GROUP) {
                d = c.set_desc[ids_pool[off + g + lane]];
                d.score = (int32_t)cnt_pool[off + g + lane];
            }
            return d;
        };
        const uint32_t min_score = (uint32_t)(unsigned long long)((double)npos[r] * tau);
        uint32_t comp_total = 0;
        for (uint32_t g = 0; g < cnt; g += K3A_GROUP) {
            const ListDesc d = g ? load_desc(g) : d_first;
            comp_total += wave_sum_u32(desc_type(d) == D_ENC_COMPLEMENT ? (uint32_t)d.score : 0u);
        }
        const uint32_t start = (BIASED ? HALF - min_score + comp_total : comp_total) * ONES;
        for (uint32_t i = lane; i < (W >> 2) * PLANES; i += 64) ((uint4*)SC)[i] = make_uint4(start, start, start, start);
        wave_lds_sync();
        for (uint32_t g = 0; g < cnt; g += K3A_GROUP) {
            const ListDesc d = g ? load_desc(g) : d_first;
            ListHeader h;
            h.size = 0;
            h.type = (int)(int8_t)(d.meta & 0xFFu); h.ncodes = d.ncodes; h.begin = d.begin; h.body = desc_body(d); h.soff = d.soff;
            const int32_t score = d.score;
            const uint32_t nblk = h.ncodes;  // gap-coded lists of both kinds
            const uint32_t incl = wave_incl_scan_u32(nblk);
            const uint32_t excl = incl - nblk;
            if ((uint32_t)lane < K3A_GROUP) {  // (lanes past the group hold empty descriptors)
                sc.h_begin[lane] = h.begin; sc.h_soff[lane] = h.type == D_ENC_BITMAP ? h.body : h.soff; sc.h_ncodes[lane] = nblk;
                sc.h_score[lane] = h.type == D_ENC_COMPLEMENT ? -score : score;
                sc.pref[lane] = incl;
            }
            const uint32_t total_blk = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            wave_lds_sync();

            uint64_t mb = __ballot(h.type == D_ENC_BITMAP);
            while (mb) {
                const int src = __builtin_ctzll(mb);
                mb &= mb - 1;
                const uint32_t s = (uint32_t)sc.h_score[src];
                const uint32_t* words = c.bmp_words + sc.h_soff[src];
                for (uint32_t w = lane; w * 32 < n; w += 64) {
                    const uint32_t x = words[w];
#pragma unroll
                    for (uint32_t q = 0; q < PLANES; ++q)  // only this lane touches these words
                        atomicAdd(&SC[q * W + w], counter_spread<BITS>(x, q, s));
                }
                wave_lds_sync();
            }

            for (uint32_t s0 = 0; s0 < total_blk; s0 += 64) {
                BlockLane bl{0u, 0u, 0u, 0u};
                const uint32_t s = s0 + lane;
                if (s < total_blk) {
                    const uint32_t i = owner_list(excl, min(K3A_GROUP, cnt - g), s);
                    const uint32_t j = s - (sc.pref[i] - sc.h_ncodes[i]);
                    const uint64_t hd = sc.h_ncodes[i] == 1 ? sc.h_soff[i] : ((const uint64_t*)(c.blk_words + sc.h_begin[i]))[j];  // (as in k2a)
                    bl.word = (uint32_t)sc.h_begin[i] + blk_rel_word(hd);
                    bl.start = blk_start(hd);
                    bl.meta = blk_width(hd) | ((blk_count(hd) - 1u) << 5);
                    bl.extra = (uint32_t)sc.h_score[i];
                }
                run_blocks(c.blk_words, bl, min(64u, total_blk - s0), lane,
                           [&](uint32_t v, uint32_t sv) {
                               atomicAdd(&SC[(v % PLANES) * W + (v >> 5)], sv << ((BITS & 31) * ((v & 31u) / PLANES)));
                           },
                           [&](uint32_t wi, uint32_t x, uint32_t sv) {  // only this lane touches these words
#pragma unroll
                               for (uint32_t q = 0; q < PLANES; ++q) atomicAdd(&SC[q * W + wi], counter_spread<BITS>(x, q, sv));
                           },
                           [](uint32_t) {});
            }
            wave_lds_sync();
        }
        if (scores_out) {  // counter = HALF - min_score + score  =>  score = counter - HALF + min_score
            for (uint32_t cc = lane; cc < n; cc += 64) {
                const uint32_t x = SC[(cc % PLANES) * W + (cc >> 5)];
                const uint32_t field = BITS == 32 ? x : ((x >> ((BITS & 31) * ((cc & 31u) / PLANES))) & ((1u << (BITS & 31)) - 1u));
                scores_out[r * (uint64_t)n + cc] = BIASED ? field - HALF + min_score : field;
            }
        }
        uint32_t pc = 0;
        const uint32_t thr_c = (256u - min_score) & 0xFFu;  // (unbiased counters only)
        const uint32_t add7 = (thr_c & 0x7Fu) * 0x01010101u, top7 = (thr_c & 0x80u) ? 0xFFFFFFFFu : 0u;
        const uint32_t all_pass = min_score == 0 ? 0xFFFFFFFFu : 0u;
        for (uint32_t w = lane; w < W; w += 64) {
            uint32_t m = 0;
#pragma unroll
            for (uint32_t q = 0; q < PLANES; ++q) {
                const uint32_t x = SC[q * W + w];
                if (BIASED) {
                    m = (m >> 1) | (x & (ONES << (BITS - 1)));  // plane q ends PLANES - 1 - q = BITS - 1 - q places below its field's top bit
                } else {  // byte >= min_score  <=>  carry out of byte + (256 - min_score); min_score = 0 keeps every colour
                    const uint32_t low = (x & 0x7F7F7F7Fu) + add7;         // carry into bit 7 of every byte
                    const uint32_t out = (x & low) | ((x ^ low) & top7);   // majority(x7, low7, bit 7 of 256 - min_score)
                    m |= (((out | all_pass) >> 7) & ONES) << q;
                }
            }
            if (w >= (n >> 5)) m &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
            bm[w] = m;
            pc += __popc(m);
        }
        pc = wave_sum_u32(pc);
        if (lane == 0) out_count[r] = pc;
        wave_lds_sync();
    }
    }
}

// ---------------------------------------------------------------------------------------------
// K3r: threshold union over dense rows -> bitmap + cardinality. Same counters as k3a_union (biased 8/16/32-bit fields,
// colour c -> plane c % PLANES, word c / 32, field (c % 32) / PLANES), but every colour set is its plain row
// (k_rows_build): no descriptors, no blocks, no complemented lists — a list adds its multiplicity to the counters of its
// members, `merge` as the reference states it (ps_threshold_union.cpp:16-40) before its complement trick. Lane = word of
// the row (rounds of 64 words), so the PLANES counter words of a word's 32 colours belong to one lane and live in
// its REGISTERS: no LDS at all; spreading a row word over them costs one shift, one mask and one multiply-add per plane.
// Ids and multiplicities arrive through scalar loads (once per round), the row words of up to four lists are in flight at once.
// ---------------------------------------------------------------------------------------------
// word at byte offset `bo` (32-bit, per lane) of a row whose base is wave-uniform: scalar base + 32-bit vector offset, the form the
// global loads take without a 64-bit address per lane and load (two VGPRs and a v_lshl_add_u64 each)
__device__ __forceinline__ uint32_t row_word(const uint32_t* row, uint32_t bo) { return *(const uint32_t*)((const char*)row + bo); }

template <int K, int BITS>
__device__ __forceinline__ void rows_spread(const uint32_t* __restrict__ rows, uint32_t W, const u32x4 id, const u32x4 mult, uint32_t wi,
                                            uint32_t (&cnt)[BITS]) {
    constexpr uint32_t ONES = BITS == 8 ? 0x01010101u : (BITS == 16 ? 0x00010001u : 1u);
    uint32_t x[K];
#pragma unroll
    for (int j = 0; j < K; ++j) x[j] = row_word(rows + (uint64_t)id[j] * W, wi << 2);
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (uint32_t q = 0; q < BITS; ++q) cnt[q] += ((x[j] >> q) & ONES) * mult[j];  // (PLANES = BITS)
}

// Threshold union of a read with few lists without any counter: whether colour c passes depends only on WHICH of the read's
// L lists contain it, a boolean function f of L bits that is monotone (multiplicities are positive). Its truth table — 2^L
// entries, one ballot: lane p adds up the multiplicities of the lists in pattern p and compares with min_score — is wave-uniform,
// and f is evaluated on whole row words as a multiplexer tree over the lists (Shannon expansion, bit-sliced over the 32
// colours of a word): a node is lo | (x_l & hi) with hi >= lo, one v_and_or; the leaves are the table's constants, so a node
// of the first level is one of {0, x_0, ~0} = (x_0 | lo) & hi. About 3 * 2^(L-1) operations per word against 24 L + 22 for
// the byte counters: 48 against 142 for five lists.
#ifndef FG_MUX_INLINE
#define FG_MUX_INLINE __forceinline__
#endif
constexpr uint32_t K3R_MUX_LISTS = 6;  // reads of up to this many lists take the multiplexer tree (2^L = one ballot)
// R row words per lane go through the tree together: the two scalars of a leaf (bits 2p and 2p + 1 of the table as masks) are made
// where the leaf stands and used at once by its 2 R instructions. (With one word per lane inside a loop over the rounds of a row the
// compiler hoisted the 2^L leaf scalars of every inlined variant out of the loop: 175 SGPR spills, 542 of 1446 static VALU
// instructions were v_writelane / v_readlane on the spill registers — round-4 review, item 3.)
template <int LEVEL, int R, uint32_t P>  // P: pattern of the lists above LEVEL
__device__ __forceinline__ void mux_tree(const uint32_t (&x)[K3R_MUX_LISTS][R], uint32_t table_lo, uint32_t table_hi, uint32_t (&out)[R]) {
    if constexpr (LEVEL == 1) {
        // table bits 2P and 2P + 1 as all-zeros / all-ones masks, made HERE (the statements keep their order): s_bfe_i32 with width 1
        uint32_t lo, hi;
        constexpr uint32_t at = (2 * P) & 31u;
        const uint32_t word = 2 * P < 32 ? table_lo : table_hi;
        asm volatile("s_bfe_i32 %0, %2, %3\n\ts_bfe_i32 %1, %2, %4" : "=&s"(lo), "=s"(hi) : "s"(word), "n"(0x10000u | at), "n"(0x10000u | (at + 1)) : "scc");
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = (x[0][r] | lo) & hi;
    } else {
        uint32_t lo[R], hi[R];
        mux_tree<LEVEL - 1, R, 2 * P>(x, table_lo, table_hi, lo);
        mux_tree<LEVEL - 1, R, 2 * P + 1>(x, table_lo, table_hi, hi);
#pragma unroll
        for (int r = 0; r < R; ++r) out[r] = lo[r] | (x[LEVEL - 1][r] & hi[r]);
    }
}
// The whole read: chunks of R rounds of 64 row words, R words per lane; returns the lane's share of the result's cardinality. `id` =
// the L lists that go through the tree; the lists in `mand` (a mask over the lanes, whose `id_l` is the lane's list) must contain
// every result colour: their rows are ANDed in (the first two with the loads of the tree's lists).
template <int L, int R>
__device__ FG_MUX_INLINE uint32_t mux_union_read(const uint32_t* __restrict__ rows, uint32_t W, uint32_t Wn, uint32_t n, const uint32_t (&id)[K3R_MUX_LISTS],
                                                   uint64_t table, uint64_t mand, uint32_t id_l, uint32_t* __restrict__ bm, int lane) {
    uint32_t pc = 0;
    uint64_t rest = mand;
    const bool m0 = rest != 0;
    const uint32_t i0 = m0 ? (uint32_t)__builtin_amdgcn_readlane((int)id_l, (int)__builtin_ctzll(rest)) : 0u;
    rest &= rest - 1;
    const bool m1 = rest != 0;
    const uint32_t i1 = m1 ? (uint32_t)__builtin_amdgcn_readlane((int)id_l, (int)__builtin_ctzll(rest)) : 0u;
    rest &= rest - 1;
    for (uint32_t w0 = 0; w0 < Wn; w0 += 64 * R) {
        uint32_t wi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wi[r] = min(w0 + 64u * r + (uint32_t)lane, W - 1) << 2;  // byte offset in a row (lanes past the row load its last word and store nothing)
        uint32_t x[K3R_MUX_LISTS][R];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const uint32_t* row = rows + (uint64_t)id[l] * W;
#pragma unroll
            for (int r = 0; r < R; ++r) x[l][r] = row_word(row, wi[r]);
        }
        uint32_t m[R];
#pragma unroll
        for (int r = 0; r < R; ++r) m[r] = 0xFFFFFFFFu;
        if (m0) {  // (wave-uniform)
            const uint32_t* row = rows + (uint64_t)i0 * W;
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] = row_word(row, wi[r]);
        }
        if (m1) {
            const uint32_t* row = rows + (uint64_t)i1 * W;
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] &= row_word(row, wi[r]);
        }
        if constexpr (L == 0) {
            const uint32_t all = 0u - (uint32_t)(table & 1ull);
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] &= all;
        } else {
            uint32_t t[R];
            mux_tree<L, R, 0u>(x, (uint32_t)table, (uint32_t)(table >> 32), t);
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] &= t[r];
        }
        for (uint64_t mm = rest; mm; mm &= mm - 1) {  // (wave-uniform; more than two mandatory lists are rare)
            const uint32_t* row = rows + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)id_l, (int)__builtin_ctzll(mm)) * W;
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] &= row_word(row, wi[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t w = w0 + 64u * r + (uint32_t)lane;
            uint32_t v = m[r];
            if (w >= (n >> 5)) v &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
            if (w < W) {
                bm[w] = v;
                pc += __popc(v);
            }
        }
    }
    return pc;
}

// Threshold union of a read with more free lists than the multiplexer tree takes (round 6; 17 % of the reads at tau = 0.8, a third at
// tau = 0.5): bit-sliced DEFICIT counters. A colour passes iff the multiplicities of the free lists that do NOT contain it add up to
// at most slack = P - min_score (and every mandatory list contains it: one AND per row word). The deficits of the 32 colours of a row
// word are B bit planes (B = 5 for slack < 32, 6 for slack < 64) that START at 2^B - 1 - slack, so a colour fails exactly when one
// of the additions carries out of the top plane (a sticky OV plane; nothing to compare at the end). Adding the wave-uniform mu under
// the mask ~x is a full adder per plane with the addend's bit as a scalar mask: and, xor3, majority = three instructions, 16 / 19 per
// list and word against the 24 of the byte counters (eight planes of shift, mask, multiply-add) — and six / seven registers per row
// word instead of eight, which is what lets a lane hold THREE row words (the whole 4546-colour row in one go, as the multiplexer
// tree does) without spilling: a read of ten lists waits for its row words three times (groups of G lists) instead of nine.
// How it got here (profiles/r6/k3r_variants_r6.txt; the byte counters take 6.4 ms at tau = 0.8 and 7.6 at 0.5):
//   1. a ripple from mu's lowest set bit up, two or three instructions per plane by scalar branches on mu's bits, three words per
//      lane, planes of 5 / 6 / 8: 12.2 / 15.1 ms (the kernel doubled, 36 vector registers spilled);
//   2. branch-free, one word per lane: 6.33 ms (the compiler makes four instructions of a plane), 6.15-6.23 with the adder spelled
//      as two v_bitop3; tau = 0.5 no better than the byte counters;
//   3. the adder specialised on the length of mu (half adders above its top bit: the lists' multiplicities have 1 / 2 / 3 / 4 / 5
//      bits in 10 / 17 / 27 / 33 / 13 % of the cases): FEWER instructions and slower, 6.29-6.37 one word per lane, 5.73 with three;
//   4. form 2 with three words per lane and groups of 3 (5 planes) / 2 (6 planes) lists: **5.36-5.43 ms at tau = 0.8, 6.09 at 0.5**
//      (groups of 4 / 3: four registers spilled, 5.42-5.43; groups of 2: 5.55; 8 waves per SIMD: 5.79). Shipped.
// The kernel is held both by its vector instructions (82 % of the cycles) and by the chain of dependent loads of a read: fewer
// instructions alone (forms 2, 3, and the counters over the free lists only) or a shorter chain alone (whole-read byte counters:
// spills) moved little; both at once took 16-20 % off.
#ifndef FG_K3R_D5_R  // (variant builds: rounds of 64 row words per lane, lists in flight for 5 / 6 planes, adders by length)
#define FG_K3R_D5_R 3
#endif
#ifndef FG_K3R_D5_G
#define FG_K3R_D5_G 3
#endif
#ifndef FG_K3R_D6_G
#define FG_K3R_D6_G 2
#endif
#ifndef FG_K3R_D5_LEN
#define FG_K3R_D5_LEN 0
#endif
template <int LEN, int B>  // LEN = bits of mu (the planes above are half adders on the carry), or B + 1: every plane a full adder under the bit's mask
__device__ __forceinline__ void deficit_add_len(uint32_t (&D)[B], uint32_t& OV, uint32_t x, uint32_t mu) {
    uint32_t c = 0;
    const uint32_t nx = ~x;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const uint32_t d = D[b];
        if (b < LEN - 1) {
            const uint32_t a = nx & (0u - ((mu >> b) & 1u));  // (s_bfe_i32: mu is wave-uniform)
            if (b == 0) {
                D[b] = d ^ a;
                c = d & a;
            } else {  // (left to itself the compiler makes four instructions of a plane: d ^ a, d & a, sum, carry)
                D[b] = __builtin_amdgcn_bitop3_b32(d, a, c, 0x96);
                c = __builtin_amdgcn_bitop3_b32(d, a, c, 0xe8);
            }
        } else if (b == LEN - 1) {
            if (b == 0) {
                D[b] = d ^ nx;
                c = d & nx;
            } else {
                D[b] = __builtin_amdgcn_bitop3_b32(d, x, c, 0x69);  // d ^ ~x ^ c
                c = __builtin_amdgcn_bitop3_b32(d, x, c, 0xb2);     // majority(d, ~x, c)
            }
        } else {
            D[b] = d ^ c;
            c = d & c;
        }
    }
    OV |= c;
}
template <int B>
__device__ __forceinline__ void deficit_add_any(uint32_t (&D)[B], uint32_t& OV, uint32_t x, uint32_t mu) {
    switch (32 - __builtin_clz(mu)) {  // (wave-uniform; 1 <= mu <= slack < 2^B)
        case 1: deficit_add_len<1, B>(D, OV, x, mu); break;
        case 2: deficit_add_len<2, B>(D, OV, x, mu); break;
        case 3: deficit_add_len<3, B>(D, OV, x, mu); break;
        case 4: deficit_add_len<4, B>(D, OV, x, mu); break;
        case 5: deficit_add_len<5, B>(D, OV, x, mu); break;
        default: deficit_add_len<B, B>(D, OV, x, mu); break;
    }
}
// The whole read: chunks of R rounds of 64 row words, R words per lane; returns the lane's share of the result's cardinality. FREE /
// MAND = masks over the lanes whose id_l / mu_l are the lists that count / that every result colour must be in.
template <int B, int R, int G>
__device__ __forceinline__ uint32_t deficit_union_read(const uint32_t* __restrict__ rows, uint32_t W, uint32_t Wn, uint32_t n, uint64_t FREE, uint64_t MAND,
                                                       uint32_t id_l, uint32_t mu_l, uint32_t slack, uint32_t* __restrict__ bm, int lane) {
    uint32_t pc = 0;
    const uint32_t start = ((1u << B) - 1u) - slack;
    for (uint32_t w0 = 0; w0 < Wn; w0 += 64 * R) {
        uint32_t wi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wi[r] = min(w0 + 64u * r + (uint32_t)lane, W - 1) << 2;  // byte offset in a row (lanes past the row load its last word and store nothing)
        uint32_t D[R][B], OV[R], m[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            OV[r] = 0;
            m[r] = 0xFFFFFFFFu;
#pragma unroll
            for (int b = 0; b < B; ++b) D[r][b] = 0u - ((start >> b) & 1u);
        }
        uint64_t mm = MAND;
        if (mm) {  // (wave-uniform) the first mandatory list's words travel with the first group's
            const uint32_t* row = rows + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)id_l, (int)__builtin_ctzll(mm)) * W;
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] = row_word(row, wi[r]);
            mm &= mm - 1;
        }
        for (uint64_t ff = FREE; ff;) {  // the row words of G free lists in flight
            uint32_t idv[G], muv[G], kk = 0;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const uint32_t a = ff ? (uint32_t)__builtin_ctzll(ff) : 0u;
                idv[j] = ff ? (uint32_t)__builtin_amdgcn_readlane((int)id_l, a) : 0u;
                muv[j] = ff ? (uint32_t)__builtin_amdgcn_readlane((int)mu_l, a) : 0u;
                kk += ff ? 1u : 0u;
                ff &= ff - 1;
            }
            uint32_t x[G][R];
#pragma unroll
            for (int j = 0; j < G; ++j)
                if ((uint32_t)j < kk) {  // (wave-uniform)
                    const uint32_t* row = rows + (uint64_t)idv[j] * W;
#pragma unroll
                    for (int r = 0; r < R; ++r) x[j][r] = row_word(row, wi[r]);
                }
#pragma unroll
            for (int j = 0; j < G; ++j)
                if ((uint32_t)j < kk) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (FG_K3R_D5_LEN) deficit_add_any<B>(D[r], OV[r], x[j][r], muv[j]);
                        else deficit_add_len<B + 1, B>(D[r], OV[r], x[j][r], muv[j]);
                    }
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) m[r] &= ~OV[r];
        for (; mm; mm &= mm - 1) {  // (wave-uniform)
            const uint32_t* row = rows + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)id_l, (int)__builtin_ctzll(mm)) * W;
#pragma unroll
            for (int r = 0; r < R; ++r) m[r] &= row_word(row, wi[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t w = w0 + 64u * r + (uint32_t)lane;
            uint32_t v = m[r];
            if (w >= (n >> 5)) v &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
            if (w < W) {
                bm[w] = v;
                pc += __popc(v);
            }
        }
    }
    return pc;
}

// Threshold union by byte counters for a whole read at once (round 6): the three rounds of 64 row words of a 4546-colour row go
// through the counters together (three words per lane, as in the multiplexer tree), the row words of FOUR lists — twelve per lane — are
// requested in one go, and the next four lists' words are requested before the current ones are spread into the counters. The loop of
// the first five rounds fetched four words per lane, waited, counted, and did that nine times for a read of ten lists: the knock-out
// builds say its reads cost 1.8 of the kernel's 6.5 ms at a sixth of the reads, and fewer instructions alone (counters over the free
// lists only; bit-sliced deficits with one word per lane) did not move that.
// MEASURED and NOT in the shipped build (-DFG_K3R_WHOLE_READ_COUNTERS): groups of 1 / 2 / 4 lists, 7 and 6 waves per SIMD: 10.2 / 11.7 /
// 14.3 ms against the 6.4-6.5 ms of the round-by-round loop (profiles/r6/k3r_variants_r6.txt). Twenty-four counter registers beside the
// words in flight do not fit the 72 (80) registers the kernel may use beside the multiplexer tree's paths: 16 to 41 vector registers
// are spilled to scratch, inside the loop. (What did shorten the chain is deficit_union_read above: its counters take six to eight
// registers per row word, so three words fit a lane.)
template <bool BIASED>
__device__ __forceinline__ uint32_t counter_union_read(const uint32_t* __restrict__ rows, uint32_t W, uint32_t Wn, uint32_t n, const uint32_t* __restrict__ ids,
                                                       const uint32_t* __restrict__ mults, uint32_t nl, uint32_t min_score, uint32_t* __restrict__ bm, int lane) {
#ifndef FG_K3R_GROUP
#define FG_K3R_GROUP 2
#endif
    constexpr int R = 3, PLANES = 8, GL = FG_K3R_GROUP;
    constexpr uint32_t ONES = 0x01010101u, HALF = 128u;
    const uint32_t start = (BIASED ? HALF - min_score : 0u) * ONES;
    const uint32_t thr_c = (256u - min_score) & 0xFFu;  // (unbiased counters only)
    const uint32_t add7 = (thr_c & 0x7Fu) * 0x01010101u, top7 = (thr_c & 0x80u) ? 0xFFFFFFFFu : 0u;
    const uint32_t all_pass = min_score == 0 ? 0xFFFFFFFFu : 0u;
    uint32_t pc = 0;
    for (uint32_t w0 = 0; w0 < Wn; w0 += 64 * R) {
        uint32_t wi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wi[r] = min(w0 + 64u * r + (uint32_t)lane, W - 1) << 2;  // byte offset in a row (lanes past the row load its last word and store nothing)
        uint32_t cnt[PLANES][R];
#pragma unroll
        for (int q = 0; q < PLANES; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) cnt[q][r] = start;
        // a group = GL lists (the last one padded with its first list at multiplicity 0: a row that is in the cache, a product that is 0)
        auto fetch = [&](uint32_t i, uint32_t (&x)[GL][R], uint32_t (&mu)[GL]) {
            const uint32_t left = nl - i;
            // ids and multiplicities through the scalar cache (the slabs have room behind a read's last id: what is read past it is not used)
            typedef const __attribute__((address_space(4))) uint32_t* s1_ptr;
            uint32_t idv[GL], muv[GL];
#pragma unroll
            for (int j = 0; j < GL; ++j) { idv[j] = *(s1_ptr)(ids + i + j); muv[j] = *(s1_ptr)(mults + i + j); }
#pragma unroll
            for (int j = 0; j < GL; ++j) {
                const bool there = (uint32_t)j < left;  // (wave-uniform)
                const uint32_t* row = rows + (uint64_t)(there ? idv[j] : idv[0]) * W;
                mu[j] = there ? muv[j] : 0u;
#pragma unroll
                for (int r = 0; r < R; ++r) x[j][r] = row_word(row, wi[r]);
            }
        };
        uint32_t xa[GL][R], ma[GL];
        fetch(0, xa, ma);
        for (uint32_t i = 0; i < nl; i += GL) {
            uint32_t xb[GL][R], mb[GL];
            const bool more = i + GL < nl;  // (wave-uniform)
            if (more) fetch(i + GL, xb, mb);
#pragma unroll
            for (int j = 0; j < GL; ++j)
#pragma unroll
                for (int q = 0; q < PLANES; ++q)
#pragma unroll
                    for (int r = 0; r < R; ++r) cnt[q][r] += ((xa[j][r] >> q) & ONES) * ma[j];
            if (more) {
#pragma unroll
                for (int j = 0; j < GL; ++j) {
                    ma[j] = mb[j];
#pragma unroll
                    for (int r = 0; r < R; ++r) xa[j][r] = xb[j][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t m = 0;
#pragma unroll
            for (int q = 0; q < PLANES; ++q) {
                const uint32_t x = cnt[q][r];
                if (BIASED) {
                    m = (m >> 1) | (x & (ONES << 7));  // plane q ends 7 - q places below its field's top bit
                } else {  // byte >= min_score  <=>  carry out of byte + (256 - min_score); min_score = 0 keeps every colour
                    const uint32_t low = (x & 0x7F7F7F7Fu) + add7;
                    const uint32_t out = (x & low) | ((x ^ low) & top7);
                    m |= (((out | all_pass) >> 7) & ONES) << q;
                }
            }
            const uint32_t w = w0 + 64u * r + (uint32_t)lane;
            if (w >= (n >> 5)) m &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
            if (w < W) {
                bm[w] = m;
                pc += __popc(m);
            }
        }
    }
    return pc;
}

template <int BITS, bool BIASED = true, bool SCORES = false>
// (7 waves per SIMD for the 8-bit counters: 72 VGPRs and 94 SGPRs leave 4 scalars spilled and no scratch, and it is the fastest of 6 / 7 / 8:
// 7.08 / 6.43 / 6.58 ms, profiles/r5/k3r_variants_r5.txt; round 6, with the deficit counters: 5.4 / 5.05-5.13 / 6.15 ms)
#ifndef FG_K3R_WAVES16  // (variant builds: waves per SIMD of the 16- and 32-bit counter instantiations)
#define FG_K3R_WAVES16 6
#endif
#ifndef FG_K3R_WAVES32
#define FG_K3R_WAVES32 4
#endif
#ifndef FG_K3R_WAVES8
#define FG_K3R_WAVES8 7
#endif
__global__ __launch_bounds__(256, BITS == 8 ? FG_K3R_WAVES8 : (BITS == 16 ? FG_K3R_WAVES16 : FG_K3R_WAVES32)) void k3r_union(const uint32_t* __restrict__ rows, uint32_t W, uint32_t n,
                                                                  const uint32_t* __restrict__ npos, const uint32_t* __restrict__ nids,
                                                                  const uint64_t* __restrict__ idoff, const uint32_t* __restrict__ ids_pool,
                                                                  const uint32_t* __restrict__ cnt_pool, double tau, uint64_t n_reads,
                                                                  uint32_t* __restrict__ out_bitmap, uint32_t* __restrict__ out_count,
                                                                  unsigned int* tickets, uint32_t* __restrict__ scores_out) {
    constexpr uint32_t PLANES = BITS, HALF = 1u << (BITS - 1);
    constexpr uint32_t ONES = BITS == 8 ? 0x01010101u : (BITS == 16 ? 0x00010001u : 1u);
    const int lane = lane_id();
    const uint32_t Wn = (n + 31) >> 5;  // words that hold colours
    typedef const __attribute__((address_space(4))) u32x4_a4* s4_ptr;  // scalar loads, as in k2r_intersect
    #ifndef FG_K3R_TICKET  // (variant builds: reads per ticket. 4 / 8 / 16 / 32: 5.85 / 5.17 / 5.06-5.11 / 5.03 ms at tau = 0.8, profiles/r6/k3r_variants_r6.txt)
#define FG_K3R_TICKET 16
#endif
    const WorkQueue wq{tickets, n_reads, FG_K3R_TICKET};
    uint64_t t_first;
    uint32_t t_count;
    while (wq.pull(t_first, t_count)) {
        const uint64_t rl = min(t_first + (uint64_t)lane, n_reads - 1);
        const uint32_t cnt_l = (uint32_t)lane < t_count ? nids[rl] : 0u;
        const uint64_t off_l = idoff[rl];
        const uint32_t np_l = npos[rl];
        const uint32_t min_l = (uint32_t)(unsigned long long)((double)np_l * tau);  // ps_threshold_union.cpp:389
#ifndef FG_K3R_NO_PREFETCH_IDS
        // the next read's ids and multiplicities are requested before the current read's rows: a read waits once, for its rows
        uint32_t nx_id = 0, nx_mu = 0;
        if (!SCORES && t_count) {
            const uint32_t nl0 = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, 0);
            const uint64_t off0 = readlane_u64(off_l, 0);
            if ((uint32_t)lane < nl0) { nx_id = ids_pool[off0 + lane]; nx_mu = cnt_pool[off0 + lane]; }
        }
#endif
        for (uint32_t ri = 0; ri < t_count; ++ri) {
            const uint64_t r = t_first + ri;
#ifndef FG_K3R_NO_PREFETCH_IDS
            const uint32_t pf_id = nx_id, pf_mu = nx_mu;
            nx_id = 0; nx_mu = 0;
            if (!SCORES && ri + 1 < t_count) {
                const uint32_t nl1 = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, ri + 1);
                const uint64_t off1 = readlane_u64(off_l, ri + 1);
                if ((uint32_t)lane < nl1) { nx_id = ids_pool[off1 + lane]; nx_mu = cnt_pool[off1 + lane]; }
            }
#endif
            // (the lane number is made opaque once per read: everything derived from it — two dozen lane masks of this loop nest — would
            // otherwise be computed once in front of the ticket loop and kept in scalar register pairs, more of them than there are
            // registers: they were spilled to vector lanes and read back instruction by instruction)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)cnt_l, ri);
            const uint32_t min_score = (uint32_t)__builtin_amdgcn_readlane((int)min_l, ri);
            const uint32_t positive = (uint32_t)__builtin_amdgcn_readlane((int)np_l, ri);
            const uint64_t off = readlane_u64(off_l, ri);
            uint32_t* bm = out_bitmap + r * W;
            if (nl == 0) {
                for (uint32_t w = ln; w < W; w += 64) bm[w] = 0;
                if (ln == 0) out_count[r] = 0;
                if (SCORES)
                    for (uint32_t cc = ln; cc < n; cc += 64) scores_out[r * (uint64_t)n + cc] = 0;
                continue;
            }
            // A colour that is not in list l scores at most P - m_l: a list with m_l > P - min_score is MANDATORY, every result colour
            // is in it (one AND per row word). What the others contribute is a monotone boolean function of which of them contain the
            // colour: with at most six of them, the multiplexer tree; no counters.
            if (!SCORES && nl <= 64u) {
                const bool has = (uint32_t)ln < nl;
#ifndef FG_K3R_NO_PREFETCH_IDS
                const uint32_t id_l = pf_id, mu_l = pf_mu;
#else
                const uint32_t id_l = has ? ids_pool[off + ln] : 0u, mu_l = has ? cnt_pool[off + ln] : 0u;
#endif
                const uint32_t slack = positive - min_score;
                const uint64_t FREE = __ballot(has && mu_l <= slack), MAND = __ballot(has && mu_l > slack);
                const uint32_t nfree = (uint32_t)__popcll(FREE);
#ifdef FG_K3R_KNOCKOUT  // (timing experiments, results wrong: 1 = the reads of the multiplexer tree are skipped, 2 = those of the byte counters)
                if ((FG_K3R_KNOCKOUT == 1) == (nfree <= K3R_MUX_LISTS)) {
                    for (uint32_t w = ln; w < W; w += 64) bm[w] = 0;
                    if (ln == 0) out_count[r] = 0;
                    continue;
                }
#endif
#ifndef FG_K3R_TREE_LISTS  // (variant builds: free lists up to which the tree is preferred to the deficit counters where both apply. 6 / 5 / 4: 5.17 / 5.16 / 5.27 ms at tau = 0.8, 5.95 / 6.10 / 6.36 at 0.5)
#define FG_K3R_TREE_LISTS 6
#endif
#ifndef FG_K3R_NO_DEFICIT
                // (seven planes, slack < 128, only in the instantiations for reads of more than 127 k-mers: in the other one they cost six spilled registers)
                constexpr bool D7 = BITS > 8 || !BIASED;
                constexpr uint32_t deficit_limit = D7 ? 128u : 64u;
#else
                constexpr uint32_t deficit_limit = 0u;
#endif
                if (nfree <= FG_K3R_TREE_LISTS || (nfree <= K3R_MUX_LISTS && slack >= deficit_limit)) {
                    uint32_t idf[K3R_MUX_LISTS], muf[K3R_MUX_LISTS];
                    uint64_t ff = FREE;
                    uint32_t free_total = 0;
#pragma unroll
                    for (uint32_t j = 0; j < K3R_MUX_LISTS; ++j) {
                        const uint32_t a = ff ? (uint32_t)__builtin_ctzll(ff) : 0u;
                        idf[j] = ff ? (uint32_t)__builtin_amdgcn_readlane((int)id_l, a) : 0u;
                        muf[j] = ff ? (uint32_t)__builtin_amdgcn_readlane((int)mu_l, a) : 0u;
                        free_total += muf[j];
                        ff &= ff - 1;
                    }
                    uint32_t score = positive - free_total;  // the mandatory lists + pattern `ln` of the others
#pragma unroll
                    for (uint32_t j = 0; j < K3R_MUX_LISTS; ++j) score += (((uint32_t)ln >> j) & 1u) ? muf[j] : 0u;
                    const uint64_t table = __ballot(score >= min_score);
                    uint32_t pcm = 0;
                    {   // three rounds of 64 row words at a time (4546 colours: the whole row). One family of instantiations: a second one
                        // for rows of a single round doubled the kernel and brought the scalar spills back (profiles/r5/k3r_variants_r5.txt)
                        switch (nfree) {
                            case 0: pcm = mux_union_read<0, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                            case 1: pcm = mux_union_read<1, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                            case 2: pcm = mux_union_read<2, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                            case 3: pcm = mux_union_read<3, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                            case 4: pcm = mux_union_read<4, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                            case 5: pcm = mux_union_read<5, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                            default: pcm = mux_union_read<6, 3>(rows, W, Wn, n, idf, table, MAND, id_l, bm, ln); break;
                        }
                    }
                    for (uint32_t w = ((Wn + 63) & ~63u) + ln; w < W; w += 64) bm[w] = 0;  // (padding words behind the last round)
                    pcm = wave_sum_u32(pcm);
                    if (ln == 0) out_count[r] = pcm;
                    continue;
                }
#ifndef FG_K3R_NO_DEFICIT  // (variant builds: the byte counters below for every read the tree does not take)
                if (slack < deficit_limit) {  // more free lists than the tree takes: five to seven planes of deficit counters (deficit_union_read)
                    uint32_t pcm;
                    if (slack < 32u) pcm = deficit_union_read<5, FG_K3R_D5_R, FG_K3R_D5_G>(rows, W, Wn, n, FREE, MAND, id_l, mu_l, slack, bm, ln);
                    else if (!D7 || slack < 64u) pcm = deficit_union_read<6, FG_K3R_D5_R, FG_K3R_D6_G>(rows, W, Wn, n, FREE, MAND, id_l, mu_l, slack, bm, ln);
                    else pcm = deficit_union_read<7, FG_K3R_D5_R, FG_K3R_D6_G>(rows, W, Wn, n, FREE, MAND, id_l, mu_l, slack, bm, ln);
                    for (uint32_t w = ((Wn + 64 * FG_K3R_D5_R - 1) / (64 * FG_K3R_D5_R) * (64 * FG_K3R_D5_R)) + ln; w < W; w += 64) bm[w] = 0;  // (padding words behind the last chunk)
                    pcm = wave_sum_u32(pcm);
                    if (ln == 0) out_count[r] = pcm;
                    continue;
                }
#endif
#ifdef FG_K3R_SPLIT_COUNTERS  // (round 6: measured, no gain — 6.49 against 6.47 ms at tau = 0.8, 7.95 against 7.63 at 0.5, profiles/r6/k3r_variants_r6.txt — and left out)
                // More free lists than the tree takes: byte counters as below, but over the FREE lists only — a mandatory list
                // costs one AND per row word at the end instead of a spread into eight planes (24 instructions per word), and the
                // threshold the counters are held against is what the free lists have to bring: min_score less the mandatory lists'
                // multiplicities. Reads of this kind have 8.3 free lists and about 10 in all at tau = 0.8.
                if (BITS == 8) {
                    const uint32_t free_sum = wave_sum_u32(has && mu_l <= slack ? mu_l : 0u);
                    const uint32_t mand_sum = positive - free_sum;
                    const uint32_t min_free = min_score > mand_sum ? min_score - mand_sum : 0u;
                    const uint32_t start_f = (BIASED ? HALF - min_free : 0u) * ONES;
                    const uint32_t thr_f = (256u - min_free) & 0xFFu;  // (unbiased counters only)
                    const uint32_t add7_f = (thr_f & 0x7Fu) * 0x01010101u, top7_f = (thr_f & 0x80u) ? 0xFFFFFFFFu : 0u;
                    const uint32_t all_pass_f = min_free == 0 ? 0xFFFFFFFFu : 0u;
                    uint32_t pcs = 0;
                    for (uint32_t w0 = 0; w0 < Wn; w0 += 64) {
                        const uint32_t w = w0 + (uint32_t)ln;
                        const uint32_t wi = min(w, W - 1);  // (lanes past the row load its last word and store nothing)
                        uint32_t cnt[PLANES];
#pragma unroll
                        for (uint32_t q = 0; q < PLANES; ++q) cnt[q] = start_f;
                        for (uint64_t ff = FREE; ff;) {  // the row words of four free lists in flight
                            u32x4 id, mu;
                            uint32_t kk = 0;
#pragma unroll
                            for (uint32_t j = 0; j < 4; ++j) {
                                const uint32_t a = ff ? (uint32_t)__builtin_ctzll(ff) : 0u;
                                id[j] = ff ? (uint32_t)__builtin_amdgcn_readlane((int)id_l, a) : 0u;
                                mu[j] = ff ? (uint32_t)__builtin_amdgcn_readlane((int)mu_l, a) : 0u;
                                kk += ff ? 1u : 0u;
                                ff &= ff - 1;
                            }
                            switch (kk) {  // (wave-uniform)
                                case 1: rows_spread<1, BITS>(rows, W, id, mu, wi, cnt); break;
                                case 2: rows_spread<2, BITS>(rows, W, id, mu, wi, cnt); break;
                                case 3: rows_spread<3, BITS>(rows, W, id, mu, wi, cnt); break;
                                default: rows_spread<4, BITS>(rows, W, id, mu, wi, cnt); break;
                            }
                        }
                        uint32_t m = 0;
#pragma unroll
                        for (uint32_t q = 0; q < PLANES; ++q) {
                            const uint32_t x = cnt[q];
                            if (BIASED) {
                                m = (m >> 1) | (x & (ONES << (BITS - 1)));
                            } else {
                                const uint32_t low = (x & 0x7F7F7F7Fu) + add7_f;
                                const uint32_t out = (x & low) | ((x ^ low) & top7_f);
                                m |= (((out | all_pass_f) >> 7) & ONES) << q;
                            }
                        }
                        for (uint64_t mm = MAND; mm; mm &= mm - 1)  // (wave-uniform)
                            m &= row_word(rows + (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)id_l, (int)__builtin_ctzll(mm)) * W, wi << 2);
                        if (w >= (n >> 5)) m &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
                        if (w < W) {
                            bm[w] = m;
                            pcs += __popc(m);
                        }
                    }
                    for (uint32_t w = ((Wn + 63) & ~63u) + ln; w < W; w += 64) bm[w] = 0;  // (padding words behind the last round)
                    pcs = wave_sum_u32(pcs);
                    if (ln == 0) out_count[r] = pcs;
                    continue;
                }
#endif
            }
#ifdef FG_K3R_WHOLE_READ_COUNTERS  // (round 6: measured — 9.8 to 14.3 ms against 6.4-6.5 — and left out; see counter_union_read)
            if (BITS == 8 && !SCORES) {
                uint32_t pcs = counter_union_read<BIASED>(rows, W, Wn, n, ids_pool + off, cnt_pool + off, nl, min_score, bm, ln);
                for (uint32_t w = ((Wn + 191) / 192 * 192) + ln; w < W; w += 64) bm[w] = 0;  // (padding words behind the last chunk of three rounds)
                pcs = wave_sum_u32(pcs);
                if (ln == 0) out_count[r] = pcs;
                continue;
            }
#endif
            const uint32_t start = (BIASED ? HALF - min_score : 0u) * ONES;
            const uint32_t thr_c = (256u - min_score) & 0xFFu;  // (unbiased counters only)
            const uint32_t add7 = (thr_c & 0x7Fu) * 0x01010101u, top7 = (thr_c & 0x80u) ? 0xFFFFFFFFu : 0u;
            const uint32_t all_pass = min_score == 0 ? 0xFFFFFFFFu : 0u;
            uint32_t pc = 0;
            // one round = 64 words of the rows, one per ln, with their PLANES counter words in this ln's registers
            for (uint32_t w0 = 0; w0 < Wn; w0 += 64) {
                const uint32_t w = w0 + (uint32_t)ln;
                const uint32_t wi = min(w, W - 1);  // (lanes past the row load its last word and store nothing)
                uint32_t cnt[PLANES];
#pragma unroll
                for (uint32_t q = 0; q < PLANES; ++q) cnt[q] = start;
                for (uint32_t i = 0; i < nl; i += 4) {  // the row words of four lists in flight
                    const u32x4 id = *(s4_ptr)(ids_pool + off + i), mu = *(s4_ptr)(cnt_pool + off + i);
                    switch (min(nl - i, 4u)) {  // (wave-uniform)
                        case 1: rows_spread<1, BITS>(rows, W, id, mu, wi, cnt); break;
                        case 2: rows_spread<2, BITS>(rows, W, id, mu, wi, cnt); break;
                        case 3: rows_spread<3, BITS>(rows, W, id, mu, wi, cnt); break;
                        default: rows_spread<4, BITS>(rows, W, id, mu, wi, cnt); break;
                    }
                }
                uint32_t m = 0;
#pragma unroll
                for (uint32_t q = 0; q < PLANES; ++q) {
                    const uint32_t x = cnt[q];
                    if (BIASED) {
                        m = (m >> 1) | (x & (ONES << (BITS - 1)));  // plane q ends PLANES - 1 - q = BITS - 1 - q places below its field's top bit
                    } else {  // byte >= min_score  <=>  carry out of byte + (256 - min_score); min_score = 0 keeps every colour
                        const uint32_t low = (x & 0x7F7F7F7Fu) + add7;
                        const uint32_t out = (x & low) | ((x ^ low) & top7);
                        m |= (((out | all_pass) >> 7) & ONES) << q;
                    }
                }
                if (w >= (n >> 5)) m &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
                if (w < W) {
                    bm[w] = m;
                    pc += __popc(m);
                    if (SCORES && w < Wn) {  // (index::kmer_matches) counter = HALF - min_score + score
#pragma unroll
                        for (uint32_t bit = 0; bit < 32; ++bit) {
                            const uint32_t cc = w * 32 + bit;
                            const uint32_t x = cnt[bit % PLANES];
                            const uint32_t field = BITS == 32 ? x : ((x >> ((BITS & 31) * (bit / PLANES))) & ((1u << (BITS & 31)) - 1u));
                            if (cc < n) scores_out[r * (uint64_t)n + cc] = BIASED ? field - HALF + min_score : field;
                        }
                    }
                }
            }
            for (uint32_t w = ((Wn + 63) & ~63u) + ln; w < W; w += 64) bm[w] = 0;  // (padding words behind the last round)
            pc = wave_sum_u32(pc);
            if (ln == 0) out_count[r] = pc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Generic colour-set kernel for the meta, differential and meta-differential codecs.
// Every colour set is a short list of ops (host/codecs_build.hpp); every op contributes a set of colours that
// is XORed into the set under construction T (members of disjoint partitions, a representative, a symmetric
// difference). Device form of an op: a 32-byte record holding a span of at most 7 plain words of the colour
// space, one lane per op, or — for large universes — the same packed blocks / bitmap chunks as the hybrid gap
// lists. Then
//   full intersection (meta_intersect / diff_intersect, ps_full_intersection.cpp:129-332): EXCL |= ~T
//   threshold union   (merge_meta / merge_diff / merge_metadiff, ps_threshold_union.cpp:42-318):
//                     score[c] += s for every c in T, keep c iff score[c] >= min_score
// The reference reaches the same sets through partition/cluster shortcuts; the results are the sets.
// ---------------------------------------------------------------------------------------------
struct DevGeneric {
    const uint64_t* set_ops_off;
    const uint32_t* set_ops;  // per colour set: span record index, or G_BLOCK_REF | block-op index
    const uint4* span;        // two uint4 per span op: {first word | count << 24, 7 words}
    const ListDesc* ops;      // block ops (GenOpDev layout: begin, soff, ncodes = #blocks)
    const uint64_t* blk_hdr;
    const uint32_t* blk_words;
    uint32_t n, w32;
};
constexpr uint32_t G_BLOCK_REF = 0x80000000u;

constexpr uint32_t G_SETS = 4;  // colour sets of a read rebuilt concurrently (one LDS plane each)
constexpr uint32_t G_SETS_UNION = 2;  // fewer for the union: its score planes take most of the LDS

template <bool UNION, int BITS, bool BIASED = true>
__global__ __launch_bounds__(256, UNION ? (BITS == 8 ? 5 : 4) : 7) void k_generic(DevGeneric g, const uint32_t* __restrict__ npos, const uint64_t* __restrict__ id_csr,
                          const ListDesc* __restrict__ desc, double tau, uint64_t n_reads,
                          uint32_t* __restrict__ out_bitmap, uint32_t* __restrict__ out_count, unsigned int* tickets,
                          uint32_t* __restrict__ scores_out) {
    // scores_out (UNION only): also store score[c] for every colour, n u32 per read (index::kmer_matches)
    // score counters as in k3a: biased BITS-bit fields (8 for reads of at most 127 k-mers, else 16), or plain 8-bit
    // fields compared byte-wise with min_score for reads of 128 to 255 k-mers (BIASED = false);
    // colour c -> plane c % PLANES, word c / 32, field (c % 32) / PLANES
    constexpr uint32_t PLANES = BITS, HALF = 1u << (BITS - 1), ONES = BITS == 8 ? 0x01010101u : 0x00010001u;
    constexpr uint32_t FIELD = (1u << BITS) - 1u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t W = g.w32, W4 = W >> 2, n = g.n;
    const uint32_t acc_bytes = UNION ? W * 4 * PLANES : W * 4;
    constexpr uint32_t GS = UNION ? G_SETS_UNION : G_SETS;
    const uint32_t per_wave = wave_scratch_bytes_compact() + GS * W * 4 + acc_bytes;
    unsigned char* mine = smem + (size_t)wv * per_wave;
    WaveScratch sc = carve_scratch(mine);
    uint32_t* T = (uint32_t*)(mine + wave_scratch_bytes_compact());  // GS planes of W words
    uint4* T4 = (uint4*)T;
    uint32_t* ACC = T + GS * W;  // EXCL (W words) or the score planes
    uint4* EX4 = (uint4*)ACC;
    const uint32_t tail_word = n >> 5, tail_mask = ~((1u << (n & 31u)) - 1u);
    const ListDesc none{0, 0, 0, 0xFFu, 0, 0};
    const WorkQueue wq{tickets, n_reads, 8};
    uint64_t t_first;
    uint32_t t_count;

    while (wq.pull(t_first, t_count))
    for (uint64_t r = t_first; r < t_first + t_count; ++r) {
        const uint64_t off = id_csr[r];
        const uint32_t cnt = (uint32_t)(id_csr[r + 1] - off);
        uint32_t* bm = out_bitmap + r * W;
        if (cnt == 0) {
            for (uint32_t w = lane; w < W; w += 64) bm[w] = 0;
            if (lane == 0) out_count[r] = 0;
            if (UNION && scores_out)
                for (uint32_t cc = lane; cc < n; cc += 64) scores_out[r * (uint64_t)n + cc] = 0;
            continue;
        }
        const uint32_t min_score = UNION ? (uint32_t)(unsigned long long)((double)npos[r] * tau) : 0u;
        if (UNION) {
            const uint32_t start = BIASED ? (HALF - min_score) * ONES : 0u;  // biased: score >= min_score  <=>  top bit of the field
            for (uint32_t i = lane; i < W4 * PLANES; i += 64) EX4[i] = make_uint4(start, start, start, start);
        } else {
            for (uint32_t g4 = lane; g4 < W4; g4 += 64) {  // colours >= n start excluded
                const uint32_t w = 4 * g4;
                EX4[g4] = make_uint4(w < tail_word ? 0u : (w == tail_word ? tail_mask : 0xFFFFFFFFu),
                                     w + 1 < tail_word ? 0u : (w + 1 == tail_word ? tail_mask : 0xFFFFFFFFu),
                                     w + 2 < tail_word ? 0u : (w + 2 == tail_word ? tail_mask : 0xFFFFFFFFu),
                                     w + 3 < tail_word ? 0u : (w + 3 == tail_word ? tail_mask : 0xFFFFFFFFu));
            }
        }
        // rounds of up to G_SETS colour sets: the (set, op) pairs of the round are spread over the lanes, so the
        // dependent fetches set -> op list -> op -> data are walked once per round, not once per set
        for (uint32_t g0 = 0; g0 < cnt; g0 += GS) {
            const uint32_t nl = min(GS, cnt - g0);
            uint64_t o0 = 0;
            uint32_t nops = 0, score = 0;
            if ((uint32_t)lane < nl) {
                const ListDesc d = desc[off + g0 + lane];
                o0 = g.set_ops_off[d.id];
                nops = (uint32_t)(g.set_ops_off[d.id + 1] - o0);
                score = (uint32_t)d.score;
            }
            for (uint32_t g4 = lane; g4 < nl * W4; g4 += 64) T4[g4] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t incl_ops = wave_incl_scan_u32(nops);
            const uint32_t excl_ops = incl_ops - nops;
            const uint32_t total_ops = (uint32_t)__builtin_amdgcn_readlane((int)incl_ops, 63);
            wave_lds_sync();
            for (uint32_t p0 = 0; p0 < total_ops; p0 += 64) {
                const uint32_t p = p0 + lane;
                ListDesc op = none;
                uint32_t plane = 0, ref = G_BLOCK_REF;  // (lanes past the end: a block reference that is never followed)
                if (p < total_ops) {
                    uint32_t e = 0;  // (v_readlane, unlike a shuffle, also reads lanes that are masked off here)
                    uint64_t first = readlane_u64(o0, 0);  // first op of the set this pair belongs to
                    for (uint32_t i = 1; i < nl; ++i) {
                        const uint32_t ei = (uint32_t)__builtin_amdgcn_readlane((int)excl_ops, i);
                        const uint64_t oi = readlane_u64(o0, i);
                        if (p >= ei) { plane = i; e = ei; first = oi; }
                    }
                    ref = g.set_ops[first + (p - e)];
                }
                const uint32_t tbase = plane * W;
                // span ops: one 32-byte record per lane, its (at most 7) words XORed in place
                if (!(ref & G_BLOCK_REF)) {
                    const uint4 a = g.span[2 * (uint64_t)ref], b = g.span[2 * (uint64_t)ref + 1];
                    uint32_t* dst = T + tbase + (a.x & 0xFFFFFFu);
                    if (a.y) atomicXor(dst + 0, a.y);
                    if (a.z) atomicXor(dst + 1, a.z);
                    if (a.w) atomicXor(dst + 2, a.w);
                    if (b.x) atomicXor(dst + 3, b.x);
                    if (b.y) atomicXor(dst + 4, b.y);
                    if (b.z) atomicXor(dst + 5, b.z);
                    if (b.w) atomicXor(dst + 6, b.w);
                } else if (p < total_ops) {
                    op = g.ops[ref & ~G_BLOCK_REF];
                }
                // block ops: the blocks of all of them through one loop, as in k2a
                const uint32_t nblk = (p < total_ops && (ref & G_BLOCK_REF)) ? op.ncodes : 0u;
                const uint32_t incl = wave_incl_scan_u32(nblk);
                const uint32_t total_blk = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                if (total_blk) {
                    sc.h_begin[lane] = op.begin; sc.h_soff[lane] = op.soff; sc.h_ncodes[lane] = nblk;
                    sc.h_score[lane] = (int32_t)(tbase * 32u);
                    sc.pref[lane] = incl;
                    wave_lds_sync();
                    for (uint32_t s0 = 0; s0 < total_blk; s0 += 64) {
                        BlockLane bl{0u, 0u, 0u, 0u};
                        const uint32_t s = s0 + lane;
                        if (s < total_blk) {
                            const uint32_t i = upper_slot(sc.pref, s);
                            const uint32_t jb = s - (sc.pref[i] - sc.h_ncodes[i]);
                            const uint64_t hd = g.blk_hdr[sc.h_soff[i] + jb];
                            bl.word = (uint32_t)sc.h_begin[i] + blk_rel_word(hd);
                            bl.start = blk_start(hd) + (uint32_t)sc.h_score[i];  // bit index relative to T
                            bl.meta = blk_width(hd) | ((blk_count(hd) - 1u) << 5);
                        }
                        run_blocks(g.blk_words, bl, min(64u, total_blk - s0), lane,
                                   [&](uint32_t v, uint32_t) { atomicXor(&T[v >> 5], 1u << (v & 31)); },
                                   [&](uint32_t wi, uint32_t x, uint32_t) { atomicXor(&T[wi], x); },
                                   [](uint32_t) {});
                    }
                    wave_lds_sync();
                }
            }
            wave_lds_sync();
            for (uint32_t l = 0; l < nl; ++l) {
                if (UNION) {
                    const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)score, l);
                    for (uint32_t w = lane; w < W; w += 64) {
                        const uint32_t x = T[l * W + w];
                        if (x) {
#pragma unroll
                            for (uint32_t q = 0; q < PLANES; ++q) atomicAdd(&ACC[q * W + w], counter_spread<BITS>(x, q, s));
                        }
                    }
                } else {
                    for (uint32_t g4 = lane; g4 < W4; g4 += 64) EX4[g4] = or_not(EX4[g4], T4[l * W4 + g4]);
                }
            }
            wave_lds_sync();
        }
        uint32_t pc = 0;
        if (UNION) {
            if (scores_out) {  // counter = HALF - min_score + score
                for (uint32_t cc = lane; cc < n; cc += 64) {
                    const uint32_t x = ACC[(cc % PLANES) * W + (cc >> 5)];
                    const uint32_t field = (x >> (BITS * ((cc & 31u) / PLANES))) & FIELD;
                    scores_out[r * (uint64_t)n + cc] = BIASED ? field - HALF + min_score : field;
                }
            }
            const uint32_t thr_c = (256u - min_score) & 0xFFu;  // (plain counters only, as in k3a)
            const uint32_t add7 = (thr_c & 0x7Fu) * 0x01010101u, top7 = (thr_c & 0x80u) ? 0xFFFFFFFFu : 0u;
            const uint32_t all_pass = min_score == 0 ? 0xFFFFFFFFu : 0u;
            for (uint32_t w = lane; w < W; w += 64) {
                uint32_t m = 0;
#pragma unroll
                for (uint32_t q = 0; q < PLANES; ++q) {
                    const uint32_t x = ACC[q * W + w];
                    if (BIASED) {
                        m = (m >> 1) | (x & (ONES << (BITS - 1)));  // plane q ends PLANES - 1 - q = BITS - 1 - q places below its field's top bit
                    } else {
                        const uint32_t low = (x & 0x7F7F7F7Fu) + add7;
                        const uint32_t out = (x & low) | ((x ^ low) & top7);
                        m |= (((out | all_pass) >> 7) & ONES) << q;
                    }
                }
                if (w >= (n >> 5)) m &= w == (n >> 5) ? (1u << (n & 31u)) - 1u : 0u;  // (only the last words hold colours >= n)
                bm[w] = m;
                pc += __popc(m);
            }
        } else {
            uint4* bm4 = (uint4*)bm;
            for (uint32_t g4 = lane; g4 < W4; g4 += 64) {
                const uint4 e = EX4[g4];
                const uint4 x = make_uint4(~e.x, ~e.y, ~e.z, ~e.w);
                bm4[g4] = x;
                pc += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w);
            }
        }
        pc = wave_sum_u32(pc);
        if (lane == 0) out_count[r] = pc;
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// sizes -> CSR offsets (three small launches)
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

__global__ __launch_bounds__(256) void scan_block_sums(const uint32_t* __restrict__ counts, uint64_t n,
                                                       uint64_t* __restrict__ block_sums, uint64_t* __restrict__ block_mapped) {
    __shared__ uint64_t s_sum[4], s_map[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t s = 0;  // 64-bit throughout: the sizes may be record bytes of a formatter, not only colour counts
    uint32_t mp = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) { uint32_t v = counts[base + i]; s += v; mp += v != 0; }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    mp = wave_sum_u32(mp);
    if (lane_id() == 0) { s_sum[threadIdx.x >> 6] = s; s_map[threadIdx.x >> 6] = mp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        block_mapped[blockIdx.x] = s_map[0] + s_map[1] + s_map[2] + s_map[3];
    }
}

// single block: exclusive scan of the block sums in place; totals[0] = sum, totals[1] = mapped reads
__global__ __launch_bounds__(256) void scan_top(uint64_t* __restrict__ block_sums, const uint64_t* __restrict__ block_mapped,
                                                uint64_t nb, uint64_t* __restrict__ totals) {
    __shared__ uint64_t s_part[256];
    __shared__ uint64_t s_carry;
    uint64_t mapped = 0;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nb; b0 += 256) {
        const uint64_t i = b0 + threadIdx.x;
        const uint64_t v = i < nb ? block_sums[i] : 0;
        if (i < nb) mapped += block_mapped[i];
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            uint64_t t = threadIdx.x >= (unsigned)o ? s_part[threadIdx.x - o] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        const uint64_t carry = s_carry;
        if (i < nb) block_sums[i] = carry + s_part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = carry + s_part[255];
        __syncthreads();
    }
    s_part[threadIdx.x] = mapped;
    __syncthreads();
    for (int o = 128; o; o >>= 1) {
        if (threadIdx.x < (unsigned)o) s_part[threadIdx.x] += s_part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = s_carry; totals[1] = s_part[0]; }
}

__global__ __launch_bounds__(256) void scan_apply(const uint32_t* __restrict__ counts, uint64_t n,
                                                  const uint64_t* __restrict__ block_sums, uint64_t* __restrict__ offsets) {
    __shared__ uint64_t s_part[256];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < n ? counts[base + i] : 0; s += v[i]; }
    s_part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        uint64_t t = threadIdx.x >= (unsigned)o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t run = block_sums[blockIdx.x] + s_part[threadIdx.x] - s;
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) { offsets[base + i] = run; run += v[i]; }
    if (base <= n - 1 && n - 1 < base + SCAN_ITEMS) offsets[n] = run;
}

// ---------------------------------------------------------------------------------------------
// K2b: bitmap -> ascending u32 colour list. Lane = one 32-colour word of the bitmap: popcount, wave
// prefix sum, every lane scatters the indices of its set bits into an LDS staging buffer at its prefix
// (16-bit entries relative to the round's first colour), then the wave copies the staged run out with
// full-width coalesced stores. 64 words (2048 colours) per round.
// ---------------------------------------------------------------------------------------------
#ifndef FG_K2B_PARTS  // (variant builds: contiguous pieces of the pass that are written at the same time)
#define FG_K2B_PARTS 8
#endif
constexpr uint32_t K2B_MAX_PARTS = 256;
constexpr uint32_t K2B_THREADS = 1024;  // 16 waves share one LDS hit histogram: 2 blocks per CU = 8 waves/SIMD
// Per-wave stage of 16-bit entries: slot i lives at entry i + 2 * (i / 32) — dense words (prefix = 32 * lane) would
// otherwise share 2 banks, and a skew of two entries keeps every aligned group of 4 slots contiguous and 4-byte aligned.
// The skew comes out of the address itself: a slot has the VIRTUAL byte address V = v0 + 4096 * wave + 2 * i and lives at
// LDS byte V + ((V >> 6) << 2) (two instructions per access, no slot counter besides V). With v0 a multiple of 1024 the
// stages of the 16 waves are 4352 bytes each and start at LDS byte v0 * 17 / 16.
constexpr uint32_t K2B_STAGE_BYTES = (2048 + 2 * 64) * 2;
__device__ __forceinline__ uint32_t k2b_stage_skew(uint32_t v) {  // v + ((v >> 6) << 2) (kept from being rewritten as shift, mask, add)
    uint32_t a;
    asm("v_lshrrev_b32 %0, 6, %1\n\tv_lshl_add_u32 %0, %0, 2, %1" : "=&v"(a) : "v"(v));
    return a;
}
// The hit histogram in front of the stages: 16-bit counters, two per word. The colours of rounds 2p and 2p + 1 share the
// words [2048 p, 2048 p + 2048): low half = even round, high half = odd round. Within a round the increment is therefore
// the same for all lanes and the counter's byte address is 4 * (stage entry) (+ 8192 p).
__host__ __device__ __forceinline__ uint32_t k2b_hist_words(uint32_t W) {
    const uint32_t rounds = (W + 63) / 64;
    return (rounds / 2) * 2048 + ((rounds & 1u) ? (W - 64 * (rounds - 1)) * 32 : 0u);
}
// bytes in front of the stages (a multiple of 1088 = 1024 * 17 / 16) and the virtual address that lands there
__host__ __device__ __forceinline__ uint32_t k2b_hist_region(uint32_t W) { return (k2b_hist_words(W) * 4 + 1087) / 1088 * 1088; }
// Inside its group of G words (2048, or what the last round of an odd number of rounds needs) the counter of entry e lives at
// word (e % 4) * (G / 4) + e / 4: a lane copies four consecutive entries out (one 16-byte store), in dense rows those are
// four consecutive colours, and the four ds_add of a wave then go to consecutive words each — with the plain layout 64 lanes
// hit 8 banks (57 % of the kernel's LDS cycles were bank conflicts on the threshold union: profiles/r3). G is a multiple of 128.
__host__ __device__ __forceinline__ uint32_t k2b_group_words(uint32_t W, uint32_t group) {
    const uint32_t rounds = (W + 63) / 64;
    return ((rounds & 1u) && group == rounds / 2) ? (W - 64 * (rounds - 1)) * 32 : 2048u;
}
__device__ __forceinline__ uint32_t k2b_hist_addr(uint32_t group_at, uint32_t G, uint32_t e) {  // byte address; group_at = 8192 * group
    return group_at + __umul24(e & 3u, G) + (e & ~3u);
}
#ifndef FG_K2B_COPY_STEPS
#define FG_K2B_COPY_STEPS 2
#endif
constexpr uint32_t K2B_COPY_STEPS = FG_K2B_COPY_STEPS;  // copy-out steps whose stage reads are in flight together (1 / 2 / 4: 6.04 / 5.97 / 5.81 ms on one box; 4 needs 70 VGPRs without the small-result rewrite)
#ifndef FG_K2B_KO  // knock-out builds (profiles/k2b_knockout.sh): 1 = no hit-counter adds, 2 = no stage scatter stores, 3 = no colour stores of the bitmap rows; results are wrong, times and LDS counters are the point
#define FG_K2B_KO 0
#endif
#define K2B_HIST_ADD(a, v) do { if (FG_K2B_KO != 1) lds_add((a), (v)); } while (0)
// (no waves-per-SIMD hint: with __launch_bounds__(K2B_THREADS, 8) the compiler schedules for registers and the same source runs at 6.8 instead
// of 6.1 ms; without it the kernel happens to need 64 VGPRs and no scratch: 8 waves per SIMD)
__device__ __forceinline__ void k2b_expand_body(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ counts,
                                                  const uint64_t* __restrict__ out_off, uint64_t n_reads, uint32_t W,
                                                  uint32_t* __restrict__ colors, unsigned int* tickets,
                                                  uint32_t* __restrict__ hit_partial, const uint64_t* __restrict__ totals,
                                                  uint64_t capacity, uint32_t block_cap, const uint32_t* __restrict__ small) {
    // small != nullptr: reads with 1..SMALL_RESULT colours have them in their slot of `small` and no bitmap row
    // launched behind the scan without a host round trip: if the colours of the pass do not fit `colors`
    // (capacity in u32), do nothing — the host enlarges the buffer and launches again
    if (totals[0] > capacity) return;
    // hit_partial != nullptr: also count, per colour, the reads of this launch that contain it. Each block
    // keeps 16-bit counters in LDS and stores them as one row of hit_partial[gridDim.x][W*32] at the end;
    // k_hits_reduce sums the rows. A block stops pulling tickets once it has taken block_cap (< 65536) reads, whatever
    // the other blocks do (tickets are dynamic: blocks that start late, or share the GPU with another stream, leave
    // more reads to the early ones): no counter can wrap. The host sizes the grid so that the caps add up to more
    // than the reads of every partition.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    if (lds_addr(smem_x) != 0u) __builtin_trap();  // (no static LDS in this kernel: byte addresses count from the dynamic block)
    uint32_t* hist = (uint32_t*)smem_x;  // k2b_hist_words(W) words at LDS byte 0, the stages behind them
    const int lane = lane_id();
    const uint32_t hist_words = hit_partial ? k2b_hist_words(W) : 0u;
    // virtual address of this wave's slot 0
    const uint32_t v_wave = (hit_partial ? k2b_hist_region(W) / 1088u * 1024u : 0u) + 4096u * (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // reads this block has taken tickets for: one word behind the stages (no static LDS: the dynamic block starts at LDS byte 0)
    uint32_t& s_taken = *(uint32_t*)(smem_x + (hit_partial ? k2b_hist_region(W) : 0u) + (K2B_THREADS / 64) * K2B_STAGE_BYTES);
    if (hit_partial) {
        for (uint32_t i = threadIdx.x; i < hist_words; i += blockDim.x) hist[i] = 0;
        if (threadIdx.x == 0) s_taken = 0;
        __syncthreads();
    }
    const WorkQueue wq{tickets, n_reads, 32, FG_K2B_PARTS};
    uint64_t t_first;
    uint32_t t_count;
    auto may_pull = [&]() -> bool {
        if (!hit_partial) return true;
        uint32_t before = 0;
        if (lane == 0) before = atomicAdd(&s_taken, 32u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)before) + 32u <= block_cap;
    };
    // the first three 64-word rounds of a read's bitmap (all of it up to 6144 colours) are requested one read
    // ahead, its size and output offset once per ticket: the per-read fetch chain is off the critical path
    auto fetch3 = [&](uint64_t r, uint32_t (&x)[3]) {
        const uint32_t* bm = bitmap + r * W;
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q) x[q] = q * 64 + lane < W ? __builtin_nontemporal_load(&bm[q * 64 + lane]) : 0u;  // read once
    };
    while (may_pull() && wq.pull(t_first, t_count)) {
    const uint64_t rl = t_first + min((uint32_t)lane, t_count - 1);
    const uint32_t cnt_raw = counts[rl];  // (both loads unconditional and in one block: one latency, not two)
    const uint64_t off_raw = out_off[rl];
    const uint32_t cnt_l = (uint32_t)lane < t_count ? cnt_raw : 0u;
    // a ticket is one contiguous piece of the CSR: read j of it begins at P0 + (colours of the ticket's reads in front of it)
    const uint64_t P0 = readlane_u64(off_raw, 0);
    const uint32_t excl_l = wave_incl_scan_u32(cnt_l) - cnt_l;
    uint64_t lg = __ballot(cnt_l != 0u), sm = 0;  // the reads whose result comes as a bitmap row; as colours
    bool is_small = false;
    if (small) {
        is_small = cnt_l != 0u && cnt_l <= SMALL_RESULT;
        sm = __ballot(is_small);
        lg = __ballot(cnt_l > SMALL_RESULT);
    }
    uint32_t cur[3], nxt[3] = {0u, 0u, 0u};
    if (lg) fetch3(t_first + (uint32_t)__builtin_ctzll(lg), nxt);  // (in flight across the small results)
    if (sm) {
        // Results of at most SMALL_RESULT colours arrive as colours (k2r_intersect / k2a_intersect): lane = (read, slot), four reads per
        // step. The k-th small read of the ticket leaves two words in LDS {colours | read << 5, offset in the ticket}; then ALL the loads of
        // the ticket (at most 32 small reads = 8 steps; unconditional, in one block: a step past the last small read loads the
        // last one's slot again), ONE wait, and the stores behind it: a wait between two stores would be a wait for the store.
        if (is_small) {
            lds_u32* e = lds32(k2b_stage_skew(v_wave + (mask_rank(sm) << 3)));
            e[0] = cnt_l | ((uint32_t)lane << 5);
            e[1] = excl_l;
        }
        wave_lds_sync();
        const uint32_t nsm = (uint32_t)__popcll(sm), slot = (uint32_t)lane & 15u;
        uint32_t info[8], at[8], v[8];
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) {
            const lds_u32* e = lds32(k2b_stage_skew(v_wave + (min(4 * g + ((uint32_t)lane >> 4), nsm - 1) << 3)));
            info[g] = e[0];
            at[g] = e[1];
        }
        wave_lds_sync();
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) v[g] = __builtin_nontemporal_load(&small[(t_first + (info[g] >> 5)) * SMALL_RESULT + slot]);
        // (the values pass through the statement: the compiler waits for the loads here and knows of no load behind it)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]));
#pragma unroll
        for (uint32_t g = 0; g < 8; ++g) {
            if (4 * g + ((uint32_t)lane >> 4) < nsm && slot < (info[g] & 31u)) {
                const uint32_t c = v[g];
                colors[P0 + at[g] + slot] = c;
                // the colour's hit counter: round c >> 11 (low / high half of the word by its parity), entry c & 2047
                if (hit_partial) K2B_HIST_ADD(k2b_hist_addr((c >> 12) * 8192u, k2b_group_words(W, c >> 12), c & 2047u), (c & 2048u) ? 0x10000u : 1u);
            }
        }
        wave_lds_sync();
    }
    while (lg) {
        const uint32_t j = (uint32_t)__builtin_ctzll(lg);
        lg &= lg - 1;
        const uint64_t r = t_first + j;
#pragma unroll
        for (uint32_t q = 0; q < 3; ++q) cur[q] = nxt[q];
        if (lg) fetch3(t_first + (uint32_t)__builtin_ctzll(lg), nxt);
        uint32_t* out = colors + P0 + (uint32_t)__builtin_amdgcn_readlane((int)excl_l, (int)j);
        const uint32_t* bm = bitmap + r * W;
        // one round = 64 words. The first three rounds have their words in registers and contain no loads: a load would make
        // the wave wait for ALL its outstanding memory operations (one in-order counter), i.e. for the stores of the round
        // before to reach L2, round after round. (Doing the same for the one wait per read that is left, with the requests
        // issued and counted by hand so that the stores stay in flight, changed nothing: measured, profiles/r2.)
        auto round = [&](const uint32_t w0, uint32_t x) {
            const uint32_t pc = __popc(x);
            const uint32_t incl = wave_incl_scan_u32(pc);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0) return;
            uint32_t va = v_wave + ((incl - pc) << 1);
            const uint32_t rel = (uint32_t)lane * 32;
            while (x) {
                if (FG_K2B_KO != 2) *lds16(k2b_stage_skew(va)) = (uint16_t)(rel | (uint32_t)__builtin_ctz(x));
                va += 2;
                x &= x - 1;
            }
            wave_lds_sync();
            const uint32_t cbase = w0 * 32;
            // hit counters of this round: which half of the word, which 2048-word group
            const uint32_t hinc = (w0 & 64u) ? 0x10000u : 1u;
            const uint32_t hoff = (w0 >> 7) * 8192u, hG = k2b_group_words(W, w0 >> 7);
            // copy-out: every lane takes 4 consecutive slots (two LDS words) and stores 4 colours at once; the last
            // total % 4 slots (all of them in a round of at most 64) go out one per lane
            const uint32_t full = total <= 64 ? 0u : total & ~3u;  // short rounds: one slot per lane, one pass
            // (K2B_COPY_STEPS steps of 256 slots at a time: their stage reads are requested together, one LDS latency per group, and the
            // counter adds — which the next stage read would wait for as well — come behind all of them)
            for (uint32_t i0 = (uint32_t)lane * 4; i0 < full; i0 += 256 * K2B_COPY_STEPS) {
                uint32_t e01[K2B_COPY_STEPS], e23[K2B_COPY_STEPS];
#pragma unroll
                for (uint32_t u = 0; u < K2B_COPY_STEPS; ++u) {
                    const uint32_t i = i0 + 256 * u;
                    const lds_u32* sp = lds32(k2b_stage_skew(v_wave + (min(i, 2044u) << 1)));  // (unconditional: a step past `full` reads what it does not use)
                    e01[u] = sp[0];
                    e23[u] = sp[1];
                }
#pragma unroll
                for (uint32_t u = 0; u < K2B_COPY_STEPS; ++u) {
                    const uint32_t i = i0 + 256 * u;
                    if (i < full) {
                        const uint32_t e0 = e01[u] & 0xFFFFu, e1 = e01[u] >> 16, e2 = e23[u] & 0xFFFFu, e3 = e23[u] >> 16;
                        if (FG_K2B_KO != 3 || e0 == 0x12345u) *(u32x4_a4*)(out + i) = u32x4{cbase + e0, cbase + e1, cbase + e2, cbase + e3};
                    }
                }
                if (hit_partial) {
#pragma unroll
                    for (uint32_t u = 0; u < K2B_COPY_STEPS; ++u) {
                        const uint32_t i = i0 + 256 * u;
                        if (i < full) {
                            const uint32_t e0 = e01[u] & 0xFFFFu, e1 = e01[u] >> 16, e2 = e23[u] & 0xFFFFu, e3 = e23[u] >> 16;
                            K2B_HIST_ADD(k2b_hist_addr(hoff, hG, e0), hinc);
                            K2B_HIST_ADD(k2b_hist_addr(hoff, hG, e1), hinc);
                            K2B_HIST_ADD(k2b_hist_addr(hoff, hG, e2), hinc);
                            K2B_HIST_ADD(k2b_hist_addr(hoff, hG, e3), hinc);
                        }
                    }
                }
            }
            if ((uint32_t)lane < total - full) {
                const uint32_t i = full + lane;
                const uint32_t e = *lds16(k2b_stage_skew(v_wave + (i << 1)));
                if (FG_K2B_KO != 3 || e == 0x12345u) out[i] = cbase + e;
                if (hit_partial) K2B_HIST_ADD(k2b_hist_addr(hoff, hG, e), hinc);
            }
            out += total;
            wave_lds_sync();
        };
        round(0, cur[0]);
        if (W > 64) round(64, cur[1]);
        if (W > 128) round(128, cur[2]);
        for (uint32_t w0 = 192; w0 < W; w0 += 64) round(w0, w0 + lane < W ? bm[w0 + lane] : 0u);
    }
    }
    if (hit_partial) {
        __syncthreads();
        uint32_t* row = hit_partial + (uint64_t)blockIdx.x * W * 32;
        const uint32_t ncol = W * 32;
        for (uint32_t i = threadIdx.x; i < hist_words; i += blockDim.x) {
            const uint32_t v = hist[i];
            const uint32_t q4 = k2b_group_words(W, i >> 11) >> 2, wi = i & 2047u;  // word wi of its group = entry (wi % q4) * 4 + wi / q4
            const uint32_t c = (i >> 11) * 4096u + (wi % q4) * 4u + wi / q4;  // colour of the low half; the high half is 2048 further
            row[c] = v & 0xFFFFu;
            if (c + 2048u < ncol) row[c + 2048u] = v >> 16;
        }
    }
}
#define FG_K2B_ARGS const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ counts, const uint64_t* __restrict__ out_off, uint64_t n_reads, uint32_t W, \
                    uint32_t* __restrict__ colors, unsigned int* tickets, uint32_t* __restrict__ hit_partial, const uint64_t* __restrict__ totals, \
                    uint64_t capacity, uint32_t block_cap, const uint32_t* __restrict__ small
__global__ __launch_bounds__(K2B_THREADS) void k2b_expand(FG_K2B_ARGS) {
    k2b_expand_body(bitmap, counts, out_off, n_reads, W, colors, tickets, hit_partial, totals, capacity, block_cap, small);
}
// the same kernel under another name: the runs that time candidate allocations of the colour lists (stage_expand's lottery) stay out of
// k2b_expand's line in kernel traces — the average of that line is the average of the passes
__global__ __launch_bounds__(K2B_THREADS) void k_probe_allocation(FG_K2B_ARGS) {
    k2b_expand_body(bitmap, counts, out_off, n_reads, W, colors, tickets, hit_partial, totals, capacity, block_cap, small);
}
#undef FG_K2B_ARGS

// ---------------------------------------------------------------------------------------------
// per-colour hit counts: hits[c] += #reads of the batch whose result contains c.
// Stage 1: each block sums its slice of reads into partial[block][W*32] (thread = one 32-colour word,
// 32 register counters, 4 reads in flight). Stage 2: column sums of the partials into the u64 totals.
// ---------------------------------------------------------------------------------------------
__global__ void k_hits(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ counts, uint64_t n_reads, uint32_t W,
                       uint32_t* __restrict__ partial, const uint32_t* __restrict__ small) {
    // small != nullptr: results of at most SMALL_RESULT colours (the empty ones included) left no row behind: k_hits_small counts those
    const uint64_t per_block = (n_reads + gridDim.x - 1) / gridDim.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * per_block, r1 = min(n_reads, r0 + per_block);
    const uint32_t floor_ = small ? SMALL_RESULT : 0u;  // a row exists when the result has more colours than this
    for (uint32_t w = threadIdx.x; w < W; w += blockDim.x) {
        uint32_t acc[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) acc[b] = 0;
        uint64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            const uint32_t x0 = counts[r] > floor_ ? bitmap[r * W + w] : 0u, x1 = counts[r + 1] > floor_ ? bitmap[(r + 1) * W + w] : 0u,
                           x2 = counts[r + 2] > floor_ ? bitmap[(r + 2) * W + w] : 0u, x3 = counts[r + 3] > floor_ ? bitmap[(r + 3) * W + w] : 0u;
#pragma unroll
            for (int b = 0; b < 32; ++b) acc[b] += ((x0 >> b) & 1u) + ((x1 >> b) & 1u) + ((x2 >> b) & 1u) + ((x3 >> b) & 1u);
        }
        for (; r < r1; ++r) {
            const uint32_t x = counts[r] > floor_ ? bitmap[r * W + w] : 0u;
#pragma unroll
            for (int b = 0; b < 32; ++b) acc[b] += (x >> b) & 1u;
        }
        uint32_t* dst = partial + ((uint64_t)blockIdx.x * W + w) * 32;
#pragma unroll
        for (int b = 0; b < 32; ++b) dst[b] = acc[b];
    }
}

// the results that travel as colours (1..SMALL_RESULT of them in the read's slot): thread = (read, slot), one atomic per colour
__global__ __launch_bounds__(256) void k_hits_small(const uint32_t* __restrict__ counts, const uint32_t* __restrict__ small, uint64_t n_reads,
                                                    unsigned long long* __restrict__ hits) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads * SMALL_RESULT; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / SMALL_RESULT;
        const uint32_t slot = (uint32_t)(i % SMALL_RESULT), cnt = counts[r];
        if (cnt <= SMALL_RESULT && slot < cnt) atomicAdd(&hits[small[i]], 1ull);
    }
}

// grid = (ceil(n/256), HITS_ROW_GROUPS): each block sums every HITS_ROW_GROUPS-th partial row for 256
// columns, then one u64 atomic per column
constexpr uint32_t HITS_ROW_GROUPS = 32;
__global__ void k_hits_reduce(const uint32_t* __restrict__ partial, uint32_t nblocks, uint32_t W, uint32_t n,
                              unsigned long long* __restrict__ hits) {
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    const uint64_t row = (uint64_t)W * 32;
    unsigned long long s = 0;
    uint32_t b = blockIdx.y;
    for (; b + 3 * HITS_ROW_GROUPS < nblocks; b += 4 * HITS_ROW_GROUPS) {
        const uint32_t x0 = partial[b * row + col], x1 = partial[(b + HITS_ROW_GROUPS) * row + col],
                       x2 = partial[(b + 2 * HITS_ROW_GROUPS) * row + col], x3 = partial[(b + 3 * HITS_ROW_GROUPS) * row + col];
        s += (unsigned long long)x0 + x1 + x2 + x3;
    }
    for (; b < nblocks; b += HITS_ROW_GROUPS) s += partial[b * row + col];
    if (s) atomicAdd(&hits[col], s);
}

__global__ void k_add_totals(unsigned long long* hits, uint32_t n, uint64_t num_reads, const uint64_t* totals) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        atomicAdd(&hits[n], (unsigned long long)num_reads);  // several streams may add to the same vector
        atomicAdd(&hits[n + 1], (unsigned long long)totals[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// Device-side output formatters (src/ps_utils.cpp:48-135): the CSR result of a pass as the bytes the reference
// writes — ascii "<id>\t<count>[\t<colour>...]\n" (psa_ascii_formatter, digits as util::vec_to_tsv) or binary
// u32 id, u32 count, u32 x count (psa_binary_formatter). One wave per read; record sizes first, an exclusive scan
// gives the byte offsets, then every lane writes the text of its colour at its prefix position.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dec_digits(uint32_t x) {
    return 1u + (x >= 10u) + (x >= 100u) + (x >= 1000u) + (x >= 10000u) + (x >= 100000u) + (x >= 1000000u) +
           (x >= 10000000u) + (x >= 100000000u) + (x >= 1000000000u);
}
// writes x in decimal, most significant digit first, at p; returns the digit count
__device__ __forceinline__ uint32_t put_dec(unsigned char* p, uint32_t x) {
    const uint32_t nd = dec_digits(x);
    for (uint32_t i = nd; i-- > 0;) {
        p[i] = (unsigned char)('0' + x % 10u);
        x /= 10u;
    }
    return nd;
}

__global__ __launch_bounds__(256) void k_fmt_ascii_sizes(const uint64_t* __restrict__ off, const uint32_t* __restrict__ colors,
                                                         uint64_t n_reads, uint32_t first_id, uint32_t* __restrict__ sizes) {
    const int lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint64_t o0 = off[r], o1 = off[r + 1];
        uint32_t b = 0;
        for (uint64_t i = o0 + lane; i < o1; i += 64) b += 1u + dec_digits(colors[i]);
        b = wave_sum_u32(b);
        if (lane == 0) sizes[r] = b + dec_digits(first_id + (uint32_t)r) + 1u + dec_digits((uint32_t)(o1 - o0)) + 1u;
    }
}

__global__ __launch_bounds__(256) void k_fmt_ascii_write(const uint64_t* __restrict__ off, const uint32_t* __restrict__ colors,
                                                         uint64_t n_reads, uint32_t first_id, const uint64_t* __restrict__ byte_off,
                                                         unsigned char* __restrict__ out) {
    const int lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint64_t o0 = off[r], o1 = off[r + 1];
        unsigned char* p = out + byte_off[r];
        const uint32_t id = first_id + (uint32_t)r, cnt = (uint32_t)(o1 - o0);
        const uint32_t hdr = dec_digits(id) + 1u + dec_digits(cnt);
        if (lane == 0) {
            const uint32_t a = put_dec(p, id);
            p[a] = '\t';
            put_dec(p + a + 1, cnt);
        }
        uint64_t at = hdr;  // bytes of the record already placed
        for (uint64_t i0 = o0; i0 < o1; i0 += 64) {
            const bool have = i0 + lane < o1;
            const uint32_t c = have ? colors[i0 + lane] : 0u;
            const uint32_t len = have ? 1u + dec_digits(c) : 0u;
            const uint32_t incl = wave_incl_scan_u32(len);
            if (have) {
                unsigned char* q = p + at + (incl - len);
                q[0] = '\t';
                put_dec(q + 1, c);
            }
            at += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) p[at] = '\n';
    }
}

__global__ __launch_bounds__(256) void k_fmt_binary_write(const uint64_t* __restrict__ off, const uint32_t* __restrict__ colors,
                                                          uint64_t n_reads, uint32_t first_id, uint32_t* __restrict__ out) {
    const int lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint64_t o0 = off[r], o1 = off[r + 1];
        uint32_t* p = out + 2 * r + o0;  // every earlier record: 2 header words + its colours
        if (lane == 0) { p[0] = first_id + (uint32_t)r; p[1] = (uint32_t)(o1 - o0); }
        for (uint64_t i = o0 + lane; i < o1; i += 64) p[2 + (i - o0)] = colors[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Device-side compressed formatter (psa_compressed_formatter, src/ps_utils.cpp:158-239): blocks of
// `u64 num_bits` + words; a record is delta(id) delta(count) followed by delta-gaps (count < 0.25 n), the raw
// n-bit bitmap (count < 0.75 n) or the delta-gaps of the missing colours. It works from the RESULT BITMAPS of the
// pass (the colour lists are not needed): sizes per read -> bit offsets inside blocks of CFMT_BLOCK_READS
// records -> byte offsets of the blocks -> every code ORed into the zeroed output at its bit position. Where the
// reference cuts blocks depends on its worker threads' buffers; any cut is the same format.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t CFMT_BLOCK_READS = 256;

__device__ __forceinline__ uint64_t delta_code(uint64_t x, uint32_t& len) {  // x < 2^32; LSB-first code of at most 43 bits
    const uint64_t y = x + 1;
    const uint32_t b = 63u - (uint32_t)__builtin_clzll(y);
    const uint32_t yb = b + 1, c = 31u - (uint32_t)__builtin_clz(yb);
    len = 2 * c + 1 + b;
    return (1ull << c) | ((uint64_t)(yb & ((1u << c) - 1u)) << (c + 1)) | ((y & ((1ull << b) - 1ull)) << (2 * c + 1));
}
__device__ __forceinline__ void put_bits(unsigned long long* out, uint64_t bitpos, uint64_t code, uint32_t len) {
    const uint32_t sh = (uint32_t)bitpos & 63u;
    atomicOr(&out[bitpos >> 6], (unsigned long long)(code << sh));
    if (sh + len > 64) atomicOr(&out[(bitpos >> 6) + 1], (unsigned long long)(code >> (64 - sh)));
}
// inclusive running maximum over the lanes
__device__ __forceinline__ int wave_incl_max_i32(int v) {
    const int lane = lane_id();
    int x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x = max(x, y);
    }
    return x;
}

// payload of the gap-coded kinds: walks the set bits of `word(w)` (the result row, complemented for the dense kind)
// in colour order; emit(gap, bit offset of its code inside the payload) when WRITE, and returns the payload bits
template <bool WRITE, typename WordFn, typename Emit>
__device__ __forceinline__ uint32_t cfmt_gap_payload(uint32_t W, int lane, WordFn word, Emit emit) {
    uint32_t total = 0;
    int carry_last = -1;  // position of the last set bit of earlier rounds
    for (uint32_t w0 = 0; w0 < W; w0 += 64) {
        const uint32_t w = w0 + lane;
        const uint32_t x = w < W ? word(w) : 0u;
        const int last = x ? (int)(w * 32 + 31 - __builtin_clz(x)) : -1;
        const int run = wave_incl_max_i32(last);  // last set bit up to and including this lane's word
        const int before = __shfl_up(run, 1);
        const int prev0 = max(carry_last, lane == 0 ? -1 : before);
        uint32_t mine = 0;
        {
            int prev = prev0;
            for (uint32_t y = x; y; y &= y - 1) {
                const int p = (int)(w * 32 + __builtin_ctz(y));
                uint32_t len;
                (void)delta_code((uint64_t)(p - prev - 1), len);
                mine += len;
                prev = p;
            }
        }
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (WRITE) {
            uint32_t at = total + incl - mine;
            int prev = prev0;
            for (uint32_t y = x; y; y &= y - 1) {
                const int p = (int)(w * 32 + __builtin_ctz(y));
                uint32_t len;
                (void)delta_code((uint64_t)(p - prev - 1), len);
                emit((uint32_t)(p - prev - 1), at);
                at += len;
                prev = p;
            }
        }
        total += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        carry_last = max(carry_last, __builtin_amdgcn_readlane(run, 63));
    }
    return total;
}

// the same payload for a result of at most 64 colours given as colours (results without a bitmap row: SMALL_RESULT)
template <bool WRITE, typename Emit>
__device__ __forceinline__ uint32_t cfmt_small_payload(const uint32_t* cols, uint32_t size, int lane, Emit emit) {
    const uint32_t c = (uint32_t)lane < size ? cols[lane] : 0u;
    const uint32_t before = (uint32_t)__shfl_up((int)c, 1);
    uint32_t gap = 0, len = 0;
    if ((uint32_t)lane < size) {
        gap = lane == 0 ? c : c - before - 1u;
        (void)delta_code(gap, len);
    }
    const uint32_t incl = wave_incl_scan_u32(len);
    if (WRITE && (uint32_t)lane < size) emit(gap, incl - len);
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}

__device__ __forceinline__ uint32_t cfmt_word(const uint32_t* row, uint32_t w, uint32_t n, bool complemented) {
    uint32_t x = row[w];
    if (complemented) {
        x = ~x;
        const uint32_t lo = w * 32;
        x &= lo >= n ? 0u : (n - lo >= 32 ? 0xFFFFFFFFu : ((1u << (n - lo)) - 1u));
    }
    return x;
}

__global__ __launch_bounds__(256) void k_cfmt_sizes(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ counts,
                                                    uint64_t n_reads, uint32_t W, uint32_t n, uint32_t sparse_thr, uint32_t dense_thr,
                                                    uint32_t first_id, uint32_t* __restrict__ bits,
                                                    const uint32_t* __restrict__ small) {
    // small != nullptr: results of at most SMALL_RESULT colours have no bitmap row; their colours are in their slot of `small`
    // (the u32 colour lists of the pass are not needed: k2b_expand runs only for consumers of the CSR)
    const int lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint32_t size = counts[r];
        uint32_t l1, l2;
        (void)delta_code((uint64_t)first_id + r, l1);
        (void)delta_code(size, l2);
        uint32_t payload = 0;
        if (size == 0) payload = 0;
        else if (small && size <= SMALL_RESULT) payload = cfmt_small_payload<false>(small + r * SMALL_RESULT, size, lane, [](uint32_t, uint32_t) {});
        else if (size >= sparse_thr && size < dense_thr) payload = n;
        else {
            const uint32_t* row = bitmap + r * W;
            const bool comp = size >= dense_thr;
            payload = cfmt_gap_payload<false>(W, lane, [&](uint32_t w) { return cfmt_word(row, w, n, comp); }, [](uint32_t, uint32_t) {});
        }
        if (lane == 0) bits[r] = l1 + l2 + payload;
    }
}

// one thread block per format block: bit offset of every record inside its block, bytes of the block (header + words)
__global__ __launch_bounds__(CFMT_BLOCK_READS) void k_cfmt_blocks(const uint32_t* __restrict__ bits, uint64_t n_reads,
                                                                   uint32_t* __restrict__ rec_off, uint32_t* __restrict__ block_bits,
                                                                   uint32_t* __restrict__ block_bytes) {
    __shared__ uint32_t wsum[CFMT_BLOCK_READS / 64];
    const uint64_t r = (uint64_t)blockIdx.x * CFMT_BLOCK_READS + threadIdx.x;
    const uint32_t b = r < n_reads ? bits[r] : 0u;
    const uint32_t incl = wave_incl_scan_u32(b);
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < (threadIdx.x >> 6); ++i) base += wsum[i];
    if (r < n_reads) rec_off[r] = base + incl - b;
    if (threadIdx.x == CFMT_BLOCK_READS - 1) {
        const uint32_t tot = base + incl;
        block_bits[blockIdx.x] = tot;
        block_bytes[blockIdx.x] = 8u + 8u * ((tot + 63u) >> 6);
    }
}

// The record of a read is assembled in LDS (64-bit ds_or at the record's own bit alignment) and flushed with plain
// stores; only its first and last word, which it may share with its neighbours, are ORed into the zeroed output.
// cap_words = LDS words per wave; a record that does not fit (never for valid thresholds: at most ~1.25 n bits) is
// ORed into the output code by code.
__global__ __launch_bounds__(256) void k_cfmt_write(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ counts,
                                                    uint64_t n_reads, uint32_t W, uint32_t n, uint32_t sparse_thr, uint32_t dense_thr,
                                                    uint32_t first_id, const uint32_t* __restrict__ rec_bits,
                                                    const uint32_t* __restrict__ rec_off,
                                                    const uint32_t* __restrict__ block_bits, const uint64_t* __restrict__ block_off,
                                                    unsigned long long* __restrict__ out, uint32_t cap_words,
                                                    const uint32_t* __restrict__ small) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    const int lane = lane_id();
    unsigned long long* buf = (unsigned long long*)smem_c + (size_t)(threadIdx.x >> 6) * cap_words;
    for (uint32_t i = lane; i < cap_words; i += 64) buf[i] = 0;
    wave_lds_sync();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint64_t blk = r / CFMT_BLOCK_READS;
        const uint64_t hdr_word = block_off[blk] >> 3;  // blocks start on 8-byte boundaries
        if (r % CFMT_BLOCK_READS == 0 && lane == 0) out[hdr_word] = block_bits[blk];
        const uint64_t pos0 = (hdr_word + 1) * 64 + rec_off[r];
        const uint64_t base_bit = pos0 & ~63ull;
        const uint32_t nwords = (uint32_t)((pos0 - base_bit) + rec_bits[r] + 63) >> 6;
        const bool staged = nwords <= cap_words;
        auto put = [&](uint64_t pos, uint64_t code, uint32_t len) {
            if (staged) {
                const uint32_t lp = (uint32_t)(pos - base_bit), sh = lp & 63u;
                atomicOr(&buf[lp >> 6], (unsigned long long)(code << sh));
                if (sh + len > 64) atomicOr(&buf[(lp >> 6) + 1], (unsigned long long)(code >> (64 - sh)));
            } else {
                put_bits(out, pos, code, len);
            }
        };
        const uint32_t size = counts[r];
        uint32_t l1, l2;
        const uint64_t c1 = delta_code((uint64_t)first_id + r, l1), c2 = delta_code(size, l2);
        if (lane == 0) {
            put(pos0, c1, l1);
            put(pos0 + l1, c2, l2);
        }
        const uint64_t pay = pos0 + l1 + l2;
        const uint32_t* row = bitmap + r * W;
        if (size == 0) {
        } else if (small && size <= SMALL_RESULT) {  // (as in k_cfmt_sizes)
            cfmt_small_payload<true>(small + r * SMALL_RESULT, size, lane, [&](uint32_t gap, uint32_t at) {
                uint32_t len;
                const uint64_t code = delta_code(gap, len);
                put(pay + at, code, len);
            });
        } else if (size >= sparse_thr && size < dense_thr) {  // the n bits of the row, shifted into place
            for (uint32_t w = lane; w * 32 < n; w += 64) {
                uint32_t x = row[w];
                const uint32_t lo = w * 32, nb = n - lo >= 32 ? 32u : n - lo;
                if (nb < 32) x &= (1u << nb) - 1u;
                if (x) put(pay + lo, x, nb);
            }
        } else {
            const bool comp = size >= dense_thr;
            cfmt_gap_payload<true>(W, lane, [&](uint32_t w) { return cfmt_word(row, w, n, comp); },
                                   [&](uint32_t gap, uint32_t at) {
                                       uint32_t len;
                                       const uint64_t code = delta_code(gap, len);
                                       put(pay + at, code, len);
                                   });
        }
        if (staged) {
            wave_lds_sync();
            unsigned long long* dst = out + (base_bit >> 6);
            for (uint32_t w = lane; w < nwords; w += 64) {
                const unsigned long long v = buf[w];
                buf[w] = 0;
                if (w == 0 || w + 1 == nwords) { if (v) atomicOr(&dst[w], v); }
                else dst[w] = v;
            }
            wave_lds_sync();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// A batch of the streaming worker loop is uploaded range by range as the reader's threads parsed it: the offsets of a range
// count from the range's first base. This adds the position of the range inside the batch (up to REBASE_RANGES ranges per launch).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t REBASE_RANGES = 64;
struct RebaseTable {
    uint64_t first_read[REBASE_RANGES + 1];  // reads [first_read[c], first_read[c + 1]) belong to range c
    uint64_t base[REBASE_RANGES];            // position of the range's first base in the batch
    uint32_t count;
};
__global__ __launch_bounds__(256) void k_offs_rebase(uint64_t* __restrict__ offs, RebaseTable t) {
    foreign_writes_acquire();
    // offs[r + 1] = end of read r
    const uint64_t r0 = t.first_read[0], r1 = t.first_read[t.count];
    for (uint64_t r = r0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < r1; r += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c = 0;
        while (c + 1 < t.count && r >= t.first_read[c + 1]) ++c;
        offs[r + 1] += t.base[c];
    }
}

// ---------------------------------------------------------------------------------------------
// algorithmic bytes of the colour-intersection stage:
//   sum over reads of  sum_c ceil(list bits / 8) + 16|C| + 4|C| + 4|R| + 8
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_account(DevColors c, const uint32_t* __restrict__ nids, const uint64_t* __restrict__ idoff,
                                                 const uint32_t* __restrict__ ids_pool, const uint32_t* __restrict__ counts,
                                                 uint64_t n_reads, unsigned long long* __restrict__ out,
                                                 const uint32_t* __restrict__ set_bytes) {
    uint64_t in_bytes = 0, out_bytes = 0;  // out[0]: list side (lists + offsets + ids), out[1]: result side
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t off = idoff[r];
        const uint32_t cnt = nids[r];
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t id = ids_pool[off + i];
            // hybrid: the list + two 8-byte offsets; other codecs: every list the set touches (+16 each)
            in_bytes += set_bytes ? (uint64_t)set_bytes[id] : (c.offsets[id + 1] - c.offsets[id] + 7) / 8 + 16;
        }
        in_bytes += 4ull * cnt;
        out_bytes += 4ull * counts[r] + 8;
    }
    for (int o = 32; o; o >>= 1) { in_bytes += __shfl_xor(in_bytes, o); out_bytes += __shfl_xor(out_bytes, o); }
    if (lane_id() == 0) {
        if (in_bytes) atomicAdd(out, (unsigned long long)in_bytes);
        if (out_bytes) atomicAdd(out + 1, (unsigned long long)out_bytes);
    }
}

// ---------------------------------------------------------------------------------------------
// --deduplicate on the device (tools/pseudoalign.cpp:91-226, src/ps_utils.cpp:307-415: the reference writes the colour-set ids of
// every read to a temporary file, sorts the records by id list and intersects every distinct list once). Here, per pass: a 64-bit
// hash of every read's id list, a radix sort of (hash, read), neighbours of the order compared EXACTLY (a read opens a group unless
// its list equals its predecessor's: two lists that collide in the hash are two groups, at worst a list is intersected twice), the
// intersection kernel runs over one list per group, and every read takes the row / colours / size of its group.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dd_hash(const uint32_t* __restrict__ nids, const uint64_t* __restrict__ idoff, const uint32_t* __restrict__ ids_pool,
                                                 uint64_t n_reads, unsigned long long* __restrict__ hash, uint32_t* __restrict__ idx) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t cnt = nids[r];
        const uint32_t* ids = ids_pool + idoff[r];
        uint64_t h = 0x9E3779B97F4A7C15ull ^ cnt;
        for (uint32_t i = 0; i < cnt; ++i) {
            h = (h ^ ids[i]) * 0xff51afd7ed558ccdULL;
            h ^= h >> 32;
        }
        hash[r] = h;
        idx[r] = (uint32_t)r;
    }
}
// head[i] = 1 when the read at position i of the order opens a group
__global__ __launch_bounds__(256) void k_dd_heads(const unsigned long long* __restrict__ hash_sorted, const uint32_t* __restrict__ idx_sorted,
                                                  const uint32_t* __restrict__ nids, const uint64_t* __restrict__ idoff, const uint32_t* __restrict__ ids_pool,
                                                  uint64_t n_reads, uint32_t* __restrict__ head) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t is_head = 1;
        if (i && hash_sorted[i] == hash_sorted[i - 1]) {
            const uint32_t r = idx_sorted[i], p = idx_sorted[i - 1];
            const uint32_t cnt = nids[r];
            if (cnt == nids[p]) {
                const uint32_t* a = ids_pool + idoff[r];
                const uint32_t* b = ids_pool + idoff[p];
                uint32_t j = 0;
                while (j < cnt && a[j] == b[j]) ++j;
                is_head = j < cnt;
            }
        }
        head[i] = is_head;
    }
}
// group_off = exclusive scan of head: position i belongs to group group_off[i] + head[i] - 1
__global__ __launch_bounds__(256) void k_dd_groups(const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ head, const uint64_t* __restrict__ group_off,
                                                   const uint32_t* __restrict__ nids, const uint64_t* __restrict__ idoff, uint64_t n_reads,
                                                   uint32_t* __restrict__ group_of, uint32_t* __restrict__ nids_u, uint64_t* __restrict__ idoff_u) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = idx_sorted[i], h = head[i];
        const uint32_t g = (uint32_t)(group_off[i] + h - 1);
        group_of[r] = g;
        if (h) { nids_u[g] = nids[r]; idoff_u[g] = idoff[r]; }
    }
}
// one wave per read: the result of its group becomes its own
__global__ __launch_bounds__(256) void k_dd_fanout(const uint32_t* __restrict__ group_of, const uint32_t* __restrict__ bitmap_u, const uint32_t* __restrict__ counts_u,
                                                   const uint32_t* __restrict__ small_u, uint64_t n_reads, uint32_t W, uint32_t* __restrict__ bitmap,
                                                   uint32_t* __restrict__ counts, uint32_t* __restrict__ small) {
    const uint32_t lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += waves) {
        const uint64_t g = group_of[r];
        const uint32_t cnt = counts_u[g];
        if (lane == 0) counts[r] = cnt;
        if (cnt == 0) continue;
        if (small_u && cnt <= SMALL_RESULT) {
            if (lane < SMALL_RESULT) small[r * SMALL_RESULT + lane] = small_u[g * SMALL_RESULT + lane];
            continue;
        }
        const u32x4* src = (const u32x4*)(bitmap_u + g * W);
        u32x4* dst = (u32x4*)(bitmap + r * W);
        for (uint32_t q = lane; q < W / 4; q += 64) dst[q] = src[q];
    }
}

// ---------------------------------------------------------------------------------------------
// Verification aid (fgpu_result_checksum; the spirit of util::check_intersection / check_union, include/util.hpp:106-208): a
// checksum of the u32 colour lists of a pass, taken twice by kernels that share nothing — from the lists themselves
// (entry p of the CSR holds colour c: v = (c + 1) * (p + 1); out = {#entries, sum of v, xor of v * odd constant}) and from
// what the colour stage left behind (result rows / small-result slots + sizes + CSR offsets: the rank of a set bit inside its
// row gives its position in the read's list). Equal triples: k2b_expand wrote every colour of every read at its place.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void checksum_fold(uint64_t cnt, uint64_t sum, uint64_t x, unsigned long long* out) {
    for (int o = 32; o; o >>= 1) { cnt += __shfl_xor(cnt, o); sum += __shfl_xor(sum, o); x ^= __shfl_xor(x, o); }
    if (lane_id() == 0) {
        if (cnt) atomicAdd(out, (unsigned long long)cnt);
        if (sum) atomicAdd(out + 1, (unsigned long long)sum);
        if (x) atomicXor(out + 2, (unsigned long long)x);
    }
}
constexpr uint64_t CHECKSUM_MIX = 0x9E3779B97F4A7C15ull;
__global__ __launch_bounds__(256) void k_checksum_lists(const uint32_t* __restrict__ colors, const uint64_t* __restrict__ last_offset, unsigned long long* __restrict__ out) {
    const uint64_t total = *last_offset;  // the CSR's own end, not the host's count
    uint64_t cnt = 0, sum = 0, x = 0;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = ((uint64_t)colors[p] + 1) * (p + 1);
        ++cnt; sum += v; x ^= v * CHECKSUM_MIX;
    }
    checksum_fold(cnt, sum, x, out);
}
// one wave per read
__global__ __launch_bounds__(256) void k_checksum_rows(const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ counts,
                                                       const uint64_t* __restrict__ offsets, uint64_t n_reads, uint32_t W,
                                                       const uint32_t* __restrict__ small, unsigned long long* __restrict__ out) {
    const uint32_t lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t cnt = 0, sum = 0, x = 0;
    for (uint64_t r = wave; r < n_reads; r += waves) {
        const uint32_t size = counts[r];
        const uint64_t p0 = offsets[r];
        if (size == 0) continue;  // (an empty result may have left no row)
        if (small && size <= SMALL_RESULT) {  // the colours themselves, ascending, in the read's slot
            if (lane < size) {
                const uint64_t v = ((uint64_t)small[r * SMALL_RESULT + lane] + 1) * (p0 + lane + 1);
                ++cnt; sum += v; x ^= v * CHECKSUM_MIX;
            }
            continue;
        }
        uint32_t before = 0;  // set bits of the row in front of this round's words
        for (uint32_t w0 = 0; w0 < W; w0 += 64) {
            uint32_t word = w0 + lane < W ? bitmap[r * W + w0 + lane] : 0u;
            const uint32_t pc = __popc(word);
            uint32_t incl = pc;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
            uint32_t rank = before + incl - pc;
            while (word) {
                const uint32_t b = __ffs(word) - 1;
                word &= word - 1;
                const uint64_t v = ((uint64_t)((w0 + lane) * 32 + b) + 1) * (p0 + rank + 1);
                ++rank; ++cnt; sum += v; x ^= v * CHECKSUM_MIX;
            }
            before += __shfl(incl, 63);
        }
    }
    checksum_fold(cnt, sum, x, out);
}

}  // namespace fg
