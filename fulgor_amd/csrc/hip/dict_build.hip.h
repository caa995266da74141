// The k-mer dictionary table built IN HBM from the 16-byte super-k-mer records when an index is opened on a device (the role of
// loading sshash::dictionary in the reference: essentials::load at tools/pseudoalign.cpp:340). The container stores the records
// (0.23 GB for the bench index), not the 1.5 GB table: building it on the host took 0.5 s of the 0.9 s an open took, on the device
// it is a few milliseconds — sort the records by (home bucket, minimizer, record number), then every hashed bucket places its own
// records by the rule of common/dict_place.h, the same code the host builder runs, so both write the same bytes (checked by
// fgpu_selfcheck, which downloads this table, compares it with the host's and walks every k-mer through it).
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "../common/dict_place.h"

namespace fg {

// sort key of a record: its minimizer (hi plane above lo plane)
__global__ __launch_bounds__(256) void k_dict_keys(const uint32_t* __restrict__ records, uint64_t nrec, uint32_t k, uint32_t m,
                                                   unsigned long long* __restrict__ key, uint32_t* __restrict__ idx) {
    const uint32_t km = k - m;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 w = ((const uint4*)records)[i];
        const uint32_t lo = (uint32_t)(rec_ctx_lo(w.x, w.z) >> km) & low_mask32(m);
        const uint32_t hi = (uint32_t)(rec_ctx_hi(w.y, w.z) >> km) & low_mask32(m);
        key[i] = lmer_key(lo, hi);
        idx[i] = (uint32_t)i;
    }
}
__global__ __launch_bounds__(256) void k_dict_homes(const unsigned long long* __restrict__ key, uint64_t nrec, uint32_t seed, uint32_t num_buckets,
                                                    uint32_t* __restrict__ home, uint32_t* __restrict__ pos) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long kk = key[i];
        home[i] = mulhi32(dict_hash((uint32_t)kk, (uint32_t)(kk >> 32), seed), num_buckets);
        pos[i] = (uint32_t)i;
    }
}
// the order by (home, key, record number): position i holds what the first sort left at pos[i]; bucket b's segment starts at bstart[b]
__global__ __launch_bounds__(256) void k_dict_gather(const uint32_t* __restrict__ home_sorted, const uint32_t* __restrict__ pos,
                                                     const unsigned long long* __restrict__ key1, const uint32_t* __restrict__ idx1, uint64_t nrec,
                                                     unsigned long long* __restrict__ key2, uint32_t* __restrict__ idx2, uint32_t* __restrict__ bstart) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = pos[i];
        key2[i] = key1[p];
        idx2[i] = idx1[p];
        if (i == 0 || home_sorted[i - 1] != home_sorted[i]) bstart[home_sorted[i]] = (uint32_t)i;
    }
}
// overflow buckets every hashed bucket needs
__global__ __launch_bounds__(256) void k_dict_count(const uint32_t* __restrict__ home, const unsigned long long* __restrict__ key, uint64_t nrec,
                                                    const uint32_t* __restrict__ bstart, uint64_t nb_hashed, uint32_t* __restrict__ nb_over) {
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb_hashed; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = bstart[b];
        uint32_t nb = 0;
        if (a != 0xFFFFFFFFu) {
            uint64_t e = a;
            while (e < nrec && home[e] == (uint32_t)b) ++e;
            const BucketPlan p = plan_bucket((uint64_t)a, e, [&](uint64_t i) { return (uint64_t)key[i]; });
            nb = (uint32_t)((p.moved + BUCKET_RECS - 1) / BUCKET_RECS);
        }
        nb_over[b] = nb;
    }
}
// every hashed bucket writes itself and its overflow run
__global__ __launch_bounds__(256) void k_dict_fill(const uint32_t* __restrict__ records, const uint32_t* __restrict__ home,
                                                   const unsigned long long* __restrict__ key, const uint32_t* __restrict__ idx, uint64_t nrec,
                                                   const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ over_off, uint64_t nb_hashed,
                                                   uint32_t* __restrict__ table) {
    const uint4 empty = make_uint4(0u, 0u, REC_W2_EMPTY, 0u);
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nb_hashed; b += (uint64_t)gridDim.x * blockDim.x) {
        uint4* bw = (uint4*)table + b * BUCKET_RECS;
        const uint32_t a = bstart[b];
        if (a == 0xFFFFFFFFu) {
            for (uint32_t s = 0; s < BUCKET_RECS; ++s) bw[s] = empty;
            continue;
        }
        uint64_t e = a;
        while (e < nrec && home[e] == (uint32_t)b) ++e;
        const auto key_at = [&](uint64_t i) { return (uint64_t)key[i]; };
        const BucketPlan p = plan_bucket((uint64_t)a, e, key_at);
        const uint64_t nb = (p.moved + BUCKET_RECS - 1) / BUCKET_RECS;
        const uint64_t run = nb_hashed + over_off[b];  // first bucket of the overflow run
        uint4* ov = (uint4*)table + run * BUCKET_RECS;
        const auto rec = [&](uint64_t i) {
            uint4 w = ((const uint4*)records)[idx[i]];
            w.w &= ~REC_SPILL;
            return w;
        };
        for (uint32_t s = p.kept_records; s < BUCKET_RECS; ++s) bw[s] = empty;
        for (uint64_t j = p.moved; j < nb * BUCKET_RECS; ++j) ov[j] = empty;
        place_bucket((uint64_t)a, e, p, key_at, [&](uint32_t slot, uint64_t i) { bw[slot] = rec(i); }, [&](uint64_t j, uint64_t i) { ov[j] = rec(i); });
        if (nb) {
            // the query reads the first REDIRECT_DIRECT buckets of the run at once; further ones hang on spill flags
            for (uint64_t b2 = REDIRECT_DIRECT - 1; b2 + 1 < nb; ++b2) table[(run + b2) * BUCKET_WORDS + (BUCKET_RECS - 1) * REC_WORDS + 3] |= REC_SPILL;
            bw[BUCKET_RECS - 1] = make_uint4(0u, (uint32_t)run, REC_W2_REDIRECT, (uint32_t)nb);
        }
    }
}

}  // namespace fg
