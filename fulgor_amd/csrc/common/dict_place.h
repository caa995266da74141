// How the super-k-mer records of one home bucket are laid out in the dictionary table: ONE statement of the rule, compiled for
// the host (host/dict_build.hpp: the table of a host-only handle, the self check) and for the device (hip/dict_build.hip.h: the
// table a device handle queries is built in HBM from the 16-byte records at open, in milliseconds instead of the half second the
// host needs), so that both write the same bytes.
//
// Input: the records sorted by (home bucket, minimizer key, record number); the segment [a, e) of that order holds the records
// whose minimizer hashes to bucket b. A KEY (one minimizer) keeps all its records together. Rule:
//   * total = e - a records; the bucket has BUCKET_RECS slots, one less if anything has to move (the last slot is the redirect);
//   * keys are taken whole, fewest records first (ties: key order), while they fit: with c = 1, 2, .. BUCKET_RECS, as many keys
//     of exactly c records as fit; the first key that does not fit ends the taking (every later one is at least as large);
//   * the kept keys fill the slots in that order (records of a key in record-number order);
//   * every other key moves to the bucket's overflow run — ceil(moved / BUCKET_RECS) consecutive buckets of the overflow region —
//     in key order.
#pragma once
#include "kmer_common.h"

namespace fg {

struct BucketPlan {
    uint32_t kept[BUCKET_RECS + 1];  // kept[c]: keys of exactly c records that stay (the first kept[c] of them in key order)
    uint32_t kept_records;           // slots used by them
    uint64_t moved;                  // records that go to the overflow run
};

// key(i): minimizer key of the i-th record of the sorted order
template <typename KeyAt>
FG_HD BucketPlan plan_bucket(uint64_t a, uint64_t e, KeyAt key) {
    BucketPlan p;
    uint32_t n_c[BUCKET_RECS + 1];
    for (uint32_t c = 0; c <= BUCKET_RECS; ++c) { n_c[c] = 0; p.kept[c] = 0; }
    const uint64_t total = e - a;
    for (uint64_t i = a; i < e;) {
        uint64_t j = i + 1;
        const uint64_t ki = key(i);
        while (j < e && key(j) == ki) ++j;
        if (j - i <= BUCKET_RECS) ++n_c[j - i];
        i = j;
    }
    const uint32_t room = total <= BUCKET_RECS ? BUCKET_RECS : BUCKET_RECS - 1;
    uint32_t slot = 0;
    for (uint32_t c = 1; c <= BUCKET_RECS; ++c) {
        const uint32_t fit = (room - slot) / c;
        const uint32_t take = n_c[c] < fit ? n_c[c] : fit;
        p.kept[c] = take;
        slot += take * c;
        if (take < n_c[c]) break;
    }
    p.kept_records = slot;
    p.moved = total - slot;
    return p;
}

// Walks the segment once more and tells where every record goes: stay(slot, i) for a kept record (slot of the home bucket),
// move(j, i) for the j-th record of the overflow run.
template <typename KeyAt, typename Stay, typename Move>
FG_HD void place_bucket(uint64_t a, uint64_t e, const BucketPlan& p, KeyAt key, Stay stay, Move move) {
    uint32_t base[BUCKET_RECS + 1], seen[BUCKET_RECS + 1];
    uint32_t at = 0;
    for (uint32_t c = 0; c <= BUCKET_RECS; ++c) { base[c] = at; at += p.kept[c] * c; seen[c] = 0; }
    uint64_t j_moved = 0;
    for (uint64_t i = a; i < e;) {
        uint64_t j = i + 1;
        const uint64_t ki = key(i);
        while (j < e && key(j) == ki) ++j;
        const uint64_t c = j - i;
        if (c <= BUCKET_RECS && seen[c] < p.kept[c]) {
            const uint32_t s0 = base[c] + seen[c] * (uint32_t)c;
            for (uint64_t r = 0; r < c; ++r) stay(s0 + (uint32_t)r, i + r);
            ++seen[c];
        } else {
            for (uint64_t r = i; r < j; ++r) move(j_moved++, r);
        }
        i = j;
    }
}

// hashed buckets for `nrec` records: `factor` buckets per record, capped so that the whole table (hashed region, tail, an overflow
// region of a quarter of the records) stays below DICT_MAX_BUCKETS; 0 = the collection does not fit
constexpr uint32_t DICT_TAIL_BUCKETS_ = 1024;  // (= DICT_TAIL_BUCKETS of host/index_model.hpp)
FG_HD uint32_t dict_hashed_buckets(uint64_t nrec, double factor, bool* capped) {
    const int64_t cap = (int64_t)DICT_MAX_BUCKETS - (int64_t)DICT_TAIL_BUCKETS_ - (int64_t)(nrec / 4) - 4096;
    if (capped) *capped = false;
    if (cap < (int64_t)(nrec / 2)) return 0;
    uint64_t want = (uint64_t)((double)nrec * factor);
    if ((int64_t)want > cap) { want = (uint64_t)cap; if (capped) *capped = true; }
    return (uint32_t)(want < 16 ? 16 : want);
}

}  // namespace fg
