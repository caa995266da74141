// Builds the GPU k-mer dictionary (Dict) from coloured unitigs.
//
// Role in the reference: sshash::dictionary::build over the unitig FASTA (src/index.cpp:268-276,
// include/builders/builder.hpp:191-199) followed by u2c (include/index.hpp:37). Design (own, see
// common/kmer_common.h): for every k-mer of every unitig pick the minimizer = the canonical m-mer
// with the smallest hash order (leftmost on ties); consecutive k-mers sharing the minimizer
// occurrence form a super-k-mer, stored as ONE 8-byte record {minimizer position, valid offset
// range, colour-set id}; records are addressed through a pilot-displaced perfect hash of the
// canonical minimizer.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <thread>
#include <string>
#include "index_model.hpp"

namespace fg {

namespace detail {

inline void pack_strings(const char* bases, uint64_t n, std::vector<uint64_t>& words) {
    words.assign((n + 31) / 32 + 2, 0);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t c = base_code((uint8_t)bases[i]);
        if (c > 3) throw std::runtime_error("unitig contains a non-ACGT character");
        uint64_t w = i >> 5;
        uint32_t b = i & 31;
        words[w] |= (uint64_t)(c & 1) << b;
        words[w] |= (uint64_t)(c >> 1) << (32 + b);
    }
}

struct KeyRec {
    uint64_t h0;
    uint64_t rec;
};

}  // namespace detail

inline void build_dict(Dict& d, uint32_t k, uint32_t m, const char* bases, uint64_t total_bases,
                       const std::vector<uint64_t>& unitig_off, const std::vector<uint32_t>& unitig_csid,
                       unsigned nthreads = 0) {
    if (k < 2 || k > 31) throw std::runtime_error("k must be in [2,31]");
    if (m < 1 || m > k || k - m > 15) throw std::runtime_error("need m <= k and k - m <= 15");
    if (total_bases >= (1ULL << 31)) throw std::runtime_error("unitig strings exceed 2^31 bases");
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    d.k = k;
    d.m = m;
    d.total_bases = total_bases;
    d.unitig_off = unitig_off;
    d.unitig_csid = unitig_csid;
    detail::pack_strings(bases, total_bases, d.strings);

    const uint64_t nu = unitig_csid.size();
    const uint32_t W = k - m + 1;  // m-mers per k-mer
    std::vector<std::vector<detail::KeyRec>> parts(nthreads);
    std::vector<uint64_t> nk_parts(nthreads, 0);
    {
        std::vector<std::thread> th;
        // split unitigs so that threads get similar numbers of bases
        std::vector<uint64_t> cut(nthreads + 1, nu);
        cut[0] = 0;
        for (unsigned t = 1; t < nthreads; ++t) {
            uint64_t target = total_bases / nthreads * t;
            cut[t] = std::lower_bound(unitig_off.begin(), unitig_off.end(), target) - unitig_off.begin();
            if (cut[t] > nu) cut[t] = nu;
        }
        for (unsigned t = 0; t < nthreads; ++t) {
            th.emplace_back([&, t]() {
                std::vector<uint64_t> hh;
                std::vector<uint8_t> fw;
                auto& out = parts[t];
                for (uint64_t u = cut[t]; u < cut[t + 1]; ++u) {
                    const uint64_t b = unitig_off[u], e = unitig_off[u + 1];
                    const uint64_t len = e - b;
                    if (len < k) throw std::runtime_error("unitig shorter than k");
                    if (unitig_csid[u] > REC_MAX_CSID) throw std::runtime_error("colour-set id exceeds record width");
                    const uint64_t nm = len - m + 1;
                    hh.resize(nm);
                    fw.resize(nm);
                    for (uint64_t i = 0; i < nm; ++i) {
                        uint64_t s = b + i;
                        uint32_t lo, hi;
                        string_lmer(d.strings[s >> 5], d.strings[(s >> 5) + 1], (uint32_t)(s & 31), m, lo, hi);
                        hh[i] = canonical_key(lo, hi, m);
                        fw[i] = is_fwd_canonical(lo, hi, m);
                    }
                    const uint64_t nk = len - k + 1;
                    nk_parts[t] += nk;
                    uint64_t run_p = ~0ULL, run_first = 0;
                    for (uint64_t s = 0; s <= nk; ++s) {
                        uint64_t p = ~0ULL;
                        if (s < nk) {
                            uint32_t best = 0xFFFFFFFFu;
                            for (uint32_t j = 0; j < W; ++j) {  // leftmost smallest order
                                uint32_t pk = (order24(hh[s + j]) << 4) | j;
                                best = pk < best ? pk : best;
                            }
                            p = s + (best & 15u);
                        }
                        if (p != run_p) {
                            if (run_p != ~0ULL) {
                                uint64_t s_last = s - 1;
                                out.push_back({hh[run_p], rec_pack((uint32_t)(b + run_p), fw[run_p] != 0, (uint32_t)(run_p - s_last),
                                                                   (uint32_t)(run_p - run_first), unitig_csid[u])});
                            }
                            run_p = p;
                            run_first = s;
                        }
                    }
                }
            });
        }
        for (auto& x : th) x.join();
    }
    std::vector<detail::KeyRec> recs;
    {
        uint64_t tot = 0;
        for (auto& p : parts) tot += p.size();
        recs.reserve(tot);
        for (auto& p : parts) {
            recs.insert(recs.end(), p.begin(), p.end());
            std::vector<detail::KeyRec>().swap(p);
        }
    }
    d.num_kmers = 0;
    for (auto x : nk_parts) d.num_kmers += x;
    std::sort(recs.begin(), recs.end(), [](const detail::KeyRec& a, const detail::KeyRec& b) {
        return a.h0 < b.h0 || (a.h0 == b.h0 && a.rec < b.rec);
    });
    // distinct keys
    std::vector<uint64_t> key_begin;  // index into recs
    for (uint64_t i = 0; i < recs.size(); ++i)
        if (i == 0 || recs[i].h0 != recs[i - 1].h0) key_begin.push_back(i);
    const uint64_t nkeys = key_begin.size();
    key_begin.push_back(recs.size());
    if (nkeys >= (1ULL << 31)) throw std::runtime_error("too many minimizers");

    d.num_slots = (uint32_t)std::max<uint64_t>(1, (uint64_t)(nkeys / 0.80) + 1);
    d.num_buckets = (uint32_t)std::max<uint64_t>(1, (nkeys + 5) / 6);  // ~6 keys per 16-bit pilot: table stays L2 resident

    std::vector<uint64_t> kh(nkeys);          // seeded hash of every key
    std::vector<uint32_t> korder(nkeys);      // keys grouped by bucket
    std::vector<uint32_t> bucket_begin, order, pos;
    std::vector<uint64_t> taken;
    bool built = false;
    for (d.seed = 1; d.seed <= 16 && !built; ++d.seed) {
        for (uint64_t i = 0; i < nkeys; ++i) kh[i] = phf_hash(recs[key_begin[i]].h0, d.seed);
        bucket_begin.assign(d.num_buckets + 1, 0);
        for (uint64_t i = 0; i < nkeys; ++i) bucket_begin[phf_bucket(kh[i], d.num_buckets) + 1]++;
        for (uint32_t b = 0; b < d.num_buckets; ++b) bucket_begin[b + 1] += bucket_begin[b];
        {
            std::vector<uint32_t> fill(bucket_begin.begin(), bucket_begin.end() - 1);
            for (uint64_t i = 0; i < nkeys; ++i) korder[fill[phf_bucket(kh[i], d.num_buckets)]++] = (uint32_t)i;
        }
        order.resize(d.num_buckets);
        for (uint32_t b = 0; b < d.num_buckets; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return bucket_begin[a + 1] - bucket_begin[a] > bucket_begin[b + 1] - bucket_begin[b];
        });
        d.pilots.assign(d.num_buckets, 0);
        d.slots.assign(d.num_slots, REC_EMPTY);
        d.overflow.clear();
        taken.assign((d.num_slots + 63) / 64, 0);
        bool ok_all = true;
        for (uint32_t b : order) {
            const uint32_t kb = bucket_begin[b], ke = bucket_begin[b + 1];
            if (kb == ke) continue;
            pos.resize(ke - kb);
            uint32_t pilot = 0;
            for (;; ++pilot) {
                if (pilot > 0xFFFFu) break;
                bool ok = true;
                for (uint32_t i = kb; i < ke && ok; ++i) {
                    uint32_t sl = phf_slot(kh[korder[i]], pilot, d.num_slots);
                    if ((taken[sl >> 6] >> (sl & 63)) & 1) ok = false;
                    for (uint32_t j = kb; j < i && ok; ++j)
                        if (pos[j - kb] == sl) ok = false;
                    pos[i - kb] = sl;
                }
                if (ok) break;
            }
            if (pilot > 0xFFFFu) { ok_all = false; break; }  // retry with the next seed
            d.pilots[b] = (uint16_t)pilot;
            for (uint32_t i = kb; i < ke; ++i) {
                const uint32_t sl = pos[i - kb];
                taken[sl >> 6] |= 1ULL << (sl & 63);
                const uint64_t rb = key_begin[korder[i]], re = key_begin[korder[i] + 1];
                if (re - rb == 1) {
                    d.slots[sl] = recs[rb].rec;
                } else {
                    if (d.overflow.size() + (re - rb) >= (1ULL << 32)) throw std::runtime_error("overflow array too large");
                    d.slots[sl] = ovf_pack((uint32_t)d.overflow.size(), (uint32_t)(re - rb));
                    for (uint64_t r = rb; r < re; ++r) d.overflow.push_back(recs[r].rec);
                }
            }
        }
        if (ok_all) { built = true; break; }
    }
    if (!built) throw std::runtime_error("perfect hash construction failed for 16 seeds");
    d.overflow.push_back(REC_EMPTY);  // keep the device array non-empty and padded for paired reads
}

// Host walk of the same structure, used ONLY by the build-time self check (verify_dict below, the
// analogue of the reference's `--check`, builder.hpp:221-277); queries never run here.
// Returns the colour-set id or 0xFFFFFFFF.
inline uint32_t dict_lookup(const Dict& d, uint32_t klo, uint32_t khi) {
    const uint32_t k = d.k, m = d.m, W = k - m + 1;
    uint32_t bestL = 0xFFFFFFFFu, bestR = 0xFFFFFFFFu;
    uint64_t hs[16];
    for (uint32_t j = 0; j < W; ++j) {
        uint32_t lo = (klo >> j) & low_mask32(m), hi = (khi >> j) & low_mask32(m);
        hs[j] = canonical_key(lo, hi, m);
        uint32_t o = order24(hs[j]) << 4;
        bestL = std::min(bestL, o | j);
        bestR = std::min(bestR, o | (15u - j));
    }
    const uint32_t jL = bestL & 15u, jR = 15u - (bestR & 15u);
    const uint32_t rlo = rc_plane(klo, k), rhi = rc_plane(khi, k);
    auto try_rec = [&](uint64_t r, uint32_t jd, uint32_t qlo, uint32_t qhi) -> uint32_t {
        if (jd < rec_jmin(r) || jd > rec_jmax(r)) return 0xFFFFFFFFu;
        uint64_t s = (uint64_t)rec_pos(r) - jd;
        uint32_t lo, hi;
        string_lmer(d.strings[s >> 5], d.strings[(s >> 5) + 1], (uint32_t)(s & 31), k, lo, hi);
        return (lo == qlo && hi == qhi) ? rec_csid(r) : 0xFFFFFFFFu;
    };
    auto probe = [&](uint64_t h0, bool doA, bool doB) -> uint32_t {
        const uint64_t h = phf_hash(h0, d.seed);
        uint32_t pilot = d.pilots[phf_bucket(h, d.num_buckets)];
        uint64_t e = d.slots[phf_slot(h, pilot, d.num_slots)];
        const uint64_t* p = &e;
        uint32_t cnt = 1;
        if (e & REC_TAG) { p = d.overflow.data() + ovf_off(e); cnt = ovf_cnt(e); }
        for (uint32_t i = 0; i < cnt; ++i) {
            if (doA) { uint32_t c = try_rec(p[i], jL, klo, khi); if (c != 0xFFFFFFFFu) return c; }
            if (doB) { uint32_t c = try_rec(p[i], k - m - jR, rlo, rhi); if (c != 0xFFFFFFFFu) return c; }
        }
        return 0xFFFFFFFFu;
    };
    if (hs[jL] == hs[jR]) return probe(hs[jL], true, true);
    uint32_t c = probe(hs[jL], true, false);
    if (c != 0xFFFFFFFFu) return c;
    return probe(hs[jR], false, true);
}

// every k-mer of every unitig must be found with its unitig's colour-set id
inline void verify_dict(const Dict& d, uint64_t stride = 1) {
    const uint32_t k = d.k;
    for (uint64_t u = 0; u < d.num_unitigs(); u += stride) {
        for (uint64_t s = d.unitig_off[u]; s + k <= d.unitig_off[u + 1]; ++s) {
            uint32_t lo, hi;
            string_lmer(d.strings[s >> 5], d.strings[(s >> 5) + 1], (uint32_t)(s & 31), k, lo, hi);
            if (dict_lookup(d, lo, hi) != d.unitig_csid[u]) throw std::runtime_error("dictionary self-check failed (fwd)");
            if (dict_lookup(d, rc_plane(lo, k), rc_plane(hi, k)) != d.unitig_csid[u])
                throw std::runtime_error("dictionary self-check failed (rc)");
        }
    }
}

}  // namespace fg
