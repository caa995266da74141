// Builds the GPU k-mer dictionary (Dict) from coloured unitigs.
//
// Role in the reference: sshash::dictionary::build over the unitig FASTA (src/index.cpp:268-276,
// include/builders/builder.hpp:191-199) followed by u2c (include/index.hpp:37). Design (own, see
// common/kmer_common.h): the minimizer of a k-mer is its canonical m-mer with the smallest hash order;
// consecutive k-mers of a unitig sharing a minimizer occurrence form a super-k-mer, stored as ONE 16-byte
// record that carries the unitig context of the occurrence (so a lookup never touches the strings) and the
// colour-set id; records live in an open-addressing table of 64-byte buckets addressed by a hash of the
// canonical minimizer. One lookup = one line fetch.
//
// Ties and strands are settled HERE so that the lookup kernel has no special cases: a query takes the
// leftmost smallest m-mer of its k-mer in its own orientation, which is the rightmost one in unitig
// orientation when the read lies on the other strand; a record therefore covers every k-mer in which its
// occurrence is A smallest-order m-mer (on ties a k-mer is covered by several records, each under its own
// occurrence). A palindromic minimizer (even m only) cannot tell the strand: it gets a record per strand flag.
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

inline uint32_t string_base(const std::vector<uint64_t>& strings, uint64_t pos) {
    const uint64_t w = strings[pos >> 5];
    const uint32_t b = pos & 31;
    return (uint32_t)((w >> b) & 1u) | (uint32_t)(((w >> (32 + b)) & 1u) << 1);
}

}  // namespace detail

inline void check_dict_params(uint32_t k, uint32_t m) {
    if (k < 2 || k > 31) throw std::runtime_error("k must be in [2,31]");
    if (m < 1 || m > k || k - m > 15) throw std::runtime_error("need m <= k and k - m <= 15");
    if (2 * k - m > REC_CTX_MAX) throw std::runtime_error("need 2k - m <= 43 (the record's context)");
}

// canonical key of the minimizer of a record, cut out of its context
inline uint64_t record_key(const uint32_t* w, uint32_t k, uint32_t m) {
    const uint32_t km = k - m;
    const uint32_t lo = (uint32_t)(rec_ctx_lo(w[0], w[2]) >> km) & low_mask32(m);
    const uint32_t hi = (uint32_t)(rec_ctx_hi(w[1], w[2]) >> km) & low_mask32(m);
    return canonical_key(lo, hi, m);
}

// Places the records into the bucket table by linear probing at record granularity: a record goes to the
// first free slot at or after its home bucket; every full bucket passed on the way is marked `spill`, which
// is what tells a query to read on. ~0.6 records per bucket on average: a query rarely needs a second line.
inline void build_dict_table(Dict& d) {
    const uint64_t nrec = d.num_records();
    if (nrec >= (1ULL << 31)) throw std::runtime_error("too many super-k-mer records");
    d.num_buckets = (uint32_t)std::max<uint64_t>(16, nrec + nrec / 2 + nrec / 8);
    const uint64_t nb_total = (uint64_t)d.num_buckets + DICT_TAIL_BUCKETS;
    d.table.assign(nb_total * BUCKET_WORDS, 0);
    for (uint64_t b = 0; b < nb_total; ++b)
        for (uint32_t r = 0; r < BUCKET_RECS; ++r) d.table[b * BUCKET_WORDS + r * REC_WORDS + 2] = REC_W2_EMPTY;
    std::vector<uint8_t> fill(nb_total, 0);
    for (uint64_t i = 0; i < nrec; ++i) {
        const uint32_t* w = &d.records[i * REC_WORDS];
        uint64_t b = mulhi32(dict_hash(record_key(w, d.k, d.m), d.seed), d.num_buckets);
        while (fill[b] == BUCKET_RECS) {
            d.table[b * BUCKET_WORDS + (BUCKET_RECS - 1) * REC_WORDS + 3] |= REC_SPILL;
            if (++b >= nb_total) throw std::runtime_error("dictionary table: probe chain ran past the tail buckets");
        }
        uint32_t* dst = &d.table[b * BUCKET_WORDS + fill[b] * REC_WORDS];
        dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3] & REC_MAX_CSID;
        ++fill[b];
    }
}

inline void build_dict(Dict& d, uint32_t k, uint32_t m, const char* bases, uint64_t total_bases,
                       const std::vector<uint64_t>& unitig_off, const std::vector<uint32_t>& unitig_csid,
                       unsigned nthreads = 0) {
    check_dict_params(k, m);
    if (nthreads == 0) nthreads = std::max(1u, std::thread::hardware_concurrency());
    d.k = k;
    d.m = m;
    d.seed = 0;
    d.total_bases = total_bases;
    d.unitig_off = unitig_off;
    d.unitig_csid = unitig_csid;
    detail::pack_strings(bases, total_bases, d.strings);

    const uint64_t nu = unitig_csid.size();
    const uint32_t km = k - m, CL = 2 * k - m;
    std::vector<std::vector<uint32_t>> parts(nthreads);
    std::vector<uint64_t> nk_parts(nthreads, 0);
    std::vector<std::string> errors(nthreads);
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
                std::vector<uint32_t> ord;
                std::vector<uint8_t> fw, pal;
                auto& out = parts[t];
                for (uint64_t u = cut[t]; u < cut[t + 1]; ++u) {
                    const uint64_t b = unitig_off[u], e = unitig_off[u + 1];
                    const int64_t len = (int64_t)(e - b);
                    if (len < (int64_t)k) { errors[t] = "unitig shorter than k"; return; }
                    if (unitig_csid[u] > REC_MAX_CSID) { errors[t] = "colour-set id exceeds record width"; return; }
                    const int64_t nm = len - m + 1, nk = len - k + 1;
                    ord.resize(nm);
                    fw.resize(nm);
                    pal.resize(nm);
                    for (int64_t i = 0; i < nm; ++i) {
                        const uint64_t s = b + i;
                        uint32_t lo, hi;
                        string_lmer(d.strings[s >> 5], d.strings[(s >> 5) + 1], (uint32_t)(s & 31), m, lo, hi);
                        ord[i] = minimizer_order(canonical_key(lo, hi, m));
                        const uint64_t kf = lmer_key(lo, hi), kr = lmer_key(rc_plane(lo, m), rc_plane(hi, m));
                        fw[i] = kf <= kr;
                        pal[i] = kf == kr;
                    }
                    nk_parts[t] += nk;
                    for (int64_t p = 0; p < nm; ++p) {
                        // k-mers [sa, sb] of the unitig contain m-mer p and no m-mer of smaller order
                        int64_t sa = std::max<int64_t>(0, p - km), sb = std::min<int64_t>(p, nk - 1);
                        for (int64_t q = p - 1; q >= sa; --q)
                            if (ord[q] < ord[p]) { sa = q + 1; break; }
                        for (int64_t q = p + 1; q <= sb + km; ++q)
                            if (ord[q] < ord[p]) { sb = q - km - 1; break; }
                        if (sa > sb) continue;
                        uint64_t clo = 0, chi = 0;  // context base c = unitig base p - km + c
                        for (uint32_t c = 0; c < CL; ++c) {
                            const int64_t x = p - km + c;
                            if (x < 0 || x >= len) continue;
                            const uint32_t code = detail::string_base(d.strings, b + x);
                            clo |= (uint64_t)(code & 1u) << c;
                            chi |= (uint64_t)(code >> 1) << c;
                        }
                        const uint32_t smin = (uint32_t)(sa - (p - km)), smax = (uint32_t)(sb - (p - km));
                        auto emit = [&](uint32_t s0, uint32_t s1, bool fwd) {
                            out.push_back((uint32_t)clo);
                            out.push_back((uint32_t)chi);
                            out.push_back(rec_w2(clo, chi, s0, s1, fwd));
                            out.push_back(unitig_csid[u]);
                        };
                        if (!pal[p]) {
                            emit(smin, smax, fw[p] != 0);
                        } else {
                            // palindromic minimizer: the flag cannot tell the strand, so one record per flag. A k-mer that is its
                            // own reverse complement (even k) would match both: the second record skips those windows.
                            emit(smin, smax, true);
                            uint32_t s0 = smin;
                            for (uint32_t s = smin; s <= smax + 1; ++s) {
                                bool self_rc = false;
                                if (s <= smax) {
                                    const uint32_t lo = (uint32_t)(clo >> s) & low_mask32(k), hi = (uint32_t)(chi >> s) & low_mask32(k);
                                    self_rc = lo == rc_plane(lo, k) && hi == rc_plane(hi, k);
                                }
                                if (s > smax || self_rc) {
                                    if (s > s0) emit(s0, s - 1, false);
                                    s0 = s + 1;
                                }
                            }
                        }
                    }
                }
            });
        }
        for (auto& x : th) x.join();
    }
    for (auto& e : errors)
        if (!e.empty()) throw std::runtime_error(e);
    {
        uint64_t tot = 0;
        for (auto& p : parts) tot += p.size();
        d.records.clear();
        d.records.reserve(tot);
        for (auto& p : parts) {
            d.records.insert(d.records.end(), p.begin(), p.end());
            std::vector<uint32_t>().swap(p);
        }
    }
    d.num_kmers = 0;
    for (auto x : nk_parts) d.num_kmers += x;
    build_dict_table(d);
}

// Host walk of the same structure, used ONLY by the build-time self check (verify_dict below, the
// analogue of the reference's `--check`, builder.hpp:221-277); queries never run here. It follows the
// lookup kernel step by step: leftmost smallest m-mer of the k-mer as given, bucket chain, every record
// compared on the strand its flag selects. Returns the number of matching records; *csid = the last match.
inline uint32_t dict_lookup(const Dict& d, uint32_t klo, uint32_t khi, uint32_t* csid) {
    const uint32_t k = d.k, m = d.m, km = k - m;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t j = 0; j <= km; ++j) {
        const uint32_t lo = (klo >> j) & low_mask32(m), hi = (khi >> j) & low_mask32(m);
        best = std::min(best, (minimizer_order(canonical_key(lo, hi, m)) << ORDER_POS_BITS) | j);
    }
    const uint32_t pm = best & ((1u << ORDER_POS_BITS) - 1u);
    const uint32_t mlo = (klo >> pm) & low_mask32(m), mhi = (khi >> pm) & low_mask32(m);
    const bool qfwd = is_fwd_canonical(mlo, mhi, m);
    const uint32_t rlo = rc_plane(klo, k), rhi = rc_plane(khi, k);
    uint32_t found = 0;
    uint64_t b = mulhi32(dict_hash(canonical_key(mlo, mhi, m), d.seed), d.num_buckets);
    for (;; ++b) {
        const uint32_t* bw = &d.table[b * BUCKET_WORDS];
        for (uint32_t r = 0; r < BUCKET_RECS; ++r) {
            const uint32_t* w = bw + r * REC_WORDS;
            const bool same = rec_fwd(w[2]) == qfwd;
            const uint32_t s = same ? km - pm : pm;  // window of this k-mer in the record's context
            if (s < rec_smin(w[2]) || s > rec_smax(w[2])) continue;
            const uint32_t lo = (uint32_t)(rec_ctx_lo(w[0], w[2]) >> s) & low_mask32(k);
            const uint32_t hi = (uint32_t)(rec_ctx_hi(w[1], w[2]) >> s) & low_mask32(k);
            if (lo == (same ? klo : rlo) && hi == (same ? khi : rhi)) {
                ++found;
                *csid = w[3] & REC_MAX_CSID;
            }
        }
        if (!(bw[(BUCKET_RECS - 1) * REC_WORDS + 3] & REC_SPILL)) break;
    }
    return found;
}

// every k-mer of every unitig must be found exactly once, on both strands, with its unitig's colour-set id
inline void verify_dict(const Dict& d, uint64_t stride = 1) {
    const uint32_t k = d.k;
    for (uint64_t u = 0; u < d.num_unitigs(); u += stride) {
        for (uint64_t s = d.unitig_off[u]; s + k <= d.unitig_off[u + 1]; ++s) {
            uint32_t lo, hi, c = 0;
            string_lmer(d.strings[s >> 5], d.strings[(s >> 5) + 1], (uint32_t)(s & 31), k, lo, hi);
            if (dict_lookup(d, lo, hi, &c) != 1 || c != d.unitig_csid[u]) throw std::runtime_error("dictionary self-check failed (fwd)");
            if (dict_lookup(d, rc_plane(lo, k), rc_plane(hi, k), &c) != 1 || c != d.unitig_csid[u])
                throw std::runtime_error("dictionary self-check failed (rc)");
        }
    }
}

// bucket statistics (tools / logs): records per bucket chain as a query sees them
struct DictStats {
    uint64_t records = 0, buckets = 0, spill_buckets = 0, max_chain = 0;
};
inline DictStats dict_stats(const Dict& d) {
    DictStats s;
    s.records = d.num_records();
    s.buckets = d.num_buckets;
    uint64_t chain = 0;
    for (uint64_t b = 0; b < (uint64_t)d.num_buckets + DICT_TAIL_BUCKETS; ++b) {
        if (d.table[b * BUCKET_WORDS + (BUCKET_RECS - 1) * REC_WORDS + 3] & REC_SPILL) {
            ++s.spill_buckets;
            s.max_chain = std::max(s.max_chain, ++chain);
        } else {
            chain = 0;
        }
    }
    return s;
}

}  // namespace fg
