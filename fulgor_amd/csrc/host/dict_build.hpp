// Builds the GPU k-mer dictionary (Dict) from coloured unitigs.
//
// Role in the reference: sshash::dictionary::build over the unitig FASTA (src/index.cpp:268-276,
// include/builders/builder.hpp:191-199) followed by u2c (include/index.hpp:37). Design (own, see
// common/kmer_common.h): the minimizer of a k-mer is its m-mer with the smallest hash order, leftmost on ties;
// consecutive k-mers sharing a minimizer occurrence form a super-k-mer, stored as ONE 16-byte record that
// carries the context of the occurrence (so a lookup never touches the strings) and the colour-set id; records
// live in a table of 64-byte buckets addressed by a hash of the minimizer. One lookup = one line fetch.
//
// Every unitig is indexed on BOTH strands (the unitig and its reverse complement are cut into super-k-mers
// independently), so that a query is compared in its own orientation only: the lookup kernel never forms a
// canonical m-mer or a reverse complement and has no strand, tie or palindrome cases. A k-mer that is its own
// reverse complement (even k only) is kept on the unitig's strand only, so that it is found once.
#pragma once
#include <atomic>
#include <cstdlib>
#include <algorithm>
#include <stdexcept>
#include <thread>
#include <string>
#include "index_model.hpp"
#include "../common/dict_place.h"

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
    if (2 * k - m > REC_CTX_MAX) throw std::runtime_error("need 2k - m <= 45 (the record's context)");
}

// the minimizer of a record, cut out of its context
inline void record_minimizer(const uint32_t* w, uint32_t k, uint32_t m, uint32_t& lo, uint32_t& hi) {
    const uint32_t km = k - m;
    lo = (uint32_t)(rec_ctx_lo(w[0], w[2]) >> km) & low_mask32(m);
    hi = (uint32_t)(rec_ctx_hi(w[1], w[2]) >> km) & low_mask32(m);
}
inline uint64_t record_key(const uint32_t* w, uint32_t k, uint32_t m) {
    uint32_t lo, hi;
    record_minimizer(w, k, m, lo, hi);
    return lmer_key(lo, hi);
}

// number of hashed buckets for the records of `d` (sets d.num_buckets); the table itself is built by build_dict_table (host) or by
// the device builder (hip/dict_build.hip.h) from the same records by the same rule (common/dict_place.h)
inline void dict_table_geometry(Dict& d) {
    const uint64_t nrec = d.num_records();
    if (nrec >= (1ULL << 31)) throw std::runtime_error("too many super-k-mer records");
    // 1.625 buckets per record (0.6 records per bucket); FULGOR_DICT_BUCKET_FACTOR overrides it (measurements: the table is rebuilt at every open)
    double factor = 1.625;
    if (const char* e = getenv("FULGOR_DICT_BUCKET_FACTOR")) { const double v = atof(e); if (v >= 1.0 && v <= 16.0) factor = v; }
    // at most 2^31 buckets in all (DICT_MAX_BUCKETS): the hashed region leaves room for an
    // overflow region of a quarter of the records; a collection that does not fit is refused here, not at the first redirect
    bool capped = false;
    const uint32_t nb = dict_hashed_buckets(nrec, factor, &capped);
    if (nb == 0)
        throw std::runtime_error("the k-mer dictionary of this collection needs more than 2^31 buckets (" + std::to_string(nrec) +
                                 " super-k-mer records): not supported by this build");
    if (capped)
        fprintf(stderr, "fulgor_amd: dictionary table capped at %u buckets for %llu records (%.2f buckets per record instead of %.2f): more keys behind redirects\n",
                nb, (unsigned long long)nrec, (double)nb / (double)nrec, factor);
    d.num_buckets = nb;
}

// Places the records into the bucket table. Hashed region: a sweep over the buckets in order; the keys hashed to a
// bucket keep all their records there while the bucket has room (keys with fewer records first). If they do not all fit, the bucket's LAST
// slot becomes its REDIRECT and the keys that are left share one run of consecutive buckets in the overflow region
// (a record is verified by its context, so records of several keys may lie side by side). ~0.6 records per bucket on average.
inline void build_dict_table(Dict& d) {
    const uint64_t nrec = d.num_records();
    dict_table_geometry(d);
    const uint64_t nb_hashed = (uint64_t)d.num_buckets + DICT_TAIL_BUCKETS;
    struct Ref { uint32_t home; uint32_t rec; uint64_t key; };
    // The table is rebuilt from the records whenever an index is opened: the sort of the records by (home bucket, key) is
    // most of the time of opening one, so it runs on several threads — the records are dealt into ranges of home buckets
    // (the hash spreads them evenly), every range is sorted on its own, and the ranges follow each other in the order.
#ifndef FG_DICT_THREADS
#define FG_DICT_THREADS 32u
#endif
    const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::min(FG_DICT_THREADS, std::thread::hardware_concurrency())), nrec / 65536 + 1);
    auto parallel = [&](uint64_t n, auto fn) {  // fn(thread, begin, end) over [0, n) in T contiguous pieces
        if (T == 1) { fn(0u, (uint64_t)0, n); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] { fn(t, n * t / T, n * (t + 1) / T); });
        for (auto& x : th) x.join();
    };
    const auto less = [](const Ref& a, const Ref& b) {
        return a.home != b.home ? a.home < b.home : (a.key != b.key ? a.key < b.key : a.rec < b.rec);
    };
    std::vector<Ref> refs(nrec);
    {
        const uint32_t P = T == 1 ? 1u : 8u * T;  // ranges of home buckets
        const auto part_of = [&](uint32_t home) { return (uint32_t)((uint64_t)home * P / d.num_buckets); };  // (home < num_buckets)
        std::vector<Ref> tmp(T == 1 ? 0 : nrec);
        std::vector<Ref>& first = T == 1 ? refs : tmp;
        std::vector<uint64_t> count((size_t)T * P, 0);
        parallel(nrec, [&](unsigned t, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i) {
                uint32_t lo, hi;
                record_minimizer(&d.records[i * REC_WORDS], d.k, d.m, lo, hi);
                first[i] = Ref{mulhi32(dict_hash(lo, hi, d.seed), d.num_buckets), (uint32_t)i, lmer_key(lo, hi)};
                ++count[(size_t)t * P + part_of(first[i].home)];
            }
        });
        if (T == 1) {
            std::sort(refs.begin(), refs.end(), less);
        } else {
            std::vector<uint64_t> start((size_t)T * P), part_begin(P + 1, 0);
            uint64_t run = 0;
            for (uint32_t p = 0; p < P; ++p) {
                part_begin[p] = run;
                for (unsigned t = 0; t < T; ++t) { start[(size_t)t * P + p] = run; run += count[(size_t)t * P + p]; }
            }
            part_begin[P] = run;
            parallel(nrec, [&](unsigned t, uint64_t b, uint64_t e) {
                uint64_t* at = &start[(size_t)t * P];
                for (uint64_t i = b; i < e; ++i) refs[at[part_of(tmp[i].home)]++] = tmp[i];
            });
            std::atomic<uint32_t> next{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < T; ++t)
                th.emplace_back([&] {
                    for (uint32_t p; (p = next.fetch_add(1)) < P;) std::sort(refs.begin() + part_begin[p], refs.begin() + part_begin[p + 1], less);
                });
            for (auto& x : th) x.join();
        }
    }
    d.table.clear();
    d.table.reserve((nb_hashed + nrec / 16 + 1024) * BUCKET_WORDS);  // room for the overflow region without a second copy
    d.table.resize(nb_hashed * BUCKET_WORDS);
    parallel(nb_hashed, [&](unsigned, uint64_t b0, uint64_t b1) {
        for (uint64_t b = b0; b < b1; ++b)
            for (uint32_t r = 0; r < BUCKET_RECS; ++r) {
                uint32_t* w = &d.table[b * BUCKET_WORDS + r * REC_WORDS];
                w[0] = w[1] = w[3] = 0;
                w[2] = REC_W2_EMPTY;
            }
    });
    // the records in table order (gathered by all threads: the sweep below then reads them front to back instead of
    // missing the cache once per record)
    std::vector<uint32_t> ordered(nrec * REC_WORDS);
    parallel(nrec, [&](unsigned, uint64_t b, uint64_t e) {
        for (uint64_t i = b; i < e; ++i) {
            const uint32_t* w = &d.records[(uint64_t)refs[i].rec * REC_WORDS];
            uint32_t* o = &ordered[i * REC_WORDS];
            o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = w[3] & ~REC_SPILL;
        }
    });
    auto put = [&](uint32_t* dst, uint64_t at_ref) {  // the record of refs[at_ref]
        const uint32_t* w = &ordered[at_ref * REC_WORDS];
        dst[0] = w[0]; dst[1] = w[1]; dst[2] = w[2]; dst[3] = w[3];
    };
    // The sweep over the hashed buckets, every thread a contiguous range of them: the overflow runs of a range go to the thread's own
    // piece of the overflow region, the redirects note the run's bucket number within that piece, and once the pieces' sizes are
    // known the numbers get their piece's start added. The pieces follow each other in bucket order: the table is the one a single
    // sweep writes (and the one the device builder writes: which record goes where is common/dict_place.h).
    // (A bucket met by a query leaves at most REDIRECT_DIRECT new buckets to look at if it is a hashed one, at most one — the next
    // of its run — if it is an overflow bucket, which bounds the lookup kernel's ring of waiting buckets: 64 runs x 3.)
    std::vector<std::vector<uint32_t>> piece(T);                 // overflow region by thread
    std::vector<std::vector<uint64_t>> redirects(T);             // hashed buckets that redirect, by thread
    std::vector<std::string> failure(T);
    parallel(nb_hashed, [&](unsigned t, uint64_t b0, uint64_t b1) {
        std::vector<uint32_t>& overflow = piece[t];
        uint64_t at = (uint64_t)(std::lower_bound(refs.begin(), refs.end(), b0, [](const Ref& r, uint64_t b) { return r.home < b; }) - refs.begin());
        const auto key = [&](uint64_t i) { return refs[i].key; };
        for (uint64_t b = b0; b < b1; ++b) {
            const uint64_t a = at;
            while (at < nrec && refs[at].home == b) ++at;
            if (at == a) continue;
            uint32_t* bw = &d.table[b * BUCKET_WORDS];
            const BucketPlan plan = plan_bucket(a, at, key);
            const uint64_t nb = (plan.moved + BUCKET_RECS - 1) / BUCKET_RECS;
            if (nb > REC_MAX_CSID) { failure[t] = "dictionary table: overflow run too long"; return; }
            const size_t o0 = overflow.size();
            if (nb) overflow.resize(o0 + nb * BUCKET_WORDS, 0);
            place_bucket(a, at, plan, key, [&](uint32_t slot, uint64_t i) { put(bw + slot * REC_WORDS, i); },
                         [&](uint64_t j, uint64_t i) { put(&overflow[o0 + j * REC_WORDS], i); });
            if (nb) {
                for (uint64_t j = plan.moved; j < nb * BUCKET_RECS; ++j) overflow[o0 + j * REC_WORDS + 2] = REC_W2_EMPTY;
                // the query reads the first REDIRECT_DIRECT buckets at once; further ones hang on spill flags
                for (uint64_t b2 = REDIRECT_DIRECT - 1; b2 + 1 < nb; ++b2) overflow[o0 + b2 * BUCKET_WORDS + (BUCKET_RECS - 1) * REC_WORDS + 3] |= REC_SPILL;
                uint32_t* dst = bw + (BUCKET_RECS - 1) * REC_WORDS;
                dst[0] = 0;
                dst[1] = (uint32_t)(o0 / BUCKET_WORDS);  // (within the thread's piece: its start is added below)
                dst[2] = REC_W2_REDIRECT;
                dst[3] = (uint32_t)nb;
                redirects[t].push_back(b);
            }
        }
    });
    for (const std::string& f : failure) if (!f.empty()) throw std::runtime_error(f);
    std::vector<uint32_t> overflow;  // the overflow region, appended behind the hashed region at the end
    {
        uint64_t ob = nb_hashed, words = 0;
        for (unsigned t = 0; t < T; ++t) words += piece[t].size();
        if (nb_hashed + words / BUCKET_WORDS >= DICT_MAX_BUCKETS) throw std::runtime_error("dictionary table exceeds 2^31 buckets");
        overflow.reserve(words);
        for (unsigned t = 0; t < T; ++t) {
            for (uint64_t b : redirects[t]) d.table[b * BUCKET_WORDS + (BUCKET_RECS - 1) * REC_WORDS + 1] += (uint32_t)ob;
            ob += piece[t].size() / BUCKET_WORDS;
            overflow.insert(overflow.end(), piece[t].begin(), piece[t].end());
            std::vector<uint32_t>().swap(piece[t]);
        }
    }
    d.table.insert(d.table.end(), overflow.begin(), overflow.end());
}

// host_table = false: the bucket table is left to the device builder (hip/dict_build.hip.h); only its geometry is fixed here
inline void build_dict(Dict& d, uint32_t k, uint32_t m, const char* bases, uint64_t total_bases,
                       const std::vector<uint64_t>& unitig_off, const std::vector<uint32_t>& unitig_csid,
                       unsigned nthreads = 0, bool host_table = true) {
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
                std::vector<uint8_t> code;  // the unitig strand being cut, one base per byte
                auto& out = parts[t];
                for (uint64_t u = cut[t]; u < cut[t + 1]; ++u) {
                    const uint64_t b = unitig_off[u], e = unitig_off[u + 1];
                    const int64_t len = (int64_t)(e - b);
                    if (len < (int64_t)k) { errors[t] = "unitig shorter than k"; return; }
                    if (unitig_csid[u] > REC_MAX_CSID) { errors[t] = "colour-set id exceeds record width"; return; }
                    const int64_t nm = len - m + 1, nk = len - k + 1;
                    nk_parts[t] += nk;
                    code.resize(len);
                    ord.resize(nm);
                    for (int strand = 0; strand < 2; ++strand) {
                        for (int64_t i = 0; i < len; ++i)
                            code[i] = strand == 0 ? (uint8_t)detail::string_base(d.strings, b + i)
                                                  : (uint8_t)(3u - detail::string_base(d.strings, b + (len - 1 - i)));
                        uint32_t lo = 0, hi = 0;  // rolling m-mer, base i of the m-mer at bit i
                        for (int64_t i = 0; i < len; ++i) {
                            lo = (lo >> 1) | ((uint32_t)(code[i] & 1u) << (m - 1));
                            hi = (hi >> 1) | ((uint32_t)(code[i] >> 1) << (m - 1));
                            if (i + 1 >= (int64_t)m) ord[i + 1 - m] = minimizer_order(lo & low_mask32(m), hi & low_mask32(m), m);
                        }
                        auto emit = [&](int64_t p, int64_t sa, int64_t sb) {  // k-mers sa..sb of the strand share the minimizer occurrence p
                            uint64_t clo = 0, chi = 0;  // context base c = strand base p - km + c
                            for (uint32_t c = 0; c < CL; ++c) {
                                const int64_t x = p - km + c;
                                if (x < 0 || x >= len) continue;
                                clo |= (uint64_t)(code[x] & 1u) << c;
                                chi |= (uint64_t)(code[x] >> 1) << c;
                            }
                            uint32_t s0 = (uint32_t)(sa - (p - km));
                            const uint32_t s1 = (uint32_t)(sb - (p - km));
                            auto put = [&](uint32_t a, uint32_t z) {
                                out.push_back((uint32_t)clo);
                                out.push_back((uint32_t)chi);
                                out.push_back(rec_w2(clo, chi, a));
                                out.push_back(rec_w3(unitig_csid[u], z));
                            };
                            if (strand == 0 || (k & 1u)) { put(s0, s1); return; }
                            // reverse strand, even k: a k-mer equal to its own reverse complement is already there
                            for (uint32_t sw = s0; sw <= s1 + 1; ++sw) {
                                bool self_rc = false;
                                if (sw <= s1) {
                                    const uint32_t wl = (uint32_t)(clo >> sw) & low_mask32(k), wh = (uint32_t)(chi >> sw) & low_mask32(k);
                                    self_rc = wl == rc_plane(wl, k) && wh == rc_plane(wh, k);
                                }
                                if (sw > s1 || self_rc) {
                                    if (sw > s0) put(s0, sw - 1);
                                    s0 = sw + 1;
                                }
                            }
                        };
                        int64_t run_p = -1, run_first = 0;
                        for (int64_t sk = 0; sk <= nk; ++sk) {
                            int64_t p = -1;
                            if (sk < nk) {
                                p = sk;
                                for (uint32_t jj = 1; jj <= km; ++jj)
                                    if (ord[sk + jj] < ord[p]) p = sk + jj;  // leftmost smallest order
                            }
                            if (p != run_p) {
                                if (run_p >= 0) emit(run_p, run_first, sk - 1);
                                run_p = p;
                                run_first = sk;
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
    if (host_table) build_dict_table(d);
    else { d.table.clear(); dict_table_geometry(d); }
}

// Host walk of the same structure, used ONLY by the build-time self check (verify_dict below, the
// analogue of the reference's `--check`, builder.hpp:221-277); queries never run here. It follows the
// lookup kernel step by step: leftmost smallest m-mer of the k-mer as given, home bucket, redirect and spill
// chains, every record compared in the k-mer's own orientation. Returns the number of matching records;
// *csid = the last match.
inline uint32_t dict_lookup(const Dict& d, uint32_t klo, uint32_t khi, uint32_t* csid) {
    const uint32_t k = d.k, m = d.m, km = k - m;
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t j = 0; j <= km; ++j) {
        const uint32_t lo = (klo >> j) & low_mask32(m), hi = (khi >> j) & low_mask32(m);
        best = std::min(best, (minimizer_order(lo, hi, m) << ORDER_POS_BITS) | j);
    }
    const uint32_t pm = best & ((1u << ORDER_POS_BITS) - 1u);
    const uint32_t mlo = (klo >> pm) & low_mask32(m), mhi = (khi >> pm) & low_mask32(m);
    uint32_t found = 0;
    const uint32_t s = km - pm;  // window of this k-mer in a record's context
    std::vector<uint64_t> visit(1, mulhi32(dict_hash(mlo, mhi, d.seed), d.num_buckets));
    for (size_t v = 0; v < visit.size(); ++v) {
        const uint32_t* bw = &d.table[visit[v] * BUCKET_WORDS];
        for (uint32_t r = 0; r < BUCKET_RECS; ++r) {
            const uint32_t* w = bw + r * REC_WORDS;
            if (r == BUCKET_RECS - 1 && (w[2] & 0x80000000u))  // the bucket's redirect
                for (uint32_t j = 0; j < std::min(w[3] & REC_MAX_CSID, REDIRECT_DIRECT); ++j) visit.push_back((uint64_t)w[1] + j);
            if (s < rec_smin(w[2]) || s > rec_smax(w[3])) continue;
            const uint32_t lo = (uint32_t)(rec_ctx_lo(w[0], w[2]) >> s) & low_mask32(k);
            const uint32_t hi = (uint32_t)(rec_ctx_hi(w[1], w[2]) >> s) & low_mask32(k);
            if (lo == klo && hi == khi) {
                ++found;
                *csid = w[3] & REC_MAX_CSID;
            }
        }
        if (bw[(BUCKET_RECS - 1) * REC_WORDS + 3] & REC_SPILL) visit.push_back(visit[v] + 1);
    }
    return found;
}

// every k-mer of every unitig must be found exactly once, on both strands, with its unitig's colour-set id
inline void verify_dict(const Dict& d, uint64_t stride = 1) {
    const uint32_t k = d.k;
    {   // what bounds the lookup kernel's ring of waiting buckets (k1_lookup, PAIRS): a hashed bucket hands a query on to at most
        // REDIRECT_DIRECT overflow buckets and never to the next hashed bucket; an overflow bucket to at most the next of its run
        const uint64_t nb_hashed = (uint64_t)d.num_buckets + DICT_TAIL_BUCKETS, nb = d.table.size() / BUCKET_WORDS;
        for (uint64_t b = 0; b < nb; ++b) {
            const uint32_t* last = &d.table[b * BUCKET_WORDS + (BUCKET_RECS - 1) * REC_WORDS];
            const bool redirect = (last[2] & 0x80000000u) != 0, spill = (last[3] & REC_SPILL) != 0;
            if (b < nb_hashed ? spill : redirect) throw std::runtime_error("dictionary self-check failed (a hashed bucket spills, or an overflow bucket redirects)");
        }
    }
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

// bucket statistics (tools / logs)
struct DictStats {
    uint64_t records = 0, buckets = 0, redirects = 0, overflow_buckets = 0, spill_buckets = 0;
};
inline DictStats dict_stats(const Dict& d) {
    DictStats s;
    s.records = d.num_records();
    s.buckets = d.num_buckets;
    const uint64_t nb_hashed = (uint64_t)d.num_buckets + DICT_TAIL_BUCKETS;
    s.overflow_buckets = d.table.size() / BUCKET_WORDS - nb_hashed;
    for (uint64_t b = 0; b < nb_hashed; ++b) {
        const uint32_t* bw = &d.table[b * BUCKET_WORDS];
        for (uint32_t r = 0; r < BUCKET_RECS; ++r) s.redirects += (bw[r * REC_WORDS + 2] & 0x80000000u) != 0;
        s.spill_buckets += (bw[(BUCKET_RECS - 1) * REC_WORDS + 3] & REC_SPILL) != 0;
    }
    return s;
}

}  // namespace fg
