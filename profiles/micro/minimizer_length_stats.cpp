// What would a dictionary with another minimizer length look like? Super-k-mer records per key, and how many of them would not fit
// their bucket, for (m, records per bucket, buckets per record). usage: mstats <index.fgidx> k
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <thread>
#include "host/index_io.hpp"
using namespace fg;
int main(int argc, char** argv) {
    HostIndex idx;
    open_index(argv[1], idx, 8);
    const Dict& d = idx.dict;
    const uint32_t k = d.k;
    printf("k %u, unitigs %llu, k-mers %llu\n", k, (unsigned long long)d.num_unitigs(), (unsigned long long)d.num_kmers);
    for (uint32_t m : {17u, 15u, 13u, 12u, 11u}) {
        const uint32_t km = k - m;
        std::vector<uint64_t> keys;  // one per record
        std::vector<uint8_t> code;
        std::vector<uint32_t> ord;
        for (uint64_t u = 0; u < d.num_unitigs(); ++u) {
            const uint64_t b = d.unitig_off[u], e = d.unitig_off[u + 1];
            const int64_t len = (int64_t)(e - b), nm = len - m + 1, nk = len - k + 1;
            code.resize(len); ord.resize(nm);
            std::vector<uint32_t> mlo(nm), mhi(nm);
            for (int strand = 0; strand < 2; ++strand) {
                for (int64_t i = 0; i < len; ++i)
                    code[i] = strand == 0 ? (uint8_t)detail::string_base(d.strings, b + i) : (uint8_t)(3u - detail::string_base(d.strings, b + (len - 1 - i)));
                uint32_t lo = 0, hi = 0;
                for (int64_t i = 0; i < len; ++i) {
                    lo = (lo >> 1) | ((uint32_t)(code[i] & 1u) << (m - 1));
                    hi = (hi >> 1) | ((uint32_t)(code[i] >> 1) << (m - 1));
                    if (i + 1 >= (int64_t)m) { ord[i + 1 - m] = minimizer_order(lo & low_mask32(m), hi & low_mask32(m), m); mlo[i + 1 - m] = lo & low_mask32(m); mhi[i + 1 - m] = hi & low_mask32(m); }
                }
                int64_t run_p = -1;
                for (int64_t sk = 0; sk < nk; ++sk) {
                    int64_t p = sk;
                    for (uint32_t jj = 1; jj <= km; ++jj) if (ord[sk + jj] < ord[p]) p = sk + jj;
                    if (p != run_p) { keys.push_back(lmer_key(mlo[p], mhi[p])); run_p = p; }
                }
            }
        }
        const uint64_t nrec = keys.size();
        std::sort(keys.begin(), keys.end());
        // records per key
        std::vector<std::pair<uint64_t, uint32_t>> kc;  // (key, count)
        for (uint64_t i = 0; i < nrec;) { uint64_t j = i; while (j < nrec && keys[j] == keys[i]) ++j; kc.push_back({keys[i], (uint32_t)(j - i)}); i = j; }
        uint64_t h[6] = {0, 0, 0, 0, 0, 0}, hr[6] = {0, 0, 0, 0, 0, 0};
        uint32_t maxc = 0;
        for (auto& x : kc) { uint32_t c = std::min(x.second, 5u); h[c]++; hr[c] += x.second; maxc = std::max(maxc, x.second); }
        printf("m %u (W %u): records %llu (%.3f per k-mer and strand), keys %llu; keys with 1/2/3/4/5+ records %.1f %.1f %.1f %.1f %.1f %%; records in such keys %.1f %.1f %.1f %.1f %.1f %%; max %u\n",
               m, km + 1, (unsigned long long)nrec, (double)nrec / (2.0 * d.num_kmers), (unsigned long long)kc.size(), 100.0 * h[1] / kc.size(), 100.0 * h[2] / kc.size(),
               100.0 * h[3] / kc.size(), 100.0 * h[4] / kc.size(), 100.0 * h[5] / kc.size(), 100.0 * hr[1] / nrec, 100.0 * hr[2] / nrec, 100.0 * hr[3] / nrec,
               100.0 * hr[4] / nrec, 100.0 * hr[5] / nrec, maxc);
        // placement: R records per bucket, factor buckets per record; a bucket whose keys' records do not all fit keeps whole keys (fewest
        // records first) in R - 1 slots and sends the rest to overflow; records in overflow = lookups that need a second fetch
        for (uint32_t R : {4u, 2u}) for (double factor : {1.625, 3.25}) {
            const uint64_t nb = (uint64_t)(nrec * factor * (R == 2 ? 1.0 : 1.0));
            std::vector<std::pair<uint32_t, uint32_t>> hk(kc.size());  // (home, count)
            for (size_t i = 0; i < kc.size(); ++i) hk[i] = {mulhi32(dict_hash((uint32_t)kc[i].first, (uint32_t)(kc[i].first >> 32), 0), (uint32_t)nb), kc[i].second};
            std::sort(hk.begin(), hk.end());
            uint64_t over = 0, redirects = 0;
            for (size_t i = 0; i < hk.size();) {
                size_t j = i; uint64_t total = 0;
                while (j < hk.size() && hk[j].first == hk[i].first) total += hk[j++].second;
                if (total > R) {
                    ++redirects;
                    std::vector<uint32_t> c; for (size_t t = i; t < j; ++t) c.push_back(hk[t].second);
                    std::sort(c.begin(), c.end());
                    uint32_t slot = 0; size_t kept = 0;
                    while (kept < c.size() && slot + c[kept] <= R - 1) slot += c[kept++];
                    for (size_t t = kept; t < c.size(); ++t) over += c[t];
                }
                i = j;
            }
            printf("   %u records per bucket of %u bytes, %.3f buckets per record (table %.0f MB): %.2f %% of the records behind a redirect, %.2f %% of the buckets redirect\n",
                   R, R * (R == 2 ? 32 : 16), factor, nb * (R == 2 ? 64.0 : 64.0) / 1e6, 100.0 * over / nrec, 100.0 * redirects / nb);
        }
    }
    return 0;
}
