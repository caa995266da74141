// K1: reads -> sorted distinct colour-set ids with multiplicities (gfx950, wave64).
//
// Replaces index::fetch_color_set_ids and the k-mer streaming half of pseudoalign_threshold_union
// (ps_full_intersection.cpp:334-374, ps_threshold_union.cpp:327-387) including u2c (index.hpp:37), and
// sshash::streaming_query::lookup_advanced behind them (call sites ps_full_intersection.cpp:341-352).
//
// The reference extends the previous hit along the unitig and looks a k-mer up from scratch only when that
// fails. The SIMT counterpart of that idea works on whole minimizer runs instead of single k-mers:
//   A  bases -> 2-bit planes by ballot; order of every m-mer; window minima by doubling: every k-mer knows the
//      position of its minimizer (leftmost smallest m-mer).
//   B  consecutive k-mers with the same minimizer occurrence form a RUN (a super-k-mer of the read, 1..k-m+1
//      k-mers, ~20 per 150-base read). Runs are compacted into a queue, one entry each; the runs of several
//      reads share the queue so that the next phase fills its 64 lanes.
//   C  lane = run: hash of the canonical minimizer -> ONE 64-byte bucket of four self-contained records
//      (common/kmer_common.h). The 2k-m read bases around the minimizer are compared with a record's context in
//      one XOR; the k-mers of the run that match are those whose window is free of mismatches, an INTERVAL of
//      window positions given by the highest mismatch below and the lowest mismatch above the core bases
//      k-m..k-1. So a run costs one line fetch and a handful of integer operations per record whatever its number
//      of k-mers, positive or negative; a non-ACGT base or the end of the read is just a mismatch.
//   E  every matching (run, record) pair is a "head" (colour-set id, number of k-mers), neighbouring runs with the
//      same id are folded into one; the heads of a read are reduced to sorted distinct ids with summed
//      multiplicities, several reads per pass.
// All special cases (ties between equal orders, strands, palindromic minimizers) are settled by the dictionary
// builder (host/dict_build.hpp); the kernel has no slow path.
#pragma once

namespace fg {

struct DevDict {
    const uint32_t* table;  // 64-byte buckets of four 16-byte records
    uint32_t num_buckets, k, m, seed;
};

// run descriptor in the queue: minimizer position | first k-mer << 10 | k-mers << 20 | read slot << 25
__device__ __forceinline__ uint32_t run_pack(uint32_t pm, uint32_t i0, uint32_t cnt, uint32_t g) {
    return pm | (i0 << 10) | (cnt << 20) | (g << 25);
}

constexpr int K1_GROUP = 8;  // reads per ticket = most reads whose runs share one pass of phase C

// Outputs per unit r (a read, or a segment of a read longer than 512 k-mers; relative to `first`): nids[r],
// npos[r] (# positive k-mers), idoff[r] = r * stride (fixed-stride slab: no allocation traffic between waves), and in
// the pools: ids ascending + how many positive k-mers had each id. KMER_OUT: also the colour-set id of every k-mer
// (0xFFFFFFFF = negative), the input of the reference's kmer_conservation / kmer_matches queries
// (src/kmer_conservation.cpp:7-54, src/kmer_matches.cpp:7-30).
// W13 fixes the number of m-mers per k-mer at 13 (k - m = 12, e.g. k = 31, m = 19) so that the window minima unroll.
// HALVES = 1..4: units of up to 128 * HALVES k-mers.
template <bool W13, int HALVES, bool KMER_OUT>
__global__ __launch_bounds__(256, HALVES <= 2 ? 8 : 4) void k1_lookup(DevDict d, const uint8_t* __restrict__ bases,
                                                                      const uint64_t* __restrict__ offs, uint64_t first, uint64_t n_reads,
                                                                      uint32_t* __restrict__ nids, uint32_t* __restrict__ npos,
                                                                      uint64_t* __restrict__ idoff, uint32_t* __restrict__ ids_pool,
                                                                      uint32_t* __restrict__ cnt_pool, uint32_t stride, unsigned int* tickets,
                                                                      uint32_t* __restrict__ kmer_out) {
    constexpr int KMAX = 128 * HALVES;      // k-mers per unit
    constexpr int NB = 2 * HALVES + 1;      // 64-base groups fetched per unit
    constexpr int NA = 2 * HALVES + 1;      // rounds of 64 m-mer positions (the last one: 16 positions)
    constexpr int PW = 2 * NB + 3;          // plane words per unit: one pad word in front, two behind
    constexpr int GROUP = HALVES == 1 ? K1_GROUP : (HALVES == 2 ? 4 : 2);
    constexpr int QCAP = KMAX;              // a unit has at most one run per k-mer
    constexpr int HCAP = KMAX;              // heads per pass; a single unit has at most one head per k-mer
    constexpr uint32_t POSM = (1u << ORDER_POS_BITS) - 1u;
    constexpr uint32_t FIRST = 0x80000000u;
    enum { M_UNIT = 0, M_QA = 1, M_QB = 2, M_NIDS = 3, M_NPOS = 4, M_HA = 5, M_HB = 6, M_WORDS = 8 };
    __shared__ uint32_t s_min[4][KMAX + 80];
    __shared__ uint32_t s_planes[4][GROUP][3][PW];
    __shared__ uint32_t s_queue[4][QCAP];
    __shared__ uint32_t s_hid[4][HCAP];
    __shared__ uint32_t s_hcnt[4][HCAP];   // k-mers | read slot << 16
    __shared__ uint32_t s_hres[4][HCAP];   // total of the id within its read | FIRST, 0 for repeats
    __shared__ uint32_t s_meta[4][GROUP][M_WORDS];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t* mn = s_min[wv];
    uint32_t* queue = s_queue[wv];
    uint32_t* hid = s_hid[wv];
    uint32_t* hcnt = s_hcnt[wv];
    uint32_t* hres = s_hres[wv];
    uint32_t (*meta)[M_WORDS] = s_meta[wv];
    const uint32_t k = d.k, m = d.m, km = k - m, W = W13 ? 13u : km + 1, CL = 2 * k - m;
    const uint32_t span = W13 ? 8u : 1u << (31 - __builtin_clz(W));  // largest power of two <= W (W <= 16)
    const uint32_t tail = W - span;
    const uint32_t maskm = low_mask32(m), maskkm = (1u << km) - 1u;
    const uint32_t clo_mask = CL >= 32 ? 0xFFFFFFFFu : (1u << CL) - 1u, chi_mask = CL > 32 ? (1u << (CL - 32)) - 1u : 0u;
    const uint32_t core_mask = low_mask32(k) & ~maskkm;  // context bases k-m .. k-1: part of every window
    const uint32_t rsh = 64 - CL;                         // reverse of a CL-bit field = 64-bit reverse >> rsh
    const WorkQueue wq{tickets, n_reads, (uint32_t)K1_GROUP};
    uint64_t t_first;
    uint32_t t_count;

    for (uint32_t i = lane; i < (uint32_t)KMAX + 80; i += 64) mn[i] = 0xFFFFFFFFu;  // positions past the last round stay "infinite"
    wave_lds_sync();

    while (wq.pull(t_first, t_count)) {
        const uint64_t myoff = (uint32_t)lane <= t_count ? offs[first + t_first + lane] : 0;
        uint64_t rb = readlane_u64(myoff, 0), re = readlane_u64(myoff, 1);  // wave-uniform: scalar registers
        uint32_t len = (uint32_t)(re - rb);
        const uint8_t* seq = bases + rb;
        // reads are padded by the host buffer: positions past the read end are masked below, not branched on
        uint32_t bb[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g) bb[g] = seq[lane + 64 * g];

        uint32_t q = 0, ng = 0;  // runs queued, read slots in use (wave-uniform)
        uint32_t j = 0;          // next read of the ticket
        bool pending = false;    // phases A and B of read j are done, its runs are not queued yet
        uint64_t LO[NB], HI[NB], NV[NB];
        uint32_t pos[NA - 1];
        uint64_t H[NA - 1];
        uint32_t R = 0, nk = 0;

        for (;;) {
            if (!pending && j < t_count) {
                const uint32_t cur_len = len;
                // ---- A: planes ----
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    const uint32_t b = bb[g];
                    // a letter (0x40..0x7F) whose low five bits select one of A, C, G, T; bits 1 and 2 of the letter give the code
                    const bool ok = (uint32_t)lane + 64 * g < cur_len && (b & 0xC0u) == 0x40u && ((0x0010008Au >> (b & 31u)) & 1u);
                    const uint64_t b1 = __ballot((b & 2u) != 0), b2 = __ballot((b & 4u) != 0);
                    LO[g] = b1 ^ b2;  // A0 C1 G2 T3
                    HI[g] = b2;
                    NV[g] = ~__ballot(ok);
                }
                if (j + 1 < t_count) {  // request the next read's bases now; they are consumed next iteration
                    rb = re;
                    re = readlane_u64(myoff, j + 2);
                    len = (uint32_t)(re - rb);
                    seq = bases + rb;
#pragma unroll
                    for (int g = 0; g < NB; ++g) bb[g] = seq[lane + 64 * g];
                }
                nk = cur_len >= k ? min(cur_len - k + 1, (uint32_t)KMAX) : 0;
                if (KMER_OUT) {
                    for (uint32_t i = lane; i < nk; i += 64) kmer_out[(t_first + j) * (uint64_t)stride + i] = NEG;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ordered before the ids written by phase C
                }
                // ---- A: order of every m-mer, window minima by doubling ----
                uint32_t v[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    // m-mer at position 64 a + lane out of a static word triple (alignbit shifts by lane & 31)
                    const uint32_t l0 = (uint32_t)LO[a], l1 = (uint32_t)(LO[a] >> 32), l2 = a + 1 < NB ? (uint32_t)LO[a + 1 < NB ? a + 1 : a] : 0u;
                    const uint32_t h0 = (uint32_t)HI[a], h1 = (uint32_t)(HI[a] >> 32), h2 = a + 1 < NB ? (uint32_t)HI[a + 1 < NB ? a + 1 : a] : 0u;
                    uint32_t lo, hi;
                    if (a == NA - 1) {  // 16 positions
                        lo = __builtin_amdgcn_alignbit(l1, l0, lane);
                        hi = __builtin_amdgcn_alignbit(h1, h0, lane);
                    } else {
                        const uint32_t la = __builtin_amdgcn_alignbit(l1, l0, lane), lb = __builtin_amdgcn_alignbit(l2, l1, lane);
                        const uint32_t ha = __builtin_amdgcn_alignbit(h1, h0, lane), hb = __builtin_amdgcn_alignbit(h2, h1, lane);
                        lo = lane < 32 ? la : lb;
                        hi = lane < 32 ? ha : hb;
                    }
                    v[a] = (minimizer_order(canonical_key(lo & maskm, hi & maskm, m)) << ORDER_POS_BITS) | (uint32_t)(64 * a + lane);
                    if (a < NA - 1 || lane < 16) mn[64 * a + lane] = v[a];
                }
                wave_lds_sync();
#pragma unroll
                for (uint32_t st = 1; st < span; st <<= 1) {  // in place: every lane reads before any lane writes
                    uint32_t nb_[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) nb_[a] = mn[64 * a + lane + st];
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        v[a] = min(v[a], nb_[a]);
                        if (a < NA - 1 || lane < 16) mn[64 * a + lane] = v[a];
                    }
                    wave_lds_sync();
                }
                // ---- B: runs of k-mers sharing a minimizer occurrence ----
                R = 0;
#pragma unroll
                for (int a = 0; a < NA - 1; ++a) {
                    // [i, i + span) and [i + W - span, i + W) cover the window
                    pos[a] = min(v[a], mn[64 * a + lane + tail]) & POSM;
                    const uint32_t carry = a ? (uint32_t)__builtin_amdgcn_readlane((int)pos[a ? a - 1 : 0], 63) : 0xFFFFFFFFu;
                    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)pos[a], 0x138, 0xF, 0xF, false);  // wave_shr:1
                    H[a] = __ballot((uint32_t)(64 * a + lane) < nk && pos[a] != prev);
                    R += (uint32_t)__popcll(H[a]);
                }
                pending = true;
            }
            // queue the runs of read j while they fit 64 lanes (a read with more runs than that goes alone)
            if (pending && (ng == 0 || (q + R <= 64u && ng < (uint32_t)GROUP))) {
                if (lane == 0) {
                    uint32_t* P = s_planes[wv][ng][0];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        P[p * PW] = p == 2 ? 0xFFFFFFFFu : 0u;
#pragma unroll
                        for (int g = 0; g < NB; ++g) {
                            const uint64_t x = p == 0 ? LO[g] : (p == 1 ? HI[g] : NV[g]);
                            P[p * PW + 1 + 2 * g] = (uint32_t)x;
                            P[p * PW + 2 + 2 * g] = (uint32_t)(x >> 32);
                        }
                        P[p * PW + 2 * NB + 1] = p == 2 ? 0xFFFFFFFFu : 0u;
                        P[p * PW + 2 * NB + 2] = p == 2 ? 0xFFFFFFFFu : 0u;
                    }
                    uint32_t* M = meta[ng];
                    M[M_UNIT] = j; M[M_QA] = q; M[M_QB] = q + R; M[M_NIDS] = 0; M[M_NPOS] = 0; M[M_HA] = 0; M[M_HB] = 0;
                }
                uint32_t nh[NA];  // first run head at or after k-mer 64 a (nk if none)
                nh[NA - 1] = nk;
#pragma unroll
                for (int a = NA - 2; a >= 0; --a) nh[a] = H[a] ? 64u * a + (uint32_t)__builtin_ctzll(H[a]) : nh[a + 1];
                uint32_t before = q;
#pragma unroll
                for (int a = 0; a < NA - 1; ++a) {
                    const uint32_t i = 64 * a + lane;
                    const uint64_t rest = lane == 63 ? 0ull : (H[a] >> (lane + 1));
                    const uint32_t next = rest ? i + 1u + (uint32_t)__builtin_ctzll(rest) : nh[a + 1];
                    if ((H[a] >> lane) & 1ull) queue[before + mask_rank(H[a])] = run_pack(pos[a], i, next - i, ng);
                    before += (uint32_t)__popcll(H[a]);
                }
                q += R;
                ++ng;
                ++j;
                pending = false;
                wave_lds_sync();
                if (q <= 64u && j < t_count) continue;  // try to add the next read
            }
            if (ng == 0) break;  // (only for an empty ticket)

            // ---- phases C + E over the queued runs. If the heads of a pass over several reads do not fit the head
            // buffer (adversarial input), the pass is repeated read slot by read slot: one unit always fits ----
            bool single = ng == 1;
            uint32_t g0 = 0, g1 = ng;
            for (;;) {
                const uint32_t qa = single ? (uint32_t)__builtin_amdgcn_readfirstlane((int)meta[g0][M_QA]) : 0u;
                const uint32_t qb = single ? (uint32_t)__builtin_amdgcn_readfirstlane((int)meta[g0][M_QB]) : q;
                uint32_t hcount = 0, hmain = 0;  // heads so far; heads that lie in per-slot order
                bool overflow = false;
                for (uint32_t c0 = qa; c0 < qb && !overflow; c0 += 64) {
                    const uint32_t e = c0 + lane;
                    const bool act = e < qb;
                    const uint32_t desc = act ? queue[e] : 0u;
                    const uint32_t pm = desc & POSM, i0 = (desc >> 10) & POSM, cnt = (desc >> 20) & 31u, g = desc >> 25;
                    // the CL read bases around the minimizer: base pm - km + c at field bit c (plane word 0 is padding)
                    const uint32_t o = pm + 32u - km;
                    const uint32_t* pl = s_planes[wv][g][0] + (o >> 5);
                    const uint32_t sh = o & 31u;
                    uint32_t f[3][2];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint32_t w0 = pl[p * PW], w1 = pl[p * PW + 1], w2 = pl[p * PW + 2];
                        f[p][0] = __builtin_amdgcn_alignbit(w1, w0, sh) & clo_mask;
                        f[p][1] = __builtin_amdgcn_alignbit(w2, w1, sh) & chi_mask;
                    }
                    // the minimizer itself: field bits km .. k-1 (k <= 31: inside the low word)
                    const uint32_t mlo = (f[0][0] >> km) & maskm, mhi = (f[1][0] >> km) & maskm;
                    const uint64_t kf = lmer_key(mlo, mhi), kr = lmer_key(rc_plane(mlo, m), rc_plane(mhi, m));
                    const bool qfwd = kf <= kr;
                    uint32_t bucket = mulhi32(dict_hash(qfwd ? kf : kr, d.seed), d.num_buckets);
                    // the same field on the other strand: reversed, bases complemented (the invalid mask only reversed)
                    uint32_t rv[3][2];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint32_t x1 = __builtin_bitreverse32(f[p][0]), x0 = __builtin_bitreverse32(f[p][1]);
                        uint32_t r0, r1;
                        if (rsh < 32) { r0 = __builtin_amdgcn_alignbit(x1, x0, rsh); r1 = x1 >> rsh; }
                        else { r0 = x1 >> (rsh - 32); r1 = 0; }
                        if (p < 2) { r0 = ~r0 & clo_mask; r1 = ~r1 & chi_mask; }
                        rv[p][0] = r0;
                        rv[p][1] = r1;
                    }
                    // windows of the run: window s of a record's context is k-mer i0 + s - runlo_f on the record's strand,
                    // k-mer i0 + runhi_r - s on the other one
                    const uint32_t runlo_f = i0 + km - pm, runhi_f = runlo_f + cnt - 1u;
                    const uint32_t runlo_r = km - runhi_f, runhi_r = km - runlo_f;
                    uint32_t slot_qa = 0, slot_qb = 0, r_unit = 0;
                    if (!single) { slot_qa = meta[g][M_QA]; slot_qb = meta[g][M_QB]; }
                    if (KMER_OUT) r_unit = meta[g][M_UNIT];
                    bool more = act;
                    bool main_round = c0 == qa;  // (wave-uniform) first bucket of the first 64 runs: heads come out in run order
                    while (__any(more)) {
                        const u32x4* bp = (const u32x4*)(d.table + (size_t)bucket * BUCKET_WORDS);
                        u32x4 rec[BUCKET_RECS];
#pragma unroll
                        for (int r = 0; r < (int)BUCKET_RECS; ++r) rec[r] = bp[r];
                        uint32_t hv[BUCKET_RECS], hc[BUCKET_RECS];
                        uint32_t mine = 0, msum = 0, mid = 0;
#pragma unroll
                        for (int r = 0; r < (int)BUCKET_RECS; ++r) {
                            const uint32_t w0 = rec[r].x, w1 = rec[r].y, w2 = rec[r].z;
                            const bool same = rec_fwd(w2) == qfwd;
                            const uint32_t a_lo0 = same ? f[0][0] : rv[0][0], a_hi0 = same ? f[1][0] : rv[1][0];
                            const uint32_t a_lo1 = same ? f[0][1] : rv[0][1], a_hi1 = same ? f[1][1] : rv[1][1];
                            const uint32_t a_nv0 = same ? f[2][0] : rv[2][0], a_nv1 = same ? f[2][1] : rv[2][1];
                            const uint32_t x0 = (a_lo0 ^ w0) | (a_hi0 ^ w1) | a_nv0;
                            const uint32_t x1 = (a_lo1 ^ (w2 & 0x7FFu)) | (a_hi1 ^ ((w2 >> 11) & 0x7FFu)) | a_nv1;
                            // mismatches below the core bound the windows from below, those above it from above
                            const uint32_t A = x0 & maskkm;
                            const uint32_t sl = 31u - (uint32_t)__builtin_clz((A << 1) | 1u);
                            const uint32_t B = __builtin_amdgcn_alignbit(x1, x0, k) & maskkm;
                            const uint32_t su = (uint32_t)__builtin_ctz(B | (1u << km));
                            const uint32_t lo = max(max(sl, rec_smin(w2)), same ? runlo_f : runlo_r);
                            const uint32_t hi = min(min(su, rec_smax(w2)), same ? runhi_f : runhi_r);
                            const bool hit = more && (x0 & core_mask) == 0 && lo <= hi;
                            hv[r] = rec[r].w & REC_MAX_CSID;
                            hc[r] = hit ? hi - lo + 1u : 0u;
                            mine += hit;
                            msum += hc[r];
                            mid = hit ? hv[r] : mid;
                            if (KMER_OUT && hit) {
                                for (uint32_t s = lo; s <= hi; ++s) {
                                    const uint32_t i = same ? s + pm - km : pm - s;
                                    kmer_out[(t_first + r_unit) * (uint64_t)stride + i] = hv[r];
                                }
                            }
                        }
                        // Heads in run order. In the main round a run with one matching record whose left neighbour (same
                        // read) matched the same id alone is folded into that neighbour: consecutive runs mostly sit on the same
                        // unitig, or on unitigs of one colour set.
                        bool follower = false;
                        uint64_t F = 0;
                        if (main_round) {
                            const uint32_t key = mine == 1 ? mid : 0xFFFFFFFFu;  // (ids have 31 bits)
                            const uint32_t pkey = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x138, 0xF, 0xF, false);  // wave_shr:1
                            const uint32_t pg = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)g, 0x138, 0xF, 0xF, false);
                            follower = mine == 1 && pkey == key && pg == g;
                            F = __ballot(follower);
                        }
                        // one scan for both: heads written (low half), k-mers of single-match runs (high half)
                        const uint32_t emit = follower ? 0u : mine;
                        const uint32_t packed = emit | ((mine == 1 ? msum : 0u) << 16);
                        const uint32_t incl = wave_incl_scan_u32(packed);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) & 0xFFFFu;
                        if (hcount + total > (uint32_t)HCAP && !single) { overflow = true; break; }
                        // a leader takes the k-mers of the followers right behind it: scan value of the last one (all lanes
                        // take part in the exchange: ds_bpermute reads 0 from lanes that are masked off)
                        const uint64_t nf = lane == 63 ? ~0ull : ~(F >> (lane + 1));
                        const uint32_t last = (uint32_t)lane + (uint32_t)__builtin_ctzll(nf);
                        const uint32_t s_last = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(last << 2), (int)incl) >> 16;
                        uint32_t at = hcount + (incl & 0xFFFFu) - emit;
                        if (mine == 1 && !follower) {
                            const uint32_t tot = s_last - (incl >> 16) + msum;
                            if (at < (uint32_t)HCAP) { hid[at] = mid; hcnt[at] = tot | (g << 16); }
                        } else if (mine > 1) {
#pragma unroll
                            for (int r = 0; r < (int)BUCKET_RECS; ++r) {
                                if (hc[r]) {
                                    if (at < (uint32_t)HCAP) { hid[at] = hv[r]; hcnt[at] = hc[r] | (g << 16); }
                                    ++at;
                                }
                            }
                        }
                        if (main_round && !single) {  // where the heads of every read slot begin and end
                            const uint32_t begin = hcount + (incl & 0xFFFFu) - emit;
                            if (act && e == slot_qa) meta[g][M_HA] = begin;
                            if (act && e + 1 == slot_qb) meta[g][M_HB] = begin + emit;
                        }
                        hcount = min(hcount + total, (uint32_t)HCAP);
                        if (main_round) hmain = hcount;
                        main_round = false;
                        more = more && (rec[BUCKET_RECS - 1].w & REC_SPILL) != 0;  // some record of this key may live in the next bucket
                        bucket += more ? 1u : 0u;
                    }
                }
                if (overflow) {  // start over, one read slot at a time
                    single = true;
                    g0 = 0;
                    g1 = 1;
                    continue;
                }
                if (single) hmain = hcount;
                if (single && lane == 0) { meta[g0][M_HA] = 0; meta[g0][M_HB] = hcount; }
                wave_lds_sync();
                // ---- E: sorted distinct ids with summed multiplicities, per read slot ----
                uint32_t maxseg = 0;
                for (uint32_t g = g0; g < g1; ++g)
                    maxseg = max(maxseg, (uint32_t)__builtin_amdgcn_readfirstlane((int)(meta[g][M_HB] - meta[g][M_HA])));
                // pass 1: total of the head's id within its read; FIRST = no earlier head of the read has that id. A head is
                // compared with the heads of its own slot and with the few heads that came out of later buckets.
                for (uint32_t h0 = 0; h0 < hcount; h0 += 64) {
                    const uint32_t h = h0 + lane;
                    const bool act = h < hcount;
                    const uint32_t vv = act ? hid[h] : 0u, gsel = act ? hcnt[h] >> 16 : g0;
                    const uint32_t ha = meta[gsel][M_HA], hb = meta[gsel][M_HB];
                    uint32_t tot = 0;
                    bool firsth = act;
                    for (uint32_t t = 0; t < maxseg; ++t) {
                        const uint32_t i = ha + t;
                        const uint32_t vi = hid[min(i, (uint32_t)HCAP - 1u)], ci = hcnt[min(i, (uint32_t)HCAP - 1u)];
                        const bool eq = i < hb && vi == vv;
                        tot += eq ? (ci & 0xFFFFu) : 0u;
                        firsth = firsth && !(eq && i < h);
                    }
                    for (uint32_t i = hmain; i < hcount; ++i) {
                        const uint32_t vi = hid[i], ci = hcnt[i];
                        const bool eq = vi == vv && (ci >> 16) == gsel;
                        tot += eq ? (ci & 0xFFFFu) : 0u;
                        firsth = firsth && !(eq && i < h);
                    }
                    if (act) hres[h] = firsth ? (tot | FIRST) : 0u;
                }
                wave_lds_sync();
                // pass 2: rank of every first head among the first heads of its read = its place in the sorted list
                for (uint32_t h0 = 0; h0 < hcount; h0 += 64) {
                    const uint32_t h = h0 + lane;
                    const bool act = h < hcount;
                    const uint32_t vv = act ? hid[h] : 0u, gsel = act ? hcnt[h] >> 16 : g0;
                    const uint32_t ha = meta[gsel][M_HA], hb = meta[gsel][M_HB];
                    const uint32_t res = act ? hres[h] : 0u;
                    uint32_t rank = 0;
                    for (uint32_t t = 0; t < maxseg; ++t) {
                        const uint32_t i = min(ha + t, (uint32_t)HCAP - 1u);
                        rank += (ha + t < hb && (hres[i] & FIRST) != 0 && hid[i] < vv) ? 1u : 0u;
                    }
                    for (uint32_t i = hmain; i < hcount; ++i)
                        rank += ((hres[i] & FIRST) != 0 && (hcnt[i] >> 16) == gsel && hid[i] < vv) ? 1u : 0u;
                    if (res & FIRST) {
                        const uint64_t base = (t_first + meta[gsel][M_UNIT]) * (uint64_t)stride;
                        ids_pool[base + rank] = vv;
                        cnt_pool[base + rank] = res & ~FIRST;
                        atomicAdd(&meta[gsel][M_NIDS], 1u);
                        atomicAdd(&meta[gsel][M_NPOS], res & ~FIRST);
                    }
                }
                wave_lds_sync();
                if ((uint32_t)lane >= g0 && (uint32_t)lane < g1) {
                    const uint64_t r = t_first + meta[lane][M_UNIT];
                    nids[r] = meta[lane][M_NIDS];
                    npos[r] = meta[lane][M_NPOS];
                    idoff[r] = r * (uint64_t)stride;
                }
                wave_lds_sync();
                if (!single || g1 >= ng) break;
                ++g0;
                ++g1;
            }
            q = 0;
            ng = 0;
            if (!pending && j >= t_count) break;
        }
    }
}

}  // namespace fg
