// K1: reads -> sorted distinct colour-set ids with multiplicities (gfx950, wave64).
//
// Replaces index::fetch_color_set_ids and the k-mer streaming half of pseudoalign_threshold_union
// (ps_full_intersection.cpp:334-374, ps_threshold_union.cpp:327-387) including u2c (index.hpp:37), and
// sshash::streaming_query::lookup_advanced behind them (call sites ps_full_intersection.cpp:341-352).
//
// The reference extends the previous hit along the unitig and looks a k-mer up from scratch only when that
// fails. The SIMT counterpart of that idea works on whole minimizer runs instead of single k-mers. A wave pulls a TICKET of
// consecutive units (reads; 32 of them for reads of up to 128 k-mers) and goes through
//   T  the bases of the whole ticket — the units lie one behind the other in the base buffer — become bit planes in LDS at once:
//      16 bases per lane and 16-byte load, the two code bits of four letters gathered with one multiplication (encode16); no
//      ballots, one global-memory wait per ticket instead of one per read.
//   A  per unit: order of every m-mer (a three-instruction multiplicative hash, cut out of the planes with one v_alignbit per
//      plane), window minima in two rounds through LDS: every k-mer knows the position of its minimizer (leftmost smallest m-mer).
//   B  consecutive k-mers with the same minimizer occurrence form a RUN (a super-k-mer of the read, 1..k-m+1
//      k-mers, 15.9 per 150-base read at m = 17). Runs are compacted into a queue, one entry each; the runs of
//      several units share the queue so that the next phase fills its 64 lanes (3.4 reads per pass).
//   C  lane = run: hash of the minimizer (as read: the dictionary holds both strands) -> ONE 64-byte bucket of four
//      self-contained records (common/kmer_common.h). The 2k-m read bases around the minimizer are compared with a record's
//      context in one XOR per plane; the k-mers of the run that match are those whose window is free of mismatches, an INTERVAL
//      of window positions given by the highest mismatch below and the lowest mismatch above the core bases
//      k-m..k-1. So a run costs one line fetch and a handful of integer operations per record whatever its number
//      of k-mers, positive or negative; a non-ACGT base or the end of the read is just a mismatch: no canonical forms, no
//      reverse complements, no strand cases.
//      A bucket whose keys do not fit keeps a redirect in its last slot: the (run, overflow bucket) pairs behind it are handed to
//      the lanes the queue left free, which borrow the run's registers with ds_bpermute and go through the same comparison in
//      the same pass.
//   E  every matching (run, record) pair is a "head" (colour-set id, number of k-mers), neighbouring runs with the
//      same id are folded into one; the place of an id in its read's sorted list is the number of heads of the read with a
//      smaller id, heads with the same id meet at the same place, where an LDS add sums their k-mers; several reads per pass.
// The kernel has no slow path; what would need one (strands, k-mers equal to their reverse complement) is settled by
// the dictionary builder (host/dict_build.hpp).
#pragma once

namespace fg {

struct DevDict {
    const uint32_t* table;  // 64-byte buckets of four 16-byte records
    uint32_t num_buckets, k, m, seed;
};

// run descriptor in the queue: minimizer position within the ticket's span of bases (13 bits) | its distance from the run's
// first k-mer << 13 (0 .. k - m) | k-mers << 17 (filled in by phase C) | read slot << 22
constexpr uint32_t RUN_POS_BITS = 13;
constexpr uint32_t RUN_POS_MASK = (1u << RUN_POS_BITS) - 1u;
__device__ __forceinline__ uint32_t run_pack(uint32_t pa, uint32_t dm, uint32_t g) { return pa | (dm << RUN_POS_BITS) | (g << 22); }
__device__ __forceinline__ uint32_t run_first_kmer(uint32_t desc) { return (desc & RUN_POS_MASK) - ((desc >> RUN_POS_BITS) & 15u); }

// ---- bases -> bit planes, 16 bases per lane (one 16-byte load), no ballots ----
// The reads of a ticket lie one behind the other in the base buffer, so a wave turns the whole span into bit planes at once:
// bit p of a plane = base p of the span. Per dword (4 letters): bits 1 and 2 of a letter give its code (A0 C1 G2 T3 =
// bit1 ^ bit2, bit2); the four bits of a dword are gathered with a multiplication (bit 8i + c of the source lands on bit
// 24 + i; two dwords share one product, the second shifted by four: its bits land on 28 + i; the stray partial products fall
// on distinct bits below 24 or leave the word), a letter is valid if, case folded, it is the letter its own code stands for
// (v_perm as a four-entry table).
__device__ __forceinline__ uint32_t gather_bits(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3, uint32_t mult) {
    const uint32_t p0 = (m0 | (m1 << 4)) * mult, p1 = (m2 | (m3 << 4)) * mult;
    return __builtin_amdgcn_perm(p1, p0, 0x0C0C0703u);  // byte 3 of p0 | byte 3 of p1 << 8
}
__device__ __forceinline__ void encode16(u32x4 x, uint32_t& lo16, uint32_t& hi16, uint32_t& bad) {
    uint32_t lm[4], hm[4];
    bad = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t a = x[j] >> 1;
        lm[j] = (x[j] ^ a) & 0x02020202u;  // bit 1 of every byte: bit1 ^ bit2 of the letter
        hm[j] = x[j] & 0x04040404u;        // bit 2 of every byte
        const uint32_t expect = __builtin_amdgcn_perm(0x47544341u, 0x47544341u, a & 0x03030303u);  // (letter >> 1) & 3: A C T G
        bad |= (x[j] & 0xDFDFDFDFu) ^ expect;
    }
    lo16 = gather_bits(lm[0], lm[1], lm[2], lm[3], 0x00810204u);  // 8 i + 1 -> 24 + i
    hi16 = gather_bits(hm[0], hm[1], hm[2], hm[3], 0x00408102u);  // 8 i + 2 -> 24 + i
}
// the same for the letters that are not nucleotides (the rare path): a byte of `d` that is not zero -> its bit
__device__ __forceinline__ uint32_t invalid16(u32x4 x) {
    uint32_t nz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t expect = __builtin_amdgcn_perm(0x47544341u, 0x47544341u, (x[j] >> 1) & 0x03030303u);
        const uint32_t d = (x[j] & 0xDFDFDFDFu) ^ expect;
        nz[j] = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) >> 6) & 0x02020202u;  // bit 1 of every byte that is not zero
    }
    return gather_bits(nz[0], nz[1], nz[2], nz[3], 0x00810204u);
}

#ifndef FG_K1_TICKET
#define FG_K1_TICKET 32  // a pass of 64 run lanes takes three or four 150-base reads (15.9 runs each at m = 17): long tickets leave few short passes behind (9: 7.47, 16: 7.25, 32: 7.11, 48: 7.10 ms per 10 M reads)
#endif
constexpr uint32_t K1_TICKET = FG_K1_TICKET;  // reads per pull from the work queue
constexpr uint32_t K1_RUN_LANES = 64;  // runs a pass of several reads may hold: all lanes (leaving 2 / 4 / 7 lanes to the overflow buckets of the first batch: 7.15 / 7.19 / 7.29 ms against 7.11)

#ifdef FG_K1_STATS  // instrumented build (profiles/k1_stats.py): how often every loop of the kernel runs
__device__ unsigned long long k1_stats[16];
#define K1_STAT(i, v) do { if (lane == 0) atomicAdd(&k1_stats[i], (unsigned long long)(v)); } while (0)
#else
#define K1_STAT(i, v) do { } while (0)
#endif

// Outputs per unit r (a read, or a segment of a read longer than 512 k-mers; relative to `first`): nids[r],
// npos[r] (# positive k-mers), idoff[r] = r * stride (fixed-stride slab: no allocation traffic between waves), and in
// the pools: ids ascending + how many positive k-mers had each id. KMER_OUT: also the colour-set id of every k-mer
// (0xFFFFFFFF = negative), the input of the reference's kmer_conservation / kmer_matches queries
// (src/kmer_conservation.cpp:7-54, src/kmer_matches.cpp:7-30).
// WFIX fixes the number of m-mers per k-mer at K1_WFIX = 15 (k - m = 14, e.g. k = 31, m = 17) so that the window minima unroll.
#ifndef FG_K1_WAVES
#define FG_K1_WAVES 6  // waves per SIMD the register allocation aims at (measured: profiles/r2)
#endif
constexpr uint32_t K1_WFIX = 15;
// WIDE: tables of more than DICT_NARROW_BUCKETS buckets (the ring keeps a pair's source lane in a word of its own).
// Q = 2, 3, 4, 6, 8: units of up to 64 * Q k-mers (round 6: 3 and 6 — reads of 159 to 222 bases at k = 31 ran the 256-k-mer
// instantiation, reads of 287 to 414 bases the 512-k-mer one).
// SHORT (Q = 2 only): every unit of the launch has at most 114 k-mers (reads of up to 144 bases): two rounds of m-mer positions instead of three.
template <bool WFIX, int Q, bool KMER_OUT, bool WIDE = false, bool SHORT = false>
__global__ __launch_bounds__(256, Q <= 4 ? FG_K1_WAVES : (Q <= 6 ? 5 : 4)) void k1_lookup(DevDict d, const uint8_t* __restrict__ bases,
                                                                      const uint64_t* __restrict__ offs, uint64_t first, uint64_t n_reads,
                                                                      uint32_t* __restrict__ nids, uint32_t* __restrict__ npos,
                                                                      uint64_t* __restrict__ idoff, uint32_t* __restrict__ ids_pool,
                                                                      uint32_t* __restrict__ cnt_pool, uint32_t stride, unsigned int* tickets,
                                                                      uint32_t* __restrict__ kmer_out) {
    foreign_writes_acquire();  // (the bases and offsets may have been written by a copy engine that the HIP runtime knows nothing about)
    constexpr int HALVES = Q <= 4 ? (Q + 1) / 2 : 4;  // 1, 2, 2, 4, 4: which set of tuning constants applies
    constexpr int KMAX = 64 * Q;            // k-mers per unit
#ifndef FG_K1_TICKET2
#define FG_K1_TICKET2 6  // (units of up to 256 k-mers; round 6: tickets of 12 / 16 are 7 % slower at 159-286 bases, profiles/r6/read_length_sweep_variants_r6.txt)
#endif
#ifndef FG_K1_TICKET4
#define FG_K1_TICKET4 2  // (units of up to 512 k-mers; round 6: 1 / 2 / 4 units per ticket: 123 / 136 / 135 G k-mers/s at 300 bases, 157 / 176 / 169 at 400, profiles/r6/read_length_sweep_variants_r6.txt)
#endif
#ifndef FG_K1_TICKET3
#define FG_K1_TICKET3 16  // (units of up to 192 k-mers; round 6: 6 / 8 / 12 / 16 units per ticket: 178 / 176 / 181 / 184 G k-mers/s at 159 bases, profiles/r6/read_length_sweep_variants_r6.txt)
#endif
#ifndef FG_K1_TICKET6
#define FG_K1_TICKET6 4
#endif
    constexpr uint32_t TICKET = Q == 2 ? K1_TICKET : (Q == 3 ? FG_K1_TICKET3 : (Q == 4 ? FG_K1_TICKET2 : (Q == 6 ? FG_K1_TICKET6 : FG_K1_TICKET4)));  // longer units: one to three per pass, and the planes of a ticket's span live in LDS: short tickets
    constexpr int NA = Q + 1;               // rounds of 64 m-mer positions (the last one: 16 positions)
    constexpr int SPAN_BASES = (int)TICKET * (KMAX + 30) + 16;  // the units of a ticket + what the 16-byte alignment of its first load adds
    constexpr int NIT = (SPAN_BASES + 1023) / 1024;              // rounds of 64 lanes x 16 bases that cover the span
    constexpr int SPW = 1 + 32 * NIT + 3;   // plane words: one pad word in front, three behind (a context reaches 14 bases past its read)
    constexpr int GROUP = Q == 2 ? 6 : (Q == 3 ? 4 : (Q == 4 ? 3 : (Q == 6 ? 2 : 1)));  // most reads whose runs share one pass of phase C
    constexpr int NSLOT = GROUP + 1;        // read slots (a ring): the reads of a pass plus the read waiting for the next one
    constexpr int QCAP = KMAX + 64;         // the runs of a pass (at most 64, or one unit: at most one run per k-mer) + those of the next read
    constexpr int HCAP = KMAX;              // heads per pass; a single unit has at most one head per k-mer
    constexpr uint32_t POSM = (1u << ORDER_POS_BITS) - 1u;
    constexpr uint32_t FIRST = 0x80000000u;
    // ring of (bucket << 6 | source lane) pairs waiting for a lane. It cannot overflow: the 64 run lanes of a pass look at hashed
    // buckets, which leave at most REDIRECT_DIRECT = 3 overflow buckets each (192 pairs), and an overflow bucket that is looked at
    // leaves at most one (the next of its run) in the place of the pair it took (host/dict_build.hpp builds it so; fgpu_selfcheck
    // verifies it)
    constexpr uint32_t PAIRS = 256;
    // M_UNIT: unit within the ticket | span position of its first base << 8; M_KEND: span position behind its last k-mer's first base
    enum { M_UNIT = 0, M_QA = 1, M_QB = 2, M_NIDS = 3, M_NPOS = 4, M_HA = 5, M_HB = 6, M_KEND = 7, M_WORDS = 8 };
    // one block of LDS per wave, every array at a constant offset from the wave's base address (one address register
    // serves them all)
    struct WaveLds {
        union {                          // (phase A and phase C never overlap)
            uint32_t mn[KMAX + 80];      // window minima of the m-mer orders
            uint32_t pairs[PAIRS];
        };
        uint32_t span[3][SPW];           // the bases of the ticket's units as bit planes: lo, hi, invalid (bit 32 + p = base p behind the ticket's first 16-byte boundary)
        uint32_t queue[QCAP + 1];        // run descriptors (+ one: a lane also looks at the entry behind its own)
        uint32_t hid[HCAP];              // heads: colour-set id
        uint32_t hcnt[HCAP];             //        k-mers | read slot << 16 | place of the slot in the pass << 20
        // (the 512-k-mer instantiation keeps hres in the place of mn — phase E runs when the window minima are long dead —: 38 instead of
        // 46 KB of LDS per block, four blocks per CU instead of three)
        uint32_t hres[HALVES >= 4 ? 1 : HCAP];  // passes of at most 64 heads: id per place; longer passes: total of the head's id within its read | FIRST, 0 for repeats
        uint32_t hsrt[64];               // (passes of at most 64 heads) k-mers per place
        uint32_t meta[NSLOT][M_WORDS];
        uint32_t psrc[WIDE ? PAIRS : 1]; // WIDE: source lanes of the ring's pairs (the bucket numbers take the whole word of `pairs`)
    };
    __shared__ WaveLds s_lds[4];
    const int lane = lane_id();
    WaveLds& L = s_lds[__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))];
    uint32_t* mn = L.mn;
    uint32_t* queue = L.queue;
    uint32_t* pairs = L.pairs;
    uint32_t* hid = L.hid;
    uint32_t* hcnt = L.hcnt;
    static_assert(HALVES < 4 || KMAX + 80 >= HCAP, "hres takes the place of mn");
    uint32_t* hres = HALVES >= 4 ? L.mn : L.hres;
    uint32_t* hsrt = L.hsrt;
    uint32_t (*meta)[M_WORDS] = L.meta;
    const uint32_t k = d.k, m = d.m, km = k - m, W = WFIX ? K1_WFIX : km + 1, CL = 2 * k - m;
    const uint32_t span = WFIX ? 8u : 1u << (31 - __builtin_clz(W));  // largest power of two <= W (W <= 16)
    const uint32_t tail = W - span;
    const uint32_t maskm = low_mask32(m), maskkm = (1u << km) - 1u;
    const uint32_t clo_mask = CL >= 32 ? 0xFFFFFFFFu : (1u << CL) - 1u, chi_mask = CL > 32 ? (1u << (CL - 32)) - 1u : 0u;
    const uint32_t maskk = low_mask32(k);
    const WorkQueue wq{tickets, n_reads, TICKET};
    uint64_t t_first;
    uint32_t t_count;

    uint32_t gs = 0;  // first read slot of the current pass (ring index)
    while (wq.pull(t_first, t_count)) {
        // the ticket's unit offsets stay in the lanes (lane i: offset of unit i; the low words are enough inside a ticket): a unit's
        // offset is one v_readlane away, not an LDS round trip
        uint32_t off_lo, off_hi;
        {
            const uint64_t x = offs[first + t_first + min((uint32_t)lane, t_count)];
            off_lo = (uint32_t)x;
            off_hi = (uint32_t)(x >> 32);
        }
        // ---- the bases of the ticket -> bit planes (every lane 16 bases per round; two lanes make a plane word) ----
        const uint32_t sb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)off_lo) & ~15u;  // the span begins at the 16-byte boundary in front of the first unit
        const uint32_t sb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)off_hi);
        {
            const uint64_t sb = ((uint64_t)sb_hi << 32) | sb_lo;
            const uint64_t se = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)off_hi, (int)t_count) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)off_lo, (int)t_count);
            // (units are at most KMAX + 30 bases long. At least one chunk: a ticket whose units are all empty — a batch that is one read
            // without bases — has se = sb, and "nchunks - 1" below would let every lane read its own chunk, 1 KB per round, off a
            // buffer that may be 1 KB long; at the end of a mapped block that is a memory fault, seen once in a dozen runs of the
            // ragged-reads test)
            const uint32_t nchunks = max(1u, min((uint32_t)((se - sb + 15u) >> 4), (uint32_t)NIT * 64u));
            const u32x4* src = (const u32x4*)(bases + sb);
            u32x4 x[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const uint32_t c = 64u * it + (uint32_t)lane;
                x[it] = src[min(c, nchunks - 1u)];  // (an empty span reads its first 16 bytes, inside the buffer's slack)
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (it > 0 && 64u * it >= nchunks) break;  // (wave-uniform) rounds behind the ticket's last base: nothing of them is looked at
                uint32_t lo16, hi16, bad;
                encode16(x[it], lo16, hi16, bad);
                uint32_t iv16 = 0;
                if (__any(bad != 0)) iv16 = invalid16(x[it]);  // (rare: a letter that is not a nucleotide among the 1024 of this round)
                const uint32_t mine = lo16 | (hi16 << 16);
                const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]: the lane's partner
                const uint32_t iother = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)iv16, 0xB1, 0xF, 0xF, false);
                if (!(lane & 1)) {
                    uint32_t* W = &L.span[0][1 + 32 * it + (lane >> 1)];
                    W[0] = __builtin_amdgcn_perm(other, mine, 0x05040100u);
                    W[SPW] = __builtin_amdgcn_perm(other, mine, 0x07060302u);
                    W[2 * SPW] = iv16 | (iother << 16);
                }
            }
        }
        wave_lds_sync();
        uint32_t q = 0, ng = 0;  // runs queued, read slots in use (wave-uniform)
        // Reads are taken one by one: phases A and B put the runs of read j behind the queued ones; if they do not fit the
        // 64 lanes of a pass any more (or the ticket is over: j = t_count), the queued reads go through phases C and E first
        // and the new runs move to the front of the queue.
        for (uint32_t j = 0; j <= t_count;) {
            const uint32_t ws = (gs + ng) % (uint32_t)NSLOT;  // slot of read j
            uint32_t R = 0xFFFFu;
            // Adversarial input (a homopolymer has one run per k-mer): the queue holds a unit's runs behind the waiting ones only up
            // to QCAP. If they do not fit, the waiting units go through phases C and E first and this unit takes its turn again.
            bool retry = false;
            if (j < t_count) {
                const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)off_lo, (int)j);
                const uint32_t cur_len = (uint32_t)__builtin_amdgcn_readlane((int)off_lo, (int)j + 1) - o0;  // (reads are shorter than 4 GB)
                // span position of the unit's first base (bit 32 of the planes is the span's first base: one pad word in front;
                // the difference of the low words is the difference: a span is far shorter than 4 GB)
                const uint32_t base = o0 - sb_lo + 32u;
                const uint32_t nk = cur_len >= k ? min(cur_len - k + 1, (uint32_t)KMAX) : 0;
                if (KMER_OUT) {
                    for (uint32_t i = lane; i < nk; i += 64) kmer_out[(t_first + j) * (uint64_t)stride + i] = NEG;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ordered before the ids written by phase C
                }
                // ---- A: order of every m-mer, window minima by doubling ----
                // rounds of 64 m-mer positions this unit needs (wave-uniform): its nk k-mers look at m-mers 0 .. nk + W - 2, and the window
                // minima read up to 15 positions past a k-mer's first m-mer; a unit at the short end of an instantiation (129 k-mers in the
                // one for up to 256) leaves the last rounds out — they were a fifth of the kernel's time at 159 bases
                const int na = HALVES == 1 ? (SHORT ? NA - 1 : NA) : max(1, min(NA, (int)((nk + 14u + 63u) >> 6)));  // m-mers 0 .. nk + 13
                const int nb = HALVES == 1 ? NA - 1 : min(NA - 1, (int)((nk + 63u) >> 6));          // k-mers 0 .. nk - 1
                mn[64 * (NA - 1) + 16 + lane] = 0xFFFFFFFFu;  // positions past the last round are "infinite"
                uint32_t v[NA];
                {
                    // m-mer at position 64 a + lane of the unit: two words of each plane of the span, cut with one v_alignbit
                    // (which takes the low five bits of its shift operand)
                    const uint32_t at = base + (uint32_t)lane;
                    const uint32_t* pw = L.span[0] + (at >> 5);
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        if (a >= na) break;
                        const uint32_t lo = __builtin_amdgcn_alignbit(pw[2 * a + 1], pw[2 * a], at);
                        const uint32_t hi = __builtin_amdgcn_alignbit(pw[SPW + 2 * a + 1], pw[SPW + 2 * a], at);
                        v[a] = (minimizer_order(lo, hi & maskm, m) << ORDER_POS_BITS) | (uint32_t)(64 * a + lane);
                        if (a < NA - 1 || lane < 16) mn[64 * a + lane] = v[a];
                    }
                }
                wave_lds_sync();
                if (WFIX) {
                    // 15 m-mers per window: minima of four neighbours first, then four of those cover the window ([i, i + 4), [i + 4, i + 8),
                    // [i + 8, i + 12), [i + 11, i + 15)): two rounds through LDS (doubling by 1, 2, 4 and the tail took four)
                    uint32_t n1[NA], n2[NA], n3[NA];
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        if (a >= na) break;
                        n1[a] = mn[64 * a + lane + 1]; n2[a] = mn[64 * a + lane + 2]; n3[a] = mn[64 * a + lane + 3];
                    }
#pragma unroll
                    for (int a = 0; a < NA; ++a) {  // in place: every lane has read before any lane writes
                        if (a >= na) break;
                        v[a] = min(min(v[a], n1[a]), min(n2[a], n3[a]));
                        if (a < NA - 1 || lane < 16) mn[64 * a + lane] = v[a];
                    }
                    wave_lds_sync();
                } else {
#pragma unroll
                    for (uint32_t st = 1; st < span; st <<= 1) {  // in place: every lane reads before any lane writes
                        uint32_t nb_[NA];
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            if (a >= na) break;
                            nb_[a] = mn[64 * a + lane + st];
                        }
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            if (a >= na) break;
                            v[a] = min(v[a], nb_[a]);
                            if (a < NA - 1 || lane < 16) mn[64 * a + lane] = v[a];
                        }
                        wave_lds_sync();
                    }
                }
#if defined(FG_K1_STOP) && FG_K1_STOP == 1  // knock-out build (profiles/k1_phase_counts.sh): the kernel up to the end of phase A
                if (lane == 0) { nids[t_first + j] = v[0] == 0x12345u; npos[t_first + j] = 0; idoff[t_first + j] = (t_first + j) * (uint64_t)stride; }
                ++j;
                continue;
#endif
                // ---- B: runs of k-mers sharing a minimizer occurrence, queued behind the runs that wait ----
                uint32_t pos[NA - 1];
                uint64_t H[NA - 1];
                R = 0;
#pragma unroll
                for (int a = 0; a < NA - 1; ++a) {
                    H[a] = 0;
                    pos[a] = 0;
                    if (a >= nb) continue;  // (wave-uniform: no k-mer of the unit in this round)
                    // [i, i + span) and [i + W - span, i + W) cover the window
                    if (WFIX) pos[a] = min(min(v[a], mn[64 * a + lane + 4]), min(mn[64 * a + lane + 8], mn[64 * a + lane + 11])) & POSM;
                    else pos[a] = min(v[a], mn[64 * a + lane + tail]) & POSM;
                    const uint32_t carry = a ? (uint32_t)__builtin_amdgcn_readlane((int)pos[a ? a - 1 : 0], 63) : 0xFFFFFFFFu;
                    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)pos[a], 0x138, 0xF, 0xF, false);  // wave_shr:1
                    H[a] = __ballot((uint32_t)(64 * a + lane) < nk && pos[a] != prev);
                    R += (uint32_t)__popcll(H[a]);
                }
                retry = ng > 0 && q + R > (uint32_t)QCAP;
                if (retry) {
                    R = 0xFFFFu;  // (nothing joins: the pass in preparation is processed as it is)
                } else {
                    if (lane == 0) {
                        uint32_t* M = meta[ws];
                        M[M_UNIT] = j | (base << 8); M[M_QA] = q; M[M_QB] = q + R; M[M_NIDS] = 0; M[M_NPOS] = 0; M[M_HA] = 0; M[M_HB] = 0; M[M_KEND] = base + nk;
                    }
                    // (a run ends where the next one of the read begins, or with the read's last k-mer: phase C works its length out)
                    uint32_t before = q;
#pragma unroll
                    for (int a = 0; a < NA - 1; ++a) {
                        if ((H[a] >> lane) & 1ull) queue[before + mask_rank(H[a])] = run_pack(base + pos[a], pos[a] - (uint32_t)(64 * a + lane), ws);
                        before += (uint32_t)__popcll(H[a]);
                    }
                    wave_lds_sync();
                }
            }
#if defined(FG_K1_STOP) && FG_K1_STOP == 2  // (knock-out build: up to the end of phase B)
            if (j < t_count && lane == 0) { nids[t_first + j] = R == 0x12345u; npos[t_first + j] = 0; idoff[t_first + j] = (t_first + j) * (uint64_t)stride; }
            ++j;
            continue;
#endif
            if (ng == 0 || (q + R <= K1_RUN_LANES && ng < (uint32_t)GROUP)) {  // the read joins the pass in preparation
                q += R;
                ++ng;
                ++j;
                continue;
            }

            // ---- phases C + E over the queued runs. If the heads of a pass over several reads do not fit the head
            // buffer (adversarial input), the pass is repeated read slot by read slot: one unit always fits ----
            K1_STAT(0, ng); K1_STAT(1, 1); K1_STAT(4, q);
            bool single = ng == 1;
            uint32_t t0 = 0, t1 = ng;  // the pass covers read slots gs + t0 .. gs + t1 - 1 of the ring
            for (;;) {
                const uint32_t gfirst = (gs + t0) % (uint32_t)NSLOT;
                const uint32_t qa = single ? (uint32_t)__builtin_amdgcn_readfirstlane((int)meta[gfirst][M_QA]) : 0u;
                const uint32_t qb = single ? (uint32_t)__builtin_amdgcn_readfirstlane((int)meta[gfirst][M_QB]) : q;
                uint32_t hcount = 0, hmain = 0;  // heads so far; heads that lie in per-slot order
                bool overflow = false;
                for (uint32_t c0 = qa; c0 < qb && !overflow; c0 += 64) {
                    const uint32_t qn = min(64u, qb - c0);  // run lanes of this chunk
                    K1_STAT(2, 1);
                    const uint32_t e = c0 + lane;
                    const bool act = (uint32_t)lane < qn;
                    // per-run state: the CL read bases around the minimizer — S[0..2] bases 0..31 of the planes lo, hi, invalid; S[3] bases
                    // 32.. of lo and hi side by side, as a record's w2 holds them; S[4] those of the invalid plane; S[5] = descriptor
                    constexpr int NS = 6;
                    uint32_t S[NS];
                    uint32_t slot_qa = 0;
                    bool slot_last = false;  // the last run of its read slot
                    {
                        uint32_t desc = act ? queue[e] : 0u;
                        const uint32_t pa = desc & RUN_POS_MASK, g = desc >> 22;
                        // the run's k-mers: up to the next run of the same read slot, or to the read's last k-mer
                        const uint32_t behind = queue[min(e + 1u, (uint32_t)QCAP)];
                        const uint32_t slot_qb = single ? qb : meta[g][M_QB];
                        if (!single) slot_qa = meta[g][M_QA];
                        const uint32_t iend = e + 1u == slot_qb ? meta[g][M_KEND] : run_first_kmer(behind);
                        desc |= (iend - run_first_kmer(desc)) << 17;
                        slot_last = e + 1u == slot_qb;
                        // span base pa - km + c at field bit c
                        const uint32_t o = pa - km;
                        const uint32_t* pl = L.span[0] + (o >> 5);
                        uint32_t up[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const uint32_t w0 = pl[p * SPW], w1 = pl[p * SPW + 1], w2 = pl[p * SPW + 2];
                            S[p] = __builtin_amdgcn_alignbit(w1, w0, o) & clo_mask;
                            up[p] = __builtin_amdgcn_alignbit(w2, w1, o) & chi_mask;
                        }
                        S[3] = up[0] | (up[1] << REC_HI_BITS);
                        S[4] = up[2];
                        S[5] = desc;
                    }
                    // the minimizer itself: field bits km .. k-1 (k <= 31: inside the low word)
                    const uint32_t mlo = (S[0] >> km) & maskm, mhi = (S[1] >> km) & maskm;
                    const uint32_t home = mulhi32(dict_hash(mlo, mhi, d.seed), d.num_buckets);
                    u32x4 rec[BUCKET_RECS];
                    {
                        const u32x4* bp = (const u32x4*)(d.table + (size_t)home * BUCKET_WORDS);
#pragma unroll
                        for (int r = 0; r < (int)BUCKET_RECS; ++r) rec[r] = bp[r];
                    }

                    // Buckets still to be looked at go through the ring `pairs` as (bucket << 6 | lane that owns the run): the overflow
                    // bucket behind this key's redirect slot, the next bucket behind a spill flag. found(...) is run on freshly loaded buckets.
                    uint32_t ptail = 0, phead = 0;  // ring counters (wave-uniform)
                    auto found = [&](bool loaded, uint32_t bucket, uint32_t src, const u32x4 last) {
                        // the bucket's redirect (its last slot): first overflow bucket, how many to read at once
                        const uint32_t target = last.y;
                        const uint32_t nbov = loaded && (int32_t)last.z < 0 ? min(last.w & REC_MAX_CSID, REDIRECT_DIRECT) : 0u;
                        const bool spill = loaded && (int32_t)last.w < 0;
                        const uint64_t ms = __ballot(spill);
                        if (__any(nbov != 0) || ms) {
#pragma unroll
                            for (uint32_t jj = 0; jj < REDIRECT_DIRECT; ++jj) {
                                const uint64_t mr = __ballot(jj < nbov);
                                if (jj < nbov) {
                                    const uint32_t at = (ptail + mask_rank(mr)) % PAIRS;
                                    if (WIDE) { pairs[at] = target + jj; L.psrc[at] = src; }
                                    else pairs[at] = ((target + jj) << 6) | src;
                                }
                                ptail += (uint32_t)__popcll(mr);
                            }
                            if (spill) {
                                const uint32_t at = (ptail + mask_rank(ms)) % PAIRS;
                                if (WIDE) { pairs[at] = bucket + 1u; L.psrc[at] = src; }
                                else pairs[at] = ((bucket + 1u) << 6) | src;
                            }
                            ptail += (uint32_t)__popcll(ms);
                            wave_lds_sync();
                        }
                    };
                    found(act, home, (uint32_t)lane, rec[BUCKET_RECS - 1]);

                    bool firstb = true;  // first batch: the run lanes look at their home bucket, free lanes take pairs
                    for (;;) {
                        // ---- lanes without a run take a waiting pair: the run's registers come over ds_bpermute ----
                        const uint32_t base = firstb ? qn : 0u;
                        const uint32_t waiting = ptail - phead;
#ifndef FG_K1_NO_TRANSPOSE
                        // Batches behind the first hold pairs only, 2.8 of them on average in a third of the passes: there a lane takes ONE
                        // record of a pair's bucket (lane = 4 * pair + slot, sixteen pairs per batch) and the comparison below runs once, not
                        // four times for three lanes' worth of work. Heads come out in the same order (pair by pair, slot by slot).
                        const bool transposed = !firstb;  // (wave-uniform)
                        const uint32_t taken = transposed ? min(waiting, 16u) : min(waiting, 64u - base);
                        const uint32_t pidx = transposed ? (uint32_t)lane >> 2 : (uint32_t)lane - base;
                        const bool ovf = transposed ? pidx < taken : (uint32_t)lane >= base && pidx < taken;
#else
                        const bool transposed = false;
                        const uint32_t taken = min(waiting, 64u - base);
                        const uint32_t pidx = (uint32_t)lane - base;
                        const bool ovf = (uint32_t)lane >= base && (uint32_t)lane - base < taken;
#endif
                        K1_STAT(3, 1); K1_STAT(5, taken);
                        uint32_t T[NS];
                        uint32_t bucket = home, src = (uint32_t)lane;
                        if (taken) {
                            if (ovf) {
                                const uint32_t at = (phead + pidx) % PAIRS;
                                const uint32_t pr = pairs[at];
                                if (WIDE) { src = L.psrc[at]; bucket = pr; }
                                else { src = pr & 63u; bucket = pr >> 6; }
                            }
#pragma unroll
                            for (int i = 0; i < NS; ++i) T[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)S[i]);
                            if (ovf) {
                                const u32x4* bp = (const u32x4*)(d.table + (size_t)bucket * BUCKET_WORDS);
                                if (transposed) {
                                    rec[0] = bp[(uint32_t)lane & 3u];
                                } else {
#pragma unroll
                                    for (int r = 0; r < (int)BUCKET_RECS; ++r) rec[r] = bp[r];
                                }
                            }
                            phead += taken;
                            if (transposed) found(ovf && ((uint32_t)lane & 3u) == BUCKET_RECS - 1, bucket, src, rec[0]);  // (the lane that holds the bucket's last slot)
                            else found(ovf, bucket, src, rec[BUCKET_RECS - 1]);
                        } else {
#pragma unroll
                            for (int i = 0; i < NS; ++i) T[i] = S[i];
                        }
                        const bool live = ovf || (firstb && act);
                        const uint32_t desc = T[NS - 1];
                        const uint32_t dm = (desc >> RUN_POS_BITS) & 15u, cnt = (desc >> 17) & 31u, g = (desc >> 22) & 7u;
                        // what a head carries besides its k-mers: read slot << 16 | place of the slot in this pass << 20
                        const uint32_t gtag = (g << 16) | ((g >= gs ? g - gs : g + (uint32_t)NSLOT - gs) << 20);
                        // windows of the run: window s of a record's context is k-mer i0 + s - runlo
                        // (a lane without work: an empty range)
                        const uint32_t runlo = live ? km - dm : 31u, runhi = live ? km - dm + cnt - 1u : 0u;
                        uint32_t hv[BUCKET_RECS], hc[BUCKET_RECS];
                        uint32_t mine = 0, msum = 0, mid = 0;
#pragma unroll
                        for (int r = 0; r < (int)BUCKET_RECS; ++r) {
                            if (transposed && r > 0) { hv[r] = 0; hc[r] = 0; continue; }  // (wave-uniform: a lane holds one record, in rec[0])
#ifdef FG_K1_SLOT_SKIP
                            // (round 6, measured and left out of the shipped build) The slots of a bucket fill from the first: at 0.6 records per
                            // bucket the last slot holds a record (or the redirect) in one bucket of a hundred, so in most passes it is empty in
                            // EVERY bucket the wave looks at and its comparison, a quarter of this loop, could be skipped for the whole wave (an
                            // empty slot has smin > smax and can only yield an empty range of windows). 5.067 against 5.072 ms per 10 M reads
                            // testing the last slot, 5.028 testing the last two (profiles/r6/k1_slot_skip_r6.txt): under one per cent —
                            // the kernel does not wait for these instructions.
#ifndef FG_K1_SKIP_FROM
#define FG_K1_SKIP_FROM 3  // (variant builds: 2 = the last two slots are tested)
#endif
                            if (r >= FG_K1_SKIP_FROM) {
                                const bool holds = live && rec_smin(rec[r].z) <= rec_smax(rec[r].w);
                                if (!__any(holds)) { hv[r] = 0; hc[r] = 0; continue; }
                            }
#endif
                            const uint32_t w0 = rec[r].x, w1 = rec[r].y, w2 = rec[r].z;
                            const uint32_t x0 = (T[0] ^ w0) | (T[1] ^ w1) | T[2];
                            // bases 32..: the two planes lie side by side in w2 as in T[3]; only bits 0 .. k - m - 2 of x1 are looked at
                            const uint32_t v1 = T[3] ^ w2;
                            const uint32_t x1 = v1 | (v1 >> REC_HI_BITS) | T[4];
                            // Mismatches below the core bound the windows from below, those above it from above; a mismatch inside the
                            // core (context bases k - m .. k - 1, part of every window) puts the lower bound above every window.
                            const uint32_t A = x0 & maskk;
                            const uint32_t sl = 31u - (uint32_t)__builtin_clz((A << 1) | 1u);
                            const uint32_t B = __builtin_amdgcn_alignbit(x1, x0, k) & maskkm;
                            const uint32_t su = (uint32_t)__builtin_ctz(B | (1u << km));
                            const uint32_t lo = max(max(sl, rec_smin(w2)), runlo);
                            const uint32_t hi = min(min(su, rec_smax(rec[r].w)), runhi);
                            hv[r] = rec[r].w & REC_MAX_CSID;
                            hc[r] = (uint32_t)max((int32_t)(hi - lo) + 1, 0);
                            const bool hit = hc[r] != 0;
                            mine += hit;
                            msum += hc[r];
                            mid = hit ? hv[r] : mid;
                            if (KMER_OUT && hit) {
                                const uint32_t mu = meta[g][M_UNIT];
                                const uint64_t row = (t_first + (mu & 0xFFu)) * (uint64_t)stride + ((desc & RUN_POS_MASK) - (mu >> 8));  // + window - (k - m) = k-mer of the unit
                                for (uint32_t s = lo; s <= hi; ++s) kmer_out[row + s - km] = hv[r];
                            }
                        }
#ifdef FG_K1_STATS
                        {
                            const uint32_t st1 = (uint32_t)__popcll(__ballot(mine == 1)), st2 = (uint32_t)__popcll(__ballot(mine >= 2)), st3 = (uint32_t)__popcll(__ballot(live));
                            K1_STAT(10, st1); K1_STAT(11, st2); K1_STAT(12, st2 ? 1 : 0); K1_STAT(13, st3);
                        }
#endif
                        // Heads in lane order: the runs first, pairs behind them. Among the run lanes of the first chunk a run with one
                        // matching record whose left neighbour (same read) matched the same id alone is folded into that neighbour:
                        // consecutive runs mostly sit on the same unitig, or on unitigs of one colour set.
                        const bool main_round = firstb && c0 == qa;  // (wave-uniform)
                        bool follower = false;
                        uint64_t F = 0;
                        if (main_round) {
                            const uint32_t key = (mine == 1 && !ovf) ? mid : 0xFFFFFFFFu;  // (ids have 31 bits)
                            const uint32_t pkey = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)key, 0x138, 0xF, 0xF, false);  // wave_shr:1
                            const uint32_t pg = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)g, 0x138, 0xF, 0xF, false);
                            follower = mine == 1 && !ovf && pkey == key && pg == g;
                            F = __ballot(follower);
                        }
                        const bool lead = mine == 1 && !ovf && main_round;  // may carry followers
                        // one scan for both: heads written (low half), k-mers of single-match runs (high half)
                        const uint32_t emit = follower ? 0u : mine;
                        const uint32_t packed = emit | ((lead ? msum : 0u) << 16);
                        const uint32_t incl = wave_incl_scan_u32(packed);
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) & 0xFFFFu;
                        if (hcount + total > (uint32_t)HCAP && !single) { overflow = true; break; }
                        // a leader takes the k-mers of the followers right behind it: scan value of the last one (all lanes
                        // take part in the exchange: ds_bpermute reads 0 from lanes that are masked off)
                        const uint64_t nf = lane == 63 ? ~0ull : ~(F >> (lane + 1));
                        const uint32_t last = (uint32_t)lane + (uint32_t)__builtin_ctzll(nf);
                        const uint32_t s_last = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(last << 2), (int)incl) >> 16;
                        uint32_t at = hcount + (incl & 0xFFFFu) - emit;
                        if (lead && !follower) {
                            const uint32_t tot = s_last - (incl >> 16) + msum;
                            if (at < (uint32_t)HCAP) { hid[at] = mid; hcnt[at] = tot | gtag; }
                        } else if (emit) {
#pragma unroll
                            for (int r = 0; r < (int)BUCKET_RECS; ++r) {
                                if (hc[r]) {
                                    if (at < (uint32_t)HCAP) { hid[at] = hv[r]; hcnt[at] = hc[r] | gtag; }
                                    ++at;
                                }
                            }
                        }
                        if (main_round) {
                            if (!single) {  // where the heads of every read slot begin and end
                                const uint32_t begin = hcount + (incl & 0xFFFFu) - emit;
                                if (act && e == slot_qa) meta[g][M_HA] = begin;
                                if (act && slot_last) meta[g][M_HB] = begin + emit;
                            }
                            // heads of the run lanes lie in per-slot order; what the pairs contribute comes behind them
                            hmain = hcount + ((uint32_t)__builtin_amdgcn_readlane((int)incl, (int)qn - 1) & 0xFFFFu);
                        }
                        hcount = min(hcount + total, (uint32_t)HCAP);
                        firstb = false;
                        if (ptail == phead) break;
                    }
                }
                K1_STAT(6, hcount); K1_STAT(9, hcount - (single ? hcount : hmain)); K1_STAT(8, overflow);
                if (overflow) {  // start over, one read slot at a time
                    single = true;
                    t0 = 0;
                    t1 = 1;
                    continue;
                }
                if (single) hmain = hcount;
                if (single && lane == 0) { meta[gfirst][M_HA] = 0; meta[gfirst][M_HB] = hcount; }
                wave_lds_sync();
                // ---- E: sorted distinct ids with summed multiplicities, per read slot ----
#if defined(FG_K1_STOP) && FG_K1_STOP == 3  // (knock-out build: without phase E)
                if (lane == 0 && hcount == 0x12345u) nids[0] = hid[0];
#else
                uint32_t maxseg = 0;
                for (uint32_t t = t0; t < t1; ++t) {
                    const uint32_t g = (gs + t) % (uint32_t)NSLOT;
                    maxseg = max(maxseg, (uint32_t)__builtin_amdgcn_readfirstlane((int)(meta[g][M_HB] - meta[g][M_HA])));
                }
                K1_STAT(7, maxseg);
                if (hcount <= 64u) {
                    // One pass of up to 64 heads. The place of an id in its read's sorted list is the number of heads of the read with a
                    // SMALLER id (counted over the read's own stretch of heads and the few heads that came out of overflow buckets):
                    // heads with the same id get the same place, where an LDS add sums their k-mers; the places that were hit,
                    // counted from the read's first place, are the output entries.
                    const bool hact = (uint32_t)lane < hcount;
                    const uint32_t vv = hact ? hid[lane] : 0u, hw = hact ? hcnt[lane] : (gfirst << 16);
                    const uint32_t gsel = (hw >> 16) & 7u, tsel = hw >> 20;
                    const uint32_t ha = meta[gsel][M_HA], seglen = meta[gsel][M_HB] - ha;
                    hsrt[lane] = 0u;  // k-mers per place
                    uint32_t rank = 0;
                    for (uint32_t t = 0; t < maxseg; ++t) {  // (ha + maxseg <= 128 <= HCAP: no clamp, what lies past the stretch is not counted)
                        const uint32_t vi = hid[ha + t];
                        rank += (uint32_t)(t < seglen) & (uint32_t)(vi < vv);
                    }
                    uint32_t before = 0;  // heads from overflow buckets that belong to earlier reads of the pass
                    for (uint32_t i = hmain; i < hcount; ++i) {
                        const uint32_t vi = hid[i], ti = hcnt[i] >> 20;
                        rank += (uint32_t)(ti == tsel) & (uint32_t)(vi < vv);
                        before += ti < tsel ? 1u : 0u;
                    }
                    if (hact) {  // (LDS operations of a wave execute in order: every lane has read hcnt before any lane overwrites it)
                        const uint32_t ss = ha + before, p = ss + rank;
                        atomicAdd(&hsrt[p], hw & 0xFFFFu);
                        hres[p] = vv;
                        hcnt[p] = gsel | (ss << 3);  // the same from every head of the read
                    }
                    wave_lds_sync();
                    // lane = place
                    const uint32_t sum = hsrt[lane], sid = hres[lane], stag = hcnt[lane];
                    const bool nz = sum != 0;
                    const uint32_t below = mask_rank(__ballot(nz));  // places hit below mine
                    const uint32_t below_ss = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((stag >> 3) << 2), (int)below);
                    if (nz) {
                        const uint32_t g = stag & 7u;
                        const uint64_t at = (t_first + (meta[g][M_UNIT] & 0xFFu)) * (uint64_t)stride + (below - below_ss);
                        ids_pool[at] = sid;
                        cnt_pool[at] = sum;
                        atomicAdd(&meta[g][M_NIDS], 1u);
                        atomicAdd(&meta[g][M_NPOS], sum);
                    }
                } else {
                    // pass 1: total of the head's id within its read; FIRST = no earlier head of the read has that id. A head is
                    // compared with the heads of its own slot and with the few heads that came out of overflow buckets.
                    for (uint32_t h0 = 0; h0 < hcount; h0 += 64) {
                        const uint32_t h = h0 + lane;
                        const bool hact = h < hcount;
                        const uint32_t vv = hact ? hid[h] : 0u, gsel = hact ? (hcnt[h] >> 16) & 7u : gfirst;
                        const uint32_t ha = meta[gsel][M_HA], hb = meta[gsel][M_HB];
                        uint32_t tot = 0;
                        bool firsth = hact;
                        for (uint32_t t = 0; t < maxseg; ++t) {
                            const uint32_t i = ha + t;
                            const uint32_t vi = hid[min(i, (uint32_t)HCAP - 1u)], ci = hcnt[min(i, (uint32_t)HCAP - 1u)];
                            const bool eq = i < hb && vi == vv;
                            tot += eq ? (ci & 0xFFFFu) : 0u;
                            firsth = firsth && !(eq && i < h);
                        }
                        for (uint32_t i = hmain; i < hcount; ++i) {
                            const uint32_t vi = hid[i], ci = hcnt[i];
                            const bool eq = vi == vv && ((ci >> 16) & 7u) == gsel;
                            tot += eq ? (ci & 0xFFFFu) : 0u;
                            firsth = firsth && !(eq && i < h);
                        }
                        if (hact) hres[h] = firsth ? (tot | FIRST) : 0u;
                    }
                    wave_lds_sync();
                    // pass 2: rank of every first head among the first heads of its read = its place in the sorted list
                    for (uint32_t h0 = 0; h0 < hcount; h0 += 64) {
                        const uint32_t h = h0 + lane;
                        const bool hact = h < hcount;
                        const uint32_t vv = hact ? hid[h] : 0u, gsel = hact ? (hcnt[h] >> 16) & 7u : gfirst;
                        const uint32_t ha = meta[gsel][M_HA], hb = meta[gsel][M_HB];
                        const uint32_t res = hact ? hres[h] : 0u;
                        uint32_t rank = 0;
                        for (uint32_t t = 0; t < maxseg; ++t) {
                            const uint32_t i = min(ha + t, (uint32_t)HCAP - 1u);
                            rank += (ha + t < hb && (hres[i] & FIRST) != 0 && hid[i] < vv) ? 1u : 0u;
                        }
                        for (uint32_t i = hmain; i < hcount; ++i)
                            rank += ((hres[i] & FIRST) != 0 && ((hcnt[i] >> 16) & 7u) == gsel && hid[i] < vv) ? 1u : 0u;
                        if (res & FIRST) {
                            const uint64_t rbase = (t_first + (meta[gsel][M_UNIT] & 0xFFu)) * (uint64_t)stride;
                            ids_pool[rbase + rank] = vv;
                            cnt_pool[rbase + rank] = res & ~FIRST;
                            atomicAdd(&meta[gsel][M_NIDS], 1u);
                            atomicAdd(&meta[gsel][M_NPOS], res & ~FIRST);
                        }
                    }
                }
#endif
                wave_lds_sync();
                if ((uint32_t)lane >= t0 && (uint32_t)lane < t1) {
                    const uint32_t g = (gs + (uint32_t)lane) % (uint32_t)NSLOT;
                    const uint64_t r = t_first + (meta[g][M_UNIT] & 0xFFu);
                    nids[r] = meta[g][M_NIDS];
                    npos[r] = meta[g][M_NPOS];
                    idoff[r] = r * (uint64_t)stride;
                }
                wave_lds_sync();
                if (!single || t1 >= ng) break;
                ++t0;
                ++t1;
            }
            gs = (gs + ng) % (uint32_t)NSLOT;  // = ws: the slot of read j
            if (retry) {  // phases A and B of the same unit again, in front of an empty queue
                q = 0;
                ng = 0;
                continue;
            }
            if (j < t_count) {  // the runs of read j move to the front of the queue
                for (uint32_t c = 0; c < R; c += 64) {
                    const uint32_t x = c + lane < R ? queue[q + c + lane] : 0u;
                    if (c + lane < R) queue[c + lane] = x;
                }
                if (lane == 0) { meta[gs][M_QA] = 0; meta[gs][M_QB] = R; }
                wave_lds_sync();
                q = R;
                ng = 1;
            }
            ++j;
        }
    }
}

}  // namespace fg
