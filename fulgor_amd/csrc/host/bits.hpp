// LSB-first bit streams and Elias gamma/delta codes (host side of the engine).
//
// Semantics follow the call sites of the un-vendored `bits` library in the reference:
// bits::bit_vector::builder::append_bits / bits::util::write_delta / read_delta
// (include/color_sets/hybrid.hpp:37-95,162-304). Layout: a bit vector is a sequence of
// 64-bit words, bit i of the stream is bit (i & 63) of word (i >> 6).
//   gamma(x): y = x+1, c = msb(y): c zeros, a one, then the low c bits of y
//   delta(x): y = x+1, b = msb(y): gamma(b), then the low b bits of y
#pragma once
#include <cstdint>
#include <vector>
#include <cassert>

namespace fg {

inline uint32_t msb64(uint64_t x) { return 63u - (uint32_t)__builtin_clzll(x); }

struct BitWriter {
    std::vector<uint64_t> words;
    uint64_t nbits = 0;

    void append(uint64_t v, uint32_t w) {  // w <= 64, v < 2^w
        if (w == 0) return;
        assert(w == 64 || (v >> w) == 0);
        uint32_t sh = nbits & 63;
        if (sh == 0) words.push_back(0);
        words.back() |= v << sh;
        if (sh + w > 64) words.push_back(v >> (64 - sh));
        nbits += w;
    }
    void gamma(uint64_t x) {
        uint64_t y = x + 1;
        uint32_t c = msb64(y);
        append(1ULL << c, c + 1);  // c zeros then a one
        append(y & ((1ULL << c) - 1), c);
    }
    void delta(uint64_t x) {
        uint64_t y = x + 1;
        uint32_t b = msb64(y);
        gamma(b);
        append(b ? (y & ((1ULL << b) - 1)) : 0, b);
    }
    // append n raw bits taken from another stream
    void append_stream(const std::vector<uint64_t>& src, uint64_t n) {
        uint64_t i = 0;
        for (; i + 64 <= n; i += 64) append(src[i >> 6], 64);
        if (i < n) append(src[i >> 6] & ((1ULL << (n - i)) - 1), (uint32_t)(n - i));
    }
};

struct BitReader {
    const uint64_t* w;
    uint64_t pos;
    BitReader(const uint64_t* words, uint64_t p) : w(words), pos(p) {}

    uint64_t peek64() const {  // 64 bits starting at pos (stream must be padded by one word)
        uint32_t sh = pos & 63;
        uint64_t a = w[pos >> 6] >> sh;
        if (sh) a |= w[(pos >> 6) + 1] << (64 - sh);
        return a;
    }
    uint64_t take(uint32_t n) {
        if (n == 0) return 0;
        uint64_t v = peek64();
        if (n < 64) v &= (1ULL << n) - 1;
        pos += n;
        return v;
    }
    uint32_t unary() {  // number of zeros before the next one; consumes the one
        uint32_t z = 0;
        for (;;) {
            uint64_t v = peek64();
            if (v) {
                uint32_t t = (uint32_t)__builtin_ctzll(v);
                pos += t + 1;
                return z + t;
            }
            z += 64;
            pos += 64;
        }
    }
    uint64_t gamma() {
        uint32_t c = unary();
        return (take(c) | (1ULL << c)) - 1;
    }
    uint64_t delta() {
        uint64_t b = gamma();
        return (take((uint32_t)b) | (1ULL << b)) - 1;
    }
};

}  // namespace fg
