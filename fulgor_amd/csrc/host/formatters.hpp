// Output formatters of the pseudoalignment path (src/ps_utils.cpp:48-243), host side, batch oriented:
// one call formats the CSR results of a batch into a byte buffer.
//   ascii      "<id>\t<count>[\t<colour>...]\n"                      psa_ascii_formatter   :48-86, util.hpp:245-261
//   binary     u32 id, u32 count, u32 x count                         psa_binary_formatter  :120-136
//   compressed per record delta(id) delta(count) then, by the hybrid 0.25/0.75 rule, delta-gaps /
//              n-bit bitmap / complement delta-gaps; records are packed into blocks
//              {u64 num_bits, words}; a block is closed once it exceeds 2^14 bytes
//              (formatter_buffer::write :31-40) and at the end; the file starts with u64 num_colors
//              psa_compressed_formatter :138-243
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "bits.hpp"

namespace fg {

inline void format_ascii(uint32_t first_id, const uint64_t* off, const uint32_t* colors, uint64_t n, std::string& out) {
    char buf[16];
    auto put = [&](uint32_t x) {
        int len = 0;
        do { buf[15 - len++] = (char)('0' + x % 10); x /= 10; } while (x);
        out.append(buf + 16 - len, len);
    };
    out.reserve(out.size() + (off[n] - off[0]) * 5 + n * 12);
    for (uint64_t r = 0; r < n; ++r) {
        put(first_id + (uint32_t)r);
        out.push_back('\t');
        put((uint32_t)(off[r + 1] - off[r]));
        for (uint64_t i = off[r]; i < off[r + 1]; ++i) { out.push_back('\t'); put(colors[i]); }
        out.push_back('\n');
    }
}

inline void format_binary(uint32_t first_id, const uint64_t* off, const uint32_t* colors, uint64_t n, std::string& out) {
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t hdr[2] = {first_id + (uint32_t)r, (uint32_t)(off[r + 1] - off[r])};
        out.append(reinterpret_cast<const char*>(hdr), 8);
        out.append(reinterpret_cast<const char*>(colors + off[r]), (size_t)hdr[1] * 4);
    }
}

// stateful: blocks span batches exactly as one reference worker's formatter_buffer would produce them
struct CompressedFormatter {
    uint32_t n = 0, sparse_thr = 0, dense_thr = 0;
    BitWriter bw;
    uint32_t pending_bytes = 0;

    void init(uint32_t num_colors, std::string& out) {  // set_num_colors, :158-164
        n = num_colors;
        const uint64_t hdr = num_colors;
        out.append(reinterpret_cast<const char*>(&hdr), 8);
        sparse_thr = (uint32_t)(0.25 * n);
        dense_thr = (uint32_t)(0.75 * n);
    }
    void flush(std::string& out) {  // :233-239
        const uint64_t num_bits = bw.nbits;
        out.append(reinterpret_cast<const char*>(&num_bits), 8);
        out.append(reinterpret_cast<const char*>(bw.words.data()), pending_bytes);
        bw = BitWriter();
        pending_bytes = 0;
    }
    void add(uint32_t id, const uint32_t* c, uint32_t size, std::string& out) {  // format, :168-230
        const size_t words_before = bw.words.size();
        bw.delta(id);
        bw.delta(size);
        if (size == 0) {
        } else if (size < sparse_thr) {
            bw.delta(c[0]);
            for (uint32_t i = 1; i < size; ++i) bw.delta(c[i] - (c[i - 1] + 1));
        } else if (size < dense_thr) {
            std::vector<uint64_t> bm((n + 63) / 64, 0);
            for (uint32_t i = 0; i < size; ++i) bm[c[i] >> 6] |= 1ULL << (c[i] & 63);
            bw.append_stream(bm, n);
        } else {
            uint32_t prev = 0, i = 0;
            bool first = true;
            for (uint32_t v = 0; v < n; ++v) {
                if (i < size && c[i] == v) { ++i; continue; }
                if (first) { bw.delta(v); first = false; }
                else bw.delta(v - (prev + 1));
                prev = v;
            }
        }
        pending_bytes += (uint32_t)((bw.words.size() - words_before) * 8);
        if (pending_bytes > (1u << 14)) flush(out);
    }
    void add_batch(uint32_t first_id, const uint64_t* off, const uint32_t* colors, uint64_t cnt, std::string& out) {
        for (uint64_t r = 0; r < cnt; ++r) add(first_id + (uint32_t)r, colors + off[r], (uint32_t)(off[r + 1] - off[r]), out);
    }
    void finish(std::string& out) { flush(out); }  // ~formatter_buffer, :42
};

}  // namespace fg
