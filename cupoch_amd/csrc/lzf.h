// lzf.h -- the LZF byte format that PCD's DATA binary_compressed carries (reference:
// io/file_format/file_pcd.cu:218,461,690 call lzf_decompress / lzf_compress of the liblzf vendored under
// third_party/liblzf).  Written from the published format, not from those sources; pinned against them both ways --
// their compressor's streams decode here, this compressor's streams decode there (the test side compiles
// third_party/liblzf/lzf_{c,d}.c in place for that; tests/test_io_and_real_data.py).  Host code.
//
// Stream = a sequence of chunks, each starting with a control byte c:
//   c < 32       literal run: the next c + 1 bytes are copied to the output;
//   c >= 32      back reference: length = (c >> 5) + 2, or 7 + 2 + (next byte) when c >> 5 == 7;
//                distance = ((c & 31) << 8 | next byte) + 1 bytes back in the OUTPUT (<= 8192);
//                the copy may overlap its own output (byte-wise, front to back).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mi {
namespace lzf {

// returns the decompressed size, or 0 when the stream is corrupt / does not fit
inline size_t decompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned c = in[ip++];
        if (c < 32) {
            const size_t run = c + 1;
            if (ip + run > in_len || op + run > out_cap) return 0;
            std::memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            size_t len = c >> 5;
            if (len == 7) {
                if (ip >= in_len) return 0;
                len += in[ip++];
            }
            if (ip >= in_len) return 0;
            const size_t dist = (((size_t)(c & 31u)) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_cap) return 0;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];
        }
    }
    return op;
}

// greedy compressor: 3-byte hash -> last position, longest match there (<= 264 bytes, <= 8192 back).
// Returns the compressed size, or 0 when out_cap is too small.
inline size_t compress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) {
    constexpr int kHashBits = 16;
    constexpr size_t kMaxOff = 8192, kMaxLen = 264, kMaxLit = 32;
    std::vector<int64_t> table((size_t)1 << kHashBits, -1);
    size_t ip = 0, op = 0, lit_start = 0;
    auto flush_literals = [&](size_t end) -> bool {
        size_t p = lit_start;
        while (p < end) {
            const size_t run = (end - p < kMaxLit) ? end - p : kMaxLit;
            if (op + 1 + run > out_cap) return false;
            out[op++] = (uint8_t)(run - 1);
            std::memcpy(out + op, in + p, run);
            op += run;
            p += run;
        }
        return true;
    };
    while (ip + 2 < in_len) {
        const uint32_t v = (uint32_t)in[ip] | ((uint32_t)in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
        const uint32_t h = (v * 2654435761u) >> (32 - kHashBits);
        const int64_t ref = table[h];
        table[h] = (int64_t)ip;
        if (ref >= 0 && ip - (size_t)ref <= kMaxOff && in[ref] == in[ip] && in[ref + 1] == in[ip + 1] &&
            in[ref + 2] == in[ip + 2]) {
            size_t len = 3;
            const size_t cap = (in_len - ip < kMaxLen) ? in_len - ip : kMaxLen;
            while (len < cap && in[ref + len] == in[ip + len]) ++len;
            if (!flush_literals(ip)) return 0;
            const size_t dist = ip - (size_t)ref - 1, l = len - 2;
            if (op + 3 > out_cap) return 0;
            if (l < 7) {
                out[op++] = (uint8_t)((l << 5) | (dist >> 8));
            } else {
                out[op++] = (uint8_t)((7u << 5) | (dist >> 8));
                out[op++] = (uint8_t)(l - 7);
            }
            out[op++] = (uint8_t)(dist & 255u);
            ip += len;
            lit_start = ip;
        } else {
            ++ip;
        }
    }
    if (!flush_literals(in_len)) return 0;
    return op;
}

}  // namespace lzf
}  // namespace mi
