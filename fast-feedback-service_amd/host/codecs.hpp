// codecs.hpp -- host decoders for the chunk formats the reference's readers hand over
// (spotfinder/spotfinder.cc:823-842): bitshuffle+LZ4 (HDF5 filter 32008 framing) and the CBF
// byte-offset codec (spotfinder/cbfread.hpp:49-106).  Own implementations of the published
// formats (LZ4 block format; bitshuffle: blocks of elements stored as bit planes); encoders are
// provided for the synthetic writers and the tests.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <span>
#include <vector>

namespace ffshost {

// ---- LZ4 block format -----------------------------------------------------------------------
// returns bytes written, or -1 on malformed input
inline long lz4_block_decompress(const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_cap) {
    const uint8_t* ip = src;
    const uint8_t* const iend = src + src_len;
    uint8_t* op = dst;
    uint8_t* const oend = dst + dst_cap;
    while (ip < iend) {
        const unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                lit += b;
            } while (b == 255);
        }
        if ((size_t)(iend - ip) < lit || (size_t)(oend - op) < lit) return -1;
        std::memcpy(op, ip, lit);
        op += lit;
        ip += lit;
        if (ip >= iend) break;  // last sequence has no match
        if (iend - ip < 2) return -1;
        const size_t off = ip[0] | ((size_t)ip[1] << 8);
        ip += 2;
        if (off == 0 || off > (size_t)(op - dst)) return -1;
        size_t ml = token & 15;
        if (ml == 15) {
            unsigned b;
            do {
                if (ip >= iend) return -1;
                b = *ip++;
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if ((size_t)(oend - op) < ml) return -1;
        const uint8_t* m = op - off;
        for (size_t i = 0; i < ml; ++i) op[i] = m[i];  // overlapping copy, byte by byte
        op += ml;
    }
    return op - dst;
}

// minimal LZ4 block *encoder* (greedy, 4-byte hash): good enough for fixtures and synth output
inline std::vector<uint8_t> lz4_block_compress(const uint8_t* src, size_t n) {
    std::vector<uint8_t> out;
    out.reserve(n + n / 255 + 16);
    std::vector<int32_t> table(1 << 14, -1);
    size_t anchor = 0, i = 0;
    auto emit = [&](size_t lit_len, size_t match_len, size_t offset, bool last) {
        size_t tok_l = lit_len < 15 ? lit_len : 15;
        size_t ml = last ? 0 : match_len - 4;
        size_t tok_m = last ? 0 : (ml < 15 ? ml : 15);
        out.push_back((uint8_t)((tok_l << 4) | tok_m));
        if (lit_len >= 15) {
            size_t r = lit_len - 15;
            while (r >= 255) { out.push_back(255); r -= 255; }
            out.push_back((uint8_t)r);
        }
        out.insert(out.end(), src + anchor, src + anchor + lit_len);
        if (!last) {
            out.push_back((uint8_t)(offset & 255));
            out.push_back((uint8_t)(offset >> 8));
            if (ml >= 15) {
                size_t r = ml - 15;
                while (r >= 255) { out.push_back(255); r -= 255; }
                out.push_back((uint8_t)r);
            }
        }
    };
    const size_t mflimit = n > 12 ? n - 12 : 0;  // last 5 bytes literals, matches end 12 before the end
    while (i < mflimit) {
        uint32_t v;
        std::memcpy(&v, src + i, 4);
        const uint32_t h = (v * 2654435761u) >> 18;
        const int32_t cand = table[h];
        table[h] = (int32_t)i;
        uint32_t w = 0;
        if (cand >= 0) std::memcpy(&w, src + cand, 4);
        if (cand >= 0 && i - (size_t)cand <= 65535 && w == v) {
            size_t ml = 4;
            while (i + ml < n - 5 && src[cand + ml] == src[i + ml]) ++ml;
            emit(i - anchor, ml, i - (size_t)cand, false);
            i += ml;
            anchor = i;
        } else {
            ++i;
        }
    }
    emit(n - anchor, 0, 0, true);
    return out;
}

// ---- bitshuffle --------------------------------------------------------------------------------
// A block of `nelem` (multiple of 8) elements of `es` bytes is stored as es*8 bit planes: plane
// (8*byte + bit) holds that bit of every element, element i at byte i/8, bit i%8 (LSB first).
inline void bitunshuffle_block(const uint8_t* in, uint8_t* out, size_t nelem, size_t es) {
    const size_t row = nelem / 8;
    std::memset(out, 0, nelem * es);
    for (size_t k = 0; k < es; ++k)
        for (size_t b = 0; b < 8; ++b) {
            const uint8_t* plane = in + (k * 8 + b) * row;
            for (size_t g = 0; g < row; ++g) {
                unsigned bits = plane[g];
                uint8_t* o = out + (g * 8) * es + k;
                while (bits) {
                    const int j = __builtin_ctz(bits);
                    bits &= bits - 1;
                    o[(size_t)j * es] |= (uint8_t)(1u << b);
                }
            }
        }
}
inline void bitshuffle_block(const uint8_t* in, uint8_t* out, size_t nelem, size_t es) {
    const size_t row = nelem / 8;
    std::memset(out, 0, nelem * es);
    for (size_t i = 0; i < nelem; ++i)
        for (size_t k = 0; k < es; ++k) {
            unsigned v = in[i * es + k];
            while (v) {
                const int b = __builtin_ctz(v);
                v &= v - 1;
                out[(k * 8 + b) * row + i / 8] |= (uint8_t)(1u << (i % 8));
            }
        }
}
inline size_t bshuf_default_block_size(size_t es) {  // bitshuffle: ~8 KiB target, multiple of 8
    size_t b = 8192 / es;
    b = b / 8 * 8;
    return b < 8 ? 8 : b;
}

// Decodes the body the reference passes to bshuf_decompress_lz4(buffer + 12, ...): a sequence of
// [4-byte big-endian compressed length][LZ4 block] per block of `block` elements, the last block
// shortened to a multiple of 8 elements, then any leftover (< 8) elements stored raw.
inline long bshuf_decompress_lz4(const uint8_t* in, size_t in_len, void* out_, size_t nelem, size_t es,
                                 size_t block = 0) {
    if (block == 0) block = bshuf_default_block_size(es);
    uint8_t* out = static_cast<uint8_t*>(out_);
    std::vector<uint8_t> tmp(block * es);
    size_t ip = 0, done = 0;
    while (nelem - done >= 8) {
        size_t this_block = nelem - done >= block ? block : (nelem - done) / 8 * 8;
        if (ip + 4 > in_len) return -1;
        const size_t clen = ((size_t)in[ip] << 24) | ((size_t)in[ip + 1] << 16) | ((size_t)in[ip + 2] << 8) | in[ip + 3];
        ip += 4;
        if (ip + clen > in_len) return -1;
        if (lz4_block_decompress(in + ip, clen, tmp.data(), this_block * es) != (long)(this_block * es)) return -1;
        ip += clen;
        bitunshuffle_block(tmp.data(), out + done * es, this_block, es);
        done += this_block;
    }
    const size_t left = (nelem - done) * es;
    if (ip + left > in_len) return -1;
    std::memcpy(out + done * es, in + ip, left);
    return (long)(ip + left);
}

// Encoder incl. the 12-byte HDF5-filter header (8-byte BE uncompressed size, 4-byte BE block bytes)
inline std::vector<uint8_t> bshuf_compress_lz4_with_header(const void* data_, size_t nelem, size_t es) {
    const uint8_t* data = static_cast<const uint8_t*>(data_);
    const size_t block = bshuf_default_block_size(es);
    std::vector<uint8_t> out(12);
    const uint64_t total = (uint64_t)nelem * es;
    for (int i = 0; i < 8; ++i) out[i] = (uint8_t)(total >> (8 * (7 - i)));
    const uint32_t bb = (uint32_t)(block * es);
    for (int i = 0; i < 4; ++i) out[8 + i] = (uint8_t)(bb >> (8 * (3 - i)));
    std::vector<uint8_t> tmp(block * es);
    size_t done = 0;
    while (nelem - done >= 8) {
        size_t this_block = nelem - done >= block ? block : (nelem - done) / 8 * 8;
        bitshuffle_block(data + done * es, tmp.data(), this_block, es);
        auto c = lz4_block_compress(tmp.data(), this_block * es);
        const uint32_t cl = (uint32_t)c.size();
        for (int i = 0; i < 4; ++i) out.push_back((uint8_t)(cl >> (8 * (3 - i))));
        out.insert(out.end(), c.begin(), c.end());
        done += this_block;
    }
    out.insert(out.end(), data + done * es, data + nelem * es);
    return out;
}

// ---- CBF byte-offset (cbfread.hpp:49-106): deltas of 1, 2 or 4 bytes with escape values ---------
template <typename Tout>
size_t byte_offset_decompress(const uint8_t* packed, size_t packed_sz, Tout* values, size_t n) {
    int32_t current = 0;
    size_t j = 0, k = 0;
    while (j < packed_sz && k < n) {
        const int8_t c = (int8_t)packed[j++];
        if (c != -128) {
            current += c;
            values[k++] = (Tout)current;   // NB: the reference never stores `current` (cbfread.hpp:69-71
            continue;                      // advances `values` without writing); we do what the codec means
        }
        if (j + 1 >= packed_sz) break;
        const int16_t s = (int16_t)(packed[j] | (packed[j + 1] << 8));
        j += 2;
        if (s != -32768) {
            current += s;
            values[k++] = (Tout)current;
            continue;
        }
        if (j + 3 >= packed_sz) break;
        const int32_t i = (int32_t)((uint32_t)packed[j] | ((uint32_t)packed[j + 1] << 8)
                                    | ((uint32_t)packed[j + 2] << 16) | ((uint32_t)packed[j + 3] << 24));
        j += 4;
        current += i;
        values[k++] = (Tout)current;
    }
    return k;
}
inline std::vector<uint8_t> byte_offset_compress(const int32_t* v, size_t n) {
    std::vector<uint8_t> out;
    int32_t cur = 0;
    for (size_t k = 0; k < n; ++k) {
        const int64_t d = (int64_t)v[k] - cur;
        if (d >= -127 && d <= 127) {
            out.push_back((uint8_t)(int8_t)d);
        } else if (d >= -32767 && d <= 32767) {
            out.push_back(0x80);
            out.push_back((uint8_t)(d & 255));
            out.push_back((uint8_t)((d >> 8) & 255));
        } else {
            out.push_back(0x80);
            out.push_back(0x00);
            out.push_back(0x80);
            for (int b = 0; b < 4; ++b) out.push_back((uint8_t)((d >> (8 * b)) & 255));
        }
        cur = v[k];
    }
    return out;
}

}  // namespace ffshost
