// kernels_decode.hpp (included by ffs_submit.hip) -- bitshuffle-LZ4 chunk decode on the GPU.
//
// What the reference does: every worker thread decompresses its frame on the CPU,
// bshuf_decompress_lz4(buffer + 12, host_image, W*H, sizeof(pixel_t), 0) (spotfinder.cc:823-842,
// h5read/src/read_chunks.cc:22-24; "we anticipate ... offload the decompression"), then copies the
// raw frame over PCIe.  Here the compressed chunk crosses PCIe (4-6x fewer bytes) and is decoded
// straight into the pitched device image the threshold kernels read.
//
// Wire format after the 12-byte header (bitshuffle's HDF5 filter, id 32008): per block of
// 8192 / elem_size elements a 4-byte big-endian compressed length and one LZ4 block; the last
// block is shortened to a multiple of 8 elements; fewer than 8 leftover elements follow raw.  A
// decoded block holds elem_size*8 bit planes: plane b = that bit of every element, element i at
// byte i/8, bit i%8.
//
// MI355X design: LZ4 is sequential inside a block but the frame has thousands of independent blocks
// (4419 per Eiger-16M frame), so one wave64 owns one block:
//   1. the wave stages its compressed block in LDS (coalesced dword loads), at the END of its one buffer:
//      the block is decoded in place (see kDecPayload);
//   2. the sequence headers are parsed wave-uniformly from LDS (scalar registers via readfirstlane);
//      literal runs and matches are copied by all 64 lanes, an overlapping match (offset < length)
//      as out[op+i] = out[op-offset + i % offset], which reads only bytes that were complete before
//      the sequence began;
//   3. the bit planes are transposed back in registers (8x8 bit-matrix transposes on 64-bit words,
//      8 pixels per lane) and go from there to the pitched image (16 consecutive bytes per lane).
// The host walks the block-length prefixes once (a pointer chase it can do while the chunk arrives)
// and passes a table of (offset, length) per block.  Every length, offset and literal run is bounds-
// checked; a malformed block raises an error flag and leaves zeros.
#pragma once
#include "ffs_device.h"

namespace ffsamd {

constexpr int kDecBlockBytes = 8192;                 // bitshuffle target block size

struct DecodeArgs {
    const uint8_t* comp;        // all chunks of the batch
    const uint2* table;         // [n_frames][blocks_per_frame + 1]: (byte offset into comp, byte length)
    uint8_t* image;             // pitched device frames
    uint64_t frame_stride;
    uint32_t pitch;
    int W, H, elem_bytes;
    uint32_t blocks_per_frame;  // LZ4 blocks; table entry [blocks_per_frame] is the raw tail
    uint32_t block_elems;       // elements per full block
    uint32_t last_block_elems;  // elements in the last LZ4 block (multiple of 8)
    uint32_t tail_elems;        // raw leftover elements (< 8)
    uint32_t* error;            // |= 4 on a malformed chunk
};

// 8 bytes of LDS starting at byte `pos`, wave-uniform (in scalar registers)
__device__ __forceinline__ unsigned long long lds_peek8(const uint32_t* s, uint32_t pos) {
    const uint32_t i = pos >> 2;
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(s[i]);
    const uint32_t w1 = __builtin_amdgcn_readfirstlane(s[i + 1]);
    const uint32_t w2 = __builtin_amdgcn_readfirstlane(s[i + 2]);
    const uint32_t sh = (pos & 3u) * 8u;
    const unsigned long long lo = ((unsigned long long)w1 << 32) | w0;
    return sh ? (lo >> sh) | ((unsigned long long)w2 << (64 - sh)) : lo;
}

// transpose of an 8x8 bit matrix held in a 64-bit word (byte r = row r, bit c = column c)
__device__ __forceinline__ unsigned long long transpose8x8(unsigned long long x) {
    unsigned long long t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}

// LDS per wave: ONE buffer.  The LZ4 block is decoded in place: the compressed bytes are staged at the END of
// the buffer and the output grows from its start; a decoder that finishes each sequence before it parses the
// next never overtakes its input when the buffer is LZ4_DECOMPRESS_INPLACE_MARGIN = (compressed size >> 8) + 32
// bytes longer than the output (lz4.h) -- checked at run time all the same (a block that would need more
// room is reported like a corrupt one).  The transposed pixels go from registers straight to the image.
// 8.3 KB per wave instead of 16.5: 19 waves per CU instead of 9 (the kernel is a latency chain per block).
constexpr int kDecPayload = kDecBlockBytes + 72;      // output + in-place margin (64) + alignment slack
constexpr int kDecBufBytes = kDecPayload + 24;        // + what the parser may peek past the end

template <int ES>  // element size in bytes: 2 or 4
__global__ __launch_bounds__(64) void k_bshuf_lz4_decode(const DecodeArgs a) {
    __shared__ uint32_t s_buf[kDecBufBytes / 4];
    const int lane = threadIdx.x;
    const uint32_t blk = blockIdx.x, frame = blockIdx.y;
    const uint2 ent = a.table[(uint64_t)frame * (a.blocks_per_frame + 1) + blk];
    const uint32_t off = __builtin_amdgcn_readfirstlane(ent.x), clen = __builtin_amdgcn_readfirstlane(ent.y);
    uint8_t* img = a.image + (uint64_t)frame * a.frame_stride;
    // first element of this block in the frame (the raw tail sits at the very end)
    const uint32_t e0 = blk == a.blocks_per_frame ? (uint32_t)a.W * (uint32_t)a.H - a.tail_elems : blk * a.block_elems;
    const uint32_t Wd = (uint32_t)a.W;

    // one element (linear index e) to the pitched image
    auto put = [&](uint32_t e, uint32_t v) {
        const uint32_t row = e / Wd, col = e - row * Wd;
        if constexpr (ES == 4) *reinterpret_cast<uint32_t*>(img + (uint64_t)row * a.pitch + (uint64_t)col * 4u) = v;
        else *reinterpret_cast<uint16_t*>(img + (uint64_t)row * a.pitch + (uint64_t)col * 2u) = (uint16_t)v;
    };
    // eight consecutive elements starting at e (a multiple of 8), as dwords where a dword does not leave its row
    auto put8 = [&](uint32_t e, const uint32_t (&w)[8 * ES / 4]) {
        uint32_t row = e / Wd, col = e - row * Wd;
        if constexpr (ES == 4) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                *reinterpret_cast<uint32_t*>(img + (uint64_t)row * a.pitch + (uint64_t)col * 4u) = w[u];
                if (++col == Wd) { col = 0; ++row; }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (col + 1 < Wd) {
                    // (e and col + row * W have the same parity; an odd width can put a pair on an odd column)
                    if ((col & 1u) == 0) {
                        *reinterpret_cast<uint32_t*>(img + (uint64_t)row * a.pitch + (uint64_t)col * 2u) = w[u];
                    } else {
                        *reinterpret_cast<uint16_t*>(img + (uint64_t)row * a.pitch + (uint64_t)col * 2u) = (uint16_t)w[u];
                        *reinterpret_cast<uint16_t*>(img + (uint64_t)row * a.pitch + (uint64_t)(col + 1) * 2u) = (uint16_t)(w[u] >> 16);
                    }
                    col += 2;
                    if (col == Wd) { col = 0; ++row; }
                } else {  // the pair straddles the end of the row
                    *reinterpret_cast<uint16_t*>(img + (uint64_t)row * a.pitch + (uint64_t)col * 2u) = (uint16_t)w[u];
                    ++row;
                    *reinterpret_cast<uint16_t*>(img + (uint64_t)row * a.pitch) = (uint16_t)(w[u] >> 16);
                    col = 1;
                    if (col == Wd) { col = 0; ++row; }
                }
            }
        }
    };

    if (blk == a.blocks_per_frame) {
        // raw tail: fewer than 8 elements, copied as they are
        if (a.tail_elems == 0) return;
        if (clen != a.tail_elems * ES) { if (lane == 0) atomicOr(a.error, 4u); return; }
        if (lane < (int)a.tail_elems) {
            uint32_t v = 0;
            for (int k = 0; k < ES; ++k) v |= (uint32_t)a.comp[(uint64_t)off + (uint32_t)lane * ES + k] << (8 * k);
            put(e0 + (uint32_t)lane, v);
        }
        return;
    }

    const uint32_t n_elems = blk + 1 == a.blocks_per_frame ? a.last_block_elems : a.block_elems;
    const uint32_t out_bytes = n_elems * ES;
    bool bad = clen == 0 || clen > (uint32_t)(kDecBlockBytes + 64);

    // ---- 1. stage the compressed block at the end of the buffer (keeping its byte misalignment sh)
    const uint32_t sh = off & 3u;
    uint32_t start = 0;
    if (!bad) {
        start = (uint32_t)kDecPayload - clen;
        start -= (start - sh) & 3u;            // start = sh (mod 4), end = start + clen <= kDecPayload
        const uint32_t* gsrc = reinterpret_cast<const uint32_t*>(a.comp + ((uint64_t)off & ~3ull));
        const uint32_t ndw = (sh + clen + 3u) >> 2, d0 = (start - sh) >> 2;
        for (uint32_t i = lane; i < ndw; i += 64) s_buf[d0 + i] = gsrc[i];
        if (lane < 5 && d0 + ndw + lane < (uint32_t)(kDecBufBytes / 4)) s_buf[d0 + ndw + lane] = 0;  // the parser may peek up to 11 bytes past the end
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);

    // ---- 2. LZ4 block decode in place (format: token, [literal length bytes], literals, offset16, [match length bytes])
    uint8_t* const out = reinterpret_cast<uint8_t*>(s_buf);
    const uint8_t* const in = reinterpret_cast<const uint8_t*>(s_buf);
    uint32_t pos = start;
    const uint32_t end = start + clen;
    uint32_t op = 0;
    while (!bad && pos < end) {
        unsigned long long w = lds_peek8(s_buf, pos);
        const uint32_t token = (uint32_t)w & 0xFFu;
        uint32_t lit = token >> 4;
        uint32_t used = 1;
        if (lit == 15u) {
            for (;;) {
                if (used == 8) { pos += 8; used = 0; w = lds_peek8(s_buf, pos); }
                const uint32_t b = (uint32_t)(w >> (8 * used)) & 0xFFu;
                ++used;
                lit += b;
                if (b != 255u) break;
                if (pos + used > end) { bad = true; break; }
            }
        }
        pos += used;
        // (op <= pos: the output never passes the read cursor; checked, not assumed)
        if (bad || pos + lit > end || op + lit > out_bytes || op > pos) { bad = true; break; }
        // dest lies below source: every 64-byte piece is read before it is written, later pieces are untouched
        for (uint32_t i = lane; i < lit; i += 64) out[op + i] = in[pos + i];
        pos += lit;
        op += lit;
        if (pos >= end) break;  // the last sequence carries literals only
        w = lds_peek8(s_buf, pos);
        const uint32_t offset = (uint32_t)w & 0xFFFFu;
        uint32_t mlen = (token & 15u) + 4u;
        used = 2;
        if ((token & 15u) == 15u) {
            for (;;) {
                if (used == 8) { pos += 8; used = 0; w = lds_peek8(s_buf, pos); }
                const uint32_t b = (uint32_t)(w >> (8 * used)) & 0xFFu;
                ++used;
                mlen += b;
                if (b != 255u) break;
                if (pos + used > end) { bad = true; break; }
            }
        }
        pos += used;
        if (bad || offset == 0 || offset > op || op + mlen > out_bytes || pos > end
            || (pos < end && op + mlen > pos)) { bad = true; break; }  // (last clause: would overwrite unread input)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);  // the literals above must have landed before a match reads them
        const uint32_t src = op - offset;
        if (offset >= mlen) {
            for (uint32_t i = lane; i < mlen; i += 64) out[op + i] = out[src + i];
        } else if (offset == 1u) {
            const uint8_t v = out[src];
            for (uint32_t i = lane; i < mlen; i += 64) out[op + i] = v;
        } else {
            for (uint32_t i = lane; i < mlen; i += 64) out[op + i] = out[src + i % offset];
        }
        op += mlen;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (op != out_bytes) bad = true;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    if (bad && lane == 0) atomicOr(a.error, 4u);

    // ---- 3. bit-unshuffle: lane g rebuilds elements 8g .. 8g+7 from byte g of every plane and stores them
    const uint32_t row_bytes = n_elems / 8;  // bytes per plane
    for (uint32_t g = lane; g < row_bytes; g += 64) {
        uint32_t px[8 * ES / 4];
        if (bad) {
#pragma unroll
            for (int u = 0; u < 8 * ES / 4; ++u) px[u] = 0;  // a malformed block leaves zeros
        } else {
            unsigned long long t[ES];
#pragma unroll
            for (int k = 0; k < ES; ++k) {
                unsigned long long x = 0;
#pragma unroll
                for (int b = 0; b < 8; ++b) x |= (unsigned long long)out[(uint32_t)(k * 8 + b) * row_bytes + g] << (8 * b);
                t[k] = transpose8x8(x);  // byte u = byte k of element 8g + u
            }
            if constexpr (ES == 2) {
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    const uint32_t e_lo = ((uint32_t)(t[0] >> (8 * u)) & 0xFFu) | (((uint32_t)(t[1] >> (8 * u)) & 0xFFu) << 8);
                    const uint32_t e_hi = ((uint32_t)(t[0] >> (8 * u + 8)) & 0xFFu) | (((uint32_t)(t[1] >> (8 * u + 8)) & 0xFFu) << 8);
                    px[u / 2] = e_lo | (e_hi << 16);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    px[u] = ((uint32_t)(t[0] >> (8 * u)) & 0xFFu) | (((uint32_t)(t[1] >> (8 * u)) & 0xFFu) << 8)
                            | (((uint32_t)(t[2] >> (8 * u)) & 0xFFu) << 16) | (((uint32_t)(t[3] >> (8 * u)) & 0xFFu) << 24);
            }
        }
        put8(e0 + 8u * g, px);
    }
}
template __global__ void k_bshuf_lz4_decode<2>(const DecodeArgs);
template __global__ void k_bshuf_lz4_decode<4>(const DecodeArgs);

}  // namespace ffsamd
