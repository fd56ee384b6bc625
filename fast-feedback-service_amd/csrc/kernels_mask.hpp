// kernels_mask.hpp (included by ffs_context.hip only) -- one-off kernels on the valid-pixel mask: the tables the
// streaming threshold kernels read instead of the mask (they depend on the mask alone, so they are built once per
// mask) and the resolution mask (spotfinder/kernels/masking.cu:37-73,99-147).
#pragma once
#include "ffs_device.h"

namespace ffsamd {

// ---- tables that depend on the mask alone -------------------------------------------------------------
// One thread per (group, row).  mmap[y][x] = number of valid pixels in the 7x7 window clipped to the
// image (the oracle's m, standalone.cc:126-141); ginfo[y][g] byte 0 = mask bits of row y,
// ginfo[y + 3][g] bytes 1, 2 = min / max of m over the VALID pixels of group g in row y (max = 0: none),
// byte 3 = the mask bits of row y once more (row y is the centre row when row y + 3 comes in).
__global__ __launch_bounds__(256) void k_build_maps(const uint8_t* maskbits, uint32_t mpitch, int W, int H, int pitch_px,
                                                    uint8_t* mmap, uint8_t* ginfo, uint32_t gpitch_bytes) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (g * 8 >= pitch_px) return;
    // 24 mask bits per row: columns 8g-8 .. 8g+15
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int dy = -3; dy <= 3; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const uint8_t* row = maskbits + (uint64_t)yy * mpitch;
        uint32_t b = (uint32_t)row[g] << 8;
        if (g > 0) b |= row[g - 1];
        if ((uint32_t)(g + 1) < mpitch) b |= (uint32_t)row[g + 1] << 16;
        for (int j = 0; j < 8; ++j) cnt[j] += __popc((b >> (j + 5)) & 0x7Fu);  // columns 8g+j-3 .. 8g+j+3
    }
    const uint32_t own = maskbits[(uint64_t)y * mpitch + g];
    uint32_t mn = 255, mx = 0;
    for (int j = 0; j < 8; ++j) {
        mmap[(uint64_t)y * pitch_px + g * 8 + j] = (uint8_t)cnt[j];
        if ((own >> j) & 1u) { mn = min(mn, cnt[j]); mx = max(mx, cnt[j]); }
    }
    if (mx == 0) mn = 0;
    ginfo[(uint64_t)y * gpitch_bytes + g * 4] = (uint8_t)own;
    ginfo[(uint64_t)(y + kInfoExtraRows) * gpitch_bytes + g * 4 + 1] = (uint8_t)mn;
    ginfo[(uint64_t)(y + kInfoExtraRows) * gpitch_bytes + g * 4 + 2] = (uint8_t)mx;
    ginfo[(uint64_t)(y + kInfoExtraRows) * gpitch_bytes + g * 4 + 3] = (uint8_t)own;  // the centre row's mask bits again
}

// The same tables for 32-bit pixels: lane groups of FOUR pixels (16 bytes), one ginfo dword per group (mask bits
// in bits 0-3).  The counts are those of the mask alone; the oracle also drops neighbours >= 2^24 from its sums
// and counts (standalone.cc:78,90) -- k_stream_u32 sends every window that holds such a pixel to the gather path.
__global__ __launch_bounds__(256) void k_build_maps4(const uint8_t* maskbits, uint32_t mpitch, int W, int H, int pitch_px,
                                                     uint8_t* mmap, uint8_t* ginfo, uint32_t gpitch_bytes) {
    const int g = blockIdx.x * 256 + threadIdx.x;   // group of 4 pixels: columns 4g .. 4g+3
    const int y = blockIdx.y;
    if (g * 4 >= pitch_px) return;
    uint32_t cnt[4] = {0, 0, 0, 0};
    const int byte0 = (g * 4) >> 3, sh = (g * 4) & 7;   // the group's bits sit at bit `sh` (0 or 4) of byte0
    for (int dy = -3; dy <= 3; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const uint8_t* row = maskbits + (uint64_t)yy * mpitch;
        uint32_t b = (uint32_t)row[byte0] << 8;
        if (byte0 > 0) b |= row[byte0 - 1];
        if ((uint32_t)(byte0 + 1) < mpitch) b |= (uint32_t)row[byte0 + 1] << 16;
        // bit 8 + sh + j is pixel j of the group; its window is bits (8 + sh + j - 3) .. (8 + sh + j + 3)
        for (int j = 0; j < 4; ++j) cnt[j] += __popc((b >> (5 + sh + j)) & 0x7Fu);
    }
    const uint32_t own = (maskbits[(uint64_t)y * mpitch + byte0] >> sh) & 0xFu;
    uint32_t mn = 255, mx = 0;
    for (int j = 0; j < 4; ++j) {
        mmap[(uint64_t)y * pitch_px + g * 4 + j] = (uint8_t)cnt[j];
        if ((own >> j) & 1u) { mn = min(mn, cnt[j]); mx = max(mx, cnt[j]); }
    }
    if (mx == 0) mn = 0;
    ginfo[(uint64_t)y * gpitch_bytes + g * 4] = (uint8_t)own;
    ginfo[(uint64_t)(y + kInfoExtraRows) * gpitch_bytes + g * 4 + 1] = (uint8_t)mn;
    ginfo[(uint64_t)(y + kInfoExtraRows) * gpitch_bytes + g * 4 + 2] = (uint8_t)mx;
    ginfo[(uint64_t)(y + kInfoExtraRows) * gpitch_bytes + g * 4 + 3] = (uint8_t)own;
}


// Resolution mask: spotfinder/kernels/masking.cu:37-73 (float32 distance / d-spacing), :99-147.
// One thread per mask byte (8 pixels); only clears bits, as the reference only ever masks.
__global__ void k_resolution_mask(uint8_t* maskbits, uint32_t mpitch, int W, int H, float wavelength,
                                  float distance, float cx, float cy, float psx, float psy,
                                  float dmin, float dmax) {
    const int bx = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (bx * 8 >= W || y >= H) return;
    uint8_t* p = maskbits + (uint64_t)y * mpitch + bx;
    uint32_t b = *p;
    for (int j = 0; j < 8; ++j) {
        const int x = bx * 8 + j;
        if (x >= W || !((b >> j) & 1u)) continue;  // masking.cu:120-126
        const float dx = (((float)x + 0.5f) - cx) * psx;  // :50-52
        const float dy = (((float)y + 0.5f) - cy) * psy;
        const float r = sqrtf(dx * dx + dy * dy);
        const float theta = 0.5f * atanf(r / distance);  // :71
        const float res = wavelength / (2.0f * sinf(theta));  // :72
        if ((dmin > 0 && res < dmin) || (dmax > 0 && res > dmax)) b &= ~(1u << j);  // :133-142
    }
    *p = (uint8_t)b;
}

}  // namespace ffsamd
