// ref_shim.cc -- extern "C" door onto the REFERENCE's own CPU spot-finder.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together with
// the reference's baseline/spotfinder/standalone.cc *from where that file lies*
// under /root/reference (see oracle/Makefile, target _ref/libffs_ref.so).  No
// reference source is copied into this repository; the built library lives in
// oracle/_ref/ (git-ignored) and is used to (a) pin oracle/ffs_oracle.c and
// (b) generate tests/golden/ fixtures, and (c) as bench.py's cpu_baseline
// ("kind": "reference").
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <span>

#include "standalone.h"  // reference: baseline/spotfinder/standalone.h:8-29

extern "C" {

void *ffs_ref_create(size_t width, size_t height) {
    return new StandaloneSpotfinder<double>(width, height);
}

void ffs_ref_destroy(void *h) {
    delete static_cast<StandaloneSpotfinder<double> *>(h);
}

// StandaloneSpotfinder<double>::standard_dispersion(image, uint8 mask)
// (baseline/spotfinder/standalone.cc:258-270); dst receives W*H bytes 0/1.
int ffs_ref_standard_dispersion(void *h, const double *image, const uint8_t *mask,
                                size_t width, size_t height, uint8_t *dst) {
    auto *sf = static_cast<StandaloneSpotfinder<double> *>(h);
    const size_t n = width * height;
    auto res = sf->standard_dispersion(std::span<const double>(image, n),
                                       std::span<const uint8_t>(mask, n));
    std::memcpy(dst, res.data(), n);
    return 0;
}
}
