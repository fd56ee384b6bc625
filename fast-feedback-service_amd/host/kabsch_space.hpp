// kabsch_space.hpp -- spot variances in Kabsch space, the post-processing of the 3D reflections that
// the reference runs on the host after find_3d_components (spotfinder/spotfinder.cc:1152-1215 around
// Reflection3D::variances_in_kabsch_space, connected_components/connected_components.cc:159-203).
// The per-signal data come from ffs_stack3d_signals (same order as Reflection3D::signals_).
//
// Geometry: the reference builds dx2's Panel(distance, beam centre, pixel size, image size) and
// Scan({1, n}, {start, width}); dx2 is an absent submodule, so the flat single panel that constructor
// describes is written out here: fast axis +x, slow axis -y, origin (-bx*px, +by*py, -distance) mm,
// px_to_mm = pixel * pixel size, lab = origin + x*fast + y*slow; image_range[0] = 1.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

#include "ffs_hip.h"

namespace ffshost {

struct KabschGeometry {
    double distance_mm, beam_center_x_px, beam_center_y_px, pixel_size_x_mm, pixel_size_y_mm;
    double wavelength, oscillation_start, oscillation_width;
};

struct KabschVariances {
    std::vector<double> sigma_b_variance, sigma_m_variance;
    std::vector<int> bbox_depth;
    double est_sigma_b_deg = 0, est_sigma_m_deg = 0;  // the two log lines, spotfinder.cc:1201-1214
    int n_sigma_m = 0;
};

using Vec3 = std::array<double, 3>;
inline Vec3 cross(const Vec3& a, const Vec3& b) {
    return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}
inline double dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 normalized(Vec3 a) {
    const double n = std::sqrt(dot(a, a));
    return {a[0] / n, a[1] / n, a[2] / n};
}
inline Vec3 lab_coord(const KabschGeometry& g, double xpx, double ypx) {
    const double xmm = xpx * g.pixel_size_x_mm, ymm = ypx * g.pixel_size_y_mm;
    return {-g.beam_center_x_px * g.pixel_size_x_mm + xmm, g.beam_center_y_px * g.pixel_size_y_mm - ymm, -g.distance_mm};
}

inline KabschVariances kabsch_variances(const KabschGeometry& g, const ffs_reflection* refl, uint32_t n_refl,
                                        const uint32_t* sx, const uint32_t* sy, const int32_t* sz, const uint32_t* si,
                                        const int32_t* sr, uint64_t n_sig) {
    constexpr double deg_to_rad = M_PI / 180.0, rad_to_deg = 180.0 / M_PI;
    constexpr int image_range_0 = 1, min_bbox_depth = 5;
    struct Frame { Vec3 s1, e1, e2; double mags1, zeta, phi, varx = 0, vary = 0, varz = 0, total = 0; };
    std::vector<Frame> fr(n_refl);
    const Vec3 s0 = {0.0, 0.0, -1.0 / g.wavelength}, m2 = {1.0, 0.0, 0.0};
    for (uint32_t r = 0; r < n_refl; ++r) {  // spotfinder.cc:1187-1194, cc.cc:166-175
        Frame& f = fr[r];
        f.s1 = lab_coord(g, (double)refl[r].com_x, (double)refl[r].com_y);
        f.e1 = normalized(cross(f.s1, s0));
        f.e2 = normalized(cross(f.s1, f.e1));
        f.mags1 = std::sqrt(dot(f.s1, f.s1));
        f.zeta = dot(m2, f.e1);
        f.phi = (g.oscillation_start + ((double)refl[r].com_z - image_range_0) * g.oscillation_width) * deg_to_rad;
    }
    for (uint64_t v = 0; v < n_sig; ++v) {  // cc.cc:177-193
        if (sr[v] < 0) continue;
        Frame& f = fr[(size_t)sr[v]];
        const double x = (double)sx[v] + 0.5, y = (double)sy[v] + 0.5, z = (double)sz[v] + 0.5;
        const Vec3 s1p = lab_coord(g, x, y);
        const Vec3 d = {s1p[0] - f.s1[0], s1p[1] - f.s1[1], s1p[2] - f.s1[2]};
        const double eps1 = dot(f.e1, d) / f.mags1;
        const double eps2 = dot(f.e2, d) / f.mags1;
        const double phi_dash = (g.oscillation_start + (z - image_range_0) * g.oscillation_width) * deg_to_rad;
        const double eps3 = (phi_dash - f.phi) * f.zeta;
        const double inten = (double)si[v];
        f.varx += inten * eps1 * eps1;
        f.vary += inten * eps2 * eps2;
        f.varz += inten * eps3 * eps3;
        f.total += inten;
    }
    KabschVariances out;
    double sum_b = 0, sum_m = 0;
    for (uint32_t r = 0; r < n_refl; ++r) {  // cc.cc:194-199, spotfinder.cc:1195-1200
        const Frame& f = fr[r];
        const double varx = f.varx / f.total, vary = f.vary / f.total, varz = f.varz / f.total;
        const int depth = refl[r].z_max - refl[r].z_min + 1;
        out.sigma_b_variance.push_back((varx + vary) / 2.0);
        out.sigma_m_variance.push_back(varz);
        out.bbox_depth.push_back(depth);
        sum_b += out.sigma_b_variance.back();
        if (depth >= min_bbox_depth) {
            sum_m += varz;
            ++out.n_sigma_m;
        }
    }
    if (n_refl) out.est_sigma_b_deg = std::sqrt(sum_b / n_refl) * rad_to_deg;
    if (out.n_sigma_m) out.est_sigma_m_deg = std::sqrt(sum_m / out.n_sigma_m) * rad_to_deg;
    return out;
}

}  // namespace ffshost
