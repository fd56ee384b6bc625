// h5_writer.cc -- writes an NXmx/Eiger style HDF5 data set from any Reader (test + demo fixture
// generator behind `ffs_hosttool mkh5`; part of libffs_h5.so).  Three layouts, the ones the
// reference's reader resolves (h5read/src/h5read.c:905-990):
//   "vds-links"  master: /entry/data/data_%06d external links -> <stem>_%06d.h5:/entry/data/data,
//                /entry/data/data = virtual dataset over those links (file name ".")   [DLS Eiger]
//   "vds-files"  /entry/data/data = virtual dataset naming the data files directly
//   "plain"      /entry/data/data = one chunked dataset in the master
// Chunks are written pre-compressed (H5Dwrite_chunk): 12-byte bitshuffle header + LZ4 blocks,
// exactly what a detector's filter-32008 pipeline leaves on disk.
#ifdef FFS_HAVE_HDF5
#include <hdf5.h>

#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "codecs.hpp"
#include "reader.hpp"

namespace ffshost {
namespace {
void put_scalar(hid_t file, const char* path, double v) {
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t d = H5Dcreate2(file, path, H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v);
    H5Dclose(d);
    H5Sclose(sp);
}
void mkgroup(hid_t file, const char* path) { H5Gclose(H5Gcreate2(file, path, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)); }

hid_t create_frames(hid_t file, const char* name, hid_t type, hsize_t n, hsize_t H, hsize_t W) {
    hsize_t dims[3] = {n, H, W}, chunk[3] = {1, H, W};
    hid_t sp = H5Screate_simple(3, dims, nullptr);
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE);
    H5Pset_chunk(pl, 3, chunk);
    const unsigned cd[2] = {0, 2};  // bitshuffle: default block size, LZ4
    H5Pset_filter(pl, (H5Z_filter_t)32008, H5Z_FLAG_OPTIONAL, 2, cd);
    hid_t d = H5Dcreate2(file, name, type, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    if (d < 0) {  // library refuses an unregistered filter: store the same bytes without naming it
        H5Premove_filter(pl, H5Z_FILTER_ALL);
        d = H5Dcreate2(file, name, type, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    }
    H5Pclose(pl);
    H5Sclose(sp);
    if (d < 0) throw std::runtime_error("mkh5: cannot create frame dataset");
    return d;
}
}  // namespace

void h5_write_nxmx(Reader& r, const std::string& master_path, const std::string& layout, size_t frames_per_file,
                   size_t n_written) {
    const hsize_t H = r.image_shape()[0], W = r.image_shape()[1];
    const size_t es = r.get_element_size(), N = r.get_number_of_images();
    if (n_written > N) n_written = N;
    if (frames_per_file == 0) frames_per_file = N;
    const hid_t type = es == 2 ? H5T_NATIVE_UINT16 : H5T_NATIVE_UINT32;
    std::string stem = master_path, dirpart;
    if (auto p = stem.rfind("_master.h5"); p != std::string::npos) stem = stem.substr(0, p);
    else if (auto q = stem.rfind(".h5"); q != std::string::npos) stem = stem.substr(0, q);
    const size_t slash = stem.rfind('/');
    const std::string base = slash == std::string::npos ? stem : stem.substr(slash + 1);

    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    H5Pset_libver_bounds(fapl, H5F_LIBVER_V110, H5F_LIBVER_LATEST);  // virtual datasets need >= 1.10
    hid_t master = H5Fcreate(master_path.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    if (master < 0) throw std::runtime_error("mkh5: cannot create " + master_path);
    for (const char* g : {"/entry", "/entry/data", "/entry/instrument", "/entry/instrument/detector",
                          "/entry/instrument/beam", "/entry/sample", "/entry/sample/sample_omega"})
        mkgroup(master, g);
    put_scalar(master, "/entry/instrument/detector/saturation_value", (double)r.get_trusted_range()[1]);
    put_scalar(master, "/entry/instrument/beam/incident_wavelength", r.get_wavelength().value_or(0.976f));
    const auto ps = r.get_pixel_size().value_or(std::array<float, 2>{7.5e-5f, 7.5e-5f});
    const auto bc = r.get_beam_center().value_or(std::array<float, 2>{H / 2.0f, W / 2.0f});
    put_scalar(master, "/entry/instrument/detector/y_pixel_size", ps[0]);
    put_scalar(master, "/entry/instrument/detector/x_pixel_size", ps[1]);
    put_scalar(master, "/entry/instrument/detector/beam_center_y", bc[0]);
    put_scalar(master, "/entry/instrument/detector/beam_center_x", bc[1]);
    put_scalar(master, "/entry/instrument/detector/distance", r.get_detector_distance().value_or(0.3f));
    {
        std::vector<double> om(N);
        const auto osc = r.get_oscillation();
        for (size_t i = 0; i < N; ++i) om[i] = osc[0] + osc[1] * (double)i;
        hsize_t n = N;
        hid_t sp = H5Screate_simple(1, &n, nullptr);
        hid_t d = H5Dcreate2(master, "/entry/sample/sample_omega/omega", H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, om.data());
        H5Dclose(d);
        H5Sclose(sp);
    }
    if (auto m = r.get_mask()) {  // pixel_mask: 0 = good (uint32 as the detector writes it)
        std::vector<uint32_t> pm(H * W);
        for (size_t i = 0; i < H * W; ++i) pm[i] = (*m)[i] ? 0u : 1u;
        hsize_t d2[2] = {H, W};
        hid_t sp = H5Screate_simple(2, d2, nullptr);
        hid_t d = H5Dcreate2(master, "/entry/instrument/detector/pixel_mask", H5T_NATIVE_UINT32, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(d, H5T_NATIVE_UINT32, H5S_ALL, H5S_ALL, H5P_DEFAULT, pm.data());
        H5Dclose(d);
        H5Sclose(sp);
    }

    std::vector<uint8_t> raw(H * W * es);
    auto write_frames = [&](hid_t dset, size_t first, size_t count) {
        for (size_t i = 0; i < count && first + i < n_written; ++i) {
            r.get_raw_chunk(first + i, raw);
            auto c = bshuf_compress_lz4_with_header(raw.data(), H * W, es);
            hsize_t off[3] = {i, 0, 0};
            if (H5Dwrite_chunk(dset, H5P_DEFAULT, 0, off, c.size(), c.data()) < 0)
                throw std::runtime_error("mkh5: H5Dwrite_chunk failed");
        }
    };

    if (layout == "plain") {
        hid_t d = create_frames(master, "/entry/data/data", type, N, H, W);
        write_frames(d, 0, N);
        H5Dclose(d);
    } else if (layout == "vds-links" || layout == "vds-files") {
        hsize_t vdims[3] = {N, H, W};
        hid_t vspace = H5Screate_simple(3, vdims, nullptr);
        hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
        size_t file_no = 0;
        for (size_t first = 0; first < N; first += frames_per_file, ++file_no) {
            const size_t count = std::min(frames_per_file, N - first);
            char suffix[32], link[64];
            std::snprintf(suffix, sizeof suffix, "_%06zu.h5", file_no + 1);
            std::snprintf(link, sizeof link, "/entry/data/data_%06zu", file_no + 1);
            const std::string data_file = stem + suffix, data_rel = base + suffix;
            if (first < n_written || n_written == N) {
                hid_t f = H5Fcreate(data_file.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
                mkgroup(f, "/entry");
                mkgroup(f, "/entry/data");
                hid_t d = create_frames(f, "/entry/data/data", type, count, H, W);
                write_frames(d, first, count);
                H5Dclose(d);
                H5Fclose(f);
            }  // else: data file "not written yet" (live collection)
            hsize_t sdims[3] = {count, H, W}, start[3] = {first, 0, 0}, one[3] = {1, 1, 1};
            hid_t sspace = H5Screate_simple(3, sdims, nullptr);
            H5Sselect_hyperslab(vspace, H5S_SELECT_SET, start, nullptr, one, sdims);
            if (layout == "vds-links") {
                H5Lcreate_external(data_rel.c_str(), "/entry/data/data", master, link, H5P_DEFAULT, H5P_DEFAULT);
                H5Pset_virtual(dcpl, vspace, ".", link, sspace);
            } else {
                H5Pset_virtual(dcpl, vspace, data_rel.c_str(), "/entry/data/data", sspace);
            }
            H5Sclose(sspace);
        }
        H5Sselect_all(vspace);
        hid_t d = H5Dcreate2(master, "/entry/data/data", type, vspace, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        if (d < 0) throw std::runtime_error("mkh5: cannot create the virtual dataset");
        H5Dclose(d);
        H5Pclose(dcpl);
        H5Sclose(vspace);
    } else {
        throw std::runtime_error("mkh5: layout must be vds-links, vds-files or plain");
    }
    H5Fclose(master);
    H5Pclose(fapl);
}

// results_ffs.h5: the reflection table of spotfinder.cc:1219-1300 (dx2 ReflectionTable::write into
// "dials/processing/group_0"): one dataset per column under the group.
namespace {
void put_column(hid_t grp, const char* name, hid_t type, hsize_t n, hsize_t width, const void* data) {
    hsize_t dims[2] = {n, width};
    hid_t sp = H5Screate_simple(width > 1 ? 2 : 1, dims, nullptr);
    hid_t d = H5Dcreate2(grp, name, type, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    if (d < 0) throw std::runtime_error(std::string("results_ffs.h5: cannot create column ") + name);
    if (n > 0) H5Dwrite(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    H5Dclose(d);
    H5Sclose(sp);
}
}  // namespace

void h5_write_reflection_table(const std::string& path, const std::string& group, const std::vector<double>& xyz,
                               const std::vector<int>& id, const std::vector<double>* sigma_b_variance,
                               const std::vector<double>* sigma_m_variance, const std::vector<int>* spot_extent_z) {
    const hsize_t n = id.size();
    if (xyz.size() != 3 * n) throw std::runtime_error("results_ffs.h5: xyzobs.px.value and id disagree in length");
    hid_t f = H5Fcreate(path.c_str(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) throw std::runtime_error("cannot create " + path);
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lcpl, 1);
    hid_t grp = H5Gcreate2(f, ("/" + group).c_str(), lcpl, H5P_DEFAULT, H5P_DEFAULT);
    H5Pclose(lcpl);
    if (grp < 0) {
        H5Fclose(f);
        throw std::runtime_error("results_ffs.h5: cannot create group " + group);
    }
    put_column(grp, "xyzobs.px.value", H5T_NATIVE_DOUBLE, n, 3, xyz.data());
    put_column(grp, "id", H5T_NATIVE_INT, n, 1, id.data());
    if (sigma_b_variance) put_column(grp, "sigma_b_variance", H5T_NATIVE_DOUBLE, n, 1, sigma_b_variance->data());
    if (sigma_m_variance) put_column(grp, "sigma_m_variance", H5T_NATIVE_DOUBLE, n, 1, sigma_m_variance->data());
    if (spot_extent_z) put_column(grp, "spot_extent_z", H5T_NATIVE_INT, n, 1, spot_extent_z->data());
    {   // number of rows, as an attribute of the group
        hid_t sp = H5Screate(H5S_SCALAR);
        hid_t a = H5Acreate2(grp, "num_reflections", H5T_NATIVE_HSIZE, sp, H5P_DEFAULT, H5P_DEFAULT);
        H5Awrite(a, H5T_NATIVE_HSIZE, &n);
        H5Aclose(a);
        H5Sclose(sp);
    }
    H5Gclose(grp);
    H5Fclose(f);
}

// "<name> <rows> <cols> min.. max.. mean.." for every dataset of a group (what the reference's tests
// read back with h5py, tests/test_spotfinder.py:96-104)
void h5_print_group_stats(const std::string& path, const std::string& group) {
    hid_t f = H5Fopen(path.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) throw std::runtime_error("cannot open " + path);
    hid_t grp = H5Gopen2(f, ("/" + group).c_str(), H5P_DEFAULT);
    if (grp < 0) {
        H5Fclose(f);
        throw std::runtime_error("no group " + group);
    }
    H5G_info_t info;
    H5Gget_info(grp, &info);
    for (hsize_t i = 0; i < info.nlinks; ++i) {
        char name[256];
        H5Lget_name_by_idx(grp, ".", H5_INDEX_NAME, H5_ITER_INC, i, name, sizeof name, H5P_DEFAULT);
        hid_t d = H5Dopen2(grp, name, H5P_DEFAULT);
        if (d < 0) continue;
        hid_t sp = H5Dget_space(d);
        hsize_t dims[2] = {0, 1};
        const int nd = H5Sget_simple_extent_dims(sp, dims, nullptr);
        const size_t rows = dims[0], cols = nd > 1 ? dims[1] : 1;
        std::vector<double> v(rows * cols);
        if (!v.empty()) H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data());
        std::printf("%s %zu %zu", name, rows, cols);
        for (int what = 0; what < 3; ++what)
            for (size_t c = 0; c < cols; ++c) {
                double acc = what == 0 ? 1e300 : what == 1 ? -1e300 : 0.0;
                for (size_t r = 0; r < rows; ++r) {
                    const double x = v[r * cols + c];
                    acc = what == 0 ? std::min(acc, x) : what == 1 ? std::max(acc, x) : acc + x;
                }
                std::printf(" %.17g", what == 2 && rows ? acc / rows : acc);
            }
        std::printf("\n");
        H5Sclose(sp);
        H5Dclose(d);
    }
    H5Gclose(grp);
    H5Fclose(f);
}
}  // namespace ffshost
#else
#include <stdexcept>
#include <string>
#include "reader.hpp"
namespace ffshost {
void h5_write_nxmx(Reader&, const std::string&, const std::string&, size_t, size_t) {
    throw std::runtime_error("mkh5 needs an HDF5-enabled build");
}
void h5_write_reflection_table(const std::string&, const std::string&, const std::vector<double>&, const std::vector<int>&,
                               const std::vector<double>*, const std::vector<double>*, const std::vector<int>*) {
    throw std::runtime_error("results_ffs.h5 needs an HDF5-enabled build");
}
void h5_print_group_stats(const std::string&, const std::string&) {
    throw std::runtime_error("needs an HDF5-enabled build");
}
}  // namespace ffshost
#endif
