// h5_reader.cc -- NXmx / Eiger HDF5 frame source (compiled when HDF5 headers are available;
// -DFFS_HAVE_HDF5).  Same behaviour as the reference's H5Read (h5read/src/h5read.c):
//   * /entry/data/data is the frame stack: a virtual dataset whose sources are data files or
//     external links (h5read.c:905-990), or a plain chunked dataset
//   * frames are handed over as RAW chunks (H5Dread_chunk, h5read.c:428-456): bitshuffle-LZ4 with
//     the 12-byte filter header, decoded by the driver (or on the GPU)
//   * availability of a frame = its chunk has storage, after H5Drefresh for SWMR files (:379-420)
//   * pixel_mask == 0 -> valid (:556-640); metadata paths as in :795-900
#ifdef FFS_HAVE_HDF5
#include <hdf5.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "reader.hpp"

namespace ffshost {

namespace {
struct ErrSilence {  // HDF5 prints a stack trace for every failed probe otherwise (h5read.c:395-399)
    H5E_auto2_t fn;
    void* data;
    ErrSilence() {
        H5Eget_auto2(H5E_DEFAULT, &fn, &data);
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    }
    ~ErrSilence() { H5Eset_auto2(H5E_DEFAULT, fn, data); }
};

std::string dirname_of(const std::string& p) {
    const size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

bool read_scalar_double(hid_t file, const char* path, double& out) {
    ErrSilence q;
    hid_t d = H5Dopen2(file, path, H5P_DEFAULT);
    if (d < 0) return false;
    double v = 0;
    const bool ok = H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v) >= 0;
    H5Dclose(d);
    if (ok) out = v;
    return ok;
}
}  // namespace

class H5Read : public Reader {
    struct Source {
        std::string filename, dsetname;
        hsize_t start = 0, frames = 0;
        hid_t file = -1, dset = -1;
    };
    hid_t master_ = -1;
    std::vector<Source> src_;
    size_t n_images_ = 0;
    std::array<size_t, 2> shape_{};
    h5read_dtype dtype_ = H5READ_DTYPE_UINT16;
    std::vector<uint8_t> mask_;
    std::array<int64_t, 2> trusted_{0, 65535};
    std::optional<float> wavelength_, distance_;
    std::optional<std::array<float, 2>> pixel_size_, beam_center_;
    std::array<float, 2> osc_{0, 0};

    bool open_source(Source& s) {
        if (s.dset >= 0) return true;
        ErrSilence q;
        if (s.filename.empty()) {
            s.dset = H5Dopen2(master_, s.dsetname.c_str(), H5P_DEFAULT);
        } else {
            s.file = H5Fopen(s.filename.c_str(), H5F_ACC_RDONLY | H5F_ACC_SWMR_READ, H5P_DEFAULT);
            if (s.file < 0) s.file = H5Fopen(s.filename.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
            if (s.file < 0) return false;  // data file not written yet
            s.dset = H5Dopen2(s.file, s.dsetname.c_str(), H5P_DEFAULT);
        }
        return s.dset >= 0;
    }
    Source* find(size_t index, hsize_t& local) {
        for (auto& s : src_)
            if (index >= s.start && index < s.start + s.frames) {
                local = index - s.start;
                return &s;
            }
        return nullptr;
    }

  public:
    explicit H5Read(const std::string& filename) {
        {
            ErrSilence q;
            master_ = H5Fopen(filename.c_str(), H5F_ACC_RDONLY | H5F_ACC_SWMR_READ, H5P_DEFAULT);
            if (master_ < 0) master_ = H5Fopen(filename.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
        }
        if (master_ < 0) throw std::runtime_error("Error: Reading " + filename);
        const std::string root = dirname_of(filename);
        hid_t data = H5Dopen2(master_, "/entry/data/data", H5P_DEFAULT);
        if (data < 0) throw std::runtime_error("Error: Reading H5 entry /entry/data/data");
        {  // setup_data, h5read.c:1067-1091
            hid_t type = H5Dget_type(data), space = H5Dget_space(data);
            if (H5Sget_simple_extent_ndims(space) != 3) throw std::runtime_error("VDS data not three dimensional");
            hsize_t dims[3];
            H5Sget_simple_extent_dims(space, dims, nullptr);
            n_images_ = dims[0];
            shape_ = {(size_t)dims[1], (size_t)dims[2]};
            const size_t sz = H5Tget_size(type);
            if (H5Tget_class(type) != H5T_INTEGER || (sz != 2 && sz != 4))
                throw std::runtime_error("Error: only 16- and 32-bit integer pixel data are handled");
            dtype_ = sz == 2 ? H5READ_DTYPE_UINT16 : H5READ_DTYPE_UINT32;
            trusted_[1] = sz == 2 ? 65535 : (int64_t)0xFFFFFFFFll;
            H5Tclose(type);
            H5Sclose(space);
        }
        hid_t plist = H5Dget_create_plist(data);
        if (H5Pget_layout(plist) == H5D_VIRTUAL) {  // vds_info, h5read.c:905-990
            size_t count = 0;
            H5Pget_virtual_count(plist, &count);
            for (size_t j = 0; j < count; ++j) {
                Source s;
                hid_t vspace = H5Pget_virtual_vspace(plist, j);
                hsize_t start[3], stride[3], cnt[3], block[3];
                H5Sget_regular_hyperslab(vspace, start, stride, cnt, block);
                H5Sclose(vspace);
                s.start = start[0];
                s.frames = block[0] * cnt[0];
                char fn[4096], dn[4096];
                H5Pget_virtual_filename(plist, j, fn, sizeof fn);
                H5Pget_virtual_dsetname(plist, j, dn, sizeof dn);
                s.dsetname = dn;
                if (std::strcmp(fn, ".") == 0) {
                    // source in the master itself, possibly an external link: dereference it
                    H5L_info_t info;
                    if (H5Lget_info(master_, dn, &info, H5P_DEFAULT) >= 0 && info.type == H5L_TYPE_EXTERNAL) {
                        std::vector<char> buf(info.u.val_size + 1);
                        H5Lget_val(master_, dn, buf.data(), buf.size(), H5P_DEFAULT);
                        unsigned flags;
                        const char *nameptr, *dsetptr;
                        H5Lunpack_elink_val(buf.data(), info.u.val_size, &flags, &nameptr, &dsetptr);
                        s.filename = root + "/" + nameptr;
                        s.dsetname = dsetptr;
                    }
                } else {
                    s.filename = root + "/" + fn;
                }
                src_.push_back(std::move(s));
            }
        } else {
            Source s;
            s.dsetname = "/entry/data/data";
            s.frames = n_images_;
            src_.push_back(std::move(s));
        }
        H5Pclose(plist);
        H5Dclose(data);

        // trusted range / wavelength / geometry / oscillation, h5read.c:795-900
        double v;
        if (read_scalar_double(master_, "/entry/instrument/detector/saturation_value", v)) trusted_[1] = (int64_t)v;
        if (read_scalar_double(master_, "/entry/instrument/detector/underload_value", v)) trusted_[0] = (int64_t)v;
        if (read_scalar_double(master_, "/entry/instrument/beam/incident_wavelength", v)) wavelength_ = (float)v;
        double px = -1, py = -1, bx = -1, by = -1, dist = -1;
        read_scalar_double(master_, "/entry/instrument/detector/x_pixel_size", px);
        read_scalar_double(master_, "/entry/instrument/detector/y_pixel_size", py);
        read_scalar_double(master_, "/entry/instrument/detector/beam_center_x", bx);
        read_scalar_double(master_, "/entry/instrument/detector/beam_center_y", by);
        read_scalar_double(master_, "/entry/instrument/detector/distance", dist);
        pixel_size_ = {{(float)py, (float)px}};
        beam_center_ = {{(float)by, (float)bx}};
        distance_ = (float)dist;
        {
            ErrSilence q;
            hid_t om = H5Dopen2(master_, "/entry/sample/sample_omega/omega", H5P_DEFAULT);
            if (om >= 0) {
                hid_t sp = H5Dget_space(om);
                const hssize_t n = H5Sget_simple_extent_npoints(sp);
                if (n >= 2) {
                    std::vector<double> o((size_t)n);
                    if (H5Dread(om, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, o.data()) >= 0)
                        osc_ = {(float)o[0], (float)(o[1] - o[0])};
                }
                H5Sclose(sp);
                H5Dclose(om);
            }
        }
        {  // read_mask, h5read.c:561-640: pixel_mask == 0 -> 1
            ErrSilence q;
            hid_t md = H5Dopen2(master_, "/entry/instrument/detector/pixel_mask", H5P_DEFAULT);
            if (md >= 0) {
                hid_t sp = H5Dget_space(md);
                const size_t n = (size_t)H5Sget_simple_extent_npoints(sp);
                std::vector<uint64_t> raw(n);
                if (n == shape_[0] * shape_[1]
                    && H5Dread(md, H5T_NATIVE_UINT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.data()) >= 0) {
                    mask_.resize(n);
                    for (size_t i = 0; i < n; ++i) mask_[i] = raw[i] == 0;
                }
                H5Sclose(sp);
                H5Dclose(md);
            } else {
                std::fprintf(stdout, "Warning: no mask data found at /entry/instrument/detector/pixel_mask\n");
            }
        }
    }
    ~H5Read() override {
        for (auto& s : src_) {
            if (s.dset >= 0) H5Dclose(s.dset);
            if (s.file >= 0) H5Fclose(s.file);
        }
        if (master_ >= 0) H5Fclose(master_);
    }
    bool is_image_available(size_t index) override {  // h5read_get_chunk_size > 0
        hsize_t local;
        Source* s = find(index, local);
        if (!s || !open_source(*s)) return false;
        ErrSilence q;
        hsize_t off[3] = {local, 0, 0}, size = 0;
        H5Dget_chunk_storage_size(s->dset, off, &size);
        if (size == 0) {
            H5Drefresh(s->dset);
            H5Dget_chunk_storage_size(s->dset, off, &size);
        }
        return size > 0;
    }
    std::span<uint8_t> get_raw_chunk(size_t index, std::span<uint8_t> dst) override {
        hsize_t local;
        Source* s = find(index, local);
        if (!s || !open_source(*s)) return {dst.data(), 0};
        hsize_t off[3] = {local, 0, 0}, size = 0;
        H5Dget_chunk_storage_size(s->dset, off, &size);
        if (size == 0 || size > dst.size()) return {dst.data(), 0};
        uint32_t filters = 0;
        if (H5Dread_chunk(s->dset, H5P_DEFAULT, off, &filters, dst.data()) < 0) return {dst.data(), 0};
        return {dst.data(), (size_t)size};
    }
    ChunkCompression get_raw_chunk_compression() override { return BITSHUFFLE_LZ4; }
    size_t get_number_of_images() const override { return n_images_; }
    h5read_dtype get_dtype() const override { return dtype_; }
    std::array<int64_t, 2> get_trusted_range() const override { return trusted_; }
    std::array<size_t, 2> image_shape() const override { return shape_; }
    std::optional<std::span<const uint8_t>> get_mask() const override {
        if (mask_.empty()) return std::nullopt;
        return {{mask_.data(), mask_.size()}};
    }
    std::optional<float> get_wavelength() const override { return wavelength_; }
    std::optional<std::array<float, 2>> get_pixel_size() const override { return pixel_size_; }
    std::optional<std::array<float, 2>> get_beam_center() const override { return beam_center_; }
    std::optional<float> get_detector_distance() const override { return distance_; }
    std::array<float, 2> get_oscillation() const override { return osc_; }
};

std::unique_ptr<Reader> make_h5_reader(const std::string& filename) { return std::make_unique<H5Read>(filename); }
bool h5_supported() { return true; }

}  // namespace ffshost

template <> bool is_ready_for_read<ffshost::H5Read>(const std::string& filename) {  // h5read.h:327-336
    ffshost::ErrSilence q;
    hid_t f = H5Fopen(filename.c_str(), H5F_ACC_RDONLY | H5F_ACC_SWMR_READ, H5P_DEFAULT);
    if (f < 0) f = H5Fopen(filename.c_str(), H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return false;
    H5Fclose(f);
    return true;
}
#else
#include "reader.hpp"
#include <stdexcept>
namespace ffshost {
std::unique_ptr<Reader> make_h5_reader(const std::string&) {
    throw std::runtime_error("HDF5/NeXus input needs an HDF5-enabled build (hdf5.h was not found at build time)");
}
bool h5_supported() { return false; }
}  // namespace ffshost
template <> bool is_ready_for_read<ffshost::H5Read>(const std::string&) { return true; }
#endif
