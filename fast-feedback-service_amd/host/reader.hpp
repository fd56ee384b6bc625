// reader.hpp -- frame-source plugin surface of the spotfinder driver.
//
// A reader written for the reference compiles against this header unchanged: the abstract `class Reader`, the
// `h5read_dtype` values its get_dtype() returns and the free template `is_ready_for_read<T>` are declared here exactly
// as h5read/include/h5read.h:22-32,173-204,327-336 declares them, in the global namespace (SHMRead: spotfinder/shmread.hpp:10-67,
// CBFRead: spotfinder/cbfread.hpp:116-164 derive from it and specialise the template).  Two additions, both with
// defaults so that such a reader need not know them: ChunkCompression::NONE for sources that hand over raw pixels
// (the synthetic reader), appended after the reference's enumerators, and reentrant().
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <optional>
#include <span>
#include <string>
#include <vector>

#ifndef _H5READ_H   // (with the reference's own <h5read.h> in the same unit, its definitions are the ones in force)
typedef enum {      // h5read.h:22-32, same values
    H5READ_DTYPE_UNKNOWN = 0,
    H5READ_DTYPE_UINT8,
    H5READ_DTYPE_UINT16,
    H5READ_DTYPE_UINT32,
    H5READ_DTYPE_INT8,
    H5READ_DTYPE_INT16,
    H5READ_DTYPE_INT32,
    H5READ_DTYPE_FLOAT32,
    H5READ_DTYPE_FLOAT64,
} h5read_dtype;

inline size_t h5read_dtype_size(h5read_dtype dtype) {   // h5read.h:61, h5read.c
    switch (dtype) {
    case H5READ_DTYPE_UINT8: case H5READ_DTYPE_INT8: return 1;
    case H5READ_DTYPE_UINT16: case H5READ_DTYPE_INT16: return 2;
    case H5READ_DTYPE_UINT32: case H5READ_DTYPE_INT32: case H5READ_DTYPE_FLOAT32: return 4;
    case H5READ_DTYPE_FLOAT64: return 8;
    default: return 0;
    }
}

/// Base class object to provide a unified reader interface (h5read.h:173-204)
class Reader {
  public:
    enum ChunkCompression { BITSHUFFLE_LZ4, BYTE_OFFSET_32, NONE };
    virtual ~Reader() = default;
    virtual bool is_image_available(size_t index) = 0;
    virtual std::span<uint8_t> get_raw_chunk(size_t index, std::span<uint8_t> destination) = 0;
    virtual ChunkCompression get_raw_chunk_compression() = 0;
    virtual size_t get_number_of_images() const = 0;
    size_t get_element_size() const { return h5read_dtype_size(this->get_dtype()); }
    virtual h5read_dtype get_dtype() const = 0;
    virtual std::array<int64_t, 2> get_trusted_range() const = 0;
    virtual std::array<size_t, 2> image_shape() const = 0;  // (slow, fast)
    virtual std::optional<std::span<const uint8_t>> get_mask() const = 0;  // 1 = valid
    virtual std::optional<float> get_wavelength() const = 0;
    virtual std::optional<std::array<float, 2>> get_pixel_size() const = 0;   // (y, x) m
    virtual std::optional<std::array<float, 2>> get_beam_center() const = 0;  // (y, x) px
    virtual std::optional<float> get_detector_distance() const = 0;           // m
    virtual std::array<float, 2> get_oscillation() const = 0;                 // (start, width) deg
    // Addition to the reference's interface: true when is_image_available / get_raw_chunk may be called
    // from several worker threads at once (one file per frame, no shared state).  The driver then skips
    // the reader mutex the reference takes around every call (spotfinder.cc:763-765), which otherwise
    // caps the whole pipeline at the speed of one thread copying chunks.
    virtual bool reentrant() const { return false; }
};

/// Is the source at `path` complete enough to open?  Specialised per reader class (h5read.h:327-336, shmread.cc:90-95,
/// cbfread.cc:127-134); the driver polls it (wait_for_ready_for_read, spotfinder.cc:137-175).
template <typename T>
bool is_ready_for_read(const std::string& path);
#endif  // _H5READ_H

namespace ffshost {

using ::Reader;
// the driver's own implementations (host/readers.cc, host/h5_reader.cc); opaque here, made by the factories below
class SynthRead;
class CBFRead;
class SHMRead;
class H5Read;

// spotfinder <file>: directory -> SHMRead, *.cbf -> CBFRead, "synth:..." -> SynthRead,
// anything else -> H5Read (spotfinder/spotfinder.cc:443-465)
std::unique_ptr<Reader> make_synth_reader(const std::string& spec);
std::unique_ptr<Reader> make_cbf_reader(const std::string& templ, size_t num_images, size_t first_index);
std::unique_ptr<Reader> make_shm_reader(const std::string& dir);
std::unique_ptr<Reader> make_h5_reader(const std::string& master_file);  // NXmx / Eiger HDF5
bool h5_supported();
// fixture writer (ffs_hosttool mkh5): frames [0, n_written) get chunks, later ones stay unwritten
void h5_write_nxmx(Reader& source, const std::string& master_file, const std::string& layout, size_t frames_per_file,
                   size_t n_written);
// results_ffs.h5 (spotfinder.cc:1219-1300): columns under `group`; the optional ones are written when given
void h5_write_reflection_table(const std::string& path, const std::string& group, const std::vector<double>& xyzobs_px,
                               const std::vector<int>& id, const std::vector<double>* sigma_b_variance,
                               const std::vector<double>* sigma_m_variance, const std::vector<int>* spot_extent_z);
void h5_print_group_stats(const std::string& path, const std::string& group);

}  // namespace ffshost

template <> bool is_ready_for_read<ffshost::SHMRead>(const std::string& dir);
template <> bool is_ready_for_read<ffshost::CBFRead>(const std::string& templ);
template <> bool is_ready_for_read<ffshost::H5Read>(const std::string& master_file);
