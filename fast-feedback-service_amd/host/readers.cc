// readers.cc -- frame sources behind the Reader plugin surface (reader.hpp):
//   SynthRead  "synth:<workload>[:n_images[:seed]]"  deterministic frames from libffs_synth
//   CBFRead    "<prefix>####.cbf"                    spotfinder/cbfread.{hpp,cc}
//   SHMRead    "<directory>"                         spotfinder/shmread.{hpp,cc}
// H5Read (NXmx/VDS, h5read/src/h5read.c) needs HDF5 and is a "next" row.
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <stdexcept>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "codecs.hpp"
#include "ffs_synth.h"
#include "minijson.hpp"
#include "reader.hpp"

namespace fs = std::filesystem;

namespace ffshost {

// ---- synthetic --------------------------------------------------------------------------------------
class SynthRead : public Reader {
    ffs_synth_params p_{};
    size_t n_images_ = 10;
    std::vector<uint8_t> mask_;
    float osc_width_ = 0.f;

  public:
    explicit SynthRead(const std::string& spec) {
        // synth:<eiger16m|jungfrau9m|plumbing1k|sweep16m|tiny>[:n[:seed]]
        std::vector<std::string> parts;
        std::stringstream ss(spec);
        std::string tok;
        while (std::getline(ss, tok, ':')) parts.push_back(tok);
        const std::string kind = parts.size() > 1 ? parts[1] : "eiger16m";
        if (parts.size() > 2 && !parts[2].empty()) n_images_ = std::stoul(parts[2]);
        const uint64_t seed = parts.size() > 3 ? std::stoull(parts[3]) : 0;
        p_.sigma_min = 0.8; p_.sigma_max = 1.6; p_.peak_min = 30; p_.peak_max = 5000;
        if (kind == "eiger16m" || kind == "sweep16m") {
            p_.width = 4148; p_.height = 4362; p_.pixel_bytes = 2; p_.background = 2.0;
            p_.n_spots = 1500; p_.max_value = 65535; p_.seed = seed ? seed : 2000;
            mask_.resize((size_t)p_.width * p_.height);
            ffs_synth_mask_modules(mask_.data(), p_.width, p_.height, 1028, 512, 12, 38);
            if (kind == "sweep16m") {
                p_.n_spots = 800; p_.n_frames = (uint32_t)n_images_; p_.sigma_z_min = 0.5; p_.sigma_z_max = 2.0;
                p_.seed = seed ? seed : 5000;
                osc_width_ = 0.1f;
            }
        } else if (kind == "jungfrau9m") {
            p_.width = 3072; p_.height = 3072; p_.pixel_bytes = 4; p_.background = 5.0;
            p_.n_spots = 1000; p_.peak_max = 200000; p_.max_value = (1u << 24) - 1; p_.seed = seed ? seed : 4000;
            mask_.resize((size_t)p_.width * p_.height);
            ffs_synth_mask_modules(mask_.data(), p_.width, p_.height, 1024, 512, 0, 0);
        } else if (kind == "plumbing1k") {
            p_.width = 1024; p_.height = 1024; p_.pixel_bytes = 2; p_.background = 3.0;
            p_.n_spots = 150; p_.max_value = 65535; p_.seed = seed ? seed : 1000;
            mask_.assign((size_t)p_.width * p_.height, 1);
            ffs_synth_mask_rect(mask_.data(), 1024, 1024, 500, 512, 0, 1024);
            ffs_synth_mask_rect(mask_.data(), 1024, 1024, 0, 1024, 480, 518);
            ffs_synth_mask_dead_pixels(mask_.data(), 1024, 1024, 1000, 50);
        } else if (kind == "tiny" || kind == "tinysweep") {
            p_.width = 300; p_.height = 200; p_.pixel_bytes = 2; p_.background = 2.0;
            p_.n_spots = 40; p_.max_value = 65535; p_.seed = seed ? seed : 7;
            mask_.assign((size_t)p_.width * p_.height, 1);
            if (kind == "tinysweep") {
                p_.n_frames = (uint32_t)n_images_; p_.sigma_z_min = 0.5; p_.sigma_z_max = 2.0;
                osc_width_ = 0.1f;
            }
        } else {
            throw std::runtime_error("unknown synthetic workload '" + kind + "'");
        }
    }
    bool is_image_available(size_t index) override { return index < n_images_; }
    std::span<uint8_t> get_raw_chunk(size_t index, std::span<uint8_t> dst) override {
        const size_t bytes = (size_t)p_.width * p_.height * p_.pixel_bytes;
        if (dst.size() < bytes) return {dst.data(), 0};
        ffs_synth_frame(&p_, (uint32_t)index, dst.data());
        return {dst.data(), bytes};
    }
    bool reentrant() const override { return true; }  // frames are a pure function of (parameters, index)
    ChunkCompression get_raw_chunk_compression() override { return NONE; }
    size_t get_number_of_images() const override { return n_images_; }
    h5read_dtype get_dtype() const override { return p_.pixel_bytes == 2 ? H5READ_DTYPE_UINT16 : H5READ_DTYPE_UINT32; }
    std::array<int64_t, 2> get_trusted_range() const override {
        return {0, p_.pixel_bytes == 2 ? 65535 : (int64_t)0xFFFFFFFFll};
    }
    std::array<size_t, 2> image_shape() const override { return {p_.height, p_.width}; }
    std::optional<std::span<const uint8_t>> get_mask() const override { return {{mask_.data(), mask_.size()}}; }
    std::optional<float> get_wavelength() const override { return 0.976f; }
    std::optional<std::array<float, 2>> get_pixel_size() const override { return {{0.75e-4f, 0.75e-4f}}; }
    std::optional<std::array<float, 2>> get_beam_center() const override {
        return {{p_.height / 2.0f, p_.width / 2.0f}};
    }
    std::optional<float> get_detector_distance() const override { return 0.3f; }
    std::array<float, 2> get_oscillation() const override { return {0.f, osc_width_}; }
};

std::unique_ptr<Reader> make_synth_reader(const std::string& spec) { return std::make_unique<SynthRead>(spec); }

// ---- CBF (spotfinder/cbfread.cc) ----------------------------------------------------------------------
static std::string expand_template(const std::string& templ, size_t index) {  // cbfread.cc:16-22
    const size_t a = templ.find('#'), b = templ.rfind('#');
    if (a == std::string::npos) return templ;
    char num[64];
    std::snprintf(num, sizeof num, "%0*zu", (int)(b - a + 1), index);
    return templ.substr(0, a) + num + templ.substr(b + 1);
}

class CBFRead : public Reader {
    size_t n_images_, first_;
    std::array<size_t, 2> shape_{0, 0};
    std::string templ_;
    std::vector<uint8_t> mask_;

  public:
    CBFRead(const std::string& templ, size_t n, size_t first) : n_images_(n), first_(first), templ_(templ) {
        if (first > 1) {  // cbfread.cc:37-40
            std::printf("Error: Can only handle CBF start index of 0 or 1\n");
            std::exit(1);
        }
        std::ifstream f(expand_template(templ, first));
        if (!f) throw std::runtime_error("cannot open " + expand_template(templ, first));
        std::string line;
        int got = 0;
        auto value_of = [](const std::string& l) {  // get_value_contents, cbfread.cc:27-33
            const size_t c = l.find(':');
            return std::stoul(l.substr(c == std::string::npos ? l.find(' ') : c + 1));
        };
        while (got < 2 && std::getline(f, line)) {  // cbfread.cc:49-59
            if (line.rfind("X-Binary-Size-Fastest-Dimension", 0) == 0) { shape_[1] = value_of(line); ++got; }
            else if (line.rfind("X-Binary-Size-Second-Dimension", 0) == 0) { shape_[0] = value_of(line); ++got; }
        }
        if (got < 2) throw std::runtime_error("CBF header lacks X-Binary-Size-* dimensions");
        // mask from the first image: the reference pushes (value < 0) (cbfread.cc:78-80), which marks
        // the negative (= flagged) pixels as VALID; we use the evident intent, valid = (value >= 0)
        const size_t npx = shape_[0] * shape_[1];
        std::vector<uint8_t> buf(npx * 4);
        auto chunk = get_raw_chunk(first_, buf);
        std::vector<int32_t> img(npx, 0);
        byte_offset_decompress(chunk.data(), chunk.size(), img.data(), npx);
        mask_.resize(npx);
        for (size_t i = 0; i < npx; ++i) mask_[i] = img[i] >= 0;
    }
    h5read_dtype get_dtype() const override { return H5READ_DTYPE_UINT16; }
    // `index` is the driver's offset image number (image_num + --start-index, spotfinder.cc:756) and
    // is used as the file number as it stands.  (The reference adds the start index a second time
    // inside CBFRead, cbfread.cc:87-96, so with --start-index 1 it opens file 2 for image 0.)
    bool is_image_available(size_t index) override { return fs::exists(expand_template(templ_, index)); }
    std::span<uint8_t> get_raw_chunk(size_t index, std::span<uint8_t> dst) override {  // cbfread.cc:94-128
        std::ifstream f(expand_template(templ_, index), std::ios::binary);
        std::string data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        static const std::string marker = "\x0c\x1a\x04\xd5";
        const size_t at = data.find(marker);
        if (at == std::string::npos) return {dst.data(), 0};
        const size_t start = at + marker.size(), n = data.size() - start;
        if (n > dst.size()) return {dst.data(), 0};
        std::memcpy(dst.data(), data.data() + start, n);
        return {dst.data(), n};
    }
    ChunkCompression get_raw_chunk_compression() override { return BYTE_OFFSET_32; }
    size_t get_number_of_images() const override { return n_images_; }
    std::array<size_t, 2> image_shape() const override { return shape_; }
    std::optional<std::span<const uint8_t>> get_mask() const override { return {{mask_.data(), mask_.size()}}; }
    std::array<int64_t, 2> get_trusted_range() const override { return {0, 65535}; }
    std::optional<float> get_wavelength() const override { return std::nullopt; }
    std::optional<std::array<float, 2>> get_pixel_size() const override { return std::nullopt; }
    std::optional<std::array<float, 2>> get_beam_center() const override { return std::nullopt; }
    std::optional<float> get_detector_distance() const override { return std::nullopt; }
    std::array<float, 2> get_oscillation() const override { return {0, 0}; }
};
std::unique_ptr<Reader> make_cbf_reader(const std::string& t, size_t n, size_t first) {
    return std::make_unique<CBFRead>(t, n, first);
}

// ---- /dev/shm directory (spotfinder/shmread.cc) ---------------------------------------------------------
class SHMRead : public Reader {
    size_t n_images_ = 0;
    std::array<size_t, 2> shape_{};
    std::string base_;
    std::vector<uint8_t> mask_;
    std::array<int64_t, 2> trusted_{};
    std::optional<float> wavelength_;
    std::array<float, 2> beam_center_{}, pixel_size_{}, osc_{};
    float distance_ = 0;
    h5read_dtype dtype_ = H5READ_DTYPE_UINT16;

    std::string image_path(size_t index) const {
        char b[64];
        std::snprintf(b, sizeof b, "/image_%06zu_2", index);
        return base_ + b;
    }

  public:
    explicit SHMRead(const std::string& path) : base_(path) {
        std::ifstream f(path + "/start_1");
        std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        const JsonValue d = JsonParser(text).parse();
        n_images_ = (size_t)d.at("nimages").number() * (size_t)d.at("ntrigger").number();  // shmread.cc:19-20
        shape_ = {(size_t)d.at("y_pixels_in_detector").number(), (size_t)d.at("x_pixels_in_detector").number()};
        const int depth = (int)d.at("bit_depth_image").number();
        if (depth == 16) dtype_ = H5READ_DTYPE_UINT16;
        else if (depth == 32) dtype_ = H5READ_DTYPE_UINT32;
        else throw std::runtime_error("Data is unhandled bit-depth: " + std::to_string(depth) + "-bit");
        trusted_ = {0, (int64_t)d.at("countrate_correction_count_cutoff").number()};
        if (d.contains("wavelength")) wavelength_ = (float)d.at("wavelength").number();
        distance_ = (float)d.at("detector_distance").number() / 1000;  // shmread.cc:45
        pixel_size_ = {(float)d.at("y_pixel_size").number(), (float)d.at("x_pixel_size").number()};
        beam_center_ = {(float)d.at("beam_center_y").number(), (float)d.at("beam_center_x").number()};
        if (d.contains("omega_start") && d.contains("omega_increment"))
            osc_ = {(float)d.at("omega_start").number(), (float)d.at("omega_increment").number()};
        // start_5: int32 pixel_mask, 0 = good -> mask = !v (shmread.cc:58-75)
        const size_t npx = shape_[0] * shape_[1];
        const std::string mfile = base_ + "/start_5";
        if (fs::file_size(mfile) != npx * 4) throw std::runtime_error("Error: Mask file does not match expected size");
        std::vector<int32_t> raw(npx);
        std::ifstream fm(mfile, std::ios::binary);
        fm.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)(npx * 4));
        mask_.resize(npx);
        for (size_t i = 0; i < npx; ++i) mask_[i] = !raw[i];
    }
    bool is_image_available(size_t index) override { return fs::exists(image_path(index)); }
    std::span<uint8_t> get_raw_chunk(size_t index, std::span<uint8_t> dst) override {
        const int fd = ::open(image_path(index).c_str(), O_RDONLY);
        if (fd < 0) return {dst.data(), 0};
        size_t got = 0;
#if defined(__x86_64__)
        // The destination is the driver's pinned staging area, which the GPU's DMA engine reads next: the chunk goes through
        // a cache-resident bounce buffer and on with non-temporal stores, so that it lands in DRAM and neither fills this
        // core's cache with lines the DMA then has to pull out of it nor evicts the bounce buffer (tools/ubench/shm_read.cc
        // modes 5 / 7: the DMA beside eight such readers keeps 56 GB/s, beside read() into the area itself 40-46).
        if (stream_copy_ && (reinterpret_cast<uintptr_t>(dst.data()) & 31u) == 0) {
            constexpr size_t kBounce = 256u << 10;
            static thread_local std::unique_ptr<uint8_t[]> bounce_store(new uint8_t[kBounce + 64]);
            uint8_t* bounce = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(bounce_store.get()) + 63) & ~(uintptr_t)63);
            while (got < dst.size()) {
                const ssize_t r = ::read(fd, bounce, std::min(kBounce, dst.size() - got));
                if (r <= 0) break;
                stream_copy(dst.data() + got, bounce, (size_t)r);
                got += (size_t)r;
                if (((size_t)r & 31u) != 0) break;   // (a short read ends the file: the next piece would start unaligned)
            }
            _mm_sfence();
            if (got < dst.size() && (got & 31u) != 0) {   // anything after an odd-sized piece (never for a whole file: EOF follows)
                for (;;) {
                    const ssize_t r = ::read(fd, dst.data() + got, dst.size() - got);
                    if (r <= 0) break;
                    got += (size_t)r;
                }
            }
            ::close(fd);
            return {dst.data(), got};
        }
#endif
        while (got < dst.size()) {
            const ssize_t r = ::read(fd, dst.data() + got, dst.size() - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        ::close(fd);
        return {dst.data(), got};
    }
#if defined(__x86_64__)
    __attribute__((target("avx2"))) static void stream_copy(uint8_t* d, const uint8_t* s, size_t n) {   // d 32-byte aligned
        const size_t whole = n & ~(size_t)31;
        for (size_t o = 0; o < whole; o += 32)
            _mm256_stream_si256(reinterpret_cast<__m256i*>(d + o), _mm256_load_si256(reinterpret_cast<const __m256i*>(s + o)));
        if (n != whole) std::memcpy(d + whole, s + whole, n - whole);
    }
    const bool stream_copy_ = __builtin_cpu_supports("avx2") && std::getenv("FFS_SHM_PLAIN_READ") == nullptr;
#endif
    bool reentrant() const override { return true; }  // one file per frame
    ChunkCompression get_raw_chunk_compression() override { return BITSHUFFLE_LZ4; }
    h5read_dtype get_dtype() const override { return dtype_; }
    size_t get_number_of_images() const override { return n_images_; }
    std::array<size_t, 2> image_shape() const override { return shape_; }
    std::optional<std::span<const uint8_t>> get_mask() const override { return {{mask_.data(), mask_.size()}}; }
    std::array<int64_t, 2> get_trusted_range() const override { return trusted_; }
    std::optional<float> get_wavelength() const override { return wavelength_; }
    std::optional<std::array<float, 2>> get_pixel_size() const override { return {pixel_size_}; }
    std::optional<std::array<float, 2>> get_beam_center() const override { return {beam_center_}; }
    std::optional<float> get_detector_distance() const override { return distance_; }
    std::array<float, 2> get_oscillation() const override { return osc_; }
};
std::unique_ptr<Reader> make_shm_reader(const std::string& dir) { return std::make_unique<SHMRead>(dir); }

}  // namespace ffshost

template <> bool is_ready_for_read<ffshost::SHMRead>(const std::string& dir) {  // shmread.cc:90-95
    return fs::exists(dir + "/start_1") && fs::exists(dir + "/start_4");
}
template <> bool is_ready_for_read<ffshost::CBFRead>(const std::string& t) {  // cbfread.cc:127-134
    return fs::exists(ffshost::expand_template(t, 1));
}
