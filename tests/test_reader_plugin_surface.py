"""CPU: a reader written the way the reference's SHMRead / CBFRead are (global `class Reader`, `h5read_dtype`
get_dtype(), a specialisation of `is_ready_for_read<T>`: h5read/include/h5read.h:173-204,327-336,
spotfinder/shmread.hpp:10-67) compiles against host/reader.hpp unchanged, and can be handed to code that takes the
driver's `Reader&`."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include "reader.hpp"       // in place of <h5read.h>
    #include <cstdio>
    #include <cstring>

    // shaped like spotfinder/shmread.hpp: no knowledge of NONE or reentrant()
    class MyRead : public Reader {
        std::vector<uint8_t> mask_ = std::vector<uint8_t>(12, 1);
      public:
        bool is_image_available(size_t index) override { return index < 3; }
        std::span<uint8_t> get_raw_chunk(size_t, std::span<uint8_t> destination) override {
            std::memset(destination.data(), 7, 24);
            return destination.subspan(0, 24);
        }
        ChunkCompression get_raw_chunk_compression() override { return Reader::ChunkCompression::BITSHUFFLE_LZ4; }
        size_t get_number_of_images() const override { return 3; }
        h5read_dtype get_dtype() const override { return H5READ_DTYPE_UINT16; }
        std::array<int64_t, 2> get_trusted_range() const override { return {0, 65535}; }
        std::array<size_t, 2> image_shape() const override { return {3, 4}; }
        std::optional<std::span<const uint8_t>> get_mask() const override { return {{mask_.data(), mask_.size()}}; }
        std::optional<float> get_wavelength() const override { return 0.976f; }
        std::optional<std::array<float, 2>> get_pixel_size() const override { return {{75e-6f, 75e-6f}}; }
        std::optional<std::array<float, 2>> get_beam_center() const override { return {{1.5f, 2.0f}}; }
        std::optional<float> get_detector_distance() const override { return 0.2f; }
        std::array<float, 2> get_oscillation() const override { return {0.f, 0.1f}; }
    };
    template <>
    bool is_ready_for_read<MyRead>(const std::string& path) { return !path.empty(); }

    static size_t drive(ffshost::Reader& r) {   // what the driver does with any reader
        std::vector<uint8_t> buf(64);
        return r.get_element_size() * 100 + r.get_raw_chunk(0, buf).size() + (r.reentrant() ? 1000 : 0);
    }
    int main() {
        MyRead r;
        static_assert(H5READ_DTYPE_UINT16 == 2 && H5READ_DTYPE_UINT32 == 3 && H5READ_DTYPE_FLOAT64 == 8, "h5read.h:22-32");
        static_assert(Reader::BITSHUFFLE_LZ4 == 0 && Reader::BYTE_OFFSET_32 == 1, "h5read.h:175-178");
        std::printf("%zu %d\n", drive(r), (int)is_ready_for_read<MyRead>("x"));
        return 0;
    }
''')


def test_reference_shaped_reader_compiles_and_runs(tmp_path):
    src = tmp_path / "myread.cc"
    src.write_text(SRC)
    exe = tmp_path / "myread"
    inc = os.path.join(ROOT, "fast-feedback-service_amd", "host")
    subprocess.run(["g++", "-std=c++20", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert out == ["224", "1"]       # 2-byte pixels, 24-byte chunk, not reentrant by default; ready
