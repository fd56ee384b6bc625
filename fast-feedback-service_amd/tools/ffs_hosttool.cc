// ffs_hosttool -- host-side helper for tests and demos (no GPU):
//   ffs_hosttool selftest                 round-trip / known-answer checks of host/codecs.hpp
//   ffs_hosttool mkshm <synth:spec> <dir> write an Eiger-stream style directory the SHM reader takes
//                                         (start_1 JSON header, start_4, start_5 int32 mask,
//                                         image_%06d_2 bitshuffle-LZ4 chunks; spotfinder/shmread.cc)
//   ffs_hosttool mkcbf <synth:spec> <prefix>  write <prefix>0001.cbf ... (miniCBF, byte-offset codec)
//   ffs_hosttool mkh5 <synth:spec> <master.h5> [layout [frames_per_file [n_written]]]
//                                         NXmx master + data files (host/h5_writer.cc)
//   ffs_hosttool h5info <master.h5>       what the HDF5 reader sees; synthinfo prints the same checksums
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iterator>
#include <random>

#include "codecs.hpp"
#include "reader.hpp"

using namespace ffshost;

static int fail(const char* what) {
    std::printf("FAIL: %s\n", what);
    return 1;
}

static int selftest() {
    // LZ4 known answer: "aaaaaaaaaaaaaaaaaaaa" (20 x 'a') = token 0x1B? build by hand:
    // literals "a" (1), match offset 1 length 14+4... use a hand-assembled block instead:
    //   token 0x1F: 1 literal, match len 15+ext ; literal 'a'; offset 0x0001; ext 0 -> match 19 bytes
    {
        const uint8_t blk[] = {0x1F, 'a', 0x01, 0x00, 0x00, 0x50, 'b', 'c', 'd', 'e', 'f'};
        uint8_t out[64] = {0};
        const long n = lz4_block_decompress(blk, sizeof blk, out, sizeof out);
        if (n != 25 || std::memcmp(out, "aaaaaaaaaaaaaaaaaaaabcdef", 25) != 0) return fail("lz4 known answer");
    }
    std::mt19937_64 rng(12345);
    for (int round = 0; round < 40; ++round) {
        const size_t n = 1 + rng() % 70000;
        std::vector<uint8_t> data(n);
        const int mode = round % 4;
        for (size_t i = 0; i < n; ++i)
            data[i] = mode == 0 ? (uint8_t)rng() : mode == 1 ? (uint8_t)(i / 97) : mode == 2 ? (uint8_t)((rng() % 50) == 0) : 0;
        auto c = lz4_block_compress(data.data(), n);
        std::vector<uint8_t> back(n + 8);
        if (lz4_block_decompress(c.data(), c.size(), back.data(), n) != (long)n || std::memcmp(back.data(), data.data(), n))
            return fail("lz4 round trip");
    }
    for (size_t es : {2u, 4u})
        for (size_t nelem : {8u, 24u, 4096u, 4097u, 10007u, 70001u}) {
            std::vector<uint8_t> data(nelem * es);
            for (size_t i = 0; i < nelem; ++i) {
                const uint32_t v = (rng() % 100 == 0) ? (uint32_t)rng() : (uint32_t)(rng() % 7);
                std::memcpy(&data[i * es], &v, es);
            }
            if (nelem % 8 == 0) {
                std::vector<uint8_t> sh(nelem * es), un(nelem * es);
                bitshuffle_block(data.data(), sh.data(), nelem, es);
                bitunshuffle_block(sh.data(), un.data(), nelem, es);
                if (un != data) return fail("bitshuffle round trip");
                // plane 0 holds the LSBs: element i at byte i/8, bit i%8
                for (size_t i = 0; i < nelem; ++i)
                    if (((sh[i / 8] >> (i % 8)) & 1) != (data[i * es] & 1)) return fail("bit-plane layout");
            }
            auto c = bshuf_compress_lz4_with_header(data.data(), nelem, es);
            std::vector<uint8_t> back(nelem * es);
            if (bshuf_decompress_lz4(c.data() + 12, c.size() - 12, back.data(), nelem, es) < 0 || back != data)
                return fail("bitshuffle-lz4 round trip");
            uint64_t total = 0;
            for (int i = 0; i < 8; ++i) total = (total << 8) | c[i];
            if (total != nelem * es) return fail("bitshuffle header");
        }
    {
        std::vector<int32_t> v(5000);
        for (auto& x : v) {
            const int m = (int)(rng() % 20);
            x = m == 0 ? (int32_t)(rng() % 100000) - 20000 : m == 1 ? -1 : (int32_t)(rng() % 12);
        }
        auto c = byte_offset_compress(v.data(), v.size());
        std::vector<int32_t> back(v.size());
        if (byte_offset_decompress(c.data(), c.size(), back.data(), back.size()) != v.size() || back != v)
            return fail("byte-offset round trip");
        // known answer: deltas 1, 300 (escape to 16 bit), -70000 (escape to 32 bit)
        const uint8_t ka[] = {0x01, 0x80, 0x2C, 0x01, 0x80, 0x00, 0x80, 0x90, 0xEE, 0xFE, 0xFF};
        int32_t o[3];
        if (byte_offset_decompress(ka, sizeof ka, o, 3) != 3 || o[0] != 1 || o[1] != 301 || o[2] != 301 - 70000)
            return fail("byte-offset known answer");
    }
    std::printf("codecs selftest ok\n");
    return 0;
}

static int mkshm(const std::string& spec, const std::string& dir) {
    auto r = make_synth_reader(spec);
    std::filesystem::create_directories(dir);
    const size_t H = r->image_shape()[0], W = r->image_shape()[1], es = r->get_element_size();
    {
        std::ofstream f(dir + "/start_1");
        f << "{\"nimages\": " << r->get_number_of_images() << ", \"ntrigger\": 1, \"y_pixels_in_detector\": " << H
          << ", \"x_pixels_in_detector\": " << W << ", \"bit_depth_image\": " << es * 8
          << ", \"countrate_correction_count_cutoff\": " << r->get_trusted_range()[1] << ", \"wavelength\": 0.976"
          << ", \"detector_distance\": 300.0, \"y_pixel_size\": 7.5e-05, \"x_pixel_size\": 7.5e-05"
          << ", \"beam_center_y\": " << H / 2.0 << ", \"beam_center_x\": " << W / 2.0;
        if (r->get_oscillation()[1] > 0) f << ", \"omega_start\": 0.0, \"omega_increment\": " << r->get_oscillation()[1];
        f << "}\n";
    }
    { std::ofstream f(dir + "/start_4"); f << "\n"; }
    {
        std::vector<int32_t> m(W * H);
        auto mask = *r->get_mask();
        for (size_t i = 0; i < W * H; ++i) m[i] = mask[i] ? 0 : 1;  // pixel_mask: 0 = good
        std::ofstream f(dir + "/start_5", std::ios::binary);
        f.write(reinterpret_cast<const char*>(m.data()), (std::streamsize)(m.size() * 4));
    }
    std::vector<uint8_t> buf(W * H * es);
    for (size_t i = 0; i < r->get_number_of_images(); ++i) {
        r->get_raw_chunk(i, buf);
        auto c = bshuf_compress_lz4_with_header(buf.data(), W * H, es);
        char name[64];
        std::snprintf(name, sizeof name, "/image_%06zu_2", i);
        std::ofstream f(dir + name, std::ios::binary);
        f.write(reinterpret_cast<const char*>(c.data()), (std::streamsize)c.size());
    }
    return 0;
}

static int mkcbf(const std::string& spec, const std::string& prefix) {
    auto r = make_synth_reader(spec);
    const size_t H = r->image_shape()[0], W = r->image_shape()[1];
    if (r->get_element_size() != 2) return fail("mkcbf writes 16-bit sources only");
    std::vector<uint8_t> buf(W * H * 2);
    auto mask = *r->get_mask();
    for (size_t i = 0; i < r->get_number_of_images(); ++i) {
        r->get_raw_chunk(i, buf);
        std::vector<int32_t> v(W * H);
        for (size_t k = 0; k < W * H; ++k) v[k] = mask[k] ? reinterpret_cast<uint16_t*>(buf.data())[k] : -1;  // flagged = -1
        auto c = byte_offset_compress(v.data(), v.size());
        char name[32];
        std::snprintf(name, sizeof name, "%04zu.cbf", i + 1);
        std::ofstream f(prefix + name, std::ios::binary);
        f << "###CBF: VERSION 1.5\r\n\r\n_array_data.data\r\n;\r\n--CIF-BINARY-FORMAT-SECTION--\r\n"
          << "Content-Type: application/octet-stream;\r\n     conversions=\"x-CBF_BYTE_OFFSET\"\r\n"
          << "X-Binary-Size: " << c.size() << "\r\nX-Binary-Element-Type: \"signed 32-bit integer\"\r\n"
          << "X-Binary-Size-Fastest-Dimension: " << W << "\r\nX-Binary-Size-Second-Dimension: " << H << "\r\n\r\n"
          << "\x0c\x1a\x04\xd5";
        f.write(reinterpret_cast<const char*>(c.data()), (std::streamsize)c.size());
    }
    return 0;
}

// mkh5 <synth:spec> <master.h5> [layout [frames_per_file [n_written]]]
static int mkh5(int argc, char** argv) {
    auto r = make_synth_reader(argv[2]);
    const std::string layout = argc > 4 ? argv[4] : "vds-links";
    const size_t per_file = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 0;
    const size_t n_written = argc > 6 ? std::strtoull(argv[6], nullptr, 10) : r->get_number_of_images();
    h5_write_nxmx(*r, argv[3], layout, per_file, n_written);
    return 0;
}

// h5info <master.h5>: what the H5 reader sees (metadata + per-frame availability and a pixel checksum)
static int h5info(const std::string& file) {
    auto r = make_h5_reader(file);
    const size_t H = r->image_shape()[0], W = r->image_shape()[1], es = r->get_element_size();
    std::printf("images %zu shape %zu %zu bytes %zu trusted %lld %lld\n", r->get_number_of_images(), H, W, es,
                (long long)r->get_trusted_range()[0], (long long)r->get_trusted_range()[1]);
    std::printf("wavelength %.6g distance %.6g pixel %.6g %.6g beam %.6g %.6g osc %.6g %.6g\n", *r->get_wavelength(),
                *r->get_detector_distance(), (*r->get_pixel_size())[0], (*r->get_pixel_size())[1],
                (*r->get_beam_center())[0], (*r->get_beam_center())[1], r->get_oscillation()[0], r->get_oscillation()[1]);
    if (auto m = r->get_mask()) {
        size_t valid = 0;
        for (auto v : *m) valid += v;
        std::printf("mask valid %zu\n", valid);
    } else {
        std::printf("mask none\n");
    }
    std::vector<uint8_t> chunk(W * H * es + 4096), px(W * H * es);
    for (size_t i = 0; i < r->get_number_of_images(); ++i) {
        if (!r->is_image_available(i)) {
            std::printf("frame %zu unavailable\n", i);
            continue;
        }
        auto c = r->get_raw_chunk(i, chunk);
        if (c.size() < 12 || bshuf_decompress_lz4(c.data() + 12, c.size() - 12, px.data(), W * H, es) < 0)
            return fail("chunk does not decode");
        uint64_t h = 1469598103934665603ull;  // FNV-1a over the pixel bytes
        for (auto b : px) h = (h ^ b) * 1099511628211ull;
        std::printf("frame %zu chunk %zu fnv %016llx\n", i, c.size(), (unsigned long long)h);
    }
    return 0;
}

// synthinfo <synth:spec>: the same per-frame checksums straight from the synthetic source
static int synthinfo(const std::string& spec) {
    auto r = make_synth_reader(spec);
    const size_t H = r->image_shape()[0], W = r->image_shape()[1], es = r->get_element_size();
    std::vector<uint8_t> px(W * H * es);
    for (size_t i = 0; i < r->get_number_of_images(); ++i) {
        r->get_raw_chunk(i, px);
        uint64_t h = 1469598103934665603ull;
        for (auto b : px) h = (h ^ b) * 1099511628211ull;
        std::printf("frame %zu fnv %016llx\n", i, (unsigned long long)h);
    }
    return 0;
}

// chunkfnv <chunk file> <nelem> <elem bytes>: decode one bitshuffle-LZ4 chunk with the host codec and
// print the FNV-1a of the pixels (cross-check for chunk writers)
static int chunkfnv(const std::string& path, size_t nelem, size_t es) {
    std::ifstream f(path, std::ios::binary);
    std::vector<uint8_t> c((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> px(nelem * es);
    if (c.size() < 12 || bshuf_decompress_lz4(c.data() + 12, c.size() - 12, px.data(), nelem, es) < 0)
        return fail("chunk does not decode");
    uint64_t h = 1469598103934665603ull;
    for (auto b : px) h = (h ^ b) * 1099511628211ull;
    std::printf("fnv %016llx\n", (unsigned long long)h);
    return 0;
}

int main(int argc, char** argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    try {
        if (cmd == "selftest") return selftest();
        if (cmd == "mkshm" && argc == 4) return mkshm(argv[2], argv[3]);
        if (cmd == "mkcbf" && argc == 4) return mkcbf(argv[2], argv[3]);
        if (cmd == "mkh5" && argc >= 4) return mkh5(argc, argv);
        if (cmd == "h5info" && argc == 3) return h5info(argv[2]);
        if (cmd == "synthinfo" && argc == 3) return synthinfo(argv[2]);
        if (cmd == "chunkfnv" && argc == 5)
            return chunkfnv(argv[2], std::strtoull(argv[3], nullptr, 10), std::strtoull(argv[4], nullptr, 10));
        if (cmd == "h5stats" && argc == 4) { h5_print_group_stats(argv[2], argv[3]); return 0; }
        if (cmd == "h5support") { std::printf("%d\n", h5_supported() ? 1 : 0); return 0; }
    } catch (const std::exception& e) {
        std::printf("Error: %s\n", e.what());
        return 1;
    }
    std::printf("usage: ffs_hosttool selftest | mkshm <synth:spec> <dir> | mkcbf <synth:spec> <prefix> |\n"
                "       mkh5 <synth:spec> <master.h5> [vds-links|vds-files|plain [frames_per_file [n_written]]] |\n"
                "       h5info <master.h5> | synthinfo <synth:spec> | h5support | h5stats <file.h5> <group> |\n"
                "       chunkfnv <chunk> <nelem> <elem bytes>\n");
    return 2;
}
