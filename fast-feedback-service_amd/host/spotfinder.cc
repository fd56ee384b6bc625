// spotfinder.cc -- the `spotfinder` driver for libffs_hip.so: same command line, stdout phrases,
// --pipe_fd JSON lines, output files and exit codes as the reference's spotfinder/spotfinder.cc
// (flags :291-398, JSON :997-1008, per-image lines :1055-1087, 3D stage :1101-1148, summary
// :1308-1329), rebuilt around batches: every worker thread owns one ffs_stream, pulls a run of
// frame numbers from the shared counter (the reference pulls one, :752), decodes them into the
// stream's pinned buffer and submits the batch.  All device work goes through include/ffs_hip.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <condition_variable>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <pthread.h>
#include <sys/resource.h>
#include <sched.h>
#include <unistd.h>
#include <vector>

#include "codecs.hpp"
#include "ffs_hip.h"
#include "minijson.hpp"
#include "kabsch_space.hpp"
#include "reader.hpp"

using namespace ffshost;
using namespace std::chrono_literals;
namespace fs = std::filesystem;

#ifndef FFS_VERSION
#define FFS_VERSION "ffs-mi355x 0.1 (gfx950)"
#endif

static std::atomic<bool> g_stop{false};
extern "C" void stop_processing(int) {  // spotfinder.cc:43-54
    if (g_stop.load()) std::_Exit(1);
    static const char msg[] = "Running interrupted by user request\n";
    (void)!write(STDOUT_FILENO, msg, sizeof msg - 1);
    g_stop.store(true);
}

// ---- arguments (spotfinder.cc:291-398, src/ffs/arg_parser.cc, src/ffs/cuda_arg_parser.cc) ---------
struct Args {
    std::string file;
    bool sample = false, validate = false, writeout = false, save_h5 = false, output_for_index = false;
    bool verbose = false, strict_dtype = false;
    uint32_t threads = 1, images = 0, min_spot_size = 3, min_spot_size_3d = 3, start_index = 0, batch = 0, assemblies = 0;
    bool images_set = false, wavelength_set = false, detector_set = false;
    float max_sep = 2.0f, timeout = 30.0f, dmin = -1.f, dmax = -1.f, wavelength = 0.f, slot_margin = 2.0f;
    int pipe_fd = -1, device = 0;
    std::vector<int> devices;  // --devices / --gpus: the frame queue is dealt to all of them
    std::string algorithm = "dispersion", detector_json, gather = "host";
    std::string max_valid = "trusted";   // trusted | none | N
    uint32_t min_count = 2;
    bool cpu_decode = false, no_numa_pinning = false, single_buffer = false, all_threads = false, read_only = false, clean_exit = false;
};

static void usage() {
    std::printf(
      "Usage: spotfinder [-h] [--version] [-v] [-d DEVICE] [--list-devices] [--sample | FILE.nxs]\n"
      "                  [-n NUM] [--validate] [--images NUM] [--writeout] [--min-spot-size N]\n"
      "                  [--min-spot-size-3d N] [--max-peak-centroid-separation N] [--start-index N]\n"
      "                  [-t S] [-fd FD] [-a ALGO] [--dmin MIN D] [--dmax MAX D] [-w \xce\xbb] [--detector JSON]\n"
      "                  [-h5] [--output-for-index] [--batch N] [--cpu-decode] [--strict-dtype]\n"
      "                  [--max-valid trusted|none|N] [--min-count N]\n"
      "                  [--devices D0,D1,... | --gpus N] [--no-numa-pinning] [--single-buffer] [--all-threads] [--read-only] [--clean-exit]\n"
      "--max-valid: a centre pixel above this value is never strong (the reference's kernels test it against the\n"
      "              data set's trusted maximum).  trusted (default) = the frame source's trusted-range maximum when it is\n"
      "              below the pixel type's maximum, none = no test (the CPU baseline's behaviour), N = this value\n"
      "--min-count: valid pixels a 7x7 window needs (default 2, the CPU baseline's; the reference's kernels use 3)\n"
      "--validate:  every image is also decided by an independent path (every valid pixel's window gathered from memory,\n"
      "              no streaming kernel) and the two strong-pixel masks are compared: Match / Mismatch per image\n"
      "--devices / --gpus: one context and worker pool per GPU, all pulling frames from the one queue\n"
      "              (-n threads are dealt round-robin to the GPUs, at least one each); rotation sweeps send\n"
      "              their strong-pixel lists to the first GPU's 3D stack (RCCL over xGMI, else peer copies)\n"
      "--all-threads: every one of the -n threads reads (default: at most eight per GPU when chunks are decoded there)\n"
      "--gather host|rccl: with several GPUs and --output-for-index, where the spot centres of a round of batches (one per GPU) are\n"
      "              collected: read from each context's host arrays (default: seven times cheaper inside one process), or gathered\n"
      "              over RCCL to the first GPU (counts by all-gather, rows by send / recv, one copy to the host)\n"
      "--clean-exit: destroy streams and contexts and let the runtime tear itself down before the process ends (default: the\n"
      "              process leaves as soon as its last result is out -- a request's wall time ends there)\n"
      "--batch N:   frames per GPU batch (default 16 chunks / 4 decoded frames); a batch is filled by all readers of its GPU\n"
      "--single-buffer: one batch per GPU at a time (default: four of chunks / three of decoded frames, filled while the others are on the GPU)\n"
      "--read-only: (diagnostic) read every chunk into the staging areas and submit nothing\n"
      "environment, A/B only: FFS_SHM_PLAIN_READ=1 -- /dev/shm chunks by read() straight into the staging area instead of\n"
      "              through a cache-resident bounce buffer and non-temporal stores\n"
      "--cpu-decode: decompress bitshuffle-LZ4 chunks on the worker thread (the reference's way) instead\n"
      "              of sending them to the GPU as they are\n"
      "FILE: NXmx .nxs/.h5 (needs an HDF5 build), a /dev/shm directory, a ####.cbf template, or\n"
      "      synth:<eiger16m|jungfrau9m|plumbing1k|sweep16m|tiny|tinysweep>[:n_images[:seed]]\n");
}

[[noreturn]] static void arg_error(const std::string& m) {  // arg_parser.cc:72-77
    std::printf("Error: %s\n", m.c_str());
    usage();
    std::exit(1);
}

static void list_devices() {  // cuda_arg_parser.cc:39-53
    const int n = ffs_device_count();
    for (int i = 0; i < n; ++i) {
        char name[256];
        ffs_device_name(i, name, sizeof name);
        std::printf("%d: %s\n", i, name);
    }
    std::exit(0);
}

static Args parse_args(int argc, char** argv) {
    std::vector<std::string> a(argv + 1, argv + argc);
    if (fs::exists("common.args")) {  // arg_parser.cc:57-71
        std::ifstream f("common.args");
        std::string line;
        while (std::getline(f, line))
            if (!line.empty() && std::find(a.begin(), a.end(), line) == a.end()) a.push_back(line);
    }
    Args r;
    if (const char* e = std::getenv("SPOTFINDER_TIMEOUT")) {  // spotfinder.cc:293-301
        try { r.timeout = std::stof(e); } catch (...) { std::printf("Ignoring invalid SPOTFINDER_TIMEOUT value: %s\n", e); }
    }
    auto need = [&](size_t& i, const std::string& flag) -> const std::string& {
        if (i + 1 >= a.size()) arg_error("Too few arguments for '" + flag + "'.");
        return a[++i];
    };
    auto u32 = [&](const std::string& v, const std::string& flag) {
        try { size_t used; long long x = std::stoll(v, &used); if (used != v.size() || x < 0) throw 1; return (uint32_t)x; }
        catch (...) { arg_error("pattern not found for '" + flag + "': " + v); }
    };
    auto f32 = [&](const std::string& v, const std::string& flag) {
        try { size_t used; float x = std::stof(v, &used); if (used != v.size()) throw 1; return x; }
        catch (...) { arg_error("pattern not found for '" + flag + "': " + v); }
    };
    for (size_t i = 0; i < a.size(); ++i) {
        const std::string& s = a[i];
        if (s == "-h" || s == "--help") { usage(); std::exit(0); }
        else if (s == "--version") { std::printf("%s\n", FFS_VERSION); std::exit(0); }
        else if (s == "-v" || s == "--verbose") r.verbose = true;
        else if (s == "--list-devices") list_devices();
        else if (s == "-d" || s == "--device") r.device = (int)u32(need(i, s), s);
        else if (s == "--sample") r.sample = true;
        else if (s == "-n" || s == "--threads") r.threads = u32(need(i, s), s);
        else if (s == "--validate") r.validate = true;
        else if (s == "--images") { r.images = u32(need(i, s), s); r.images_set = true; }
        else if (s == "--writeout") r.writeout = true;
        else if (s == "--min-spot-size") r.min_spot_size = u32(need(i, s), s);
        else if (s == "--min-spot-size-3d") r.min_spot_size_3d = u32(need(i, s), s);
        else if (s == "--max-peak-centroid-separation") r.max_sep = f32(need(i, s), s);
        else if (s == "--start-index") r.start_index = u32(need(i, s), s);
        else if (s == "-t" || s == "--timeout") r.timeout = f32(need(i, s), s);
        else if (s == "-fd" || s == "--pipe_fd") r.pipe_fd = std::stoi(need(i, s));
        else if (s == "-a" || s == "--algorithm") r.algorithm = need(i, s);
        else if (s == "--cpu-decode") r.cpu_decode = true;
        else if (s == "--dmin") r.dmin = f32(need(i, s), s);
        else if (s == "--dmax") r.dmax = f32(need(i, s), s);
        else if (s == "-w" || s == "--wavelength" || s == "-\xce\xbb") { r.wavelength = f32(need(i, s), s); r.wavelength_set = true; }
        else if (s == "--detector") { r.detector_json = need(i, s); r.detector_set = true; }
        else if (s == "-h5" || s == "--save-h5") r.save_h5 = true;
        else if (s == "--output-for-index") r.output_for_index = true;
        else if (s == "--batch") r.batch = u32(need(i, s), s);
        else if (s == "--assemblies") r.assemblies = u32(need(i, s), s);   // batches in flight per GPU (tuning; default below)
        else if (s == "--slot-margin") r.slot_margin = f32(need(i, s), s);  // per cent of head room per chunk slot (tuning)
        else if (s == "--gpus") { const uint32_t n = u32(need(i, s), s); r.devices.clear(); for (uint32_t d = 0; d < n; ++d) r.devices.push_back((int)d); }
        else if (s == "--devices") {
            r.devices.clear();
            std::stringstream ss(need(i, s));
            std::string tok;
            while (std::getline(ss, tok, ',')) r.devices.push_back((int)u32(tok, s));
        }
        else if (s == "--gather") { r.gather = need(i, s); if (r.gather != "host" && r.gather != "rccl") arg_error("--gather takes host or rccl"); }
        else if (s == "--strict-dtype") r.strict_dtype = true;
        else if (s == "--max-valid") {
            r.max_valid = need(i, s);
            if (r.max_valid != "trusted" && r.max_valid != "none") (void)u32(r.max_valid, s);
        }
        else if (s == "--min-count") { r.min_count = u32(need(i, s), s); if (r.min_count < 2) arg_error("--min-count must be at least 2"); }
        else if (s == "--no-numa-pinning") r.no_numa_pinning = true;
        else if (s == "--single-buffer") r.single_buffer = true;
        else if (s == "--read-only") r.read_only = true;   // diagnostic: frames are read into the staging buffers and not submitted
        else if (s == "--all-threads") r.all_threads = true;
        else if (s == "--clean-exit") r.clean_exit = true;
        else if (!s.empty() && s[0] == '-' && s.size() > 1) arg_error("Unknown argument: " + s);
        else if (r.file.empty()) r.file = s;
        else arg_error("Maximum number of positional arguments exceeded");
    }
    const bool implicit_sample = std::getenv("H5READ_IMPLICIT_SAMPLE") != nullptr;  // spotfinder.cc:268-283
    if (r.sample && !r.file.empty()) arg_error("Argument 'FILE.nxs' not allowed with '--sample'");
    if (!r.sample && r.file.empty() && !implicit_sample) arg_error("One of the arguments '--sample' or 'FILE.nxs' is required");
    if (r.file.empty()) r.sample = true;
    return r;
}

struct DetectorGeometry {  // spotfinder/kernels/masking.cuh:16-80
    float pixel_size_x = 0, pixel_size_y = 0, beam_center_x = 0, beam_center_y = 0, distance = 0;
};

static DetectorGeometry detector_from_json(const std::string& text) {
    const JsonValue j = JsonParser(text).parse();
    for (const char* k : {"pixel_size_x", "pixel_size_y", "beam_center_x", "beam_center_y", "distance"})
        (void)j.at(k);  // throws "Key ... is missing from the input JSON"
    DetectorGeometry d;
    d.pixel_size_x = (float)j.at("pixel_size_x").number() / 1000.0f;  // mm -> m
    d.pixel_size_y = (float)j.at("pixel_size_y").number() / 1000.0f;
    d.beam_center_x = (float)j.at("beam_center_x").number() / (d.pixel_size_x * 1000);  // mm -> px
    d.beam_center_y = (float)j.at("beam_center_y").number() / (d.pixel_size_y * 1000);
    d.distance = (float)j.at("distance").number() / 1000.0f;
    return d;
}

// thread-safe writer of JSON lines to the inherited pipe (PipeHandler, spotfinder.cc:208-255)
class PipeHandler {
    int fd_;
    std::mutex m_;
  public:
    explicit PipeHandler(int fd) : fd_(fd) { std::printf("PipeHandler initialized with pipe_fd: %d\n", fd); }
    ~PipeHandler() { close(fd_); }
    void send(const std::string& line) { send_lines(line + "\n"); }
    // whole lines, each ending in a newline: one write for a batch's worth (a write per line was 7 000 system calls a second on
    // the one thread that also frees the GPU's staging areas)
    void send_lines(const std::string& s) {
        std::lock_guard<std::mutex> lock(m_);
        size_t at = 0;
        while (at < s.size()) {
            const ssize_t w = write(fd_, s.c_str() + at, s.size() - at);
            if (w == -1) { if (errno == EINTR) continue; std::cerr << "Error writing to pipe: " << std::strerror(errno) << std::endl; return; }
            at += (size_t)w;
        }
    }
};

// minimal PNG (stored deflate blocks) for --writeout, in place of lodepng
static void write_png_rgb(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    auto crc32 = [](const uint8_t* d, size_t n, uint32_t c) {
        static uint32_t table[256];
        static bool init = false;
        if (!init) {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t k = i;
                for (int j = 0; j < 8; ++j) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1;
                table[i] = k;
            }
            init = true;
        }
        c = ~c;
        for (size_t i = 0; i < n; ++i) c = table[(c ^ d[i]) & 255] ^ (c >> 8);
        return ~c;
    };
    std::ofstream f(path, std::ios::binary);
    auto be32 = [](uint32_t v, uint8_t* o) { o[0] = v >> 24; o[1] = v >> 16; o[2] = v >> 8; o[3] = v; };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        uint8_t len[4];
        be32((uint32_t)data.size(), len);
        f.write((const char*)len, 4);
        std::vector<uint8_t> td(type, type + 4);
        td.insert(td.end(), data.begin(), data.end());
        f.write((const char*)td.data(), (std::streamsize)td.size());
        uint8_t c[4];
        be32(crc32(td.data(), td.size(), 0), c);
        f.write((const char*)c, 4);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write((const char*)sig, 8);
    std::vector<uint8_t> ihdr(13);
    be32(w, &ihdr[0]);
    be32(h, &ihdr[4]);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr);
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (3 * w + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(n & 255); z.push_back(n >> 8); z.push_back(~n & 255); z.push_back((~n >> 8) & 255);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; ++i) { a = (a + raw[off + i]) % 65521; b = (b + a) % 65521; }
        off += n;
    }
    uint8_t ad[4];
    be32((b << 16) | a, ad);
    z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
}

static void write_mask_png(const std::string& path, const uint8_t* mask, uint32_t w, uint32_t h) {
    std::vector<uint8_t> img((size_t)w * h * 3, 255);  // spotfinder.cc:627-645
    for (size_t k = 0; k < (size_t)w * h; ++k)
        if (!mask[k]) { img[3 * k + 1] = 0; img[3 * k + 2] = 0; }
    write_png_rgb(path, img.data(), w, h);
}

template <typename T>
static std::string fmt_num(T v) { std::ostringstream o; o << v; return o.str(); }  // iostream default, :1138-1147

#define FFS_CHECK(ctx, expr)                                                       \
    do {                                                                           \
        if ((expr) != FFS_OK) {                                                    \
            std::printf("Error: %s\n", ffs_last_error(ctx));                       \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    const auto process_start = std::chrono::steady_clock::now();
    std::printf("Spotfinder version: %s\n", FFS_VERSION);
    Args args = parse_args(argc, argv);
    const std::string file = args.file;
    // -v: where the process's wall time goes outside the timed loop (the service starts one process per request, service.py:497,
    // so start-up and tear-down are what a short request sees): "[+ ms since main] step (ms it took)"
    auto stamp_prev = process_start;
    auto stamp = [&](const char* what) {
        if (!args.verbose) return;
        const auto t = std::chrono::steady_clock::now();
        std::printf("[%7.1f ms] %s (%.1f ms)\n", std::chrono::duration<double, std::milli>(t - process_start).count(), what,
                    std::chrono::duration<double, std::milli>(t - stamp_prev).count());
        stamp_prev = t;
    };

    int algorithm = FFS_ALGO_DISPERSION;
    {  // DispersionAlgorithm, spotfinder.cc:180-203
        std::string lower = args.algorithm;
        std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
        if (lower == "dispersion") std::printf("Algorithm: Dispersion\n");
        else if (lower == "dispersion_extended") {
            std::printf("Algorithm: Dispersion Extended\n");
            algorithm = FFS_ALGO_DISPERSION_EXTENDED;
        } else {
            std::printf("Error: Invalid algorithm specified\n");
            return 1;
        }
    }
    if (args.threads < 1) {
        std::printf("Error: Thread count must be >= 1\n");
        return 1;
    }
    struct JoinedThread {   // (an early `return` below must not meet a joinable std::thread)
        std::thread th;
        ~JoinedThread() { if (th.joinable()) th.join(); }
    };
    // The frame source is opened (header, the 72 MB pixel mask of an Eiger-16M stream directory: 35 ms) on a helper thread WHILE this
    // one initialises the HIP runtime (70-160 ms): nothing in it needs the GPU, and a request's latency is the wall time of the process
    // (DESIGN.md section 5b).  It prints only when it has to wait for its files or fails.
    std::unique_ptr<Reader> reader_ptr;
    int reader_rc = 0;
    std::atomic<bool> give_up{false};   // (no GPU after all: the helper stops waiting for files and says nothing more)
    auto open_reader = [&]() -> int {
        // ---- choose the reader (spotfinder.cc:438-466)
        auto wait_ready = [&](const std::string& path, auto checker) {  // wait_for_ready_for_read, :137-175
            const auto t0 = std::chrono::steady_clock::now();
            bool waited = false;
            while (!checker(path)) {
                if (give_up.load()) return;
                const double w = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                std::printf("\rWaiting for \033[1;35m%s\033[0m to be ready for read [%4.1f s] ", path.c_str(), w);
                std::fflush(stdout);
                waited = true;
                if (w > args.timeout) {
                    std::printf("\nError: Waited too long for read availability\n");
                    std::exit(1);
                }
                std::this_thread::sleep_for(80ms);
            }
            if (waited) std::printf("\n");
        };
        try {
            if (args.sample || file.rfind("synth:", 0) == 0) {
                reader_ptr = make_synth_reader(args.sample ? "synth:eiger16m:6" : file);
            } else {
                if (!fs::exists(file) && file.find('#') == std::string::npos)
                    wait_ready(file, [](const std::string& s) { return fs::exists(s); });
                if (fs::is_directory(file)) {
                    wait_ready(file, is_ready_for_read<SHMRead>);
                    reader_ptr = make_shm_reader(file);
                } else if (file.size() > 4 && file.compare(file.size() - 4, 4, ".cbf") == 0) {
                    if (!args.images_set) {
                        std::printf("Error: CBF reading must specify --images\n");
                        return 1;
                    }
                    reader_ptr = make_cbf_reader(file, args.images, args.start_index);
                } else {
                    wait_ready(file, is_ready_for_read<H5Read>);
                    reader_ptr = make_h5_reader(file);
                }
            }
        } catch (const std::exception& e) {
            if (!give_up.load()) std::printf("Error: %s\n", e.what());
            return 1;
        }
        return 0;
    };
    JoinedThread reader_holder;
    reader_holder.th = std::thread([&] { reader_rc = open_reader(); });
    stamp("arguments parsed");
    if (ffs_device_count() < 1) {  // cuda_arg_parser.cc:56-61
        give_up.store(true);
        std::printf("\033[1;31mError: Could not select GPU device\033[0m\n");
        return 1;
    }
    stamp("HIP runtime initialised (first device query)");
    {
        char name[256];
        if (ffs_device_name(args.device, name, sizeof name) != FFS_OK) {
            give_up.store(true);
            std::printf("\033[1;31mError: Could not select GPU device\033[0m\n");
            return 1;
        }
        std::printf("Using %s\n", name);
    }

    reader_holder.th.join();   // (opened beside the runtime's initialisation: see above)
    if (reader_rc != 0) return reader_rc;
    stamp("frame source opened (beside the runtime's initialisation)");
    Reader& reader = *reader_ptr;
    std::mutex reader_mutex;

    const size_t bytes_per_pixel = reader.get_element_size();
    {
        // The reference builds one binary per pixel width and exits with the data's bit depth on a
        // mismatch (spotfinder.cc:468-476) so that the service relaunches spotfinder32
        // (service.py:503-507).  This binary handles both widths; --strict-dtype (or being invoked
        // as spotfinder / spotfinder32 with FFS_STRICT_DTYPE set) restores the exit-code protocol.
        const std::string self = fs::path(argv[0]).filename().string();
        const size_t expect = self == "spotfinder32" ? 4 : 2;
        if ((args.strict_dtype || std::getenv("FFS_STRICT_DTYPE")) && bytes_per_pixel != expect) {
            std::printf("Error: Data type mismatch; This executable only accepts %zu bit != %zu\n", expect * 8,
                        bytes_per_pixel * 8);
            return (int)(bytes_per_pixel * 8);
        }
    }
    const uint32_t num_images = args.images_set ? args.images : (uint32_t)reader.get_number_of_images();
    const uint32_t height = (uint32_t)reader.image_shape()[0], width = (uint32_t)reader.image_shape()[1];
    // The reference hands the frame source's trusted maximum to every launch (spotfinder.cc:482,868,879) and its kernels
    // refuse centre pixels above it (kernels/thresholding.cu:208-215).  Here: --max-valid trusted (default) does the same
    // whenever that maximum says something, i.e. lies below the pixel type's own maximum (the reference narrows the int64 to
    // pixel_t, spotfinder.cu:155,167 -- a cut-off above 65535 on 16-bit data wraps there; here it means "no pixel is above it").
    const int64_t trusted_px_max = reader.get_trusted_range()[1];
    const int64_t type_max = bytes_per_pixel == 2 ? 65535ll : 4294967295ll;
    int64_t max_valid = -1;   // < 0: no test
    if (args.max_valid == "trusted") max_valid = (trusted_px_max >= 0 && trusted_px_max < type_max) ? trusted_px_max : -1;
    else if (args.max_valid != "none") max_valid = std::stoll(args.max_valid);

    // ---- detector geometry / wavelength (spotfinder.cc:484-587) -----------------------------------
    DetectorGeometry detector;
    if (args.detector_set) {
        try {
            detector = detector_from_json(args.detector_json);
        } catch (const std::exception& e) {
            std::printf("Error: %s\n", e.what());
            return 1;
        }
    } else {
        const auto bc = reader.get_beam_center();
        const auto ps = reader.get_pixel_size();
        const auto dd = reader.get_detector_distance();
        if (!bc) { std::printf("Error: No beam center available from file. Please pass detector metadata with --distance.\n"); return 1; }
        if (!ps) { std::printf("Error: No pixel size available from file. Please pass detector metadata with --distance.\n"); return 1; }
        if (!dd) { std::printf("Error: No detector distance available from file. Please pass metadata with --distance.\n"); return 1; }
        detector.distance = *dd;
        detector.beam_center_x = (*bc)[1];
        detector.beam_center_y = (*bc)[0];
        detector.pixel_size_x = (*ps)[1];
        detector.pixel_size_y = (*ps)[0];
    }
    float wavelength;
    if (args.wavelength_set) {
        wavelength = args.wavelength;
    } else {
        const auto w = reader.get_wavelength();
        if (!w) {
            std::printf("Error: No wavelength provided. Please pass wavelength using: --wavelength\n");
            return 1;
        }
        wavelength = *w;
        std::printf("Got wavelength from file: %f \xc3\x85\n", wavelength);
    }
    std::printf("Detector geometry:\n    Distance:    %.1f mm\n    Beam Center: %.1f px %.1f px\nBeam Wavelength: %.2f \xc3\x85\n",
                detector.distance * 1000, detector.beam_center_x, detector.beam_center_y, wavelength);
    const auto [oscillation_start, oscillation_width] = reader.get_oscillation();
    if (oscillation_width > 0)
        std::printf("Oscillation:  Start: %.2f\xc2\xb0  Width: %.2f\xc2\xb0\n", oscillation_start, oscillation_width);

    std::signal(SIGINT, stop_processing);
    stamp("frame source opened, header read");

    // ---- device context -------------------------------------------------------------------------------
    // bitshuffle-LZ4 chunks go to the GPU as they are (read straight into the pinned staging area) unless the pixels are
    // needed on the host (--writeout) or --cpu-decode asks for the reference's way
    const bool gpu_decode = reader.get_raw_chunk_compression() == Reader::BITSHUFFLE_LZ4 && !args.cpu_decode && !args.writeout;
    // frames per GPU batch.  Chunks: 16 for long runs (120 MB of staging per batch for Eiger-16M: the threshold kernels are tuned
    // for 16-32 frames and PCIe, not the GPU, is the limit), 8 for runs below 2048 images per GPU -- a stream's device buffers
    // (37 MB per frame of the batch) are prepared by the driver on FIRST USE at ~60 GB/s, so four assemblies of 16 frames cost a
    // run its first 40 ms, of 8 frames 20 ms (1000 frames: 6.0 k frames/s against 5.2-5.5 k).  Decoded frames are five times
    // larger: 4.  Never more than half the data set per GPU.
    const uint32_t n_dev_arg = (uint32_t)std::max<size_t>(1, args.devices.size());
    const uint32_t batch_dflt = !gpu_decode ? 4u : (num_images >= 2048u * n_dev_arg ? 16u : 8u);
    const uint32_t batch = args.batch ? args.batch : std::max<uint32_t>(1, std::min<uint32_t>(batch_dflt, num_images / (2 * n_dev_arg)));
    std::printf("Image:       %4u x %4u = %u px\n", width, height, width * height);
    std::printf("GPU batches: %u frames per submit, filled by all readers of a GPU; %s\n", batch,
                args.single_buffer ? "one batch in flight" : (gpu_decode ? "four batches in flight per GPU" : "three batches in flight per GPU"));
    std::printf("Running with %u CPU threads\n", args.threads);

    // One context per GPU (the reference has one device, -d: src/ffs/cuda_arg_parser.cc:30-61).  With
    // --devices / --gpus the one frame queue (next_image below) feeds the worker threads of all of them.
    std::vector<int> devices = args.devices.empty() ? std::vector<int>{args.device} : args.devices;
    {
        const int have = ffs_device_count();
        for (int d : devices)
            if (d < 0 || d >= have) {
                std::printf("Error: device %d does not exist (%d visible)\n", d, have);
                return 1;
            }
    }
    const uint32_t n_dev = (uint32_t)devices.size();
    if (args.threads < n_dev) args.threads = n_dev;  // at least one worker per GPU
    // The exchange between GPUs (RCCL communicators: 2-5 s to load the library and initialise, measured in round 5 -- it was what two
    // contexts cost a 1000-image request) serves rotation sweeps only -- every frame's strong-pixel list has to reach the GPU that owns
    // the 3D stack -- so stills never pay it, and a sweep initialises it on a helper thread beside the contexts and the first batches
    // (joined before the first batch is added to the stack).
    const bool gather_rccl = n_dev > 1 && args.gather == "rccl" && args.output_for_index && !(oscillation_width > 0);
    const bool need_exchange = n_dev > 1 && (oscillation_width > 0 || gather_rccl);
    JoinedThread multi_holder;
    std::thread& multi_thread = multi_holder.th;
    std::once_flag multi_joined;
    auto join_multi = [&] { std::call_once(multi_joined, [&] { if (multi_thread.joinable()) multi_thread.join(); }); };
    if (n_dev > 1) {
        std::string list;
        for (int d : devices) list += (list.empty() ? "" : ", ") + std::to_string(d);
        // (--gather rccl names the transport: contexts that share a GPU then send to their own rank, which a one-GPU box can rehearse)
        if (need_exchange) multi_thread = std::thread([&devices, n_dev, gather_rccl] { (void)ffs_multi_init(devices.data(), (int)n_dev, gather_rccl ? "rccl" : nullptr); });
        const char* env = std::getenv("FFS_GATHER");
        std::printf("GPUs:        %s (frame queue shared; exchange of rotation lists: %s)\n", list.c_str(),
                    oscillation_width > 0 ? (env ? env : "rccl") : "none needed (stills)");
        if (gather_rccl) std::printf("Spot lists:  gathered over RCCL to GPU %d, one round of batches (one per GPU) at a time\n", devices[0]);
        stamp("exchange between GPUs set going");
    }
    std::vector<ffs_ctx*> ctxs(n_dev, nullptr);
    for (uint32_t di = 0; di < n_dev; ++di) {
        if (ffs_ctx_create(devices[di], width, height, (int)bytes_per_pixel, batch, 0, &ctxs[di]) != FFS_OK) {
            std::printf("Error: %s\n", ffs_last_error(nullptr));
            return 1;
        }
        stamp("ffs_ctx_create (code object, mask tables, shared HIP streams)");
    }
    ffs_ctx* ctx = ctxs[0];  // owns the 3D stack; also the context the one-off steps below report from
    {  // upload_mask, spotfinder.cc:61-108
        size_t valid = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const auto mask = reader.get_mask();
        if (mask) for (uint8_t v : *mask) valid += v != 0;
        else valid = (size_t)width * height;
        for (ffs_ctx* cx : ctxs) FFS_CHECK(cx, ffs_ctx_set_mask(cx, mask ? mask->data() : nullptr));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Uploaded mask (%.2f Mpx) in %.2f ms (%.1f GBps)\n", valid / 1e6, ms,
                    (double)width * height * n_dev / (ms * 1e-3) / 1e9);
    }
    if (const auto mask = reader.get_mask(); args.writeout && mask) write_mask_png("mask_source.png", mask->data(), width, height);
    if (args.dmin > 0 || args.dmax > 0) {  // spotfinder.cc:648-683
        for (ffs_ctx* cx : ctxs)
            FFS_CHECK(cx, ffs_ctx_apply_resolution_mask(cx, wavelength, detector.distance, detector.beam_center_x,
                                                        detector.beam_center_y, detector.pixel_size_x,
                                                        detector.pixel_size_y, args.dmin, args.dmax));
        if (args.writeout) {
            std::vector<uint8_t> m((size_t)width * height);
            FFS_CHECK(ctx, ffs_ctx_get_mask(ctx, m.data()));
            write_mask_png("mask_calculated.png", m.data(), width, height);
        }
    }
    const bool rotation = oscillation_width > 0;
    ffs_params prm;
    ffs_default_params(&prm);
    prm.min_spot_size = args.min_spot_size;
    prm.min_spot_size_3d = args.min_spot_size_3d;
    prm.max_peak_centroid_separation = args.max_sep;
    prm.want_reflections = (!rotation && (args.save_h5 || args.output_for_index)) ? 1 : 0;
    prm.want_strong_mask = args.writeout ? 1 : 0;
    prm.algorithm = algorithm;
    prm.max_valid = max_valid;
    prm.min_count = (int32_t)args.min_count;
    if (args.validate) prm.want_strong_mask = 1;   // the masks are what --validate compares (spotfinder.cc:1012-1053)
    for (ffs_ctx* cx : ctxs) FFS_CHECK(cx, ffs_ctx_set_params(cx, &prm));
    if (max_valid >= 0) std::printf("Trusted range: centre pixels above %lld are not spots\n", (long long)max_valid);
    // --validate (spotfinder.cc:1012-1053 compares every image with the CPU baseline, which is test infrastructure here and
    // not linked into the product): every batch also goes through a second context per GPU whose threshold stage shares
    // nothing with the hot path's but the predicate -- no streaming kernel, no screen, no queue: the window of EVERY valid
    // pixel is gathered from memory (tuning "threshold_path" = 2; the extended algorithm: its plain one-pixel-per-lane first
    // pass and the grid-wide sparse kernels) -- and the two strong-pixel masks are compared image by image.
    std::vector<ffs_ctx*> vctxs(n_dev, nullptr);
    if (args.validate) {
        for (uint32_t di = 0; di < n_dev; ++di) {
            if (ffs_ctx_create(devices[di], width, height, (int)bytes_per_pixel, batch, 0, &vctxs[di]) != FFS_OK) {
                std::printf("Error: %s\n", ffs_last_error(nullptr));
                return 1;
            }
            ffs_ctx* v = vctxs[di];
            FFS_CHECK(v, ffs_ctx_set_tuning(v, "threshold_path", 2));
            FFS_CHECK(v, ffs_ctx_set_tuning(v, "ext_first_pass", 0));
            FFS_CHECK(v, ffs_ctx_set_tuning(v, "sparse_stage", 1));
            FFS_CHECK(v, ffs_ctx_set_tuning(v, "strong_log", 0));
            std::vector<uint8_t> m((size_t)width * height);   // the mask as the first context holds it (resolution filter included)
            FFS_CHECK(ctxs[di], ffs_ctx_get_mask(ctxs[di], m.data()));
            FFS_CHECK(v, ffs_ctx_set_mask(v, m.data()));
            FFS_CHECK(v, ffs_ctx_set_params(v, &prm));
        }
        std::printf("Validation: every image is also decided by the gather path (every valid pixel's window summed from memory)\n");
    }
    if (args.save_h5 && !h5_supported()) {
        std::printf("Error: --save-h5 needs an HDF5-enabled build\n");
        return 1;
    }

    std::printf("Dataset type: %s\n", rotation ? "Rotation set" : "Still set");
    ffs_stack3d* stack = nullptr;
    std::mutex print_mutex;
    if (rotation) FFS_CHECK(ctx, ffs_stack3d_create(ctx, 0, &stack));

    std::unique_ptr<PipeHandler> pipe;
    if (args.pipe_fd != -1) pipe = std::make_unique<PipeHandler>(args.pipe_fd);
    stamp("masks uploaded, parameters set: the timed loop starts");

    const auto all_start = std::chrono::steady_clock::now();
    std::atomic<uint32_t> next_image{0};
    std::atomic<uint32_t> completed{0};
    std::mutex retired_mutex;
    std::vector<ffs_stream*> retired_streams;   // the workers' streams, destroyed after the summary
    std::atomic<double> time_waiting_acc{0.0};
    std::atomic<int> failed{0};
    std::map<uint32_t, std::vector<float>> reflection_centers_2d;  // spotfinder.cc:706-708
    std::mutex reflection_centers_2d_mutex;

    // Workers are dealt round-robin to the GPUs; each is kept on the CPUs of its GPU's NUMA node (where the node is known
    // and leaves it CPUs of this process's affinity set), so that the frames it reads, its pinned staging buffer -- first
    // touched by hipHostMalloc on this thread -- and the GPU's PCIe root sit on one socket.  --no-numa-pinning turns it off.
    std::vector<cpu_set_t> node_cpus(n_dev);
    std::vector<bool> node_known(n_dev, false);
    if (!args.no_numa_pinning) {
        cpu_set_t allowed;
        CPU_ZERO(&allowed);
        sched_getaffinity(0, sizeof allowed, &allowed);
        for (uint32_t di = 0; di < n_dev; ++di) {
            const int node = ffs_device_numa_node(devices[di]);
            if (node < 0) continue;
            std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
            std::string list;
            if (!std::getline(f, list)) continue;
            cpu_set_t set;
            CPU_ZERO(&set);
            std::stringstream ss(list);
            std::string tok;
            int n_set = 0;
            while (std::getline(ss, tok, ',')) {
                const size_t dash = tok.find('-');
                const int lo = std::atoi(tok.c_str()), hi = dash == std::string::npos ? lo : std::atoi(tok.c_str() + dash + 1);
                for (int cpu = lo; cpu <= hi && cpu < CPU_SETSIZE; ++cpu)
                    if (CPU_ISSET(cpu, &allowed)) { CPU_SET(cpu, &set); ++n_set; }
            }
            if (n_set > 0) { node_cpus[di] = set; node_known[di] = true; }
        }
        if (args.verbose)
            for (uint32_t di = 0; di < n_dev; ++di)
                std::printf("GPU %d: NUMA node %d%s\n", devices[di], ffs_device_numa_node(devices[di]),
                            node_known[di] ? ", workers pinned to its CPUs" : " (workers not pinned)");
    }

    // ---- batches assembled from several readers (round 4) ---------------------------------------------------------------
    // The reference gives every worker thread one frame at a time: read, decompress, copy, kernel, copy back, connected
    // components, in sequence (spotfinder.cc:751-1008).  Rounds 2-3 gave every worker batches of its own -- four frames, an
    // eighth of what the kernels are tuned for, and sixteen small submissions in flight.  Now a GPU batch is an ASSEMBLY that all
    // the GPU's reader threads fill together: global batch b holds images b B .. b B + B - 1, goes to GPU b mod n_dev, and sits
    // in assembly (b / n_dev) mod K of that GPU (an ffs_stream with its pinned staging area cut into B slots).  A reader takes
    // the next slot number from the GPU's counter, reads that image's chunk into its slot, and whoever fills a batch's last
    // slot submits it.  One collector thread per GPU waits for the batches in order, hands out their results (in frame order)
    // and frees the assembly for batch b + K n_dev.  Readers never wait for the GPU unless all K assemblies are in flight.
    struct Assembly {
        ffs_stream* s = nullptr;
        ffs_stream* v = nullptr;       // --validate: the same batch on the validation context
        uint8_t* host = nullptr;       // the stream's pinned staging area (allocated when the assembly is first claimed)
        size_t host_bytes = 0, slot_bytes = 0;
        size_t over_at = 0, over_used = 0;   // overflow area behind the slots
        std::unique_ptr<std::mutex> over_mu = std::make_unique<std::mutex>();
        std::vector<const void*> chunk_ptr;
        std::vector<size_t> chunk_len;
        std::vector<std::vector<uint8_t>> spill;   // chunks that did not fit their slot (their whole batch then goes up from here)
        std::vector<uint8_t> slot_filled;          // (under the GPU's mutex) which slots of the batch hold their image
        // state, under the GPU's mutex
        int64_t batch = -1;            // the global batch this assembly holds, -1: free
        uint64_t next_q = 0;           // the GPU-local batch number it serves next (claims happen in order)
        uint32_t n = 0, filled = 0;
        bool ready = false;            // staging area in place: slots may be filled
        bool creating = false;         // somebody is making the stream and pinning the staging area
        bool submitted = false, skipped = false;
        int submitted_by = 0;
    };
    struct Gpu {
        ffs_ctx* ctx = nullptr;
        ffs_ctx* vctx = nullptr;
        uint32_t index = 0;
        std::vector<Assembly> as;
        std::mutex mu;
        std::condition_variable cv;
        std::atomic<uint64_t> next_slot{0};
    };
    const size_t frame_bytes = (size_t)width * height * bytes_per_pixel;
    const uint32_t K = args.single_buffer ? 1u : args.assemblies ? std::min(args.assemblies, 32u) : (gpu_decode ? 4u : 3u);
    const uint64_t total_batches = ((uint64_t)num_images + batch - 1) / batch;
    std::vector<std::unique_ptr<Gpu>> gpus;
    for (uint32_t di = 0; di < n_dev; ++di) {
        auto g = std::make_unique<Gpu>();
        g->ctx = ctxs[di];
        g->vctx = args.validate ? vctxs[di] : nullptr;
        g->index = di;
        g->as.resize(K);
        for (uint32_t k = 0; k < K; ++k) {
            Assembly& A = g->as[k];
            A.next_q = k;
            A.chunk_ptr.resize(batch);
            A.chunk_len.resize(batch);
            A.spill.resize(batch);
            A.slot_filled.assign(batch, 0);
        }
        gpus.push_back(std::move(g));
    }
    std::atomic<bool> readers_done{false};   // no batch will be submitted any more: collectors stop at the first one that is missing
    std::atomic<size_t> chunk_estimate{0};   // staging bytes per compressed chunk, from the first chunk anybody reads
    std::atomic<uint32_t> validate_mismatches{0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    auto wake_all = [&]() { for (auto& g : gpus) { std::lock_guard<std::mutex> lock(g->mu); g->cv.notify_all(); } };
    auto fail = [&](const char* what, ffs_ctx* cx) {
        std::printf("Error: %s%s\n", what, cx ? ffs_last_error(cx) : "");
        failed = 1;
        wake_all();
    };

    // ---- --gather rccl: the spot centres of one ROUND of batches (local batch q of every GPU) through ffs_multi_gather_rows ----------
    // The collectors meet once per round; the last to arrive runs the collective for all (counts by ncclAllGather, rows by ncclSend /
    // ncclRecv to the first GPU, one copy to the host) and everybody takes its own batch's rows from the gathered table.  A round
    // that cannot fill up (the data set ends, an interrupt) is served from the host arrays, as `--gather host` serves all of them.
    struct GatherRound {
        std::mutex mu;
        std::condition_variable cv;
        std::vector<ffs_stream*> streams;
        std::vector<uint32_t> n_rows, row_at;   // per GPU: rows of its batch, where they start in `rows`
        std::vector<float> rows;
        uint32_t arrived = 0;
        uint64_t round = 0;                      // rounds served so far
        bool ok = false;                         // the round just served went through RCCL
    } gr;
    gr.streams.assign(n_dev, nullptr);
    gr.n_rows.assign(n_dev, 0);
    gr.row_at.assign(n_dev, 0);
    const uint32_t gather_cap = 1u << 20;
    if (gather_rccl) gr.rows.resize((size_t)gather_cap * 4);
    std::vector<int> gpu_rank(n_dev, 0);   // rank of a GPU's device in the communicator: distinct devices in order of appearance (ffs_multi_init)
    {
        std::vector<int> distinct;
        for (uint32_t di = 0; di < n_dev; ++di) {
            auto it = std::find(distinct.begin(), distinct.end(), devices[di]);
            if (it == distinct.end()) { distinct.push_back(devices[di]); gpu_rank[di] = (int)distinct.size() - 1; }
            else gpu_rank[di] = (int)(it - distinct.begin());
        }
    }
    std::atomic<uint64_t> rccl_rounds{0}, host_rounds{0};
    // -> pointer to this GPU's rows (frame id bits, x, y, z) of the round, or nullptr: read them from the host arrays
    auto gather_round = [&](uint32_t di, uint64_t q, ffs_stream* st, uint32_t my_rows) -> const float* {
        const bool full_round = (q + 1) * n_dev <= total_batches;   // every GPU has a local batch q
        if (!full_round) { host_rounds += 1; return nullptr; }
        std::unique_lock<std::mutex> lock(gr.mu);
        // (waits in slices: a failure or the end of the readers is announced on the GPUs' condition variables, not on this one)
        while (!(gr.round == q || failed.load() || readers_done.load() || g_stop.load())) gr.cv.wait_for(lock, 20ms);   // the previous round has been taken by everybody
        if (gr.round != q) { host_rounds += 1; return nullptr; }
        gr.streams[di] = st;
        gr.n_rows[di] = my_rows;
        if (++gr.arrived == n_dev) {
            join_multi();
            // rows arrive rank after rank, inside a rank in the order of `streams` (= GPU order)
            uint32_t at = 0;
            for (int r = 0; r < (int)n_dev; ++r)
                for (uint32_t g = 0; g < n_dev; ++g)
                    if (gpu_rank[g] == r) { gr.row_at[g] = at; at += gr.n_rows[g]; }
            uint32_t got = 0;
            gr.ok = at <= gather_cap && ffs_multi_gather_rows(gr.streams.data(), n_dev, 0, gr.rows.data(), gather_cap, &got) == FFS_OK && got == at;
            (gr.ok ? rccl_rounds : host_rounds) += 1;
            gr.arrived = 0;
            gr.round = q + 1;
            gr.cv.notify_all();
        } else {
            while (!(gr.round > q || failed.load() || readers_done.load() || g_stop.load())) gr.cv.wait_for(lock, 20ms);
            if (gr.round <= q) {   // the round cannot fill up any more (the readers have stopped): everybody reads the host arrays
                gr.arrived = 0;
                host_rounds += 1;
                return nullptr;
            }
        }
        return gr.ok ? gr.rows.data() + (size_t)gr.row_at[di] * 4 : nullptr;
    };

    // ---- the collector of one GPU: results of its batches, in order (the reference's post-processing of an image, :901-1087) ----
    auto collector = [&](uint32_t di) {
        Gpu& G = *gpus[di];
        ffs_ctx* ctx = G.ctx;
        if (node_known[di]) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &node_cpus[di]);
        double t_wait = 0, t_emit = 0;
        uint32_t n_batches = 0;
        for (uint64_t q = 0;; ++q) {
            const uint64_t b = q * n_dev + di;
            if (b >= total_batches) break;
            Assembly& A = G.as[q % K];
            {
                std::unique_lock<std::mutex> lock(G.mu);
                // (an interrupt or a time-out stops the READERS; what has been submitted -- and, below, every image that was read --
                // still comes out, as the reference's workers finish the image they hold: spotfinder.cc:770-790)
                G.cv.wait(lock, [&] { return (A.batch == (int64_t)b && A.submitted) || readers_done.load() || failed.load(); });
                if (failed.load() || !(A.batch == (int64_t)b && A.submitted)) break;
            }
            const uint32_t thread_id = (uint32_t)A.submitted_by;
            if (!A.skipped) {
                const ffs_frame_result* res = nullptr;
                uint32_t nres = 0;
                const auto t_w0 = now();
                const int wrc = ffs_wait(A.s, &res, &nres);
                t_wait += secs(t_w0, now());
                const auto t_e0 = now();
                if (wrc != FFS_OK) { fail("", ctx); break; }
                float tm[5] = {0};
                ffs_stream_timings(A.s, tm);
                const ffs_frame_result* vres = nullptr;
                if (A.v) {
                    uint32_t nv = 0;
                    if (ffs_wait(A.v, &vres, &nv) != FFS_OK || nv != nres) { fail("validation pass: ", G.vctx); break; }
                }
                if (rotation) {
                    // key = image number read (rotation_slices[offset_image_num], :913-918); the stack has its own lock (the
                    // reference's rotation_slices_mutex), held only while the transfer is enqueued
                    join_multi();   // (the exchange's communicators, initialised beside the run so far)
                    if (ffs_stack3d_add_batch(stack, A.s) != FFS_OK) { fail("", ctx); break; }
                }
                const float* round_rows = nullptr;   // --gather rccl: this batch's centre rows as they came back from the collective
                if (gather_rccl) {
                    uint32_t my_rows = 0;
                    for (uint32_t i = 0; i < nres; ++i) my_rows += res[i].n_reflections;
                    round_rows = gather_round(di, q, A.s, my_rows);
                }
                // what this batch prints and sends goes out in one piece each (the collector is the one thread between the GPU and a
                // free staging area: a printf and a write per image, into pipes a Python caller drains, were on that path)
                std::string text, json_lines;
                char line_buf[512];
                for (uint32_t i = 0; i < nres; ++i) {
                    const ffs_frame_result& r = res[i];
                    const uint32_t image_num = (uint32_t)r.frame_id;
                    if (args.writeout && r.strong_mask) {  // :937-994
                        const uint8_t* px = A.host + (size_t)i * frame_bytes;
                        std::vector<uint8_t> img((size_t)width * height * 3);
                        for (size_t k = 0; k < (size_t)width * height; ++k) {
                            const float v = bytes_per_pixel == 2 ? (float)reinterpret_cast<const uint16_t*>(px)[k]
                                                                 : (float)reinterpret_cast<const uint32_t*>(px)[k];
                            const uint8_t g = (uint8_t)std::max(0.0f, 255.99f - v * 10);
                            img[3 * k] = img[3 * k + 1] = img[3 * k + 2] = g;
                        }
                        auto put = [&](long x, long y) {
                            if (x >= 0 && y >= 0 && x < (long)width && y < (long)height) {
                                const size_t k = (size_t)y * width + x;
                                img[3 * k] = 0; img[3 * k + 1] = 0; img[3 * k + 2] = 255;
                            }
                        };
                        for (uint32_t bi = 0; bi < r.n_boxes; ++bi) {
                            const ffs_box& bx = r.boxes[bi];
                            for (int e = 5; e <= 7; ++e) {
                                for (long x = (long)bx.l - e; x <= (long)bx.r + e; ++x) { put(x, (long)bx.t - e); put(x, (long)bx.b + e); }
                                for (long y = (long)bx.t - e; y <= (long)bx.b + e; ++y) { put((long)bx.l - e, y); put((long)bx.r + e, y); }
                            }
                        }
                        char name[64];
                        std::snprintf(name, sizeof name, "pixels_%05u.txt", image_num);
                        std::ofstream out(name);
                        for (uint32_t y = 0, k = 0; y < height; ++y)
                            for (uint32_t x = 0; x < width; ++x, ++k)
                                if (r.strong_mask[k]) {
                                    img[3 * k] = 255; img[3 * k + 1] = 0; img[3 * k + 2] = 0;
                                    char line[32];
                                    std::snprintf(line, sizeof line, "%4u, %4u\n", x, y);
                                    out << line;
                                }
                        std::snprintf(name, sizeof name, "image_%05u.png", image_num);
                        write_png_rgb(name, img.data(), width, height);
                    }
                    if (args.save_h5 && !rotation) {  // :919-933
                        std::vector<float> coms;
                        for (uint32_t qq = 0; qq < r.n_reflections; ++qq) {
                            coms.push_back(r.reflections[qq].com_x);
                            coms.push_back(r.reflections[qq].com_y);
                            coms.push_back(r.reflections[qq].com_z);
                        }
                        std::lock_guard<std::mutex> lock(reflection_centers_2d_mutex);
                        reflection_centers_2d[image_num + args.start_index] = std::move(coms);
                    }
                    if (pipe) {  // keys in alphabetical order, as nlohmann dumps them (:997-1008)
                        std::string j = "{\"file\":" + json_escape(file) + ",\"file-number\":" + std::to_string(image_num)
                                        + ",\"n_spots_total\":" + std::to_string(r.n_boxes)
                                        + ",\"num_strong_pixels\":" + std::to_string(r.num_strong_pixels);
                        if (args.output_for_index) {
                            j += ",\"spot_centers\":[";
                            for (uint32_t qq = 0; qq < r.n_reflections; ++qq) {
                                if (qq) j += ",";
                                if (round_rows) {   // (frame id bits, x, y, z) rows in frame order: the next n_reflections are this image's
                                    j += json_number(round_rows[1]) + "," + json_number(round_rows[2]) + "," + json_number(round_rows[3]);
                                    round_rows += 4;
                                } else {
                                    j += json_number(r.reflections[qq].com_x) + "," + json_number(r.reflections[qq].com_y) + ","
                                         + json_number(r.reflections[qq].com_z);
                                }
                            }
                            j += "]";
                        }
                        json_lines += j;
                        json_lines += "}\n";
                    }
                    if (vres) {  // :1012-1053
                        const ffs_frame_result& v = vres[i];
                        const bool same = r.strong_mask && v.strong_mask && std::memcmp(r.strong_mask, v.strong_mask, (size_t)width * height) == 0
                                          && r.num_strong_pixels == v.num_strong_pixels && r.n_boxes == v.n_boxes;
                        if (same) std::snprintf(line_buf, sizeof line_buf, "Thread %2u, Image %4u: Compared: \033[32mMatch %u px\033[0m\n", thread_id, image_num, r.num_strong_pixels);
                        else {
                            std::snprintf(line_buf, sizeof line_buf, "Thread %2u, Image %4u: Compared: \033[1;31mMismatch (%u px from kernel)\033[0m\n", thread_id, image_num, r.num_strong_pixels);
                            validate_mismatches += 1;
                        }
                        text += line_buf;
                    }
                    std::snprintf(line_buf, sizeof line_buf, "Extracted %u spots\n", r.n_components);  // connected_components.cc:119
                    text += line_buf;
                    if (prm.min_spot_size > 0) {
                        std::snprintf(line_buf, sizeof line_buf, "Removed %u spots with size < %u pixels\n", r.n_components - r.n_boxes, prm.min_spot_size);
                        text += line_buf;
                    }
                    if (prm.want_reflections && r.n_filtered_sep > 0) {
                        std::snprintf(line_buf, sizeof line_buf, "Filtered %u spots with peak-centroid distance > %s\n", r.n_filtered_sep, fmt_num(prm.max_peak_centroid_separation).c_str());
                        text += line_buf;
                    }
                    if (args.threads == 1) {  // :1056-1076 (timings are per batch here)
                        std::snprintf(line_buf, sizeof line_buf, "Thread %2u finished image %4u\n       Copy: %5.1f ms\n     Kernel: %5.1f ms\n  Post Copy: %5.1f ms\n"
                                    "       Post: %5.1f ms\n             \xe2\x95\x90\xe2\x95\x90\xe2\x95\x90\xe2\x95\x90\xe2\x95\x90\xe2\x95\x90\xe2\x95\x90\xe2\x95\x90\n"
                                    "     Total:  %5.1f ms (%.1f GBps)\n    %u strong pixels\n    %u filtered reflections (%u pixels)\n",
                                    thread_id, image_num, tm[0] / nres, tm[1] / nres, tm[3] / nres, tm[2] / nres, tm[4] / nres,
                                    (double)frame_bytes * nres / (tm[4] * 1e-3) / 1e9, r.num_strong_pixels, r.n_boxes,
                                    r.num_strong_pixels_filtered);
                    } else {  // :1078-1085
                        std::snprintf(line_buf, sizeof line_buf, "Thread %2u finished image %4u with %5u strong pixels, %4u filtered reflections (%u pixels)\n",
                                    thread_id, image_num, r.num_strong_pixels, r.n_boxes, r.num_strong_pixels_filtered);
                    }
                    text += line_buf;
                    completed += 1;
                }
                if (pipe && !json_lines.empty()) pipe->send_lines(json_lines);
                {
                    std::lock_guard<std::mutex> lock(print_mutex);
                    std::fwrite(text.data(), 1, text.size(), stdout);
                }
                t_emit += secs(t_e0, now());
            } else {
                completed += A.n;   // --read-only: nothing was submitted
            }
            ++n_batches;
            if (args.verbose && n_batches <= 5 && !A.skipped) {
                float tm2[5] = {0};
                ffs_stream_timings(A.s, tm2);
                std::lock_guard<std::mutex> lock(print_mutex);
                std::printf("GPU %d collector: batch %u out %.1f ms after the start (device: copy+decode %.2f, threshold %.2f, sparse %.2f, total %.2f ms)\n",
                            devices[di], n_batches - 1, secs(all_start, now()) * 1e3, tm2[0], tm2[1], tm2[2], tm2[4]);
            }
            {
                std::lock_guard<std::mutex> lock(G.mu);
                A.batch = -1;
                A.next_q += K;
                A.submitted = false;
                G.cv.notify_all();
            }
        }
        if (args.verbose) {
            std::lock_guard<std::mutex> lock(print_mutex);
            std::printf("GPU %d collector: %u batches; waiting for the GPU %.0f ms, results out %.0f ms, done %.0f ms after the start\n",
                        devices[di], n_batches, t_wait * 1e3, t_emit * 1e3, secs(all_start, now()) * 1e3);
        }
    };

    // the first n frames of an assembly's batch go to the GPU (n = all of them, or what had been read when the readers stopped)
    auto submit_batch = [&](Gpu& G, Assembly& A, uint32_t n, uint32_t first, int thread_id) -> bool {
        if (args.read_only) {
            A.skipped = true;
        } else {
            bool spilled = false;
            for (uint32_t i = 0; i < n; ++i) spilled = spilled || !A.spill[i].empty();
            if (gpu_decode && spilled)   // (all chunks of a batch lie in the staging area or none: the others go through the heap too)
                for (uint32_t i = 0; i < n; ++i)
                    if (A.spill[i].empty()) {
                        const uint8_t* p = static_cast<const uint8_t*>(A.chunk_ptr[i]);
                        A.spill[i].assign(p, p + A.chunk_len[i]);
                        A.chunk_ptr[i] = A.spill[i].data();
                    }
            const int sub = gpu_decode ? ffs_submit_compressed(A.s, A.chunk_ptr.data(), A.chunk_len.data(), n, first)
                                       : ffs_submit(A.s, A.host, n, first);
            if (sub != FFS_OK) { fail("", G.ctx); return false; }
            if (A.v) {   // the same input through the validation context
                const int vsub = gpu_decode ? ffs_submit_compressed(A.v, A.chunk_ptr.data(), A.chunk_len.data(), n, first)
                                            : ffs_submit(A.v, A.host, n, first);
                if (vsub != FFS_OK) { fail("validation pass: ", G.vctx); return false; }
            }
        }
        std::lock_guard<std::mutex> lock(G.mu);
        A.n = n;
        A.submitted_by = thread_id;
        A.submitted = true;
        G.cv.notify_all();
        return true;
    };

    // An assembly's stream(s) and pinned staging area, made by whoever asks first (the others wait).  The first K readers of a GPU
    // each make one assembly as soon as the size of a chunk is known, side by side: made one after the other, each when its first
    // batch was claimed, the first four batches of a run went through at one per 10 ms (a 150 MB staging area takes 6 ms to pin).
    auto ensure_assembly = [&](Gpu& G, Assembly& A, uint32_t index, int thread_id) -> bool {
        std::unique_lock<std::mutex> lock(G.mu);
        if (A.ready) return true;
        if (A.creating) {
            G.cv.wait(lock, [&] { return A.ready || g_stop.load() || failed.load(); });
            return A.ready;
        }
        A.creating = true;
        lock.unlock();
        const auto t_c0 = now();
        bool ok = ffs_stream_create(G.ctx, &A.s) == FFS_OK && (!G.vctx || ffs_stream_create(G.vctx, &A.v) == FFS_OK);
        const auto t_c1 = now();
        // chunks: B tight slots (the first chunk's size + 2 %: what lies in consecutive slots crosses PCIe as ONE copy --
        // a copy per chunk cost 5 % of the frame rate) and behind them an overflow area for the chunks that do not fit theirs
        A.slot_bytes = gpu_decode ? chunk_estimate.load() : frame_bytes;
        A.over_at = (size_t)batch * A.slot_bytes;
        const size_t over = gpu_decode ? std::max((size_t)batch * A.slot_bytes / 4, std::min(3 * A.slot_bytes, frame_bytes + 4096)) : 0;
        void* v = nullptr;
        ok = ok && ffs_stream_reserve_host(A.s, A.over_at + over) == FFS_OK && ffs_stream_host_buffer(A.s, &v, &A.host_bytes) == FFS_OK;
        A.host = static_cast<uint8_t*>(v);
        if (!ok) { fail("", G.ctx); return false; }
        lock.lock();
        A.ready = true;
        G.cv.notify_all();
        lock.unlock();
        if (args.verbose) {
            std::lock_guard<std::mutex> pl(print_mutex);
            std::printf("Thread %2d: assembly %u of GPU %d ready (stream %.1f ms, %.0f MB of staging %.1f ms) %.0f ms after the start\n", thread_id,
                        index, devices[G.index], secs(t_c0, t_c1) * 1e3, A.host_bytes / 1e6, secs(t_c1, now()) * 1e3, secs(all_start, now()) * 1e3);
        }
        return true;
    };

    // ---- a reader: chunks from the frame source into the slots of its GPU's assemblies ----------------------------------------
    auto reader_thread = [&](int thread_id) {
        const uint32_t di = (uint32_t)thread_id % n_dev;
        Gpu& G = *gpus[di];
        ffs_ctx* ctx = G.ctx;
        if (node_known[di]) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &node_cpus[di]);
        // scratch: a chunk whose size nobody knows yet; chunks the CPU decodes.  NOT value-initialised: a vector of this size
        // zero-fills 72 MB (10 ms of page faults on the clock, for a chunk of 7 MB)
        const size_t raw_bytes = frame_bytes * (bytes_per_pixel == 2 ? 2 : 1) + 4096;
        std::unique_ptr<uint8_t[]> raw;
        auto scratch = [&]() -> std::span<uint8_t> {
            if (!raw) raw.reset(new uint8_t[raw_bytes]);
            return {raw.get(), raw_bytes};
        };
        double t_read = 0, t_chunk = 0, t_submit = 0, t_blocked = 0;
        uint32_t n_read = 0, n_submitted = 0;
        bool made_mine = false;
        auto last_received = now();
        while (!g_stop.load() && !failed.load()) {
            const uint64_t j = G.next_slot.fetch_add(1);
            const uint64_t q = j / batch, b = q * n_dev + di;
            const uint32_t k = (uint32_t)(j % batch);
            if (b >= total_batches) break;
            const uint32_t first = (uint32_t)(b * batch);
            const uint32_t n_in_batch = std::min<uint32_t>(batch, num_images - first);
            if (k >= n_in_batch) continue;   // (the last batch is short)
            const uint32_t image_num = first + k;
            const uint32_t offset_image_num = image_num + args.start_index;  // :756
            Assembly& A = G.as[q % K];

            // the first chunk anybody reads sizes the staging areas (chunks that the GPU decodes: B slots of that size + 25 %)
            std::span<uint8_t> chunk;
            bool have_chunk = false;
            auto read_into = [&](std::span<uint8_t> dst) -> bool {   // false: stopped
                // readers are not thread-safe in general (:763-765); those that say they are skip the lock
                std::unique_lock<std::mutex> lock(reader_mutex, std::defer_lock);
                if (!reader.reentrant()) lock.lock();
                const auto w0 = now();
                while (!reader.is_image_available(offset_image_num) && !g_stop.load()) {
                    if (secs(last_received, now()) > args.timeout) {  // :776-787
                        std::printf("Timeout waiting for image %u\n", offset_image_num);
                        g_stop.store(true);
                        wake_all();
                        break;
                    }
                    std::this_thread::sleep_for(100ms);
                }
                if (g_stop.load()) return false;
                last_received = now();
                time_waiting_acc.fetch_add(secs(w0, last_received));
                for (;;) {  // zero-length reads on /dev/shm: retry (:805-821)
                    const auto c0 = now();
                    chunk = reader.get_raw_chunk(offset_image_num, dst);
                    t_chunk += secs(c0, now());
                    if (chunk.size() != 0) break;
                    std::printf("\033[1mRace Condition?!?? Got buffer size 0 for image %u. Sleeping.\033[0m\n", image_num);
                    std::this_thread::sleep_for(100ms);
                    if (g_stop.load() || failed.load()) return false;
                }
                return true;
            };
            const auto t_fill = now();
            if (gpu_decode && chunk_estimate.load() == 0) {
                if (!read_into(scratch())) break;
                have_chunk = true;
                size_t expect = 0;
                chunk_estimate.compare_exchange_strong(expect, ((chunk.size() + (size_t)(chunk.size() * (double)args.slot_margin / 100.0) + 16384) + 63) & ~(size_t)63);
            }

            if (!made_mine) {   // the GPU's first K readers make one assembly each, side by side (see ensure_assembly)
                made_mine = true;
                const uint32_t local = (uint32_t)thread_id / n_dev;
                if (local < K && !ensure_assembly(G, G.as[local], local, thread_id)) break;
            }
            // this batch's assembly: claimed by the first of its readers to get here (in order: batch q - K must have been collected)
            {
                const auto t_b0 = now();
                std::unique_lock<std::mutex> lock(G.mu);
                G.cv.wait(lock, [&] { return A.batch == (int64_t)b || (A.batch == -1 && A.next_q == q) || g_stop.load() || failed.load(); });
                if (g_stop.load() || failed.load()) break;
                if (A.batch == -1) {
                    A.batch = (int64_t)b;
                    A.n = n_in_batch;
                    A.filled = 0;
                    A.submitted = false;
                    A.skipped = false;
                    std::fill(A.slot_filled.begin(), A.slot_filled.end(), (uint8_t)0);
                    A.over_used = 0;   // (before the lock is dropped below: the batch's other readers take the overflow area as soon as they see `ready`)
                    if (!A.ready) {   // first use (usually made ahead, below): the stream(s) and the pinned staging area, outside the lock
                        lock.unlock();
                        if (!ensure_assembly(G, A, (uint32_t)(q % K), thread_id)) break;
                        lock.lock();
                    }
                    else {   // (a batch that went up through the heap may have made the library grow -- and move -- the staging area)
                        void* v = nullptr;
                        (void)ffs_stream_host_buffer(A.s, &v, &A.host_bytes);
                        A.host = static_cast<uint8_t*>(v);
                    }
                } else if (!A.ready) {
                    G.cv.wait(lock, [&] { return A.ready || g_stop.load() || failed.load(); });
                    if (!A.ready) break;
                }
                t_blocked += secs(t_b0, now());
            }

            uint8_t* slot = A.host + (size_t)k * A.slot_bytes;
            A.spill[k].clear();
            if (gpu_decode) {
                if (have_chunk && chunk.size() <= A.slot_bytes) {
                    std::memcpy(slot, chunk.data(), chunk.size());
                    chunk = {slot, chunk.size()};
                } else if (!have_chunk) {
                    if (!read_into({slot, A.slot_bytes})) break;
                    have_chunk = true;
                }
                if (chunk.data() != slot || chunk.size() >= A.slot_bytes) {
                    // larger than its slot (the read may have been cut): into the overflow area of the same staging buffer, one such
                    // chunk at a time (its size is only known once it has been read: the area's free end is its buffer) ...
                    bool placed = false;
                    {
                        std::lock_guard<std::mutex> over_lock(*A.over_mu);
                        const size_t at = A.over_at + A.over_used;
                        const size_t room = at < A.host_bytes ? A.host_bytes - at : 0;
                        if (room >= 2 * A.slot_bytes) {
                            if (!read_into({A.host + at, room})) break;
                            if (chunk.size() < room) {
                                placed = true;
                                A.over_used += (chunk.size() + 63) & ~(size_t)63;
                            }
                        }
                    }
                    if (!placed) {   // ... or, when that is full too, through the heap -- and so will its batch
                        A.spill[k].resize(raw_bytes);
                        if (!read_into(A.spill[k])) break;
                        A.spill[k].resize(chunk.size());
                        chunk = {A.spill[k].data(), chunk.size()};
                    }
                }
                A.chunk_ptr[k] = chunk.data();
                A.chunk_len[k] = chunk.size();
            } else {
                if (!read_into(scratch())) break;
                switch (reader.get_raw_chunk_compression()) {  // decode outside the lock (:823-842)
                case Reader::BITSHUFFLE_LZ4:
                    if (chunk.size() < 12 || bshuf_decompress_lz4(chunk.data() + 12, chunk.size() - 12, slot, (size_t)width * height, bytes_per_pixel) < 0) {
                        std::printf("Error: corrupt bitshuffle-LZ4 chunk for image %u\n", image_num);
                        failed = 1;
                        wake_all();
                    }
                    break;
                case Reader::BYTE_OFFSET_32:
                    if (bytes_per_pixel == 2) byte_offset_decompress(chunk.data(), chunk.size(), reinterpret_cast<uint16_t*>(slot), (size_t)width * height);
                    else byte_offset_decompress(chunk.data(), chunk.size(), reinterpret_cast<uint32_t*>(slot), (size_t)width * height);
                    break;
                case Reader::NONE:
                    std::memcpy(slot, chunk.data(), std::min(chunk.size(), frame_bytes));
                    break;
                }
            }
            if (failed.load()) break;
            ++n_read;
            t_read += secs(t_fill, now());

            bool last = false;
            {
                std::lock_guard<std::mutex> lock(G.mu);
                A.slot_filled[k] = 1;
                last = ++A.filled == A.n;
            }
            if (!last) continue;
            // the batch is complete: whoever filled its last slot sends it off
            const auto t_s0 = now();
            if (!submit_batch(G, A, A.n, first, thread_id)) break;
            ++n_submitted;
            t_submit += secs(t_s0, now());
            if (args.verbose && b < 5) {
                std::lock_guard<std::mutex> lock(print_mutex);
                std::printf("Thread %2d: batch %llu submitted %.1f ms after the start (the call took %.2f ms)\n", thread_id, (unsigned long long)b,
                            secs(all_start, now()) * 1e3, secs(t_s0, now()) * 1e3);
            }
        }
        if (args.verbose) {
            std::lock_guard<std::mutex> lock(print_mutex);
            std::printf("Thread %2d: %u chunks read in %.0f ms (%.0f ms of it in get_raw_chunk), waiting for a free assembly %.0f ms, %u batches submitted (%.0f ms), "
                        "done %.0f ms after the start\n", thread_id, n_read, t_read * 1e3, t_chunk * 1e3, t_blocked * 1e3, n_submitted, t_submit * 1e3,
                        secs(all_start, now()) * 1e3);
        }
    };
    {
        // How many of the -n threads read.  The reference needs one thread per frame in flight because its threads decompress
        // (service.py passes --threads 40); here a reader only moves chunks from the frame source into pinned memory, and eight
        // per GPU saturate PCIe (tools/cli_profile.sh) -- beyond that they contend for the page cache's locks.  Threads that
        // decompress on the host (--cpu-decode, CBF, --writeout) are all used.
        uint32_t n_workers = args.threads;
        if (gpu_decode && !args.all_threads) n_workers = std::min<uint32_t>(n_workers, 8 * n_dev);
        n_workers = std::max(n_workers, n_dev);
        if (args.verbose && n_workers != args.threads) std::printf("Workers: %u of the %u threads read for the GPU(s)\n", n_workers, args.threads);
        std::vector<std::thread> threads;
        for (uint32_t di = 0; di < n_dev; ++di) threads.emplace_back(collector, di);
        for (uint32_t t = 0; t < n_workers; ++t) threads.emplace_back(reader_thread, (int)t);
        // The signal handler only sets g_stop (nothing else is safe there), so somebody has to tell the threads parked on a GPU's
        // condition variable: readers waiting for a free assembly are woken by the collector only when a batch COMPLETES, and after an
        // interrupt none may -- the readers that hold its slots return without filling them.  (The reference's workers poll: :770-790.)
        std::atomic<bool> watch_over{false};
        std::thread stop_watcher([&] {
            bool told = false;
            while (!watch_over.load()) {
                if (!told && (g_stop.load() || failed.load())) {
                    wake_all();
                    told = true;
                }
                std::this_thread::sleep_for(20ms);
            }
        });
        for (size_t t = n_dev; t < threads.size(); ++t) threads[t].join();   // the readers
        watch_over.store(true);
        stop_watcher.join();
        // Readers that stopped early (a time-out: the data set ended before --images; an interrupt) leave batches half filled.  The
        // reference's workers finish the image they hold, so every image that WAS read still goes through: the leading filled
        // slots of such a batch are submitted as a shorter batch (images arrive in order; one behind a missing image is dropped).
        if (!failed.load())
            for (uint32_t di = 0; di < n_dev; ++di) {
                Gpu& G = *gpus[di];
                for (Assembly& A : G.as) {
                    uint32_t prefix = 0, first = 0;
                    {
                        std::lock_guard<std::mutex> lock(G.mu);
                        if (A.batch < 0 || A.submitted || !A.ready) continue;
                        while (prefix < A.n && A.slot_filled[prefix]) ++prefix;
                        first = (uint32_t)((uint64_t)A.batch * batch);
                    }
                    if (prefix > 0) (void)submit_batch(G, A, prefix, first, 0);
                }
            }
        readers_done.store(true);
        wake_all();                                                          // (collectors waiting for a batch nobody will submit)
        for (uint32_t di = 0; di < n_dev; ++di) threads[di].join();
        // every result is out: the streams' buffers are released after the totals are printed, not on the clock
        for (auto& g : gpus)
            for (Assembly& A : g->as) {
                if (A.s) retired_streams.push_back(A.s);
                if (A.v) retired_streams.push_back(A.v);
            }
    }
    if (failed.load()) return 1;

    // ---- 3D connected components (spotfinder.cc:1099-1148) ------------------------------------------
    const auto t_joined = std::chrono::steady_clock::now();
    if (args.verbose) std::printf("Workers joined %.0f ms after the start\n", std::chrono::duration<double>(t_joined - all_start).count() * 1e3);
    if (rotation) {
        std::printf("Processing 3D spots\n");
        const ffs_reflection* refl = nullptr;
        uint32_t n = 0, n_calc = 0, f_size = 0, f_sep = 0;
        FFS_CHECK(ctx, ffs_stack3d_finish(stack, &refl, &n, &n_calc, &f_size, &f_sep));
        if (args.verbose) std::printf("3D finish: %.1f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_joined).count() * 1e3);
        std::printf("Calculated %u spots\n", n_calc);  // connected_components.cc:453-454
        if (f_size > 0) std::printf("Filtered %u spots with size < %u pixels\n", f_size, prm.min_spot_size_3d);
        if (f_sep > 0) std::printf("Filtered %u spots with peak-centroid distance > %s\n", f_sep, fmt_num(prm.max_peak_centroid_separation).c_str());
        std::printf("Found %u spots\n", n);
        if (args.writeout) {
            std::ofstream out("3d_reflections.txt");
            for (uint32_t i = 0; i < n; ++i) {
                const ffs_reflection& r = refl[i];
                out << "X: [" << r.x_min << ", " << r.x_max << "] "
                    << "Y: [" << r.y_min << ", " << r.y_max << "] "
                    << "Z: [" << r.z_min << ", " << r.z_max << "] "
                    << "COM: (" << r.com_x << ", " << r.com_y << ", " << r.com_z << ")\n";
            }
        }
        {   // spot variances for integration (:1152-1215)
            const uint32_t *sx, *sy, *si;
            const int32_t *sz, *sr;
            uint64_t n_sig = 0;
            FFS_CHECK(ctx, ffs_stack3d_signals(stack, &sx, &sy, &sz, &si, &sr, &n_sig));
            const KabschGeometry geom{detector.distance * 1000.0, detector.beam_center_x, detector.beam_center_y,
                                      detector.pixel_size_x * 1000.0, detector.pixel_size_y * 1000.0, wavelength,
                                      oscillation_start, oscillation_width};
            const KabschVariances kv = kabsch_variances(geom, refl, n, sx, sy, sz, si, sr, n_sig);
            if (n) std::printf("Estimated sigma_b (degrees): %.6f\n", kv.est_sigma_b_deg);
            if (kv.n_sigma_m)
                std::printf("Estimated sigma_m (degrees): %.6f, calculated on %d spots\n", kv.est_sigma_m_deg, kv.n_sigma_m);
            if (args.save_h5) {  // :1217-1262
                try {
                    std::vector<double> flat;
                    for (uint32_t i = 0; i < n; ++i) {
                        flat.push_back(refl[i].com_x);
                        flat.push_back(refl[i].com_y);
                        flat.push_back(refl[i].com_z);
                    }
                    const std::vector<int> id(n, 0);
                    h5_write_reflection_table("results_ffs.h5", "dials/processing/group_0", flat, id, &kv.sigma_b_variance,
                                              &kv.sigma_m_variance, &kv.bbox_depth);
                    std::printf("Successfully wrote 3D reflections to HDF5 file\n");
                } catch (const std::exception& e) {
                    std::printf("Error writing data to HDF5 file: %s\n", e.what());
                }
            }
        }
        if (args.verbose) std::printf("3D analysis in all: %.1f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_joined).count() * 1e3);
        std::printf("3D spot analysis complete\n");
        ffs_stack3d_destroy(stack);
    } else if (args.save_h5) {  // :1265-1306
        std::printf("Processing 2D spots\n");
        try {
            std::vector<double> flat;
            std::vector<int> ids;
            int id = 0;
            for (const auto& kv : reflection_centers_2d) {  // std::map: ascending image number
                for (float v : kv.second) flat.push_back((double)v);
                ids.insert(ids.end(), kv.second.size() / 3, id);
                ++id;
            }
            h5_write_reflection_table("results_ffs.h5", "dials/processing/group_0", flat, ids, nullptr, nullptr, nullptr);
            std::printf("Successfully wrote %zu 2D reflections to HDF5 file\n", ids.size());
        } catch (const std::exception& e) {
            std::printf("Error writing data to HDF5 file: %s\n", e.what());
        }
        std::printf("2D spot analysis complete\n");
    }

    const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - all_start).count();
    const uint32_t done = completed.load();
    std::printf("\n%d images in %.2f s (\033[1;34m%.2f GBps\033[0m) (\033[1;34m%.1f fps\033[0m)\n", (int)done, total,
                (double)width * height * bytes_per_pixel * done / total / 1e9, done / total);
    if (args.verbose) {   // CPU seconds against wall seconds: under a cgroup CPU quota, threads beyond it stall everyone
        struct rusage ru{};
        getrusage(RUSAGE_SELF, &ru);
        const double user = ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, sys = ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
        const double since_launch = std::chrono::duration<double>(std::chrono::steady_clock::now() - process_start).count();
        std::printf("CPU time of the process: %.2f s user + %.2f s system over %.2f s since launch (%.1f cores busy on average)\n", user, sys,
                    since_launch, (user + sys) / since_launch);
    }
    if (gather_rccl)
        std::printf("Spot lists: %llu rounds of %u batches gathered over RCCL, %llu batch results read from the host arrays\n",
                    (unsigned long long)rccl_rounds.load(), n_dev, (unsigned long long)host_rounds.load());
    if (args.validate)
        std::printf("Validation: %u of %u images differ between the hot path and the gather path\n", validate_mismatches.load(), done);
    const double time_waiting = time_waiting_acc.load();
    if (time_waiting < 10) std::printf("Total time waiting for images to appear: %.0f ms\n", time_waiting * 1000);
    else std::printf("Total time waiting for images to appear: %.2f s\n", time_waiting);
    pipe.reset();
    join_multi();
    stamp("timed loop and reports done");
    const int exit_code = (args.validate && validate_mismatches.load()) ? 1 : 0;
    if (!args.clean_exit) {
        // Every result is out and nothing is in flight: the process ends here.  Destroying streams and contexts (35-90 ms: pinned
        // staging areas are unregistered, device memory freed) and the runtime's own exit handlers (~100 ms) do nothing a dying
        // process needs -- the kernel driver releases the GPU's resources -- and the service starts one process per request
        // (service.py:497): they were a fifth of a 1000-image request's wall time.  --clean-exit keeps the full teardown.
        std::fflush(stdout);
        std::fflush(stderr);
        std::_Exit(exit_code);
    }
    for (ffs_stream* st : retired_streams) ffs_stream_destroy(st);
    stamp("streams destroyed");
    for (ffs_ctx* cx : ctxs) ffs_ctx_destroy(cx);
    for (ffs_ctx* cx : vctxs) if (cx) ffs_ctx_destroy(cx);
    stamp("contexts destroyed: main returns (the runtime's own exit handlers follow)");
    std::fflush(stdout);
    return exit_code;
}
