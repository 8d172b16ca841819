// surface_check.cpp -- a "user of the reference" written against include/blur_amd.hpp with the
// reference's global names.  Modes:
//   host                                       host-side functions only (no GPU), self-checking
//   blur in.raw rows cols sigma out.raw        pffft_(cv::Mat-like&, sigma) on the GPU
//   box  in.raw w h ch ksize passes out.raw    fastboxblur(...) on the GPU
#define BLUR_AMD_GLOBAL_NAMES
#include "blur_amd.hpp"

#include <cstdio>
#include <fstream>
#include <iterator>

struct FakeMat {               // the three members pffft_() uses of cv::Mat (Source.cpp:434,459-461)
    unsigned char* data;
    int size[2];
};

static std::vector<uint8_t> slurp(const char* p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static int host_checks()
{
    REQUIRE(gaussian_window(5) == 31 && gaussian_window(20) == 131 && gaussian_window(50) == 331);
    REQUIRE(gaussian_window(200, 1024) == 1025);
    REQUIRE(nearestTransformSize(3970) == 4000 && nearestTransformSize(2290) == 2304 && nearestTransformSize(1) == 32);
    REQUIRE(isValidSize(576) && !isValidSize(608));
    std::vector<float> k;
    getGaussian(k, 1.0, 3, 8);                         // README.md:93-101 centring
    REQUIRE(k.size() == 8 && k[2] == 0 && k[6] == 0 && k[1] == k[7] && k[0] > k[1]);
    getGaussian(k, 2.0);
    REQUIRE((int)k.size() == gaussian_window(2.0));
    AlignedVector<float> ak(16), ak2;                  // the reference's aligned container (Source.cpp:465)
    getGaussian(ak, 1.0, 3, 8);
    REQUIRE(ak.size() == 8 && ak[0] == k.size() * 0 + ak[0] && (reinterpret_cast<uintptr_t>(ak.data()) & 63) == 0);
    ak2 = ak;                                          // Source.cpp:503 needs allocator equality
    REQUIRE(ak2 == ak);
    // README.md:49-52: length 7, pad 6 -> g f e d c b | A..G | f e d c b a
    const uint8_t row[7] = { 1, 2, 3, 4, 5, 6, 7 };
    uint8_t padded[19];
    const int sz[2] = { 1, 7 };
    Reflect_101<uint8_t, 1>(row, padded, 0, 0, 6, 6, sz);
    const uint8_t want[19] = { 7, 6, 5, 4, 3, 2, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4, 3, 2, 1 };
    REQUIRE(std::memcmp(padded, want, 19) == 0);
    // de/interleave round trip with the +0.5 / truncate rule
    std::vector<uint8_t> img(3 * 10), back(3 * 10);
    for (int i = 0; i < 30; ++i) img[i] = (uint8_t)(i * 7);
    std::vector<float> p0(10), p1(10), p2(10);
    float* pl[3] = { p0.data(), p1.data(), p2.data() };
    deinterleave_BGR(img.data(), pl, 10u);
    REQUIRE(p1[3] == img[3 * 3 + 1]);
    p0[0] += 0.49f;                                    // rounds back down
    const float* cpl[3] = { p0.data(), p1.data(), p2.data() };
    interleave_BGR(cpl, back.data(), 10u);
    REQUIRE(back == img);
    // flip_block twice is the identity
    std::vector<float> a(5 * 3), b(5 * 3), c(5 * 3);
    for (int i = 0; i < 15; ++i) a[i] = (float)i;
    flip_block<float, 1>(a.data(), b.data(), 5, 3);
    REQUIRE(b[1] == a[5]);                             // out[x*h + y] = in[y*w + x]
    flip_block<float, 1>(b.data(), c.data(), 3, 5);
    REQUIRE(a == c);
    // hybrid_loop (Utils.hpp:16-55): every index exactly once in every build mode (-DMYLOOP: std::thread blocks),
    // the two-argument form receives a thread number
    std::vector<int> seen(1000, 0), tids(1000, -1);
    hybrid_loop(1000, [&](int i) { ++seen[i]; });
    hybrid_loop(1000, [&](int i, int tid) { tids[i] = tid; });
    for (int i = 0; i < 1000; ++i) REQUIRE(seen[i] == 1 && tids[i] >= 0);
    hybrid_loop(0, [&](int) { std::abort(); });        // an empty range runs nothing (the reference divides by zero here)
    hybrid_loop(3u, [&](unsigned i) { ++seen[i]; });   // fewer indices than threads
    REQUIRE(seen[0] == 2 && seen[2] == 2 && seen[3] == 1);
    // image files (the role of cv::imread / cv::imwrite, Source.cpp:623,635): binary PPM round trip, comments in the header
    {
        Image im;
        im.size[0] = 5; im.size[1] = 7;
        im.pixels.resize(5 * 7 * 3);
        for (size_t i = 0; i < im.pixels.size(); ++i) im.pixels[i] = (uint8_t)(i * 37 + 11);
        im.data = im.pixels.data();
        const std::string path = "/tmp/blur_amd_surface_check.ppm";
        REQUIRE(imwrite(path, im));
        Image back2 = imread(path);
        REQUIRE(back2.size[0] == 5 && back2.size[1] == 7 && back2.pixels == im.pixels && back2.data == back2.pixels.data());
        FILE* f = std::fopen(path.c_str(), "wb");
        std::fputs("P6\n# a comment\n7 5\n# another\n255\n", f);
        std::fwrite(im.pixels.data(), 1, im.pixels.size(), f);
        std::fclose(f);
        Image back3 = imread(path);
        REQUIRE(back3.pixels == im.pixels && back3.channels() == 3);
        REQUIRE(imread("/nonexistent/file.ppm").empty());
    }
    std::printf("host ok\n");
    return 0;
}

int main(int argc, char** argv)
{
    const std::string mode = argc > 1 ? argv[1] : "host";
    try {
        if (mode == "host") return host_checks();
        if (mode == "blur" && argc == 7) {
            std::vector<uint8_t> img = slurp(argv[2]);
            FakeMat m{ img.data(), { std::atoi(argv[3]), std::atoi(argv[4]) } };
            if (img.size() != (size_t)m.size[0] * m.size[1] * 3) { std::printf("bad input size\n"); return 2; }
            pffft_(m, std::atof(argv[5]));
            std::ofstream(argv[6], std::ios::binary).write((const char*)img.data(), img.size());
            return 0;
        }
        if ((mode == "pocket1d" || mode == "pocket2d") && argc == 7) {
            std::vector<uint8_t> img = slurp(argv[2]);
            FakeMat m{ img.data(), { std::atoi(argv[3]), std::atoi(argv[4]) } };
            if (img.size() != (size_t)m.size[0] * m.size[1] * 3) { std::printf("bad input size\n"); return 2; }
            if (mode == "pocket1d") pocketfft_1D(m, std::atof(argv[5]));
            else pocketfft_2D(m, std::atof(argv[5]));
            std::ofstream(argv[6], std::ios::binary).write((const char*)img.data(), img.size());
            return 0;
        }
        if ((mode == "whole2d" || mode == "dftimage") && argc == 7) {
            std::vector<uint8_t> img = slurp(argv[2]);
            const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
            if (img.size() != (size_t)rows * cols * 3) { std::printf("bad input size\n"); return 2; }
            pocketfft_2D_whole(img.data(), rows, cols, std::atof(argv[5]), mode == "dftimage");
            std::ofstream(argv[6], std::ios::binary).write((const char*)img.data(), img.size());
            return 0;
        }
        if (mode == "box" && argc == 9) {
            std::vector<uint8_t> img = slurp(argv[2]);
            fastboxblur(img.data(), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]));
            std::ofstream(argv[8], std::ios::binary).write((const char*)img.data(), img.size());
            return 0;
        }
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    std::printf("usage: surface_check host | blur in rows cols sigma out | box in w h ch ksize passes out\n");
    return 2;
}
