// blur_amd.hpp -- the reference's C++ call surface for the FFT-blur hot path, over the C ABI.
//
// A user of michelerenzullo/Blur_algorithms calls free functions from Source.cpp / Utils.hpp.
// This header offers the same names with the same argument meaning; the heavy ones run on the
// MI355X through libblur_amd.so (include/blur_amd.h), the light ones are host code written from
// the reference's semantics (not its text) and checked against it in tests/.
//
//   gaussian_window(sigma, max_width)                       Source.cpp:60-73
//   getGaussian(kernel, sigma, width, FFT_length)           Source.cpp:75-102
//   isValidSize(N) / nearestTransformSize(N)                Utils.hpp:141-157
//   deinterleave_BGR(in, planes, total) / interleave_BGR    Utils.hpp:159-210
//   Reflect_101<T,C>(in, out, top, bottom, left, right, sz) Utils.hpp:212-243
//   hybrid_loop(end, op)                                    Utils.hpp:16-55
//   PFAlloc<T>, AlignedVector<T>                            Utils.hpp:57-138, Source.cpp:58
//   flip_block<T,C>(in, out, w, h)                          call sites Source.cpp:540,562
//   fastboxblur(in, w, h, channels, ksize, passes)          call site  Source.cpp:587
//   pffft_(image, sigma)                                    Source.cpp:429-570
//   pocketfft_1D(image, sigma) / pocketfft_2D(image, sigma) Source.cpp:280-392 / 143-277 (same engine, no Nyquist quirk;
//                                                           `#define DFT_image` -> the log-spectrum picture, :235-252)
//
// Everything lives in namespace blur_amd::compat; define BLUR_AMD_GLOBAL_NAMES before including
// to also get the names in the global namespace, as the reference has them.
// Errors: the reference returns void and checks nothing; here a failing GPU call throws
// std::runtime_error with the library's message.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <type_traits>
#if defined(_OPENMP)
#include <omp.h>
#endif
#include <thread>
#include <vector>

#include "blur_amd.h"

namespace blur_amd {
namespace compat {

inline void check(blur_ctx* ctx, int rc, const char* what)
{
    if (rc != BLUR_OK) throw std::runtime_error(std::string(what) + ": " + blur_last_error(ctx));
}

// process-wide context on device BLUR_AMD_DEVICE (default 0), created on first use
inline blur_ctx* default_ctx()
{
    struct Holder {
        blur_ctx* c = nullptr;
        Holder()
        {
            const char* d = std::getenv("BLUR_AMD_DEVICE");
            const int rc = blur_ctx_create(&c, d ? std::atoi(d) : 0);
            if (rc != BLUR_OK) throw std::runtime_error(std::string("blur_ctx_create: ") + blur_last_error(nullptr));
        }
        ~Holder() { if (c) blur_ctx_destroy(c); }
    };
    static Holder h;
    return h.c;
}

// ---- PFAlloc / AlignedVector (Utils.hpp:57-138, Source.cpp:58): 64-byte aligned float storage.
// The GPU path does not need it; it is here so that code written against the reference's types compiles.
template <class T> struct PFAlloc {
    using value_type = T;
    PFAlloc() noexcept = default;
    template <class U> PFAlloc(const PFAlloc<U>&) noexcept {}
    T* allocate(std::size_t n)
    {
        void* p = nullptr;
        if (posix_memalign(&p, 64, (n ? n : 1) * sizeof(T)) != 0) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, std::size_t) noexcept { std::free(p); }
    template <class U> bool operator==(const PFAlloc<U>&) const noexcept { return true; }
    template <class U> bool operator!=(const PFAlloc<U>&) const noexcept { return false; }
};
template <typename T> using AlignedVector = std::vector<T, PFAlloc<T>>;

// ---- sizing ---------------------------------------------------------------------------
inline int gaussian_window(const double sigma, const int max_width = 0) { return blur_gaussian_window(sigma, max_width); }
inline int isValidSize(int N) { return blur_is_valid_size(N); }
inline int nearestTransformSize(int N) { return blur_nearest_transform_size(N); }

// T: any contiguous float container with resize()/data() (std::vector<float>, AlignedVector<float>)
template <typename T> void getGaussian(T& kernel, const double sigma, int width = 0, int FFT_length = 0)
{
    if (!width) width = gaussian_window(sigma);
    kernel.resize(FFT_length ? FFT_length : width);
    std::vector<float> tmp(std::max(width, FFT_length));
    if (blur_get_gaussian(tmp.data(), sigma, width, FFT_length) != BLUR_OK) throw std::invalid_argument("getGaussian: bad arguments");
    std::copy(tmp.begin(), tmp.begin() + kernel.size(), kernel.data());
}

// ---- the parallel-for of the reference (Utils.hpp:16-55) ------------------------------------
// Same three modes, chosen the same way: -DMYLOOP (the published build, .vscode/tasks.json:21) splits [0, end) into
// one contiguous block per hardware thread and runs them on fresh std::threads; OpenMP when compiled with it; else
// serial.  A two-argument operation receives the number of the thread that runs the index.  (end == 0 runs nothing:
// the reference's MYLOOP branch divides by zero there, Utils.hpp:42-44.)
template <typename T, typename op> void hybrid_loop(T end, op operation)
{
    auto call = [&](T i, int tid) {
        if constexpr (std::is_invocable_v<op, T>) operation(i);
        else operation(i, tid);
    };
#if defined(MYLOOP) || defined(__EMSCRIPTEN_THREADS__)
    if (!(end > 0)) return;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned long long n = static_cast<unsigned long long>(end);
    const unsigned long long block = (n + hw - 1) / hw;
    const unsigned nthreads = static_cast<unsigned>((n + block - 1) / block);          // never more threads than blocks
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t)
        pool.emplace_back([&call, t, block, n] {
            const unsigned long long b = t * block, e = std::min(n, b + block);
            for (unsigned long long i = b; i < e; ++i) call(static_cast<T>(i), static_cast<int>(t));
        });
    for (auto& th : pool) th.join();
#elif defined(_OPENMP)
#pragma omp parallel for
    for (T i = 0; i < end; ++i) call(i, omp_get_thread_num());
#else
    for (T i = 0; i < end; ++i) call(i, 0);
#endif
}

// ---- layout utilities (host) --------------------------------------------------------------
// float -> integer conversions add 0.5 and truncate, as the reference does (Utils.hpp:163,189)
template <typename T, typename U>
void deinterleave_BGR(const T* const interleaved_BGR, U** const deinterleaved_BGR, const uint32_t total_size)
{
    constexpr float round = (std::is_integral_v<U> && !std::is_integral_v<T>) ? 0.5f : 0.f;
    hybrid_loop(total_size, [&](uint32_t x) {
        for (int c = 0; c < 3; ++c) deinterleaved_BGR[c][x] = static_cast<U>(interleaved_BGR[3 * static_cast<size_t>(x) + c] + round);
    });
}

template <typename T, typename U>
void interleave_BGR(const U** const deinterleaved_BGR, T* const interleaved_BGR, const uint32_t total_size)
{
    constexpr float round = (std::is_integral_v<T> && !std::is_integral_v<U>) ? 0.5f : 0.f;
    hybrid_loop(total_size, [&](uint32_t x) {
        for (int c = 0; c < 3; ++c) interleaved_BGR[3 * static_cast<size_t>(x) + c] = static_cast<T>(deinterleaved_BGR[c][x] + round);
    });
}

// reflect-101 border of an interleaved C-channel image; pads are clamped to dim-1 like the
// reference does (Utils.hpp:217-220).  original_size = {rows, cols}.
template <typename T, int C>
void Reflect_101(const T* const input, T* output, int pad_top, int pad_bottom, int pad_left, int pad_right, const int* original_size)
{
    const int rows = original_size[0], cols = original_size[1];
    pad_top = std::min(pad_top, rows - 1);
    pad_bottom = std::min(pad_bottom, rows - 1);
    pad_left = std::min(pad_left, cols - 1);
    pad_right = std::min(pad_right, cols - 1);
    const int out_rows = rows + pad_top + pad_bottom, out_cols = cols + pad_left + pad_right;
    auto mirror = [](int i, int n) { i = i < 0 ? -i : i; return i >= n ? 2 * (n - 1) - i : i; };
    hybrid_loop(out_rows, [&](int i) {
        const T* src_row = input + static_cast<size_t>(mirror(i - pad_top, rows)) * cols * C;
        T* dst_row = output + static_cast<size_t>(i) * out_cols * C;
        for (int j = 0; j < out_cols; ++j) {
            const T* s = src_row + static_cast<size_t>(mirror(j - pad_left, cols)) * C;
            for (int c = 0; c < C; ++c) dst_row[static_cast<size_t>(j) * C + c] = s[c];
        }
    });
}

// out-of-place transpose of an h x w image with C interleaved channels: out[x*h + y] = in[y*w + x]
template <typename T, int C> void flip_block(const T* in, T* out, const int w, const int h)
{
    constexpr int B = 64;
    hybrid_loop((h + B - 1) / B, [&](int by) {
        for (int x0 = 0; x0 < w; x0 += B)
            for (int y = by * B; y < std::min(h, by * B + B); ++y)
                for (int x = x0; x < std::min(w, x0 + B); ++x)
                    for (int c = 0; c < C; ++c)
                        out[(static_cast<size_t>(x) * h + y) * C + c] = in[(static_cast<size_t>(y) * w + x) * C + c];
    });
}

// ---- GPU entry points -------------------------------------------------------------------------
// fastboxblur(in, w, h, channels, ksize, passes): in place on host memory
inline void fastboxblur(uint8_t* in, int w, int h, int channels, int ksize, int passes, blur_ctx* ctx = nullptr)
{
    if (!ctx) ctx = default_ctx();
    check(ctx, blur_fastboxblur_u8_host(ctx, in, w, h, channels, ksize, passes), "fastboxblur");
}

// pffft_ on raw host memory: rows*cols*3 interleaved u8, in place
inline void pffft_(uint8_t* data, int rows, int cols, double sigma, blur_ctx* ctx = nullptr, const blur_opts* opts = nullptr)
{
    if (!ctx) ctx = default_ctx();
    check(ctx, blur_gaussian_u8c3_host(ctx, data, data, rows, cols, sigma, opts), "pffft_");
}

// pffft_(cv::Mat& image, double nsmooth): anything with .data, .size[0] (rows), .size[1] (cols)
template <class Mat, class = decltype(std::declval<Mat&>().size[0])> void pffft_(Mat& image, double nsmooth)
{
    pffft_(reinterpret_cast<uint8_t*>(image.data), static_cast<int>(image.size[0]), static_cast<int>(image.size[1]), nsmooth);
}

// pocketfft_1D(image, sigma) (Source.cpp:280-392) and pocketfft_2D(image, sigma) (Source.cpp:143-277): the two
// pocketfft paths multiply all N/2+1 bins with the kernel's own spectrum (no Nyquist-slot quirk) and, inside the
// cropped image, both equal the linear convolution of the reflect-101 extended image -- the engine's
// nyquist_quirk = 0 mode (tests/ check it against a scipy.fft = pocketfft restatement of both).
template <class Mat, class = decltype(std::declval<Mat&>().size[0])> void pocketfft_1D(Mat& image, double nsmooth)
{
    blur_opts o;
    blur_opts_default(&o);
    o.nyquist_quirk = 0;
    pffft_(reinterpret_cast<uint8_t*>(image.data), static_cast<int>(image.size[0]), static_cast<int>(image.size[1]), nsmooth, nullptr, &o);
}
// The whole padded image as ONE 2D transform, with pocketfft_2D's own sizes and borders (blur_pocketfft2d_u8c3_host);
// dft_image = the reference's `#define DFT_image` build (Source.cpp:235-252): the picture becomes the fft-shifted log
// spectrum 20 log10(|Re F| + 1e-5) of every padded plane.
inline void pocketfft_2D_whole(uint8_t* data, int rows, int cols, double sigma, bool dft_image = false, blur_ctx* ctx = nullptr)
{
    if (!ctx) ctx = default_ctx();
    check(ctx, blur_pocketfft2d_u8c3_host(ctx, data, data, rows, cols, sigma, dft_image ? 1 : 0), "pocketfft_2D");
}
// pocketfft_2D(image, sigma).  Like the reference, the function follows the DFT_image macro of the translation unit
// that includes this header; without it the blur runs on the (faster, equivalent) 1D-tiled engine, or on the
// whole-image kernels when BLUR_AMD_POCKETFFT_2D_WHOLE_IMAGE is defined.
template <class Mat, class = decltype(std::declval<Mat&>().size[0])> void pocketfft_2D(Mat& image, double nsmooth)
{
#if defined(DFT_image)
    pocketfft_2D_whole(reinterpret_cast<uint8_t*>(image.data), static_cast<int>(image.size[0]), static_cast<int>(image.size[1]), nsmooth, true);
#elif defined(BLUR_AMD_POCKETFFT_2D_WHOLE_IMAGE)
    pocketfft_2D_whole(reinterpret_cast<uint8_t*>(image.data), static_cast<int>(image.size[0]), static_cast<int>(image.size[1]), nsmooth, false);
#else
    pocketfft_1D(image, nsmooth);
#endif
}

// ---- image in / out (the role cv::imread / cv::imwrite play in main(), Source.cpp:623,635) ----------------------
// Binary PPM (P6, maxval 255): the one format that needs no codec library.  An Image has what pffft_(Mat&, double)
// asks of a cv::Mat: .data, .size[0] = rows, .size[1] = cols, channels().
struct Image {
    std::vector<uint8_t> pixels;
    int size[2] = { 0, 0 };
    uint8_t* data = nullptr;                            // = pixels.data(), like cv::Mat::data; kept right by the copy / move members
    Image() = default;
    Image(const Image& o) : pixels(o.pixels), data(pixels.empty() ? nullptr : pixels.data()) { size[0] = o.size[0]; size[1] = o.size[1]; }
    Image(Image&& o) noexcept : pixels(std::move(o.pixels)), data(pixels.empty() ? nullptr : pixels.data())
    {
        size[0] = o.size[0]; size[1] = o.size[1];
        o.data = nullptr; o.size[0] = o.size[1] = 0;
    }
    Image& operator=(Image o)
    {
        pixels.swap(o.pixels);
        size[0] = o.size[0]; size[1] = o.size[1];
        data = pixels.empty() ? nullptr : pixels.data();
        return *this;
    }
    int channels() const { return 3; }
    bool empty() const { return pixels.empty(); }
};

inline Image imread(const std::string& path)
{
    Image im;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return im;
    auto token = [&](int& v) {                          // decimal number, comments (# ... end of line) skipped
        int c = std::fgetc(f);
        while (c == '#' || c == ' ' || c == '\t' || c == '\r' || c == '\n') {
            if (c == '#') while (c != '\n' && c != EOF) c = std::fgetc(f);
            else c = std::fgetc(f);
        }
        if (c < '0' || c > '9') return false;
        v = 0;
        while (c >= '0' && c <= '9') {
            if (v > 100000000) return false;            // a hostile header: no image side has nine digits
            v = v * 10 + (c - '0');
            c = std::fgetc(f);
        }
        return true;                                    // the one whitespace byte after the number is consumed
    };
    int w = 0, h = 0, maxv = 0;
    const bool ok = std::fgetc(f) == 'P' && std::fgetc(f) == '6' && token(w) && token(h) && token(maxv) && w > 0 && h > 0 && maxv == 255 &&
                    static_cast<long long>(w) * h <= (1ll << 30);
    if (ok) {
        im.pixels.resize(static_cast<size_t>(w) * h * 3);
        if (std::fread(im.pixels.data(), 1, im.pixels.size(), f) == im.pixels.size()) {
            im.size[0] = h;
            im.size[1] = w;
            im.data = im.pixels.data();
        } else im.pixels.clear();
    }
    std::fclose(f);
    return im;
}

inline bool imwrite(const std::string& path, const Image& im)
{
    if (im.empty()) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", im.size[1], im.size[0]);
    const bool ok = std::fwrite(im.data, 1, im.pixels.size(), f) == im.pixels.size();
    return std::fclose(f) == 0 && ok;
}

}  // namespace compat
}  // namespace blur_amd

#ifdef BLUR_AMD_GLOBAL_NAMES
using blur_amd::compat::AlignedVector;
using blur_amd::compat::Image;
using blur_amd::compat::imread;
using blur_amd::compat::imwrite;
using blur_amd::compat::deinterleave_BGR;
using blur_amd::compat::fastboxblur;
using blur_amd::compat::flip_block;
using blur_amd::compat::gaussian_window;
using blur_amd::compat::getGaussian;
using blur_amd::compat::hybrid_loop;
using blur_amd::compat::interleave_BGR;
using blur_amd::compat::isValidSize;
using blur_amd::compat::nearestTransformSize;
using blur_amd::compat::pffft_;
using blur_amd::compat::pocketfft_1D;
using blur_amd::compat::pocketfft_2D;
using blur_amd::compat::pocketfft_2D_whole;
using blur_amd::compat::Reflect_101;
#endif
