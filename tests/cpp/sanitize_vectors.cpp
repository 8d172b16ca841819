// sanitize_vectors.cpp -- TEST INFRASTRUCTURE for tests/test_sanitizers.py: the host arithmetic of the product (csrc/host_math.cpp) and
// the oracle's C restatement (oracle/blur_oracle.c, boxblur_oracle.c) run over their vectors under AddressSanitizer + UBSan:
// sizing for every n <= 13000 against the oracle, kernels bit for bit, the Toeplitz fragments, the multiplier tables, reflect /
// de-interleave round trips, one small float32 blur and one box blur of the port.  Exit code 0 and "sanitize ok" = clean.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "host_math.hpp"

extern "C" {
int ora_gaussian_window(double sigma, int max_width);
void ora_get_gaussian(float* kernel, double sigma, int width, int fft_length);
int ora_is_valid_size(int n);
int ora_nearest_transform_size(int n);
int ora_pffft_blur_u8c3_f32(const uint8_t* src, uint8_t* dst, int rows, int cols, double sigma);
int ora_pffft_blur_u8c3_f64(const uint8_t* src, uint8_t* dst, int rows, int cols, double sigma, int quirk, float* planes);
int ora_fastboxblur_u8(uint8_t* inout, int w, int h, int channels, int ksize, int passes);
}

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main()
{
    using namespace blur_amd;
    for (int n = 0; n <= 13000; ++n) {
        REQUIRE(is_valid_size(n) == ora_is_valid_size(n));
        REQUIRE(nearest_transform_size(n) == ora_nearest_transform_size(n));
    }
    const double sigmas[] = { 0.3, 1.0, 2.5, 5.0, 20.0, 38.7298, 50.0, 106.77 };
    for (double s : sigmas) {
        for (int mw : { 0, 7, 64, 1024 }) REQUIRE(gaussian_window(s, mw) == ora_gaussian_window(s, mw));
        const int w = gaussian_window(s, 0);
        for (int n : { 0, nearest_transform_size(w + 500) }) {
            std::vector<float> a(n ? n : w), b(n ? n : w);
            get_gaussian(a.data(), s, w, n);
            ora_get_gaussian(b.data(), s, w, n);
            REQUIRE(std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
        }
        // Toeplitz fragments: every element is a tap or zero, hi + lo reproduce tap * 2^14 to 2^-22 relative
        if (w <= 337) {
            const int pad = (w - 1) / 2, nkb = mx_nkb(pad);
            std::vector<float> taps(w);
            get_gaussian(taps.data(), s, w, 0);
            std::vector<uint16_t> fr(static_cast<size_t>(2) * nkb * 512);
            mx_fragments(taps.data(), pad, nkb, fr.data());
            for (int kb = 0; kb < nkb; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int wpos = 16 * kb + 8 * (l >> 5) + j, o = l & 31, t = wpos - o - mx_pada(pad);
                        const size_t e = (static_cast<size_t>(kb) * 64 + l) * 8 + j;
                        const float v = f16_to_f32(fr[e]) + f16_to_f32(fr[static_cast<size_t>(nkb) * 512 + e]);
                        const float want = (t >= -pad && t <= pad) ? taps[t + pad] * 16384.f : 0.f;
                        REQUIRE(std::fabs(v - want) <= std::fabs(want) * 4.8e-7f + 1e-12f);
                    }
        }
    }
    // binary16 conversions round-trip on every finite half
    for (uint32_t h = 0; h < 0x10000u; ++h) {
        if (((h >> 10) & 31u) == 31u) continue;
        REQUIRE(f32_to_f16(f16_to_f32(static_cast<uint16_t>(h))) == h);
    }
    // multiplier tables of the wave-resident kernels: finite, even, DC = sum of the kernel / n
    {
        const int n = 2304, n_ref = 2304, w = gaussian_window(20.0, 0);
        std::vector<float> k(n), m(n);
        get_gaussian(k.data(), 20.0, w, n);
        wr_multipliers(k.data(), n, n_ref, true, m.data());
        for (int f = 1; f < n / 2; ++f) REQUIRE(m[f] == m[n - f] && std::isfinite(m[f]));
        REQUIRE(std::fabs(m[0] * n - 1.f) < 1e-5f);
    }
    // the oracle's float32 port and box blur on small images (all its buffers under the sanitizers)
    {
        const int rows = 61, cols = 77;
        std::vector<uint8_t> src(static_cast<size_t>(rows) * cols * 3), dst(src.size());
        for (size_t i = 0; i < src.size(); ++i) src[i] = static_cast<uint8_t>((i * 37u) ^ (i >> 3));
        REQUIRE(ora_pffft_blur_u8c3_f32(src.data(), dst.data(), rows, cols, 3.0) == 0);
        std::vector<float> planes(static_cast<size_t>(3) * rows * cols);
        REQUIRE(ora_pffft_blur_u8c3_f64(src.data(), dst.data(), rows, cols, 7.5, 1, planes.data()) == 0);
        REQUIRE(ora_pffft_blur_u8c3_f64(src.data(), dst.data(), rows, cols, 2.0, 0, nullptr) == 0);
        std::vector<uint8_t> box(src);
        REQUIRE(ora_fastboxblur_u8(box.data(), cols, rows, 3, 9, 3) == 0);
        REQUIRE(ora_fastboxblur_u8(box.data(), cols, rows, 3, 41, 2) == 0);
    }
    std::printf("sanitize ok\n");
    return 0;
}
