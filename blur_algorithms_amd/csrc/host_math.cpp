// host_math.cpp -- see host_math.hpp.  Written from the reference's semantics
// (Source.cpp:60-102,434-457; Utils.hpp:141-157), not from its text.
#include "host_math.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace blur_amd {

int gaussian_window(double sigma, int max_width)
{
    // the reference narrows the double expression to float, then works in float and
    // truncates (Source.cpp:64-65); an even width is bumped to the next odd one (:68)
    const float radius = static_cast<float>(sigma * std::sqrt(2 * std::log(255)) - 1);
    int width = static_cast<int>(radius * 2 + .5f);
    if (max_width) width = std::min(width, max_width);
    return width | 1;
}

void get_gaussian(float* kernel, double sigma, int width, int fft_length)
{
    if (!width) width = gaussian_window(sigma);
    const int len = fft_length ? fft_length : width;
    const double two_s2 = 2. * sigma * sigma;
    const double norm = 3.14159265358979323846 * two_s2;
    const float half = (width - 1) / 2.f;
    std::vector<float> taps(width);
    // tap t sits at offset y = t - half (a float, as the reference's loop counter is,
    // Source.cpp:88); y*y is a float product, everything after it double, and each tap
    // is rounded to float BEFORE the normalisation (Source.cpp:89-93)
    for (int t = 0; t < width; ++t) {
        const float y = -half + static_cast<float>(t);
        taps[t] = static_cast<float>(std::exp(-(y * y) / two_s2) / norm);
    }
    double total = 0.;
    for (float v : taps) total += v;
    const double inv = 1. / total;
    for (float& v : taps) v = static_cast<float>(v * inv);
    if (!fft_length) {
        std::copy(taps.begin(), taps.end(), kernel);
        return;
    }
    // zero-extend to the FFT length with the centre tap at index 0 (Source.cpp:96-100)
    std::fill(kernel, kernel + len, 0.f);
    const int c = width / 2;
    for (int t = 0; t < width; ++t) kernel[(t - c + len) % len] = taps[t];
}

int is_valid_size(int n)
{
    // n = 32 * 2^a 3^b 5^c, where a factor is only stripped while the rest stays >= 32
    for (int p : { 5, 3, 2 })
        while (n >= p * 32 && n % p == 0) n /= p;
    return n == 32;
}

int nearest_transform_size(int n)
{
    n = std::max(n, 32);
    n = (n + 31) / 32 * 32;
    while (!is_valid_size(n)) n += 32;
    return n;
}

Sizing pffft_sizing(int rows, int cols, double sigma)
{
    Sizing s{};
    s.kSize = gaussian_window(sigma, std::max(rows, cols));
    s.pad = (s.kSize - 1) / 2;
    auto fit = [](int want, int& tz) {
        if (is_valid_size(want)) { tz = 0; return want; }
        const int n = nearest_transform_size(want);
        tz = n - want;
        return n;
    };
    s.n_col = fit(rows + 2 * s.pad, s.tz_col);
    s.n_row = fit(cols + 2 * s.pad, s.tz_row);
    return s;
}

Sizing2D pocketfft2d_sizing(int rows, int cols, double sigma)
{
    Sizing2D s{};
    s.kSize = gaussian_window(sigma, std::max(rows, cols));
    s.pad = (s.kSize - 1) / 2;
    int border[4] = { s.pad, s.pad, s.pad, s.pad };
    int sizes[2] = { rows + 2 * s.pad, cols + 2 * s.pad };
    for (int i = 0; i < 2; ++i)
        if (!is_valid_size(sizes[i])) {
            const int grown = nearest_transform_size(sizes[i]);
            const int extra = grown - sizes[i];
            sizes[i] = grown;
            border[2 * i] += extra / 2;
            // the reference adds a float to the int (`border += new_pad / 2.f + 0.5f`): int -> float, add, truncate
            border[2 * i + 1] = static_cast<int>(static_cast<float>(border[2 * i + 1]) + (extra / 2.f + 0.5f));
        }
    s.s0 = sizes[0];
    s.s1 = sizes[1];
    s.top = border[0];
    s.bottom = border[1];
    s.left = border[2];
    s.right = border[3];
    return s;
}

void kernel_multipliers(double sigma, int ksize, int n, float* m)
{
    std::vector<float> k(std::max(n, ksize));
    get_gaussian(k.data(), sigma, ksize, n);
    const int pad = ksize / 2;
    const float scaler = 1.f / n;
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int b = 0; b <= n / 2; ++b) {
        // the rotated kernel is even, so its DFT is the cosine sum (README.md:129)
        long double acc = k[0];
        for (int t = 1; t <= pad; ++t) {
            const long long ph = (static_cast<long long>(b) * t) % n;
            acc += (static_cast<long double>(k[t]) + static_cast<long double>(k[n - t])) *
                   cosl(two_pi * static_cast<long double>(ph) / n);
        }
        m[b] = static_cast<float>(acc) * scaler;
    }
}

void box_kernel_1d(float* kernel, int klen, int n)
{
    // each tap t = -(klen-1) .. klen+1 receives, for every u = -(klen-1) .. klen-1, the double
    // value clamp((klen-|u|)(klen-|t|)/klen^4, 0, 1) added into a FLOAT slot, u outermost:
    // the order of the float additions is part of the result (Source.cpp:133-138)
    const double scale = 1. / std::pow(klen, 4);
    for (int u = 1 - klen; u <= klen - 1; ++u)
        for (int t = 1 - klen; t <= klen + 1; ++t) {
            const double w = std::clamp(static_cast<double>((klen - std::abs(u)) * (klen - std::abs(t))) * scale, 0., 1.);
            float& slot = kernel[(t + n) % n];
            slot = static_cast<float>(slot + w);
        }
}

void boxfft_sizing(int rows, int cols, double nsmooth, int& klen, int& pad)
{
    const int whole = static_cast<int>(nsmooth);
    const double side = std::sqrt(static_cast<double>(std::min(whole * whole, std::min(rows - 1, cols - 1))));
    klen = static_cast<int>(side * side);
    pad = (klen - 1) / 2 * 2;
}

void kernel_multipliers_from_array(const float* k, int n, float* m)
{
    const float scaler = 1.f / n;
    const long double two_pi = 6.283185307179586476925286766559L;
    std::vector<int> nz;
    for (int i = 0; i < n; ++i) if (k[i] != 0.f) nz.push_back(i);
    for (int b = 0; b <= n / 2; ++b) {
        long double acc = 0;
        for (int i : nz) acc += static_cast<long double>(k[i]) * cosl(two_pi * static_cast<long double>((static_cast<long long>(b) * i) % n) / n);
        m[b] = static_cast<float>(acc) * scaler;
    }
}

#ifndef BLUR_GENERIC_MAX_RADIX
#define BLUR_GENERIC_MAX_RADIX 16   // largest radix of the run-time plans (the generic kernels' switch must match)
#endif

static void choose_radices(int n, std::vector<int>& out)
{
#if BLUR_GENERIC_MAX_RADIX <= 8
    {   // small-radix plans: more passes, far fewer registers in the run-time-planned kernels
        int c2 = 0, c3 = 0, c5 = 0, r = n;
        while (r % 2 == 0) { r /= 2; ++c2; }
        while (r % 3 == 0) { r /= 3; ++c3; }
        while (r % 5 == 0) { r /= 5; ++c5; }
        out.clear();
        if (r != 1) return;
        while (c3 > 0 && c2 > 0 && c2 % 3 != 0) { out.push_back(6); --c3; --c2; }
        while (c3 > 0) { out.push_back(3); --c3; }
        while (c5 > 0) { out.push_back(5); --c5; }
        while (c2 >= 3) { out.push_back(8); c2 -= 3; }
        if (c2 == 2) out.push_back(4);
        if (c2 == 1) out.push_back(2);
        std::sort(out.begin(), out.end(), std::greater<int>());
        return;
    }
#endif
    // Few LDS round trips matter more than flops: take the largest radix first.
    // Supported butterflies: 16 10 9 8 6 5 4 3 2.
    int c2 = 0, c3 = 0, c5 = 0, r = n;
    while (r % 2 == 0) { r /= 2; ++c2; }
    while (r % 3 == 0) { r /= 3; ++c3; }
    while (r % 5 == 0) { r /= 5; ++c5; }
    out.clear();
    if (r != 1) return;
    // pair 5s with 2s into radix 10, 3s with 2s into radix 6, 3s together into 9
    while (c5 > 0 && c2 > 0 && c2 % 4 != 0) { out.push_back(10); --c5; --c2; }
    while (c5 > 0) { out.push_back(5); --c5; }
    while (c3 >= 2) { out.push_back(9); c3 -= 2; }
    while (c3 > 0 && c2 > 0 && c2 % 4 != 0) { out.push_back(6); --c3; --c2; }
    while (c3 > 0) { out.push_back(3); --c3; }
    while (c2 >= 4) { out.push_back(16); c2 -= 4; }
    if (c2 == 3) out.push_back(8);
    if (c2 == 2) out.push_back(4);
    if (c2 == 1) out.push_back(2);
    // large radices first (their passes carry the most twiddles per butterfly and the
    // last pass is fused with the multiply and the first inverse pass)
    std::sort(out.begin(), out.end(), std::greater<int>());
}

bool make_plan(int n, FftPlan& p)
{
    std::vector<int> rad;
    choose_radices(n, rad);
    if (n < 2 || rad.empty()) return false;
    return make_plan_radices(n, rad.data(), static_cast<int>(rad.size()), p);
}

bool make_plan_radices(int n, const int* radices, int npass, FftPlan& p)
{
    if (n < 2 || npass < 1 || npass > kMaxPasses) return false;
    long long prod = 1;
    for (int i = 0; i < npass; ++i) prod *= radices[i];
    if (prod != n) return false;
    const std::vector<int> rad(radices, radices + npass);
    p = FftPlan{};
    p.n = n;
    p.npass = static_cast<int>(rad.size());
    const long double two_pi = 6.283185307179586476925286766559L;
    int len = n, off = 0;
    for (int i = 0; i < p.npass; ++i) {
        const int R = rad[i], m = len / R;
        p.radix[i] = R;
        p.m[i] = m;
        p.tw_off[i] = off;
        if (m > 1) {
            p.tw.resize(static_cast<size_t>(off + (R - 1) * m) * 2);
            for (int q = 1; q < R; ++q)
                for (int j = 0; j < m; ++j) {
                    const long double a = -two_pi * static_cast<long double>((static_cast<long long>(j) * q) % len) / len;
                    const size_t e = static_cast<size_t>(off + (q - 1) * m + j) * 2;
                    p.tw[e] = static_cast<float>(cosl(a));
                    p.tw[e + 1] = static_cast<float>(sinl(a));
                }
            off += (R - 1) * m;
        }
        len = m;
    }
    p.freq_of_pos.resize(n);
    for (int pos = 0; pos < n; ++pos) {
        int rem = pos, f = 0, w = 1;
        for (int i = 0; i < p.npass; ++i) {
            const int q = rem / p.m[i];
            rem -= q * p.m[i];
            f += q * w;
            w *= p.radix[i];
        }
        p.freq_of_pos[pos] = f;
    }
    return true;
}

void permuted_multipliers(const FftPlan& plan, const float* m, bool quirk, float* mperm)
{
    const int n = plan.n;
    for (int pos = 0; pos < n; ++pos) {
        int f = plan.freq_of_pos[pos];
        if (f > n / 2) f = n - f;
        if (quirk && f == n / 2) f = 0;
        mperm[pos] = m[f];
    }
}

void wr_w256(float* w256)
{
    const long double two_pi = 6.283185307179586476925286766559L;
    for (int m = 0; m < 256; ++m) {
        const long double a = -two_pi * m / 256;
        w256[2 * m] = static_cast<float>(cosl(a));
        w256[2 * m + 1] = static_cast<float>(sinl(a));
    }
}

void wr_tw0(int r0, float* tw0)
{
    const long double two_pi = 6.283185307179586476925286766559L;
    const int n = 256 * r0;
    for (int q = 1; q < r0; ++q)
        for (int j = 0; j < 256; ++j) {
            const long double a = -two_pi * static_cast<long double>((j * q) % n) / n;
            const size_t e = static_cast<size_t>((q - 1) * 256 + j) * 2;
            tw0[e] = static_cast<float>(cosl(a));
            tw0[e + 1] = static_cast<float>(sinl(a));
        }
}

void wr_multipliers(const float* karr, int n, int n_ref, bool quirk, float* mult)
{
    const long double two_pi = 6.283185307179586476925286766559L;
    std::vector<int> nz;
    for (int i = 0; i < n; ++i) if (karr[i] != 0.f) nz.push_back(i);
    long double k0 = 0, kalt = 0;
    for (int i : nz) { k0 += karr[i]; kalt += (i & 1) ? -static_cast<long double>(karr[i]) : static_cast<long double>(karr[i]); }
    for (int f = 0; f <= n / 2; ++f) {
        long double acc = 0;
        for (int i : nz) acc += static_cast<long double>(karr[i]) * cosl(two_pi * static_cast<long double>((static_cast<long long>(f) * i) % n) / n);
        if (quirk && f == n / 2) acc = kalt + (k0 - kalt) * static_cast<long double>(n) / n_ref;
        mult[f] = static_cast<float>(acc / n);
        if (f > 0 && f < n / 2) mult[n - f] = mult[f];
    }
}

uint16_t f32_to_f16(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return static_cast<uint16_t>(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));   // inf / nan
    if (x >= 0x477ff000u) return static_cast<uint16_t>(sign | 0x7c00u);                                     // rounds to >= 65520: inf
    if (x < 0x33000001u) return static_cast<uint16_t>(sign);                                                // < 2^-25 (or exactly): zero
    const int e = static_cast<int>(x >> 23) - 127;
    uint32_t mant = (x & 0x7fffffu) | 0x800000u;       // 24 bits
    int shift;                                        // bits to drop from the 24-bit significand
    uint32_t base;
    if (e >= -14) { shift = 13; base = static_cast<uint32_t>(e + 15) << 10; mant &= 0x7fffffu; }          // normal
    else { shift = 13 + (-14 - e); base = 0; }                                                             // subnormal
    const uint32_t kept = mant >> shift, rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    uint32_t h = base + kept;
    if (rem > half || (rem == half && (kept & 1u))) ++h;   // carries propagate into the exponent correctly
    return static_cast<uint16_t>(sign | h);
}

float f16_to_f32(uint16_t h)
{
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            const float v = std::ldexp(static_cast<float>(m), -24);
            std::memcpy(&x, &v, 4);
            x |= sign;
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

void mx_fragments(const float* taps, int pad, int nkb, uint16_t* out)
{
    const int pada = 8 * (nkb - 2);
    const float scale = std::ldexp(1.f, kMxScaleLog2);
    for (int kb = 0; kb < nkb; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int w = 16 * kb + 8 * (l >> 5) + j, o = l & 31, t = w - o - pada;
                uint16_t hi = 0, lo = 0;
                if (t >= -pad && t <= pad) {
                    const float v = taps[t + pad] * scale;          // exact: power of two
                    hi = f32_to_f16(v);
                    lo = f32_to_f16(v - f16_to_f32(hi));            // exact difference (Sterbenz / 11-bit remainder)
                }
                const size_t e = (static_cast<size_t>(kb) * 64 + l) * 8 + j;
                out[e] = hi;
                out[static_cast<size_t>(nkb) * 512 + e] = lo;
            }
}

}  // namespace blur_amd
