// pffft_pommier/pffft.h -- include-path shim for the un-vendored pffft submodule (.gitmodules:7-9): the
// symbols Source.cpp takes from it,
//   PFFFT_Setup, pffft_new_setup(N, PFFFT_REAL), pffft_destroy_setup          Source.cpp:477-478,565-566
//   pffft_transform_ordered(setup, in, out, work, PFFFT_FORWARD | PFFFT_BACKWARD)    Source.cpp:485,499,531,533,553,555
// as a HOST implementation written here (mixed-radix 2 / 3 / 4 / 5 float transform, header only, no SIMD), so that an unmodified
// Source.cpp compiles and runs with  -I<repo>/include/compat.  It reproduces pffft's documented interface, not its arithmetic:
//   * a real transform of N points (N a multiple of 32 with no prime factor above 5, as pffft requires) in the ORDERED layout:
//     out[0] = Re X(0), out[1] = Re X(N/2), out[2k], out[2k+1] = Re X(k), Im X(k) for k = 1 .. N/2-1 -- the layout the
//     pointwise rule of Source.cpp:414-427 is written for (slot 1 is the Nyquist bin: the quirk of Source.cpp:420-425);
//   * X(k) = sum x(n) exp(-2 pi i k n / N) forward; the backward transform is unscaled (backward(forward(x)) = N x);
//   * `work` may be null; a setup is read-only during transforms, so threads may share one (hybrid_loop, Source.cpp:520,546).
// float rounding differs from pffft's in the last bits (as pffft's SIMD and scalar builds differ from each other).
// This is the reference's own per-line CPU path for code that wants to keep it; the MI355X path replaces the whole body of
// pffft_() instead (blur_amd.hpp: pffft_(Mat&, double); batched lines: blur_convolve_lines_c32_dev), see INTEGRATION.md.
#pragma once
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdlib>
#include <vector>

typedef enum { PFFFT_FORWARD, PFFFT_BACKWARD } pffft_direction_t;
typedef enum { PFFFT_REAL, PFFFT_COMPLEX } pffft_transform_t;

struct PFFFT_Setup {
    int N;                                    // points of the transform (real or complex)
    int M;                                    // points of the complex transform underneath: N / 2 for PFFFT_REAL
    pffft_transform_t transform;
    std::vector<int> radix;                   // factors of M, outermost first
    std::vector<std::complex<float>> root;    // exp(-2 pi i j / M), j = 0 .. M-1
    std::vector<std::complex<float>> half;    // exp(-2 pi i k / N), k = 0 .. N/2 (real transforms: the split of the packed spectrum)
};

namespace pffft_shim {

typedef std::complex<float> cpx;

// out[0 .. n) = DFT of in[0], in[stride], ...; `sign` -1 forward, +1 backward; root step rs = M / n
inline void transform(const PFFFT_Setup* s, const cpx* in, size_t stride, int n, cpx* out, size_t level, int sign)
{
    if (n == 1) { out[0] = in[0]; return; }
    const int p = s->radix[level], m = n / p, rs = s->M / n;
    for (int q = 0; q < p; ++q) transform(s, in + q * stride, stride * p, m, out + static_cast<size_t>(q) * m, level + 1, sign);
    cpx t[5];
    for (int k = 0; k < m; ++k) {
        for (int q = 0; q < p; ++q) {
            cpx w = s->root[static_cast<size_t>(q) * k * rs % s->M];
            if (sign > 0) w = std::conj(w);
            const cpx v = out[static_cast<size_t>(q) * m + k];
            t[q] = cpx(v.real() * w.real() - v.imag() * w.imag(), v.real() * w.imag() + v.imag() * w.real());
        }
        if (p == 2) {
            out[k] = t[0] + t[1];
            out[k + m] = t[0] - t[1];
        } else if (p == 4) {
            const cpx a = t[0] + t[2], b = t[0] - t[2], c = t[1] + t[3], d = t[1] - t[3];
            const cpx jd = sign < 0 ? cpx(d.imag(), -d.real()) : cpx(-d.imag(), d.real());       // -i d forward, +i d backward
            out[k] = a + c;
            out[k + m] = b + jd;
            out[k + 2 * m] = a - c;
            out[k + 3 * m] = b - jd;
        } else {
            const int step = s->M / p;
            for (int r = 0; r < p; ++r) {
                cpx acc = t[0];
                for (int q = 1; q < p; ++q) {
                    cpx w = s->root[static_cast<size_t>(q) * r % p * step];
                    if (sign > 0) w = std::conj(w);
                    acc += cpx(t[q].real() * w.real() - t[q].imag() * w.imag(), t[q].real() * w.imag() + t[q].imag() * w.real());
                }
                out[k + static_cast<size_t>(r) * m] = acc;
            }
        }
    }
}

}  // namespace pffft_shim

// nullptr where pffft itself refuses: N <= 0, a prime factor above 5, or N not a multiple of 32 (real) / 16 (complex)
inline PFFFT_Setup* pffft_new_setup(int N, pffft_transform_t transform)
{
    if (N <= 0 || N % (transform == PFFFT_REAL ? 32 : 16) != 0) return nullptr;
    PFFFT_Setup* s = new PFFFT_Setup();
    s->N = N;
    s->M = transform == PFFFT_REAL ? N / 2 : N;
    s->transform = transform;
    int m = s->M;
    while (m % 4 == 0) { s->radix.push_back(4); m /= 4; }
    while (m % 2 == 0) { s->radix.push_back(2); m /= 2; }
    while (m % 3 == 0) { s->radix.push_back(3); m /= 3; }
    while (m % 5 == 0) { s->radix.push_back(5); m /= 5; }
    if (m != 1) { delete s; return nullptr; }
    const double tau = -6.283185307179586476925286766559;
    s->root.resize(s->M);
    for (int j = 0; j < s->M; ++j) s->root[j] = std::complex<float>(static_cast<float>(std::cos(tau * j / s->M)), static_cast<float>(std::sin(tau * j / s->M)));
    if (transform == PFFFT_REAL) {
        s->half.resize(N / 2 + 1);
        for (int k = 0; k <= N / 2; ++k) s->half[k] = std::complex<float>(static_cast<float>(std::cos(tau * k / N)), static_cast<float>(std::sin(tau * k / N)));
    }
    return s;
}

inline void pffft_destroy_setup(PFFFT_Setup* s) { delete s; }

inline void pffft_transform_ordered(PFFFT_Setup* setup, const float* input, float* output, float* work, pffft_direction_t direction)
{
    typedef std::complex<float> cpx;
    const int M = setup->M;
    std::vector<cpx> own;
    cpx* tmp = reinterpret_cast<cpx*>(work);
    if (!tmp) { own.resize(M); tmp = own.data(); }
    if (setup->transform == PFFFT_COMPLEX) {
        std::vector<cpx> in(reinterpret_cast<const cpx*>(input), reinterpret_cast<const cpx*>(input) + M);     // input may alias output
        pffft_shim::transform(setup, in.data(), 1, M, reinterpret_cast<cpx*>(output), 0, direction == PFFFT_FORWARD ? -1 : 1);
        return;
    }
    if (direction == PFFFT_FORWARD) {
        // two real points per complex point, then the split: X(k) = E(k) + exp(-2 pi i k / N) O(k)
        pffft_shim::transform(setup, reinterpret_cast<const cpx*>(input), 1, M, tmp, 0, -1);
        const cpx z0 = tmp[0];
        output[0] = z0.real() + z0.imag();
        output[1] = z0.real() - z0.imag();
        for (int k = 1; k < M; ++k) {
            const cpx a = tmp[k], b = std::conj(tmp[M - k]);
            const cpx e = 0.5f * (a + b), d = 0.5f * (a - b);                  // d = i O(k)
            const cpx o(d.imag(), -d.real());
            const cpx w = setup->half[k];
            output[2 * k] = e.real() + (o.real() * w.real() - o.imag() * w.imag());
            output[2 * k + 1] = e.imag() + (o.real() * w.imag() + o.imag() * w.real());
        }
    } else {
        // Z(k) = (X(k) + conj X(M - k)) + i exp(+2 pi i k / N) (X(k) - conj X(M - k)), then the unscaled inverse: N x
        std::vector<cpx> z(M);
        auto X = [&](int k) { return k == 0 ? cpx(input[0], 0.f) : k == M ? cpx(input[1], 0.f) : cpx(input[2 * k], input[2 * k + 1]); };
        for (int k = 0; k < M; ++k) {
            const cpx a = X(k), b = std::conj(X(M - k));
            const cpx e = a + b, d = a - b;
            const cpx w = std::conj(setup->half[k]);
            const cpx dw(d.real() * w.real() - d.imag() * w.imag(), d.real() * w.imag() + d.imag() * w.real());
            z[k] = cpx(e.real() - dw.imag(), e.imag() + dw.real());             // e + i dw
        }
        pffft_shim::transform(setup, z.data(), 1, M, tmp, 0, 1);
        for (int n = 0; n < M; ++n) { output[2 * n] = tmp[n].real(); output[2 * n + 1] = tmp[n].imag(); }
    }
}

// the unordered transform exists in pffft for speed; here it is the ordered one (a consistent "internal" order)
inline void pffft_transform(PFFFT_Setup* setup, const float* input, float* output, float* work, pffft_direction_t direction)
{
    pffft_transform_ordered(setup, input, output, work, direction);
}

inline void* pffft_aligned_malloc(size_t nb_bytes) { return std::aligned_alloc(64, (nb_bytes + 63) / 64 * 64); }
inline void pffft_aligned_free(void* p) { std::free(p); }
inline int pffft_simd_size() { return 1; }
