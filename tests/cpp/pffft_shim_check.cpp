// The include-path shim include/compat/pffft_pommier/pffft.h used the way Source.cpp uses pffft (lines 477-566): real ordered
// transforms against a float64 DFT, the unscaled round trip, and one padded tile through
//   pffft_transform_ordered(FORWARD) -> the pointwise rule of Source.cpp:414-427 -> pffft_transform_ordered(BACKWARD)
// against the direct convolution in float64.
#include "pffft_pommier/pffft.h"
#include <cstdio>
#include <random>
#include <vector>

static int fail(const char* what, int n, double err)
{
    std::printf("FAILED %s N=%d err=%g\n", what, n, err);
    return 1;
}

int main()
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> uni(0.f, 255.f);
    const double pi = 3.14159265358979323846;
    if (pffft_new_setup(100, PFFFT_REAL) || pffft_new_setup(32 * 7, PFFFT_REAL) || pffft_new_setup(0, PFFFT_REAL)) return fail("setup accepted a bad size", 0, 0);
    for (int N : { 32, 64, 96, 160, 288, 480, 576, 800, 1280, 2304, 4000, 4320 }) {
        PFFFT_Setup* s = pffft_new_setup(N, PFFFT_REAL);
        if (!s) return fail("setup", N, 0);
        std::vector<float> x(N), X(N), back(N), work(N);
        for (float& v : x) v = uni(rng);
        pffft_transform_ordered(s, x.data(), X.data(), work.data(), PFFFT_FORWARD);
        double worst = 0, scale = 0;
        for (int k = 0; k <= N / 2; k += (N > 1300 ? 37 : 1)) {          // every bin of the short ones, a sample of the long ones
            double re = 0, im = 0;
            for (int n = 0; n < N; ++n) {
                const double a = -2 * pi * static_cast<double>(static_cast<long long>(k) * n % N) / N;
                re += x[n] * std::cos(a);
                im += x[n] * std::sin(a);
            }
            const double gr = k == 0 ? X[0] : k == N / 2 ? X[1] : X[2 * k], gi = (k == 0 || k == N / 2) ? 0.0 : X[2 * k + 1];
            worst = std::max(worst, std::hypot(gr - re, gi - im));
            scale = std::max(scale, std::hypot(re, im));
        }
        if (worst > 2e-6 * 255.0 * N) return fail("forward", N, worst);
        pffft_transform_ordered(s, X.data(), back.data(), nullptr, PFFFT_BACKWARD);       // work may be null
        double rt = 0;
        for (int n = 0; n < N; ++n) rt = std::max(rt, static_cast<double>(std::fabs(back[n] / N - x[n])));
        if (rt > 2e-4) return fail("round trip", N, rt);
        pffft_destroy_setup(s);
    }
    // one row tile as Source.cpp:520-537 does it: length 200, kernel of 41 taps (pad 20), N = 256
    {
        const int len = 200, pad = 20, N = 256;
        std::vector<double> taps(2 * pad + 1);
        double sum = 0;
        for (int i = -pad; i <= pad; ++i) sum += taps[i + pad] = std::exp(-0.5 * i * i / 36.0);
        for (double& t : taps) t /= sum;
        std::vector<float> line(len), tile(N, 0.f), ker(N, 0.f), tf(N), kf(N), work(N);
        for (float& v : line) v = uni(rng);
        for (int p = 0; p < len + 2 * pad; ++p) { int i = p - pad; i = i < 0 ? -i : i >= len ? 2 * (len - 1) - i : i; tile[p] = line[i]; }
        for (int i = -pad; i <= pad; ++i) ker[(i + N) % N] = static_cast<float>(taps[i + pad]);
        PFFFT_Setup* s = pffft_new_setup(N, PFFFT_REAL);
        pffft_transform_ordered(s, ker.data(), kf.data(), work.data(), PFFFT_FORWARD);
        pffft_transform_ordered(s, tile.data(), tf.data(), work.data(), PFFFT_FORWARD);
        const float scaler = 1.f / N;
        tf[0] *= kf[0] * scaler;                                            // Source.cpp:420-425: slots 0 and 1 both by the DC gain
        tf[1] *= kf[0] * scaler;
        for (int k = 2; k < N; k += 2) { tf[k] *= kf[k] * scaler; tf[k + 1] *= kf[k] * scaler; }     // the kernel's spectrum is real
        pffft_transform_ordered(s, tf.data(), tile.data(), work.data(), PFFFT_BACKWARD);
        // the quirk scales the Nyquist bin with the DC gain instead of its own: remove that rank-one term before comparing
        double nyq = 0, gn = 0;
        {
            std::vector<float> t2(N, 0.f);
            for (int p = 0; p < len + 2 * pad; ++p) { int i = p - pad; i = i < 0 ? -i : i >= len ? 2 * (len - 1) - i : i; t2[p] = line[i]; nyq += (p & 1) ? -t2[p] : t2[p]; }
            for (int i = -pad; i <= pad; ++i) gn += ((i & 1) ? -1.0 : 1.0) * taps[i + pad];
        }
        double worst = 0;
        for (int x = 0; x < len; ++x) {
            double want = 0;
            for (int i = -pad; i <= pad; ++i) { int j = x + i; j = j < 0 ? -j : j >= len ? 2 * (len - 1) - j : j; want += taps[i + pad] * line[j]; }
            want += (1.0 - gn) * nyq / N * (((x + pad) & 1) ? -1.0 : 1.0);
            worst = std::max(worst, static_cast<double>(std::fabs(tile[x + pad] - want)));
        }
        pffft_destroy_setup(s);
        if (worst > 1.5e-4) return fail("tile convolution", N, worst);
    }
    std::printf("pffft shim ok\n");
    return 0;
}
