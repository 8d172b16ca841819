// engine_host_check.hip -- drives the DEVICE arithmetic of fft_engine.hpp on the CPU.
// The pass functions are __host__ __device__ and take (tid, nthreads) explicitly, so the
// exact butterfly / twiddle / permuted-multiplier code the kernels run can be checked
// without a GPU: for every pass, loop tid over a pretend workgroup (the loop end plays the
// role of __syncthreads()).  Compared against a float64 O(N^2) circular convolution.
// Usage: engine_host_check N [N...] [-r R0,R1,..]   exit code 0 = all within tolerance.
// (-r: explicit radix sequence; build with -DBLUR_ENGINE_ALL_RADICES for the large composites.)
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../blur_algorithms_amd/csrc/fft_engine.hpp"
#include "../../blur_algorithms_amd/csrc/host_math.hpp"

using namespace blur_amd;

template <int C> static double check(int n, double sigma, bool quirk, const std::vector<int>& radices = {})
{
    FftPlan plan;
    const bool ok = radices.empty() ? make_plan(n, plan) : make_plan_radices(n, radices.data(), (int)radices.size(), plan);
    if (!ok) { std::printf("N=%d: no plan\n", n); return 1e9; }
    DevPlan dp{};
    dp.n = n; dp.npass = plan.npass;
    for (int i = 0; i < plan.npass; ++i) { dp.radix[i] = plan.radix[i]; dp.m[i] = plan.m[i]; dp.tw_off[i] = plan.tw_off[i]; }
    int ksize = gaussian_window(sigma, n / 2);
    std::vector<float> m(n / 2 + 1), mperm(n);
    kernel_multipliers(sigma, ksize, n, m.data());
    permuted_multipliers(plan, m.data(), quirk, mperm.data());

    const int zs = line_stride(n);
    std::vector<float2> z(static_cast<size_t>(C) * zs, make_float2(0.f, 0.f));
    std::vector<std::complex<double>> x(static_cast<size_t>(C) * n);
    std::mt19937 rng(1234 + n);
    std::uniform_real_distribution<float> u(0.f, 255.f);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < n; ++i) {
            const float a = u(rng), b = u(rng);
            z[c * zs + phys(i)] = make_float2(a, b);
            x[c * n + i] = { a, b };
        }
    const int T = 256;
    const float2* tw = reinterpret_cast<const float2*>(plan.tw.data());
    for (int s = 0; s < 2 * dp.npass - 1; ++s) {
        int kind, i;
        schedule_step(dp, s, kind, i);
        for (int tid = 0; tid < T; ++tid)
            run_pass<C>(kind, dp.radix[i], z.data(), zs, n, dp.m[i], tw + dp.tw_off[i], mperm.data(), tid, T);
    }
    // float64 reference: y = IDFT( M .* DFT(x) ), M in natural order, unnormalised inverse
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<std::complex<double>> w(n);
    for (int k = 0; k < n; ++k) w[k] = { std::cos(two_pi * k / n), -std::sin(two_pi * k / n) };
    double worst = 0;
    for (int c = 0; c < C; ++c) {
        std::vector<std::complex<double>> X(n), y(n);
        for (int k = 0; k < n; ++k) {
            std::complex<double> acc = 0;
            for (int j = 0; j < n; ++j) acc += x[c * n + j] * w[(static_cast<long long>(k) * j) % n];
            int f = k <= n / 2 ? k : n - k;
            if (quirk && f == n / 2) f = 0;
            X[k] = acc * static_cast<double>(m[f]);
        }
        double scale = 0;
        for (int j = 0; j < n; ++j) {
            std::complex<double> acc = 0;
            for (int k = 0; k < n; ++k) acc += X[k] * std::conj(w[(static_cast<long long>(k) * j) % n]);
            y[j] = acc;
            scale = std::max(scale, std::abs(acc));
        }
        for (int j = 0; j < n; ++j) {
            const float2 g = z[c * zs + phys(j)];
            worst = std::max(worst, std::abs(std::complex<double>(g.x, g.y) - y[j]) / scale);
        }
    }
    std::printf("N=%5d C=%d passes=%d [", n, C, dp.npass);
    for (int i = 0; i < dp.npass; ++i) std::printf("%d%s", dp.radix[i], i + 1 < dp.npass ? "," : "");
    std::printf("] max rel err %.3e\n", worst);
    return worst;
}

int main(int argc, char** argv)
{
    int bad = 0;
    for (int a = 1; a < argc; ++a) {
        if (std::string(argv[a]) == "-r" && a + 1 < argc) {      // explicit radix list: -r 16,10,25
            std::vector<int> r;
            int n = 1;
            for (char* t = std::strtok(argv[++a], ","); t; t = std::strtok(nullptr, ",")) { r.push_back(std::atoi(t)); n *= r.back(); }
            if (check<1>(n, 3.0, true, r) > 2e-6) ++bad;
            if (check<3>(n, 5.0, false, r) > 2e-6) ++bad;
            continue;
        }
        const int n = std::atoi(argv[a]);
        if (check<1>(n, 3.0, true) > 2e-6) ++bad;
        if (check<3>(n, 5.0, false) > 2e-6) ++bad;
    }
    return bad ? 1 : 0;
}
