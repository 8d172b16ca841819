// fft_engine.hpp -- the in-LDS complex FFT-convolution engine (device side).
//
// Replaces, for one group of lines resident in LDS, the triple
//   pffft_transform_ordered(FORWARD) -> pffft_sorted_optimized_convolution ->
//   pffft_transform_ordered(BACKWARD)                         Source.cpp:531-533,553-555
// TWO real lines ride in one complex line (re = line a, im = line b): the kernel
// spectrum is real and even, so multiplying the complex spectrum by it convolves both
// lines at once, and the Nyquist-slot quirk of Source.cpp:420-425 is one entry of the
// multiplier table.  Forward passes are in-place decimation-in-frequency, inverse passes
// in-place decimation-in-time over the same blocks in reverse order, so the spectrum is
// never reordered; the last forward pass, the pointwise multiply and the first inverse
// pass are fused in registers.
//
// Everything here is __host__ __device__ and takes (tid, nthreads) explicitly so that the
// same arithmetic can be driven from a CPU harness (tests/cpp/engine_host_check.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace blur_amd {

#define BLUR_HD __host__ __device__ __forceinline__

constexpr int kMaxPassesDev = 8;

// what a kernel needs to know about a plan (passed by value)
struct DevPlan {
    int n;
    int npass;
    int radix[kMaxPassesDev];
    int m[kMaxPassesDev];
    int tw_off[kMaxPassesDev];
    int nxcd;                       // XCDs of the device the launch goes to (0: unknown -> 8); filled in at the launch
};

BLUR_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
BLUR_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
BLUR_HD float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
BLUR_HD float2 cmulc(float2 a, float2 b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)
BLUR_HD float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> BLUR_HD float2 rot90(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

// ---- compile-time cos/sin of 2*pi*k/R for the constant twiddles inside composite butterflies
constexpr double cx_abs(double x) { return x < 0 ? -x : x; }
constexpr double cx_sin_small(double x)
{   // |x| <= pi/4, Taylor to x^19
    double term = x, sum = x;
    for (int i = 1; i < 10; ++i) { term *= -x * x / ((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double cx_cos_small(double x)
{
    double term = 1, sum = 1;
    for (int i = 1; i < 10; ++i) { term *= -x * x / ((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
// cos / sin of 2*pi*k/R, exact symmetries handled on the integer phase
constexpr double cx_cos_frac(int k, int R)
{
    k %= R; if (k < 0) k += R;
    // reduce to first octant on the fraction k/R: angle = 2 pi k / R
    // use eighths: t = 8k/R
    const double pi = 3.14159265358979323846264338327950288;
    const int k8 = 8 * k;
    if (k8 <= R)      return cx_cos_small(2 * pi * k / R);
    if (k8 <= 3 * R)  return -cx_sin_small(2 * pi * k / R - pi / 2);
    if (k8 <= 5 * R)  return -cx_cos_small(2 * pi * k / R - pi);
    if (k8 <= 7 * R)  return cx_sin_small(2 * pi * k / R - 3 * pi / 2);
    return cx_cos_small(2 * pi * k / R - 2 * pi);
}
constexpr double cx_sin_frac(int k, int R)
{
    k %= R; if (k < 0) k += R;
    const double pi = 3.14159265358979323846264338327950288;
    const int k8 = 8 * k;
    if (k8 <= R)      return cx_sin_small(2 * pi * k / R);
    if (k8 <= 3 * R)  return cx_cos_small(2 * pi * k / R - pi / 2);
    if (k8 <= 5 * R)  return -cx_sin_small(2 * pi * k / R - pi);
    if (k8 <= 7 * R)  return -cx_cos_small(2 * pi * k / R - 3 * pi / 2);
    return cx_sin_small(2 * pi * k / R - 2 * pi);
}

// v *= exp(-/+ 2 pi i k / R) with k, R compile-time; trivial cases cost nothing
template <int K, int R, bool INV> BLUR_HD float2 ctwiddle(float2 v)
{
    constexpr int k = ((K % R) + R) % R;
    if constexpr (k == 0) return v;
    else if constexpr (4 * k == R) return rot90<INV>(v);
    else if constexpr (2 * k == R) return make_float2(-v.x, -v.y);
    else if constexpr (4 * k == 3 * R) return rot90<!INV>(v);
    else {
        constexpr float c = static_cast<float>(cx_cos_frac(k, R));
        constexpr float s = static_cast<float>(INV ? cx_sin_frac(k, R) : -cx_sin_frac(k, R));
        return make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
    }
}

// ---- butterflies: v[0..R) <- DFT_R(v) (forward: exp(-i..), INV: exp(+i..)), natural order
template <int R, bool INV> struct Bfly;

template <bool INV> struct Bfly<2, INV> {
    static BLUR_HD void run(float2* v)
    {
        const float2 a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    }
};

template <bool INV> struct Bfly<4, INV> {
    static BLUR_HD void run(float2* v)
    {
        const float2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
        const float2 c = cadd(v[1], v[3]), d = rot90<INV>(csub(v[1], v[3]));
        v[0] = cadd(a, c); v[2] = csub(a, c);
        v[1] = cadd(b, d); v[3] = csub(b, d);
    }
};

template <bool INV> struct Bfly<3, INV> {
    static BLUR_HD void run(float2* v)
    {
        constexpr float h = 0.86602540378443864676f;
        const float2 t1 = cadd(v[1], v[2]);
        const float2 t2 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
        const float2 t3 = rot90<INV>(cscale(csub(v[1], v[2]), h));
        v[0] = cadd(v[0], t1);
        v[1] = cadd(t2, t3);
        v[2] = csub(t2, t3);
    }
};

template <bool INV> struct Bfly<5, INV> {
    static BLUR_HD void run(float2* v)
    {
        constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const float2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
        const float2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
        const float2 x0 = v[0];
        const float2 p1 = make_float2(x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y);
        const float2 p2 = make_float2(x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y);
        const float2 q1 = rot90<INV>(make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
        const float2 q2 = rot90<INV>(make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
        v[0] = make_float2(x0.x + a1.x + a2.x, x0.y + a1.y + a2.y);
        v[1] = cadd(p1, q1); v[4] = csub(p1, q1);
        v[2] = cadd(p2, q2); v[3] = csub(p2, q2);
    }
};

// R = A*B: B-point DFTs over x[a + A b], constant twiddles w_R^(a kb), A-point DFTs;
// X[kb + B ka] comes out of column kb, row ka.
template <int A, int B, bool INV> struct BflyComposite {
    static BLUR_HD void run(float2* v)
    {
        constexpr int R = A * B;
        float2 y[A][B];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            float2 t[B];
#pragma unroll
            for (int b = 0; b < B; ++b) t[b] = v[a + A * b];
            Bfly<B, INV>::run(t);
#pragma unroll
            for (int b = 0; b < B; ++b) y[a][b] = t[b];
        }
        twiddle_rows<0>(y);
#pragma unroll
        for (int kb = 0; kb < B; ++kb) {
            float2 t[A];
#pragma unroll
            for (int a = 0; a < A; ++a) t[a] = y[a][kb];
            Bfly<A, INV>::run(t);
#pragma unroll
            for (int ka = 0; ka < A; ++ka) v[kb + B * ka] = t[ka];
        }
        (void)R;
    }
    // y[a][kb] *= w_R^(a*kb), all indices compile-time
    template <int IDX> static BLUR_HD void twiddle_rows(float2 (&y)[A][B])
    {
        if constexpr (IDX < A * B) {
            constexpr int a = IDX / B, kb = IDX % B;
            y[a][kb] = ctwiddle<a * kb, A * B, INV>(y[a][kb]);
            twiddle_rows<IDX + 1>(y);
        }
    }
};

template <bool INV> struct Bfly<6, INV> { static BLUR_HD void run(float2* v) { BflyComposite<2, 3, INV>::run(v); } };
template <bool INV> struct Bfly<8, INV> { static BLUR_HD void run(float2* v) { BflyComposite<2, 4, INV>::run(v); } };
template <bool INV> struct Bfly<9, INV> { static BLUR_HD void run(float2* v) { BflyComposite<3, 3, INV>::run(v); } };
template <bool INV> struct Bfly<10, INV> { static BLUR_HD void run(float2* v) { BflyComposite<2, 5, INV>::run(v); } };
template <bool INV> struct Bfly<16, INV> { static BLUR_HD void run(float2* v) { BflyComposite<4, 4, INV>::run(v); } };
// larger composites, used by the compile-time plans of fast_kernels.hpp (3 passes for every BASELINE length)
template <bool INV> struct Bfly<12, INV> { static BLUR_HD void run(float2* v) { BflyComposite<3, 4, INV>::run(v); } };
template <bool INV> struct Bfly<15, INV> { static BLUR_HD void run(float2* v) { BflyComposite<3, 5, INV>::run(v); } };
template <bool INV> struct Bfly<18, INV> { static BLUR_HD void run(float2* v) { BflyComposite<2, 9, INV>::run(v); } };
template <bool INV> struct Bfly<20, INV> { static BLUR_HD void run(float2* v) { BflyComposite<4, 5, INV>::run(v); } };
template <bool INV> struct Bfly<25, INV> { static BLUR_HD void run(float2* v) { BflyComposite<5, 5, INV>::run(v); } };

// ---- LDS addressing: element i of a line lives at phys(i).  One spare element per 32
// breaks the power-of-two strides of the late passes (bank conflicts).
__host__ __device__ constexpr int phys(int i) { return i + (i >> 5); }
__host__ __device__ constexpr int line_stride(int n) { return phys(n) + 1; }   // elements per line buffer

// ---- passes.  z: C line buffers of `zs` complex elements each.  Butterfly g of a pass
// with block length R*m: blk = g / m, j = g % m, elements base + k*m, base = blk*R*m + j.

// forward DIF pass: butterfly, then twiddle w_len^(j q)
template <int R, int C> BLUR_HD void pass_fwd(float2* z, int zs, int n, int m, const float2* tw, int tid, int nthreads)
{
    const int nb = n / R;
    for (int g = tid; g < nb; g += nthreads) {
        const int blk = g / m, j = g - blk * m;
        const int base = blk * R * m + j;
        float2 w[R];
#pragma unroll
        for (int q = 1; q < R; ++q) w[q] = tw[(q - 1) * m + j];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float2* zc = z + c * zs;
            float2 v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = zc[phys(base + k * m)];
            Bfly<R, false>::run(v);
            zc[phys(base)] = v[0];
#pragma unroll
            for (int q = 1; q < R; ++q) zc[phys(base + q * m)] = cmul(v[q], w[q]);
        }
    }
}

// inverse DIT pass: conj twiddle, then inverse butterfly
template <int R, int C> BLUR_HD void pass_inv(float2* z, int zs, int n, int m, const float2* tw, int tid, int nthreads)
{
    const int nb = n / R;
    for (int g = tid; g < nb; g += nthreads) {
        const int blk = g / m, j = g - blk * m;
        const int base = blk * R * m + j;
        float2 w[R];
#pragma unroll
        for (int q = 1; q < R; ++q) w[q] = tw[(q - 1) * m + j];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float2* zc = z + c * zs;
            float2 v[R];
            v[0] = zc[phys(base)];
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmulc(zc[phys(base + q * m)], w[q]);
            Bfly<R, true>::run(v);
#pragma unroll
            for (int k = 0; k < R; ++k) zc[phys(base + k * m)] = v[k];
        }
    }
}

// last forward pass (m == 1, no twiddles) + pointwise multiply + first inverse pass
template <int R, int C> BLUR_HD void pass_mid(float2* z, int zs, int n, const float* mperm, int tid, int nthreads)
{
    const int nb = n / R;
    for (int g = tid; g < nb; g += nthreads) {
        const int base = g * R;
        float mm[R];
#pragma unroll
        for (int q = 0; q < R; ++q) mm[q] = mperm[base + q];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float2* zc = z + c * zs;
            float2 v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = zc[phys(base + k)];
            Bfly<R, false>::run(v);
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = cscale(v[q], mm[q]);
            Bfly<R, true>::run(v);
#pragma unroll
            for (int k = 0; k < R; ++k) zc[phys(base + k)] = v[k];
        }
    }
}

// a pass with m == 1 on its own (no twiddles, no multiply): the last forward or the first inverse pass of a transform
// that is NOT fused with a pointwise multiply (the whole-image 2D path, Source.cpp:233, keeps its spectrum)
template <int R, int C, bool INV> BLUR_HD void pass_plain(float2* z, int zs, int n, int tid, int nthreads)
{
    const int nb = n / R;
    for (int g = tid; g < nb; g += nthreads) {
        const int base = g * R;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float2* zc = z + c * zs;
            float2 v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) v[k] = zc[phys(base + k)];
            Bfly<R, INV>::run(v);
#pragma unroll
            for (int k = 0; k < R; ++k) zc[phys(base + k)] = v[k];
        }
    }
}

enum PassKind { kFwd = 0, kMid = 1, kInv = 2, kFwdLast = 3, kInvFirst = 4 };

template <int R, int C>
BLUR_HD void run_pass_r(int kind, float2* z, int zs, int n, int m, const float2* tw, const float* mperm, int tid, int nthreads)
{
    if (kind == kFwd) pass_fwd<R, C>(z, zs, n, m, tw, tid, nthreads);
    else if (kind == kInv) pass_inv<R, C>(z, zs, n, m, tw, tid, nthreads);
    else if (kind == kMid) pass_mid<R, C>(z, zs, n, mperm, tid, nthreads);
    else if (kind == kFwdLast) pass_plain<R, C, false>(z, zs, n, tid, nthreads);
    else pass_plain<R, C, true>(z, zs, n, tid, nthreads);
}

// one pass of the plan for thread `tid`; the caller synchronises between passes
template <int C>
BLUR_HD void run_pass(int kind, int R, float2* z, int zs, int n, int m, const float2* tw, const float* mperm, int tid, int nthreads)
{
    switch (R) {
    case 2: run_pass_r<2, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 3: run_pass_r<3, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 4: run_pass_r<4, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 5: run_pass_r<5, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 6: run_pass_r<6, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 8: run_pass_r<8, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
#if !defined(BLUR_GENERIC_MAX_RADIX) || BLUR_GENERIC_MAX_RADIX > 8
    case 9: run_pass_r<9, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 10: run_pass_r<10, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 16: run_pass_r<16, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
#endif
#ifdef BLUR_ENGINE_ALL_RADICES   // CPU harness only: the generic kernels never plan these
    case 12: run_pass_r<12, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 15: run_pass_r<15, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 18: run_pass_r<18, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 20: run_pass_r<20, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
    case 25: run_pass_r<25, C>(kind, z, zs, n, m, tw, mperm, tid, nthreads); break;
#endif
    default: break;
    }
}

// The schedule of a plan: 2*npass-1 steps.  step s < npass-1: forward pass s;
// s == npass-1: fused middle; s > npass-1: inverse pass 2*(npass-1)-s.
BLUR_HD void schedule_step(const DevPlan& p, int s, int& kind, int& pass)
{
    if (s < p.npass - 1) { kind = kFwd; pass = s; }
    else if (s == p.npass - 1) { kind = kMid; pass = s; }
    else { kind = kInv; pass = 2 * (p.npass - 1) - s; }
}

// forward FFT, multiply by mperm, inverse FFT of C lines in LDS (unnormalised: the 1/N is
// folded into mperm, Source.cpp:423).  Ends with a barrier.
template <int C>
__device__ __forceinline__ void fftconv_lines(float2* z, int zs, const DevPlan& p, const float2* tw, const float* mperm)
{
    const int steps = 2 * p.npass - 1;
    for (int s = 0; s < steps; ++s) {
        int kind, i;
        schedule_step(p, s, kind, i);
        run_pass<C>(kind, p.radix[i], z, zs, p.n, p.m[i], tw + p.tw_off[i], mperm, threadIdx.x, blockDim.x);
        __syncthreads();
    }
}

// forward FFT only of C lines in LDS; the spectrum stays in position order (frequency freq_of_pos[pos] at
// position pos).  Ends with a barrier.
template <int C>
__device__ __forceinline__ void fft_forward_lines(float2* z, int zs, const DevPlan& p, const float2* tw)
{
    for (int i = 0; i < p.npass; ++i) {
        run_pass<C>(i == p.npass - 1 ? kFwdLast : kFwd, p.radix[i], z, zs, p.n, p.m[i], tw + p.tw_off[i], nullptr, threadIdx.x, blockDim.x);
        __syncthreads();
    }
}

// inverse FFT only (unnormalised) of C spectra held in position order.  Ends with a barrier.
template <int C>
__device__ __forceinline__ void fft_inverse_lines(float2* z, int zs, const DevPlan& p, const float2* tw)
{
    for (int i = p.npass - 1; i >= 0; --i) {
        run_pass<C>(i == p.npass - 1 ? kInvFirst : kInv, p.radix[i], z, zs, p.n, p.m[i], tw + p.tw_off[i], nullptr, threadIdx.x, blockDim.x);
        __syncthreads();
    }
}

}  // namespace blur_amd
