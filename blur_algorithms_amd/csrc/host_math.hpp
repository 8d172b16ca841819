// host_math.hpp -- host-side arithmetic of the engine: the reference's sizing and
// kernel generation (bit-identical), the kernel-spectrum multipliers and the FFT plans.
// No HIP in here; compiled into libblur_amd.so and usable from plain C++.
#pragma once
#include <cstdint>
#include <vector>

namespace blur_amd {

// Source.cpp:60-73
int gaussian_window(double sigma, int max_width = 0);
// Source.cpp:75-102 (kernel: max(width, fft_length) floats)
void get_gaussian(float* kernel, double sigma, int width, int fft_length);
// Utils.hpp:141-157
int is_valid_size(int n);
int nearest_transform_size(int n);

struct Sizing {
    int kSize, pad;
    int n_col;   // sizes[0]: FFT length of the column pass (rows + 2 pad, rounded up)
    int n_row;   // sizes[1]: FFT length of the row pass
    int tz_col, tz_row;
};
// Source.cpp:434-457
Sizing pffft_sizing(int rows, int cols, double sigma);

// pocketfft_2D's sizing, Source.cpp:149-176: one reflect-101 border per side; when a padded side is not 2^a 3^b 5^c the
// extra goes into the borders of that axis (floor before, ceil after), not into trailing zeros as in pffft_()
struct Sizing2D {
    int kSize, pad;
    int s0, s1;                       // sizes[0] (rows + top + bottom), sizes[1] (cols + left + right)
    int top, bottom, left, right;     // border[0..3]
};
Sizing2D pocketfft2d_sizing(int rows, int cols, double sigma);

// m[b] = float(Re DFT(kernel)[b]) * (1.f / n), b = 0..n/2   (Source.cpp:423,506-507)
void kernel_multipliers(double sigma, int ksize, int n, float* m);

// box_kernel (1D), Source.cpp:129-140: the tent (box * box) kernel of the `#define boxblur` mode,
// centred at index 0 of an n-periodic array (kernel: n floats, zero on entry)
void box_kernel_1d(float* kernel, int klen, int n);
// sizing of that mode, Source.cpp:437-442: kLen = kSize, pad = (kSize-1)/2 * 2
void boxfft_sizing(int rows, int cols, double nsmooth, int& klen, int& pad);
// m[b] = float(Re DFT(k)[b]) * (1.f / n) for an arbitrary n-periodic real kernel (only the real
// part of the kernel spectrum is used, Source.cpp:423)
void kernel_multipliers_from_array(const float* k, int n, float* m);

// ---- FFT plan: in-place decimation-in-frequency forward, decimation-in-time inverse.
// Forward pass i works on blocks of length len[i] = N / (radix[0]*...*radix[i-1]) with
// m[i] = len[i] / radix[i]; after all passes frequency f sits at position pos with
//   pos = q0*m[0] + q1*m[1] + ... , f = q0 + radix[0]*(q1 + radix[1]*(q2 + ...)).
// The inverse runs the same passes backwards, so no reordering is ever done: the
// pointwise multiply uses a multiplier table permuted to position order.
constexpr int kMaxPasses = 8;
struct FftPlan {
    int n = 0;
    int npass = 0;
    int radix[kMaxPasses] = {};
    int m[kMaxPasses] = {};
    int tw_off[kMaxPasses] = {};      // offset (in complex elements) of pass i's twiddles
    std::vector<float> tw;            // interleaved re,im; pass i: tw[(q-1)*m + j] = exp(-2 pi i j q / len_i)
    std::vector<int> freq_of_pos;     // n entries
};
// returns false if n has a prime factor other than 2, 3, 5
bool make_plan(int n, FftPlan& plan);
// same with an explicit radix sequence (the compile-time plans of fast_kernels.hpp)
bool make_plan_radices(int n, const int* radices, int npass, FftPlan& plan);

// multiplier table in POSITION order for a complex FFT of length n:
// mperm[pos] = m[min(f, n-f)], f = freq_of_pos[pos]; with `quirk`, f == n/2 uses m[0]
// (Source.cpp:420-425, the imaginary slot of i = 0 holds the Nyquist bin).
void permuted_multipliers(const FftPlan& plan, const float* m, bool quirk, float* mperm);

// ---- wave-resident engine (wr_kernels.hpp): transform length n = 256 * r0, tables in NATURAL frequency order.
// w256[m] = exp(-2 pi i m / 256), m = 0..255 (interleaved re, im)
void wr_w256(float* w256);
// pass-0 twiddles: tw0[(q-1)*256 + j] = exp(-2 pi i j q / n), q = 1..r0-1, j = 0..255 (interleaved re, im)
void wr_tw0(int r0, float* tw0);
// Multipliers of an n-periodic real even kernel `karr` (n floats, centre at index 0), all n bins:
//   mult[f] = Re DFT_n(karr)[f] / n                                              Source.cpp:423,506-507
// With `quirk` the reference scales its Nyquist bin (slot 1 of pffft's ordered layout) with the DC gain
// (Source.cpp:420-425); its transform length is n_ref, ours is n: the extra term (K[0] - K[n_ref/2]) / n_ref is
// added to bin n/2 (the alternating sum of the taps does not depend on the even length they are wrapped to).
void wr_multipliers(const float* karr, int n, int n_ref, bool quirk, float* mult);

// ---- matrix-core engine (mx_kernels.hpp): the convolution as a banded Toeplitz product on f16 MFMA ----------------
// IEEE binary16 <-> binary32 on the host (round to nearest even; no _Float16 needed from the host compiler)
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);
constexpr int kMxScaleLog2 = 14;      // taps are scaled by 2^14 before they are split into two halves
// padding of a 32-output tile's window to whole 8-element operand fragments: PADA = pad rounded up to 8,
// window = 32 + 2 PADA = 16 * nkb elements
inline int mx_pada(int pad) { return (pad + 7) & ~7; }
inline int mx_nkb(int pad) { return 2 + mx_pada(pad) / 8; }
// Operand fragments of the Toeplitz matrix Tz[w][o] = taps[w - o - PADA + pad] (zero outside the taps), PADA = 8 (nkb - 2)
// for the kernel's nkb >= mx_nkb(pad), w = 0..16 nkb - 1
// the window position, o = 0..31 the output: lane l of a wave holds, for block kb, the eight values
// Tz[16 kb + 8 (l >> 5) + j][l & 31], j = 0..7 -- the B operand of v_mfma_f32_32x32x16_f16 when the data is A (row pass)
// and equally the A operand when the data is B (column pass).  out: [2][nkb][64][8] binary16; part 0 = hi
// = f16(tap * 2^14), part 1 = lo = f16(tap * 2^14 - hi).  taps: 2 pad + 1 floats, centre at index pad.
void mx_fragments(const float* taps, int pad, int nkb, uint16_t* out);

}  // namespace blur_amd
