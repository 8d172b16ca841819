// row role of FFT length 4000 (4K frames, sigma 20: 3840 columns + 2*65 pad + 30 zeros).
// One workgroup of 768 threads per CU transforms the three channel lines of a row pair at once (fast_rowpass3_u8,
// pass-0 twiddles in registers: 116 VGPRs).  16 x 25 x 10: 4000 has no 3-pass split with radices <= 16, and the
// radix-16 first pass is exactly one round (750 butterflies).  Measured per 4K frame: 57.9 us; 16 x 10 x 25 and
// 16 x 10 x 5 x 5 in the same kernel 68 and 70 us; the earlier kernel (one line per 256-thread workgroup, three
// workgroups per CU, 16 x 10 x 5 x 5, removed) 72 us.
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4000, 0, 768, 16, 25, 10)
