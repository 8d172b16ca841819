// row role of FFT length 4000 (4K frames, sigma 20: 3840 columns + 2*60 pad + 40 zeros).
// Flags 32 = channels together (fast_rowpass3_u8): one workgroup of 768 threads per CU transforms the three channel
// lines of a row pair at once (pass-0 twiddles in registers: 116 VGPRs); 16 x 25 x 10 because 4000 has no 3-pass split with radices <= 16 and the radix-16
// first pass is exactly one round (750 butterflies).  Measured per 4K frame: 70.1 us against 72.2 for
// 16 x 10 x 5 x 5 on three 256-thread workgroups per CU (fast_rowpass_u8), 74.6 without the staged input.
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4000, 32, 768, 16, 25, 10)
