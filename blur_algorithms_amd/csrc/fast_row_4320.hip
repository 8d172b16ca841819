// row role of FFT length 4320 (4K frames, sigma 50: 3840 columns + 2*150 pad + zeros).
// Flags 32 = channels together (fast_rowpass3_u8), 18 x 16 x 15 on 768 threads: radix 18 first makes pass 0 one round
// (720 butterflies for the three lines).  Measured per 4K frame at sigma 50: 65.5 us against 84.6 for the same plan on
// three 256-thread workgroups per CU (fast_rowpass_u8).
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4320, 32, 768, 18, 16, 15)
