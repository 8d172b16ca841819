// row role of FFT length 4320 (4K frames, sigma 50: 3840 columns + 2*150 pad + zeros).
// 18 x 16 x 15 on 768 threads: radix 18 first makes pass 0 one round (720 butterflies for the three channel lines).
// Measured per 4K frame at sigma 50: 62.8 us (the removed one-line-per-workgroup kernel with the same plan: 84.6).
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4320, 0, 768, 18, 16, 15)
