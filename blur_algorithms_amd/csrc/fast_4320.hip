// FFT length 4320 = 16 x 18 x 15: compile-time specialised row / column kernels (fast_kernels.hpp)
// BLUR_FAST_INSTANCE(N, LDS padding, threads of the row kernel, threads of the column kernel, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_INSTANCE(4320, 1, 320, 540, 16,18,15)
