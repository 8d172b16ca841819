// FFT length 1280 = 16 x 16 x 5: compile-time specialised row / column kernels (fast_kernels.hpp)
// BLUR_FAST_INSTANCE(N, LDS padding, threads of the row kernel, threads of the column kernel, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_INSTANCE(1280, 1, 128, 320, 16,16,5)
