// FFT length 2560 = 16 x 16 x 10: compile-time specialised row / column kernels (fast_kernels.hpp)
// BLUR_FAST_INSTANCE(N, LDS padding, threads of the row kernel, threads of the column kernel, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_INSTANCE(2560, 1, 192, 640, 16,16,10)
