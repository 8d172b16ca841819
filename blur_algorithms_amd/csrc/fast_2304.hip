// FFT length 2304 = 16 x 16 x 9: compile-time specialised row / column kernels (fast_kernels.hpp)
// BLUR_FAST_INSTANCE(N, LDS padding, threads of the row kernel, threads of the column kernel, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_INSTANCE(2304, 1, 192, 576, 16,16,9)
