// FFT length 4000 = 16 x 10 x 5 x 5: compile-time specialised row / column kernels (fast_kernels.hpp)
// BLUR_FAST_INSTANCE(N, LDS padding, threads of the row kernel, threads of the column kernel, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_INSTANCE(4000, 0, 256, 512, 16,10,5,5)
