// FFT length 1280 = 16 x 16 x 5, col pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_COL(N, LDS padding, threads per workgroup, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_COL(1280, 1, 320, 16,16,5)
