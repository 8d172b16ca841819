// FFT length 2560 = 10 x 16 x 16, col pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_COL(N, LDS padding, threads per workgroup, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_COL(2560, 1, 512, 10,16,16)
