// FFT length 2304 = 9 x 16 x 16, column pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_COL(N, LDS padding, threads per workgroup, wave-local inner passes, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_COL(2304, 1, 512, 1, 9,16,16)
