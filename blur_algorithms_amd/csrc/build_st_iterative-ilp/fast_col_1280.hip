// FFT length 1280 = 16 x 16 x 5, column pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_COL(N, LDS padding, threads per workgroup, wave-local inner passes, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_COL(1280, 1, 320, 0, 16,16,5)
