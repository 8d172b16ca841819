// FFT length 2560 = 10 x 16 x 16, column pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_COL(N, LDS padding, threads per workgroup, wave-local inner passes, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_COL(2560, 1, 512, 0, 10,16,16)
