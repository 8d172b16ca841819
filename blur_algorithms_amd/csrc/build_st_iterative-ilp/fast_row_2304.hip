// FFT length 2304 = 16 x 9 x 4 x 4, row pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_ROW(N, LDS padding, threads per workgroup, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_ROW(2304, 0, 192, 16,9,4,4)
