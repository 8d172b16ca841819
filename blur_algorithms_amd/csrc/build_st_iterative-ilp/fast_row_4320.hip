// FFT length 4320 = 16 x 10 x 9 x 3, row pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_ROW(N, LDS padding, threads per workgroup, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4320, 0, 320, 16,10,9,3)
