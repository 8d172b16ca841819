// FFT length 4000 = 16 x 10 x 5 x 5, row pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_ROW(N, LDS padding, threads per workgroup, radices...)
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4000, 0, 256, 16,10,5,5)
