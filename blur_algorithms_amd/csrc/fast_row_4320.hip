// FFT length 4320 = 18 x 16 x 15, row pass: compile-time specialised kernel (fast_kernels.hpp)
// BLUR_FAST_ROW(N, LDS padding, threads per workgroup, radices...)
// radix 18 first: 240 pass-0 butterflies fit one 256-thread workgroup (3 workgroups per CU);
// 16 x 10 x 9 x 3 with 320 threads held only 2 per CU and ran 1.35x slower
#include "fast_kernels.hpp"
BLUR_FAST_ROW(4320, 0, 256, 18, 16, 15)
