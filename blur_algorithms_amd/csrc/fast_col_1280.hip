// column role of FFT length 1280 (1080p frames, sigma 20: 1080 rows + 2*65 pad), strips of 8 columns.
// 8 x 10 x 16 on 640 threads: every pass is at most one round of butterflies for the 4 lines of a strip
// (640, 512, 320), the radix-8 first pass keeps the kernel at 136 VGPRs (10 waves per CU, 3 on two of the SIMDs),
// flags 5 = LDS padding + pass-0 twiddles in LDS.  Measured per 1080p frame: 17.4 us against 22.0 for 16 x 16 x 5 on
// 480 threads.
#include "fast_kernels.hpp"
BLUR_FAST_COL(1280, 5, 640, 8, 10, 16)
