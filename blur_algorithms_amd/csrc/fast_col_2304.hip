// column role of FFT length 2304 (4K frames, sigma 20: 2160 rows + 2*60 pad): 9 x 16 x 16, strips of 8 columns.
// Flags 7 = LDS padding + register diet (radix-16 twiddles read at use, pass-0 twiddles in LDS): 768 threads =
// 12 waves per CU at <= 168 VGPRs with the strip prefetch still in registers (measured 60.9 us per 4K frame
// against 65.9 for 512 threads with register twiddles).
#include "fast_kernels.hpp"
BLUR_FAST_COL(2304, 7, 768, 0, 9, 16, 16)
