// column role of FFT length 2560 (4K frames, sigma 50: 2160 rows + 2*150 pad + zeros), strips of 8 columns.
// 10 x 16 x 16 on 768 threads.  Its LDS (4 lines + pixel stage + tables = 149 KB) has no room for pass-0 twiddles, so
// they stay in registers; flag 64 keeps the per-thread offsets out of the loop-invariant set instead, which is what
// lets 12 waves fit (156 VGPRs against 256 with the offsets hoisted).  Measured per 4K frame at sigma 50: 61.8 us
// against 72.3 on 512 threads.
#include "fast_kernels.hpp"
BLUR_FAST_COL(2560, 65, 768, 10, 16, 16)
