// column role of FFT length 2304 (4K frames, sigma 20: 2160 rows + 2*65 pad), strips of 8 columns.
// 12 x 12 x 16 on 768 threads: with 4 lines per workgroup every pass is ONE round of butterflies (768, 768, 576), and
// flags 5 = LDS padding + pass-0 twiddles in LDS keep the kernel at <= 168 VGPRs, i.e. 12 waves per CU, with the strip
// prefetch still in registers.  Measured per 4K frame: 54.6 us against 65.9 for 9 x 16 x 16 on 512 threads with
// register twiddles (8 waves per CU) and 60.9 for 9 x 16 x 16 on 768 threads.
#include "fast_kernels.hpp"
BLUR_FAST_COL(2304, 5, 768, 12, 12, 16)
