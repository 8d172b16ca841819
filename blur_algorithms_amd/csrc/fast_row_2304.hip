// row role of FFT length 2304 (1080p frames, sigma 20: 1920 columns + 2*65 pad + zeros).
// Flags 1 = LDS padding.  12 x 12 x 16 on 576 threads makes every pass one round of butterflies for the three
// channel lines (576, 576, 432).  Measured per 1080p frame: 17.9 us (the removed one-line-per-workgroup kernel with
// 16 x 9 x 4 x 4 on 192 threads: 24.2).
#include "fast_kernels.hpp"
BLUR_FAST_ROW(2304, 1, 576, 12, 12, 16)
