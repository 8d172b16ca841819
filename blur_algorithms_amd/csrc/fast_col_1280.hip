// column role of FFT length 1280 (1080p frames, sigma 20: 1080 rows + 2*60 pad): 16 x 16 x 5, strips of 8 columns.
// Flags 3 = LDS padding + late radix-16 twiddles; 480 threads (one pass-0 butterfly group of 80 per line, 6 groups).
#include "fast_kernels.hpp"
BLUR_FAST_COL(1280, 3, 480, 0, 16, 16, 5)
