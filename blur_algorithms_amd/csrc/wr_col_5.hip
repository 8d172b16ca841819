// column role, N = 1280 = 5 * 256: strips of 8 columns, 20 sub-blocks per (strip, channel) task
#include "wr_kernels.hpp"
BLUR_WR_COL(5, 4, 320)
