// column role, N = 2048 = 8 * 256: strips of 8 columns, 32 sub-blocks per (strip, channel) task
#include "wr_kernels.hpp"
BLUR_WR_COL(8, 4, 512)
