// column role, N = 1024 = 4 * 256: strips of 8 columns, 16 sub-blocks per (strip, channel) task
#include "wr_kernels.hpp"
BLUR_WR_COL(4, 4, 256)
