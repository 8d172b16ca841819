// column role, N = 1536 = 6 * 256: strips of 8 columns, 24 sub-blocks per (strip, channel) task
#include "wr_kernels.hpp"
BLUR_WR_COL(6, 4, 384)
