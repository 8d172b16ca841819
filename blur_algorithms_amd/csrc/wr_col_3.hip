// column role, N = 768 = 3 * 256: strips of 8 columns, 12 sub-blocks per (strip, channel) task
#include "wr_kernels.hpp"
BLUR_WR_COL(3, 4, 192)
