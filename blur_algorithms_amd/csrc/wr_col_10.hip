// column role, N = 2560 = 10 * 256: strips of 8 columns, 40 sub-blocks per (strip, channel) task
#include "wr_kernels.hpp"
BLUR_WR_COL(10, 4, 640)
