// column role, N = 2304 = 9 * 256 (4K frames, sigma 20: 2160 rows + 2 * 65 pad = 2290): strips of 8 columns, the 36
// sub-blocks of a (strip, channel) task are one round of nine waves
#include "wr_kernels.hpp"
BLUR_WR_COL(9, 4, 576)
