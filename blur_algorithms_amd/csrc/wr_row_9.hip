// row role, N = 2304 = 9 * 256: the three channel lines of a row pair together, 27 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(9, 768)
