// row role, N = 2048 = 8 * 256: the three channel lines of a row pair together, 24 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(8, 768)
