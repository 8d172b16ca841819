// row role, N = 1280 = 5 * 256: the three channel lines of a row pair together, 15 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(5, 768)
