// row role, N = 1024 = 4 * 256: the three channel lines of a row pair together, 12 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(4, 768)
