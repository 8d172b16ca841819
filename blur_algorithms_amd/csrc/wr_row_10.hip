// row role, N = 2560 = 10 * 256: the three channel lines of a row pair together, 30 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(10, 768)
