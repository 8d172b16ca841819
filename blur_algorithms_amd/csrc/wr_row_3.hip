// row role, N = 768 = 3 * 256: the three channel lines of a row pair together, 9 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(3, 768)
