// row role, N = 1536 = 6 * 256: the three channel lines of a row pair together, 18 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(6, 768)
