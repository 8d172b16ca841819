// row role, N = 3840 = 15 * 256: the three channel lines of a row pair together, 45 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(15, 768)
