// row role, N = 3072 = 12 * 256: the three channel lines of a row pair together, 36 sub-blocks per unit
#include "wr_kernels.hpp"
BLUR_WR_ROW(12, 768)
