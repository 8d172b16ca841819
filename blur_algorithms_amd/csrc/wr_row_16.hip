// row role, N = 4096 = 16 * 256 (4K frames, sigma 20: 3840 columns + 2 * 65 pad = 3970): the 48 sub-blocks of the three
// channel lines of a row pair are one round of twelve waves
#include "wr_kernels.hpp"
BLUR_WR_ROW(16, 768)
