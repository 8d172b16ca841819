// column role, N = 4096 = 16 * 256, TWO complex lines (a strip of 4 columns) per task: the lines and the byte stage of columns of up to
// 4096 points fit LDS (the C = 4 kernels stop at 2560); round 4, for whole images up to rows + 2 pad = 4096 and for the bands of the
// tiled path
#include "wr_kernels.hpp"
BLUR_WR_COL(16, 2, 512)
