// fused matrix-core engine for wide windows, 19 window blocks of 16 positions: pad 121..136 (2 pad + 1 taps); one channel per workgroup
#include "fw_kernels.hpp"
BLUR_FW(19)
