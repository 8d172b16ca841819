// fused matrix-core engine for wide windows, 23 window blocks of 16 positions: pad 153..168 (2 pad + 1 taps); one channel per workgroup
#include "fw_kernels.hpp"
BLUR_FW(23)
