// fused matrix-core engine for wide windows, 13 window blocks of 16 positions: pad 73..88 (2 pad + 1 taps); one channel per workgroup
#include "fw_kernels.hpp"
BLUR_FW(13)
