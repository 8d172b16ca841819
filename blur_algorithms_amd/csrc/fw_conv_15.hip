// fused matrix-core engine for wide windows, 15 window blocks of 16 positions: pad 89..104 (2 pad + 1 taps); one channel per workgroup
#include "fw_kernels.hpp"
BLUR_FW(15)
