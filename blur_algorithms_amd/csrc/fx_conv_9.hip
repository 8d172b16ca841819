// fused matrix-core engine, 9 window blocks of 16 positions: pad <= 56 (2 pad + 1 taps)
#include "fx_kernels.hpp"
BLUR_FX(9)
