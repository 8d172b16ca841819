// fused matrix-core engine, 3 window blocks of 16 positions: pad <= 8 (2 pad + 1 taps)
#include "fx_kernels.hpp"
BLUR_FX(3)
